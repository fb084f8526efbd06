"""CPU oracle of the occlusion-mesh depth maps (TEST INFRASTRUCTURE ONLY; small meshes -- plain Python loops).

The reference renders occlusion meshes with OpenGL (src/opengl/renderer.cc) and cannot be reproduced bit for bit by any
software implementation; this oracle pins the SPECIFICATION the HIP rasteriser implements:
  vertex stage     the vertex shaders' arithmetic per camera model (renderer.cc:226-262, 470-495, 630-653), f32
  projection       p = f * x'/z + c (SetupProjection :919-974 in this code base's pixel-centre convention)
  near plane       triangles crossing z = min_depth are clipped on the vertex stage's output (x', y', z), before the division
                   (OpenGL clips in clip space): cut = inside vertex + t * (outside - inside), t = (near - z_in) / (z_out - z_in),
                   f32; one vertex cut off -> the quad is split along (first cut, second kept vertex)
  rasterisation    samples at integer pixel coordinates, f64 edge functions, top-left rule
  depth            perspective-correct interpolation of camera-space z, nearest fragment, 0 where nothing was drawn
  boundaries       ComputeEdgeNormalsList / FilterEdgeList (occlusion_geometry.cc:488-645) and MaskOutOcclusionBoundaries
                   (:284-402) with visibility and pixel tests against the UNMASKED map (the reference's result depends on
                   OpenMP timing)
"""
import numpy as np

from . import reg_binding as rb


def _atanf(x):
    """atanf of include/e3d_libm.h (the implementation kernels and oracle share), elementwise on an f32 array."""
    from . import binding
    return binding.libm_eval("atanf", np.ascontiguousarray(x, np.float32)).reshape(np.shape(x))


def _atan2f_1(r):
    """atan2f(r, 1.f) of include/e3d_libm.h."""
    from . import binding
    r = np.ascontiguousarray(r, np.float32)
    return binding.libm_eval("atan2f", r, np.ones_like(r)).reshape(r.shape)


F = np.float32


def project_vertices(model, cam, R, t, verts, shaded=False):
    """pixel position and camera-space z of every vertex; shaded=True also returns the vertex stage's (x', y')"""
    R = np.asarray(R, F); t = np.asarray(t, F); v = np.asarray(verts, F)
    X = (R[0, 0] * v[:, 0] + (R[0, 1] * v[:, 1] + R[0, 2] * v[:, 2])) + t[0]
    Y = (R[1, 0] * v[:, 0] + (R[1, 1] * v[:, 1] + R[1, 2] * v[:, 2])) + t[1]
    Z = (R[2, 0] * v[:, 0] + (R[2, 1] * v[:, 1] + R[2, 2] * v[:, 2])) + t[2]
    lx, ly = X.copy(), Y.copy()
    if model in (6, 7, 8):
        # SimpleRadial / Radial / Polynomial shaders (renderer.cc:402-413, 326-337, 288-300): the radial factor (99 outside the cut-off)
        # times the camera-space x, y.  SimpleRadial has its own r2 expression.
        with np.errstate(all="ignore"):
            q = [F(cam.p[4 + i]) for i in range(3)]
            if model == 6:
                r2 = ((X * X + Y * Y) / (Z * Z)).astype(F)
                fac = np.where(r2 > F(cam.cutoff2), F(99.0), F(1.0) + r2 * q[0]).astype(F)
            else:
                nx, ny = X / Z, Y / Z
                r2 = (nx * nx + ny * ny).astype(F)
                poly = F(1.0) + r2 * (q[0] + r2 * q[1]) if model == 7 else F(1.0) + r2 * (q[0] + r2 * (q[1] + r2 * q[2]))
                fac = np.where(r2 <= F(cam.cutoff2), poly, F(99.0)).astype(F)
            lx = (fac * X).astype(F); ly = (fac * Y).astype(F)
    elif model in (11, 12):
        # RadialFisheye / SimpleRadialFisheye shaders (renderer.cc:361-378, 434-452): r2 after the fisheye warp against the outer camera's cut-off
        with np.errstate(all="ignore"):
            nx, ny = (X / Z).astype(F), (Y / Z).astype(F)
            r2 = (nx * nx + ny * ny).astype(F)
            r = np.sqrt(r2).astype(F)
            big = r > F(1e-6)
            th = np.where(big, _atan2f_1(r) / r, F(1.0)).astype(F)
            if model == 11:
                wx, wy = (th * nx).astype(F), (th * ny).astype(F)
                nx2, ny2 = np.where(big, wx, nx).astype(F), np.where(big, wy, ny).astype(F)
                r2 = np.where(big, nx2 * nx2 + ny2 * ny2, r2).astype(F)
                poly = F(1.0) + r2 * (F(cam.p[4]) + r2 * F(cam.p[5]))
            else:
                r2 = np.where(big, (r2 * th).astype(F) * th, r2).astype(F)
                nx2, ny2 = np.where(big, nx * th, nx).astype(F), np.where(big, ny * th, ny).astype(F)
                poly = F(1.0) + r2 * F(cam.p[4])
            fac = np.where(r2 <= F(cam.cutoff2), poly, F(99.0)).astype(F)
            lx = ((Z * fac) * nx2).astype(F); ly = ((Z * fac) * ny2).astype(F)
    elif model == 10:
        # FullOpenCV shader (renderer.cc:528-543)
        with np.errstate(all="ignore"):
            nx, ny = (X / Z).astype(F), (Y / Z).astype(F)
            x2, xy, y2 = nx * nx, nx * ny, ny * ny
            r2 = (x2 + y2).astype(F)
            k1, k2, p1, p2, k3, k4, k5, k6 = [F(cam.p[4 + i]) for i in range(8)]
            radial = (F(1.0) + r2 * (k1 + r2 * (k2 + r2 * k3))) / (F(1.0) + r2 * (k4 + r2 * (k5 + r2 * k6)))
            dx = Z * (radial * nx + F(2.0) * p1 * xy + p2 * (r2 + F(2.0) * x2))
            dy = Z * (radial * ny + F(2.0) * p2 * xy + p1 * (r2 + F(2.0) * y2))
            inside = r2 <= F(cam.cutoff2)
            lx = np.where(inside, dx, X * F(99.0)).astype(F); ly = np.where(inside, dy, Y * F(99.0)).astype(F)
    elif model not in (0, 5):
        with np.errstate(all="ignore"):
            nx, ny = X / Z, Y / Z
            r2 = nx * nx + ny * ny
            inside = r2 <= F(cam.cutoff2)
            if model == 4:
                # FisheyeFOV shader (renderer.cc:153-160); the camera class's guard on the optical axis, where GLSL's 0 / 0 is undefined
                r = (np.sqrt(X * X + Y * Y).astype(F) / Z).astype(F)
                fac = np.where(r < F(1e-6), F(1.0), _atanf((r * F(cam.p[5])).astype(F)) / (r * F(cam.p[4]))).astype(F)
                lx = (fac * X).astype(F); ly = (fac * Y).astype(F)
                px = F(cam.p[0]) * (lx / Z) + F(cam.p[2]); py = F(cam.p[1]) * (ly / Z) + F(cam.p[3])
                return (px.astype(F), py.astype(F), Z.astype(F)) + ((lx.astype(F), ly.astype(F)) if shaded else ())
            if model == 3:
                # FisheyePolynomial4 shader (renderer.cc:187-205): r2 becomes the radial factor, 99 outside the cut-off
                r = np.sqrt(r2)
                th = np.where(r > F(1e-6), _atan2f_1(r.astype(F)) / r, F(1.0)).astype(F)
                fx_, fy_ = np.where(r > F(1e-6), th * nx, nx).astype(F), np.where(r > F(1e-6), th * ny, ny).astype(F)
                rr = np.where(r > F(1e-6), th * th * r2, r2).astype(F)
                k1, k2, k3, k4 = [F(cam.p[4 + i]) for i in range(4)]
                fac = F(1.0) + rr * (k1 + rr * (k2 + rr * (k3 + rr * k4)))
                fac = np.where(inside, fac, F(99.0)).astype(F)
                lx = ((Z * fac) * fx_).astype(F); ly = ((Z * fac) * fy_).astype(F)
                px = F(cam.p[0]) * (lx / Z) + F(cam.p[2]); py = F(cam.p[1]) * (ly / Z) + F(cam.p[3])
                return (px.astype(F), py.astype(F), Z.astype(F)) + ((lx.astype(F), ly.astype(F)) if shaded else ())
            if model in (2, 9):          # THIN_PRISM_FISHEYE, FISHEYE_POLYNOMIAL_2_TANGENTIAL_2 (renderer.cc:236-258)
                r = np.sqrt(r2)
                th = np.where(r > F(1e-6), _atan2f_1(r.astype(F)) / r, F(1.0)).astype(F)
                nx = np.where(r > F(1e-6), th * nx, nx); ny = np.where(r > F(1e-6), th * ny, ny)
            x2, xy, y2 = nx * nx, nx * ny, ny * ny
            r2 = x2 + y2
            q = [F(cam.p[4 + i]) for i in range(cam.n_params - 4)]
            k1, k2, p1, p2 = q[:4]
            if model in (1, 9):
                radial = F(1.0) + r2 * (k1 + r2 * k2)
                dx = Z * (radial * nx + F(2.0) * p1 * xy + p2 * (r2 + F(2.0) * x2))
                dy = Z * (radial * ny + F(2.0) * p2 * xy + p1 * (r2 + F(2.0) * y2))
            else:
                k3, k4, sx1, sy1 = q[4:8]
                radial = F(1.0) + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)))
                dx = Z * (radial * nx + F(2.0) * p1 * xy + p2 * (r2 + F(2.0) * x2) + sx1 * r2)
                dy = Z * (radial * ny + F(2.0) * p2 * xy + p1 * (r2 + F(2.0) * y2) + sy1 * r2)
            lx = np.where(inside, dx, X * F(99.0)).astype(F); ly = np.where(inside, dy, Y * F(99.0)).astype(F)
    with np.errstate(all="ignore"):
        px = F(cam.p[0]) * (lx / Z) + F(cam.p[2]); py = F(cam.p[1]) * (ly / Z) + F(cam.p[3])
    return (px.astype(F), py.astype(F), Z.astype(F)) + ((lx.astype(F), ly.astype(F)) if shaded else ())


def _inside(ax, ay, bx, by, px, py):
    dx, dy = bx - ax, by - ay
    e = dx * (py - ay) - dy * (px - ax)
    return (e > 0) | ((e == 0) & ((dy < 0) | ((dy == 0) & (dx > 0)))), e


def _near_cut(pin, pout, near, proj):
    """(px, py, z) of the point where the edge inside -> outside meets the near plane; p = (px, py, z, lx, ly), f32"""
    fx, fy, cx, cy = proj
    with np.errstate(all="ignore"):
        t = (near - pin[2]) / (pout[2] - pin[2])
        lx = pin[3] + t * (pout[3] - pin[3]); ly = pin[4] + t * (pout[4] - pin[4])
        return (fx * (lx / near) + cx, fy * (ly / near) + cy, near)


def near_clip(p, near, proj):
    """p: three vertices (px, py, z, lx, ly) as f32 scalars -> list of sub-triangles [(a, b, c)] of (px, py, z)"""
    inside = [bool(v[2] >= near) for v in p]
    n_in = sum(inside)
    if n_in == 3:
        return [tuple(v[:3] for v in p)]
    if n_in == 0:
        return []
    if n_in == 1:
        i = inside.index(True); j, k = (i + 1) % 3, (i + 2) % 3
        return [(p[i][:3], _near_cut(p[i], p[j], near, proj), _near_cut(p[i], p[k], near, proj))]
    i = inside.index(False); j, k = (i + 1) % 3, (i + 2) % 3
    pj, pk = _near_cut(p[j], p[i], near, proj), _near_cut(p[k], p[i], near, proj)
    return [(pj, p[j][:3], p[k][:3]), (pj, p[k][:3], pk)]


def rasterise(px, py, z, tris, W, H, min_depth=0.05, max_depth=100.0, shaded=None, proj=None):
    """shaded = (lx, ly) and proj = (fx, fy, cx, cy) enable near-plane clipping; without them every vertex must be in front of the
    near plane (triangles that are not are skipped)"""
    depth = np.full((H, W), np.inf, np.float32)
    near = F(min_depth)
    for f in range(len(tris)):
        i0, i1, i2 = (int(v) for v in tris[f])
        if shaded is not None:
            P = [(px[i], py[i], z[i], shaded[0][i], shaded[1][i]) for i in (i0, i1, i2)]
            subs = near_clip(P, near, tuple(F(v) for v in proj))
        else:
            subs = [((px[i0], py[i0], z[i0]), (px[i1], py[i1], z[i1]), (px[i2], py[i2], z[i2]))] if (z[i0] >= near and z[i1] >= near and z[i2] >= near) else []
        for A, B, C in subs:
            _raster_one(depth, A, B, C, W, H, min_depth, max_depth)
    depth[np.isinf(depth)] = 0
    return depth


def _raster_one(depth, A, B, C, W, H, min_depth, max_depth):
    za, zb, zc = A[2], B[2], C[2]
    if za > max_depth and zb > max_depth and zc > max_depth:
        return
    xs = np.array([A[0], B[0], C[0]], np.float32); ys = np.array([A[1], B[1], C[1]], np.float32)
    if not (np.all(np.isfinite(xs)) and np.all(np.isfinite(ys))):
        return
    if xs.max() < 0 or ys.max() < 0 or xs.min() > W - 1 or ys.min() > H - 1:
        return
    x0, y0 = max(0, int(np.ceil(xs.min()))), max(0, int(np.ceil(ys.min())))
    x1, y1 = min(W - 1, int(np.floor(xs.max()))), min(H - 1, int(np.floor(ys.max())))
    if x0 > x1 or y0 > y1:
        return
    ax, ay, bx, by, cx, cy = (float(v) for v in (xs[0], ys[0], xs[1], ys[1], xs[2], ys[2]))
    area = (bx - ax) * (cy - ay) - (by - ay) * (cx - ax)
    if area == 0:
        return
    if area < 0:
        bx, by, cx, cy = cx, cy, bx, by
        zb, zc = zc, zb
    yy, xx = np.mgrid[y0:y1 + 1, x0:x1 + 1].astype(np.float64)
    in0, e0 = _inside(bx, by, cx, cy, xx, yy)
    in1, e1 = _inside(cx, cy, ax, ay, xx, yy)
    in2, e2 = _inside(ax, ay, bx, by, xx, yy)
    a = e0 + e1 + e2
    ok = in0 & in1 & in2 & (a > 0)
    if not ok.any():
        return
    with np.errstate(all="ignore"):
        inv = (e0 / float(za) + e1 / float(zb) + e2 / float(zc)) / a
        zz = (1.0 / inv).astype(np.float32)
    ok &= (zz >= np.float32(min_depth)) & (zz <= np.float32(max_depth))
    sub = depth[y0:y1 + 1, x0:x1 + 1]
    sub[ok] = np.minimum(sub[ok], zz[ok])


def edge_list(verts, tris):
    """ComputeEdgeNormalsList + FilterEdgeList -> (list of (v1, v2, f1, f2 or -1, opposite), face normals f32)."""
    v = np.asarray(verts, F)
    normals = np.zeros((len(tris), 3), F)
    half = {}
    for f, (i0, i1, i2) in enumerate(np.asarray(tris, np.int64)):
        a = v[i1] - v[i0]; b = v[i2] - v[i0]
        n = np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], F)
        ln = np.sqrt(n[0] * n[0] + (n[1] * n[1] + n[2] * n[2]))
        normals[f] = n / ln if ln > 0 else n
        for p, q in ((i0, i1), (i1, i2), (i2, i0)):
            swap = p > q
            key = (min(p, q), max(p, q))
            half.setdefault(key, []).append((f, swap))
    edges = []
    for (v1, v2), faces in sorted(half.items()):
        f1 = faces[0][0]
        if len(faces) == 1:
            edges.append((v1, v2, f1, -1, 0)); continue
        factor1 = F(-1.0) if faces[0][1] else F(1.0); factor2 = F(-1.0) if faces[1][1] else F(1.0)
        f2 = faces[1][0]
        e = v[v2] - v[v1]
        n1 = normals[f1] * factor1; n2 = normals[f2] * factor2
        opposite = 1 if factor1 * factor2 > 0 else 0
        bx = n1 / np.sqrt(n1[0] * n1[0] + (n1[1] * n1[1] + n1[2] * n1[2]))
        by = np.array([bx[1] * e[2] - bx[2] * e[1], bx[2] * e[0] - bx[0] * e[2], bx[0] * e[1] - bx[1] * e[0]], F)
        by = by / np.sqrt(by[0] * by[0] + (by[1] * by[1] + by[2] * by[2]))
        n1_2 = np.array([1, 0], F)
        n2_2 = np.array([bx[0] * n2[0] + (bx[1] * n2[1] + bx[2] * n2[2]), by[0] * n2[0] + (by[1] * n2[1] + by[2] * n2[2])], F)
        if n2_2[0] < 0 and abs(n2_2[1]) < F(1e-4):
            continue
        keep = True
        if len(faces) > 2:
            cross12 = n2_2[1]
            for f3, s3 in faces[2:]:
                factor3 = F(-1.0) if s3 else F(1.0)
                c = normals[f3] * factor3
                n3 = np.array([bx[0] * c[0] + (bx[1] * c[1] + bx[2] * c[2]), by[0] * c[0] + (by[1] * c[1] + by[2] * c[2])], F)
                c13 = n1_2[0] * n3[1] - n1_2[1] * n3[0]; c23 = n2_2[0] * n3[1] - n2_2[1] * n3[0]
                sign1 = c13 * cross12 > 0; sign2 = c23 * cross12 < 0
                if sign1 and not sign2:
                    n2_2 = n3; f2 = f3; factor2 = factor3; opposite = 1 if factor1 * factor3 != 1 else 0
                elif sign2 and not sign1:
                    n1_2 = n3; f1 = f3; factor1 = factor3; opposite = 1 if factor3 * factor2 != 1 else 0
                elif not sign2:
                    keep = False; break
        if keep:
            edges.append((v1, v2, f1, f2, opposite))
    return edges, normals


def mask_boundaries(depth, edges, normals, verts, R, t, cam, splat_radius=0.03):
    R = np.asarray(R, F); t = np.asarray(t, F); v = np.asarray(verts, F)
    out = depth.copy()
    pos = -(R.T.astype(np.float64) @ t.astype(np.float64)).astype(F)
    # float32 in the kernel's order: -(R0*t0 + R3*t1 + R6*t2)
    pos = np.array([-(R[0, k] * t[0] + R[1, k] * t[1] + R[2, k] * t[2]) for k in range(3)], F)
    H, W = depth.shape

    def tr(p):
        return np.array([(R[k, 0] * p[0] + (R[k, 1] * p[1] + R[k, 2] * p[2])) + t[k] for k in range(3)], F)
    for v1, v2, f1, f2, opposite in edges:
        p1, p2 = v[v1], v[v2]
        if f2 >= 0:
            te = pos - p1
            face1 = (normals[f1][0] * te[0] + (normals[f1][1] * te[1] + normals[f1][2] * te[2])) > 0
            face2 = (normals[f2][0] * te[0] + (normals[f2][1] * te[1] + normals[f2][2] * te[2])) > 0
            if not ((opposite and face1 == face2) or (face1 != face2 and not opposite)):
                continue
        a, b = tr(p1), tr(p2)
        if a[2] <= 0 or b[2] <= 0:
            continue
        d = b - a
        count = 1 + min(int(np.sqrt(d[0] * d[0] + (d[1] * d[1] + d[2] * d[2])) / F(splat_radius) + F(0.5)), 150)
        for k in range(count):
            with np.errstate(all="ignore"):
                factor = F(k) / (F(count) - F(1.0))
            P = (a + factor * d).astype(F)
            if not (P[2] > 0):
                continue
            pxy = rb.cam_project(cam, P)
            ix, iy = int(np.trunc(pxy[0] + F(0.5))), int(np.trunc(pxy[1] + F(0.5)))
            if not (pxy[0] + F(0.5) >= 0 and pxy[1] + F(0.5) >= 0 and 0 <= ix < W and 0 <= iy < H):
                continue
            if not (depth[iy, ix] + F(0.05) >= P[2]):
                continue
            dw = rb.cam_deriv_by_world(cam, P)
            rx = np.sqrt(dw[0, 0] * dw[0, 0] + (dw[0, 1] * dw[0, 1] + dw[0, 2] * dw[0, 2])) * F(splat_radius)
            ry = np.sqrt(dw[1, 0] * dw[1, 0] + (dw[1, 1] * dw[1, 1] + dw[1, 2] * dw[1, 2])) * F(splat_radius)
            mnx = max(0, int(np.float64(F(ix) - rx) + 0.5)); mny = max(0, int(np.float64(F(iy) - ry) + 0.5))
            ex = min(W, int(np.float64(F(ix) + rx) + 1.5)); ey = min(H, int(np.float64(F(iy) + ry) + 1.5))
            if mnx < ex and mny < ey:
                sub_in = depth[mny:ey, mnx:ex]
                m = (sub_in == 0) | (sub_in + F(0.05) > P[2])
                out[mny:ey, mnx:ex][m] = -1
    return out
