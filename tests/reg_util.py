"""Shared helpers for the path-(B) tests: image pyramids and a small synthetic registration scene."""
import numpy as np


def pyramid_u8(img, n_levels):
    """Image::BuildImagePyramid (src/opt/image.cc:106-131): cv::resize(..., INTER_AREA) by 1/2 on CV_8U = exact 2x2 box
    mean with round-half-up for even sizes (recalled); odd trailing rows/cols are dropped."""
    out = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(1, n_levels):
        a = out[-1]
        h, w = (a.shape[0] // 2) * 2, (a.shape[1] // 2) * 2
        a = a[:h, :w].astype(np.uint16)
        s = a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2]
        out.append(((s + 2) >> 2).astype(np.uint8))
    return out


def look_at_pose(eye, target, up=(0, 0, 1)):
    """image_T_global (R, t) of a camera at `eye` looking at `target` (camera z forward, y down)."""
    eye = np.asarray(eye, np.float64); target = np.asarray(target, np.float64)
    z = target - eye; z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64)); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z]).astype(np.float32)
    t = (-R.astype(np.float64) @ eye).astype(np.float32)
    return R, t


def quat_from_R(R):
    """(w, x, y, z) float32 unit quaternion of a rotation matrix."""
    from scipy.spatial.transform import Rotation
    x, y, z, w = Rotation.from_matrix(np.asarray(R, np.float64)).as_quat()
    q = np.array([w, x, y, z], np.float64)
    return (q / np.linalg.norm(q)).astype(np.float32)


def quat_to_R(q):
    from oracle import binding as ob
    return ob.quat_to_R(np.asarray(q, np.float32))


def texture(u, v):
    return 120 + 55 * np.sin(7.0 * u) * np.cos(5.0 * v) + 35 * np.sin(3.0 * u + 4.0 * v) + 20 * np.cos(11.0 * v - 2.0 * u)


# distortion parameter sets of the test scenes, keyed by camera model (0 PINHOLE, 1 OPENCV, 2 THIN_PRISM_FISHEYE, 3 OPENCV_FISHEYE, 4 FOV)
DISTORTION = {
    0: [],
    1: [-0.101082, 0.0703954, 0.000438661, -0.000680887],
    2: [0.0221184, 0.0128597, 0.000531602, -0.000388873, 0.00623079, 0.0020419, -0.000805024, 4.07704e-05],
    3: [0.0221184, 0.0128597, 0.00623079, 0.0020419],
    4: [0.9],
    5: [],                                                     # SIMPLE_PINHOLE
    6: [-0.101082],                                            # SIMPLE_RADIAL (k < 0: its closed-form cut-off is finite)
    7: [-0.101082, 0.0703954],                                 # RADIAL
    8: [-0.101082, 0.0703954, 0.0123],                         # POLYNOMIAL_3
    9: [0.0221184, 0.0128597, 0.000531602, -0.000388873],      # FISHEYE_POLYNOMIAL_2_TANGENTIAL_2
    10: [-0.101082, 0.0703954, 0.000438661, -0.000680887, 0.0123, 0.021, -0.0087, 0.0031],   # FullOpenCVCamera: k1 k2 p1 p2 k3 k4 k5 k6
    11: [0.0221184, 0.0128597],                                # RadialFisheyeCamera (FisheyeBase over RadialCamera)
    12: [-0.0221184],                                          # SimpleRadialFisheyeCamera (k < 0: a finite inner cut-off)
}
UNIQUE_FOCAL = (5, 6, 7, 11, 12)  # parameter vector [f cx cy ...] (camera_base_impl.h:65-67)


def camera_params(model, fx, fy, cx, cy, distortion=None):
    """The model's parameter vector in the reference's GetParameters order; one-focal-length models take fx."""
    d = DISTORTION[model] if distortion is None else list(distortion)
    head = [fx, cx, cy] if model in UNIQUE_FOCAL else [fx, fy, cx, cy]
    return np.array(head + d, np.float32)


def expand_params(model, params):
    """[fx fy cx cy] + distortion for any model (fx = fy = f for the one-focal-length ones)."""
    p = [float(v) for v in params]
    return (p[0], p[0], p[1], p[2], p[3:]) if model in UNIQUE_FOCAL else (p[0], p[1], p[2], p[3], p[4:])



def distort_np(model, q, nx, ny):
    """float64 numpy version of the models' Distort (only for synthesising consistent test images)."""
    if model in (0, 5):
        return nx, ny
    if model in (11, 12):              # the fisheye warp, then the radial polynomial inside
        r = np.sqrt(nx * nx + ny * ny)
        f = np.where(r > 1e-6, np.arctan(r) / np.maximum(r, 1e-12), 1.0)
        nx, ny = nx * f, ny * f
    if model in (6, 7, 8, 11, 12):
        r2 = nx * nx + ny * ny
        k = list(q) + [0.0, 0.0]
        fac = 1 + r2 * (k[0] + r2 * (k[1] + r2 * k[2]))
        return nx * fac, ny * fac
    if model == 4:
        r = np.sqrt(nx * nx + ny * ny)
        f = np.where(r > 1e-6, np.arctan(r * 2 * np.tan(0.5 * q[0])) / (np.maximum(r, 1e-12) * q[0]), 1.0)
        return nx * f, ny * f
    if model in (2, 3, 9):
        r = np.sqrt(nx * nx + ny * ny)
        f = np.where(r > 1e-6, np.arctan(r) / np.maximum(r, 1e-12), 1.0)
        nx, ny = nx * f, ny * f
    x2, xy, y2 = nx * nx, nx * ny, ny * ny
    r2 = x2 + y2
    if model == 3:
        fac = 1 + r2 * (q[0] + r2 * (q[1] + r2 * (q[2] + r2 * q[3])))
        return nx * fac, ny * fac
    k1, k2, p1, p2 = q[:4]
    if model == 10:
        k3, k4, k5, k6 = q[4:8]
        radial = (1 + r2 * (k1 + r2 * (k2 + r2 * k3))) / (1 + r2 * (k4 + r2 * (k5 + r2 * k6)))
        return nx * radial + 2 * p1 * xy + p2 * (r2 + 2 * x2), ny * radial + 2 * p2 * xy + p1 * (r2 + 2 * y2)
    if model in (1, 9):
        radial = 1 + r2 * (k1 + r2 * k2)
        return nx * radial + 2 * p1 * xy + p2 * (r2 + 2 * x2), ny * radial + 2 * p2 * xy + p1 * (r2 + 2 * y2)
    k3, k4, sx1, sy1 = q[4:8]
    radial = 1 + r2 * (k1 + r2 * (k2 + r2 * (k3 + r2 * k4)))
    return (nx * radial + 2 * p1 * xy + p2 * (r2 + 2 * x2) + sx1 * r2, ny * radial + 2 * p2 * xy + p1 * (r2 + 2 * y2) + sy1 * r2)


def undistort_np(model, q, dx, dy, iters=30):
    """Inverse of distort_np by fixed-point iteration on the normalized coordinates (mild distortions only)."""
    nx, ny = dx.copy(), dy.copy()
    for _ in range(iters):
        ex, ey = distort_np(model, q, nx, ny)
        nx, ny = nx - (ex - dx), ny - (ey - dy)
    return nx, ny


def make_multi_image_scene(n_points=5000, n_images=3, width=240, height=180, n_levels=3, K=5, seed=0, perturb=0.01, model=0):
    """Planar textured wall (y = 3) seen by several pinhole cameras whose images are ray-traced from the texture, so that
    the true poses minimise the photometric cost; returns true and perturbed poses."""
    from scipy.spatial import cKDTree
    rng = np.random.RandomState(seed)
    u = rng.uniform(-1.0, 1.0, n_points); v = rng.uniform(-0.75, 0.75, n_points)
    pts = np.stack([u, np.full(n_points, 3.0), v], 1).astype(np.float32)
    _, nn = cKDTree(pts).query(pts, k=K + 1)
    nbr = nn[:, 1:].astype(np.uint32)
    tex = texture(pts[:, 0].astype(np.float64), pts[:, 2].astype(np.float64))
    fixed_desc = (tex[nbr] - tex[:, None]).astype(np.float32)
    params = camera_params(model, 210.0, 208.0, width / 2 - 0.4, height / 2 + 0.3)
    eyes = [(-0.35, -0.3, 0.1), (0.3, -0.2, -0.08), (0.02, -0.45, 0.2), (0.2, -0.5, -0.15)][:n_images]
    images = []
    for i, eye in enumerate(eyes):
        R0, t0 = look_at_pose(eye, (0.05 * i, 3, 0.02 * i))
        q = quat_from_R(R0); R = quat_to_R(q).astype(np.float64); t = t0.astype(np.float64)
        yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
        efx, efy, ecx, ecy, eq = expand_params(model, params)
        unx, uny = undistort_np(model, eq, (xx - ecx) / efx, (yy - ecy) / efy)
        d = np.stack([unx, uny, np.ones_like(xx)], -1) @ R     # R^T * dir
        o = -R.T @ t
        lam = (3.0 - o[1]) / d[..., 1]
        hit = o + lam[..., None] * d
        img = texture(hit[..., 0], hit[..., 2]).clip(0, 250)
        pyr = pyramid_u8(np.rint(img).astype(np.uint8), n_levels)
        # perturbed start pose
        dq = rng.normal(size=3) * perturb
        dt = rng.normal(size=3) * perturb
        from scipy.spatial.transform import Rotation
        Rp = Rotation.from_rotvec(dq).as_matrix() @ R
        images.append(dict(q_true=q, t_true=t0.astype(np.float32), q_init=quat_from_R(Rp), t_init=(t + dt).astype(np.float32), pyr=pyr))
    return dict(pts=pts, nbr=nbr, K=K, fixed_desc=fixed_desc, params=params, width=width, height=height, n_levels=n_levels,
                images=images, point_radius=0.01, model=model)


def plane_depth_pyramid(M, im):
    """Depth (camera z) of the wall y = 3 of make_multi_image_scene seen from the true pose, 2 x 2 means for the coarser levels."""
    R = quat_to_R(im["q_true"]).astype(np.float64); t = im["t_true"].astype(np.float64)
    yy, xx = np.mgrid[0:M["height"], 0:M["width"]].astype(np.float64)
    fx, fy, cx, cy, q = expand_params(M["model"], M["params"])
    nx, ny = undistort_np(M["model"], q, (xx - cx) / fx, (yy - cy) / fy)
    d = np.stack([nx, ny, np.ones_like(xx)], -1) @ R
    o = -R.T @ t
    maps = [((3.0 - o[1]) / d[..., 1])]
    for l in range(1, M["n_levels"]):
        p = maps[-1]
        maps.append(0.25 * (p[0::2, 0::2] + p[1::2, 0::2] + p[0::2, 1::2] + p[1::2, 1::2]))
    return [m.astype(np.float32) for m in maps]


def make_reg_scene(n_points=6000, width=320, height=240, n_levels=4, K=5, seed=0, model=0):
    """A textured, slightly wavy wall seen by a pinhole camera: points + neighbour graph + descriptors + image pyramid."""
    rng = np.random.RandomState(seed)
    u = rng.uniform(-1.2, 1.2, n_points); v = rng.uniform(-0.9, 0.9, n_points)
    pts = np.stack([u, 0.05 * np.sin(3 * u) * np.cos(2 * v) + 3.0, v], 1).astype(np.float32)      # wall near y = 3
    # K nearest neighbours in 3D (any fixed neighbour graph works for the kernels)
    from scipy.spatial import cKDTree
    _, nn = cKDTree(pts).query(pts, k=K + 1)
    nbr = nn[:, 1:].astype(np.uint32)
    yy, xx = np.mgrid[0:height, 0:width]
    img = (110 + 60 * np.sin(xx / 9.0) * np.cos(yy / 7.0) + 40 * np.sin((xx + 2 * yy) / 23.0)).clip(0, 250)
    img = img.astype(np.uint8)
    img[:12, :20] = 255                                     # an oversaturated corner
    pyr = pyramid_u8(img, n_levels)
    R0, t = look_at_pose((0.1, -0.4, 0.05), (0, 3, 0))
    q = quat_from_R(R0)
    R = quat_to_R(q)          # so3().matrix() of the stored quaternion: what both implementations actually use
    params = camera_params(model, 260.0, 255.0, width / 2 - 0.3, height / 2 + 0.2)
    fixed_desc = rng.normal(0, 8, (n_points, K)).astype(np.float32)
    var_desc = rng.normal(0, 8, (n_points, K)).astype(np.float32)
    obs_counts = rng.randint(0, 4, n_points).astype(np.int32)
    return dict(pts=pts, nbr=nbr, K=K, pyr=pyr, R=R, q=q, t=t, params=params, width=width, height=height, n_levels=n_levels,
                fixed_desc=fixed_desc, var_desc=var_desc, obs_counts=obs_counts, point_radius=0.012, model=model)


def make_rig_scene(n_points=6000, n_frames=2, width=240, height=180, n_levels=3, K=5, seed=0, perturb=0.004, model=0):
    """Two-camera rig observed in `n_frames` frames (image ids 2f = reference camera, 2f + 1 = second camera): images are
    ray-traced from the true rig geometry; start values perturb the frame poses and the rig extrinsics."""
    from scipy.spatial import cKDTree
    from scipy.spatial.transform import Rotation
    from oracle import reg_binding as rb
    rng = np.random.RandomState(seed)
    u = rng.uniform(-1.0, 1.0, n_points); v = rng.uniform(-0.75, 0.75, n_points)
    pts = np.stack([u, np.full(n_points, 3.0), v], 1).astype(np.float32)
    _, nn = cKDTree(pts).query(pts, k=K + 1)
    nbr = nn[:, 1:].astype(np.uint32)
    tex = texture(pts[:, 0].astype(np.float64), pts[:, 2].astype(np.float64))
    fixed_desc = (tex[nbr] - tex[:, None]).astype(np.float32)
    params = camera_params(model, 210.0, 208.0, width / 2 - 0.4, height / 2 + 0.3)
    # true extrinsics of camera 1: 12 cm baseline, a few degrees of rotation
    R1 = Rotation.from_rotvec([0.01, -0.04, 0.02]).as_matrix()
    q1_true = quat_from_R(R1); t1_true = np.array([-0.12, 0.01, 0.005], np.float32)
    ident = (np.array([1, 0, 0, 0], np.float32), np.zeros(3, np.float32))
    eyes = [(-0.3, -0.3, 0.1), (0.25, -0.25, -0.08), (0.0, -0.45, 0.2)][:n_frames]
    images = []
    for f, eye in enumerate(eyes):
        R0, t0 = look_at_pose(eye, (0.05 * f, 3, 0.02 * f))
        q_ref = quat_from_R(R0)
        q_dep, t_dep = rb.se3_mul(q1_true, t1_true, q_ref, t0)
        dq = rng.normal(size=3) * perturb; dt = rng.normal(size=3) * perturb
        q_init = quat_from_R(Rotation.from_rotvec(dq).as_matrix() @ quat_to_R(q_ref).astype(np.float64))
        for q, t in ((q_ref, t0), (q_dep, t_dep)):
            R = quat_to_R(q).astype(np.float64); tt = t.astype(np.float64)
            yy, xx = np.mgrid[0:height, 0:width].astype(np.float64)
            efx, efy, ecx, ecy, eq = expand_params(model, params)
            unx, uny = undistort_np(model, eq, (xx - ecx) / efx, (yy - ecy) / efy)
            d = np.stack([unx, uny, np.ones_like(xx)], -1) @ R
            o = -R.T @ tt
            lam = (3.0 - o[1]) / d[..., 1]
            hit = o + lam[..., None] * d
            img = texture(hit[..., 0], hit[..., 2]).clip(0, 250)
            images.append(dict(pyr=pyramid_u8(np.rint(img).astype(np.uint8), n_levels), q_true=q, t_true=t.astype(np.float32)))
        images[-2]["q_init"] = q_init; images[-2]["t_init"] = (t0.astype(np.float64) + dt).astype(np.float32)
    dq = rng.normal(size=3) * perturb
    rig_init = [ident, (quat_from_R(Rotation.from_rotvec(dq).as_matrix() @ R1), (t1_true + rng.normal(size=3).astype(np.float32) * perturb).astype(np.float32))]
    return dict(pts=pts, nbr=nbr, K=K, fixed_desc=fixed_desc, params=params, width=width, height=height, n_levels=n_levels,
                images=images, point_radius=0.01, model=model, rig_true=[ident, (q1_true, t1_true)], rig_init=rig_init,
                frames=[[2 * f, 2 * f + 1] for f in range(n_frames)])


# ---- the synthetic scene of the reference's Test4FrameAlignment (src/opt/test/test_alignment.cc:87-340) ---------------------------
def _render_color_depth(verts, colors, tris, R, t, W, H, fx, fy, cx, cy, min_depth, max_depth):
    """Gouraud-shaded triangle mesh through a pinhole camera: perspective-correct colour and camera-space depth at the integer
    pixel positions, nearest fragment wins, 0 where nothing is drawn (what the reference's OpenGL renderer produces for this
    scene; triangles with a vertex in front of the near plane are dropped -- in this scene they lie outside the image)."""
    P = verts.astype(np.float64) @ np.asarray(R, np.float64).T + np.asarray(t, np.float64)
    z = P[:, 2]
    with np.errstate(all="ignore"):
        px = fx * P[:, 0] / z + cx; py = fy * P[:, 1] / z + cy
    depth = np.full((H, W), np.inf); color = np.zeros((H, W, 3))
    for i0, i1, i2 in tris:
        zs = z[[i0, i1, i2]]
        if not np.all(zs > min_depth):
            continue
        xs, ys = px[[i0, i1, i2]], py[[i0, i1, i2]]
        if xs.max() < 0 or ys.max() < 0 or xs.min() > W - 1 or ys.min() > H - 1:
            continue
        x0, y0 = max(0, int(np.ceil(xs.min()))), max(0, int(np.ceil(ys.min())))
        x1, y1 = min(W - 1, int(np.floor(xs.max()))), min(H - 1, int(np.floor(ys.max())))
        if x0 > x1 or y0 > y1:
            continue
        yy, xx = np.mgrid[y0:y1 + 1, x0:x1 + 1].astype(np.float64)
        area = (xs[1] - xs[0]) * (ys[2] - ys[0]) - (ys[1] - ys[0]) * (xs[2] - xs[0])
        if area == 0:
            continue
        e0 = ((xs[2] - xs[1]) * (yy - ys[1]) - (ys[2] - ys[1]) * (xx - xs[1])) / area
        e1 = ((xs[0] - xs[2]) * (yy - ys[2]) - (ys[0] - ys[2]) * (xx - xs[2])) / area
        e2 = 1.0 - e0 - e1
        ok = (e0 >= 0) & (e1 >= 0) & (e2 >= 0)
        if not ok.any():
            continue
        w0, w1, w2 = e0 / zs[0], e1 / zs[1], e2 / zs[2]
        zz = 1.0 / (w0 + w1 + w2)
        ok &= (zz >= min_depth) & (zz <= max_depth)
        sub_d = depth[y0:y1 + 1, x0:x1 + 1]; sub_c = color[y0:y1 + 1, x0:x1 + 1]
        ok &= zz < sub_d
        c = (w0[..., None] * colors[i0] + w1[..., None] * colors[i1] + w2[..., None] * colors[i2]) * zz[..., None]
        sub_d[ok] = zz[ok]; sub_c[ok] = c[ok]
    depth[np.isinf(depth)] = 0
    return np.clip(np.rint(color), 0, 255).astype(np.uint8), depth.astype(np.float32)


def make_four_frame_scene(seed=0):
    """Heightmap of 61 x 61 vertices over 5 m x 5 m at z = 1 +- 0.05, pulled towards the camera by 6 * (distance from the centre
    in grid units), random vertex colours; two recordings 0.5 m apart in y, each with two cameras 0.1 m apart in x, pinhole
    256 x 256 with f = 128; the scan = 30 % of the pixels of the four depth maps, coloured from the images; initial poses =
    ground truth moved by (d, d, 0) * 0.002 with d = 1 for camera 0 and 3 for camera 1 (the last of the three perturbation blocks
    of the reference test overwrites the two before it)."""
    rng = np.random.RandomState(seed)
    n = 61
    gx, gy = np.meshgrid(np.arange(n), np.arange(n))
    u, v = gx / (n - 1.0) - 0.5, gy / (n - 1.0) - 0.5
    zz = 1.0 + rng.uniform(-0.05, 0.05, (n, n)) - 6 * np.sqrt(u * u + v * v)
    verts = np.stack([u * 5.0, v * 5.0, zz], -1).reshape(-1, 3).astype(np.float32)
    colors = rng.randint(0, 256, (n * n, 3)).astype(np.float64)
    tris = []
    for y in range(n - 1):
        for x in range(n - 1):
            tris += [(x + (y + 1) * n, (x + 1) + y * n, x + y * n), (x + (y + 1) * n, (x + 1) + (y + 1) * n, (x + 1) + y * n)]
    W = H = 256
    fx = fy = 128.0; cx = cy = 127.5
    image_T_global = {}
    for s, ty in ((0, -0.25), (1, 0.25)):
        for c, tx in ((0, 0.0), (1, 0.1)):
            image_T_global[(s, c)] = (np.eye(3), -np.array([tx, ty, 0.0]))        # inverse of a pure translation
    out = dict(width=W, height=H, params=np.array([fx, fy, cx, cy]), images={}, pts=[], rgb=[])
    for key, (R, t) in image_T_global.items():
        col, dep = _render_color_depth(verts, colors, tris, R, t, W, H, fx, fy, cx, cy, 0.1, 1.2 * 1.05)
        sel = (dep > 0) & (rng.uniform(0, 1, dep.shape) < 0.3)
        ys, xs = np.nonzero(sel)
        d = dep[sel].astype(np.float64)
        cam = np.stack([d * (xs - cx) / fx, d * (ys - cy) / fy, d], -1)
        out["pts"].append((cam - t) @ R)                                          # global_T_image * p
        out["rgb"].append(col[sel])
        dirn = 1.0 if key[1] == 0 else 3.0
        out["images"][key] = dict(color=col, depth=dep, R=R, t=t, t_init=t + np.array([dirn * 0.002, dirn * 0.002, 0.0]))
    out["pts"] = np.concatenate(out["pts"]).astype(np.float32)
    out["rgb"] = np.concatenate(out["rgb"]).astype(np.uint8)
    return out


def se3_log(T):
    """Sophus::SE3::log of a 4 x 4 rigid transform: [translation part (V^-1 t), rotation vector]"""
    from scipy.spatial.transform import Rotation
    w = Rotation.from_matrix(T[:3, :3]).as_rotvec()
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    Vinv = np.eye(3) - 0.5 * K + (1 / 12.0 if th < 1e-6 else (1 - th * np.cos(th / 2) / (2 * np.sin(th / 2))) / th**2) * (K @ K)
    return np.concatenate([Vinv @ T[:3, 3], w])
