"""Seeded synthetic laser-scan generator used by bench.py and the tests (SURVEY.md section 8d "synthetic inputs").

A 10 m x 10 m x 3 m "room": floor + 4 walls (5 planes) + 3 vertical cylinders.  Each scan samples the surfaces
uniformly by area, adds Gaussian range noise along the ray from the scanner, and is expressed in the scanner's
local frame (origin-centred, yawed), exactly what a .ply + MeshLab pose pair of the real pipeline holds.
Everything is torch so that 2 x 50 M points can be generated directly in HBM; on CPU it is used for the small
parity cases.  No reference data is involved.
"""
import math

import numpy as np
import torch

_CYL = [(3.0, 3.0, 0.4), (7.0, 4.0, 0.4), (5.0, 7.5, 0.4)]
_ROOM = (10.0, 10.0, 3.0)


def _rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


def rot_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


# The partial-overlap variant of the room ("~30-60 % of points find a partner"): a partition wall (a box
# 10 cm thick from the y = 0 wall to y = 5) splits the floor plan, the scanner has a maximum range, and a surface point is
# kept only if the ray from the scanner reaches it unobstructed (cylinders and the partition occlude).  Two scans on either
# side of the partition then share the open end of the room and little else.
_PART = (4.95, 5.05, 0.0, 5.0)      # x0, x1, y0, y1 (full height)
PARTIAL_MAX_RANGE = 6.5


def visible_from(p, origin, cyl, part, max_range):
    """Mask of the surface points p [n,3] (f64 torch, exact surface positions) a scanner at `origin` sees."""
    o = torch.tensor(origin, device=p.device, dtype=torch.float64)
    d = p - o
    vis = d.norm(dim=1) <= max_range
    dx, dy = d[:, 0], d[:, 1]
    a = dx * dx + dy * dy
    for (cx, cy, r) in cyl:                               # vertical cylinders: a circle in the floor plan
        fx, fy = o[0] - cx, o[1] - cy
        b = 2.0 * (fx * dx + fy * dy)
        c = fx * fx + fy * fy - r * r
        disc = b * b - 4.0 * a * c
        t0 = (-b - torch.sqrt(disc.clamp_(min=0.0))) / (2.0 * a).clamp_(min=1e-30)
        vis &= ~((disc > 0) & (t0 > 1e-9) & (t0 < 1.0 - 1e-6))
    x0, x1, y0, y1 = part                                 # the partition: slab test in the floor plan
    inv_x = 1.0 / torch.where(dx.abs() < 1e-30, torch.full_like(dx, 1e-30), dx)
    inv_y = 1.0 / torch.where(dy.abs() < 1e-30, torch.full_like(dy, 1e-30), dy)
    tx0, tx1 = (x0 - o[0]) * inv_x, (x1 - o[0]) * inv_x
    ty0, ty1 = (y0 - o[1]) * inv_y, (y1 - o[1]) * inv_y
    t_in = torch.maximum(torch.minimum(tx0, tx1), torch.minimum(ty0, ty1))
    t_out = torch.minimum(torch.maximum(tx0, tx1), torch.maximum(ty0, ty1))
    vis &= ~((t_in < t_out) & (t_out > 1e-9) & (t_in < 1.0 - 1e-6))
    return vis


def make_scan(n, origin, yaw, seed, sigma=0.002, device="cpu", room_scale=1.0, partial=False):
    """Returns (xyz_local[n,3] f32, normals_local[n,3] f32, T_true[4,4] f32 numpy).  room_scale stretches the floor plan
    (walls, cylinder positions, scanner position) in x and y: with n proportional to room_scale^2 the point density stays that
    of the 10 m room -- the weak-scaling workload of bench.py.  partial=True: the partial-overlap room (see _PART)."""
    if partial:
        return _make_scan_partial(n, origin, yaw, seed, sigma, device)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    W, D, Hh = _ROOM[0] * room_scale, _ROOM[1] * room_scale, _ROOM[2]
    cyl = [(cx * room_scale, cy * room_scale, r) for (cx, cy, r) in _CYL]
    origin = (origin[0] * room_scale, origin[1] * room_scale, origin[2])
    areas = [W * D, D * Hh, D * Hh, W * Hh, W * Hh] + [2 * math.pi * r * Hh for (_, _, r) in cyl]
    cum = torch.tensor(np.cumsum(areas) / np.sum(areas), device=device, dtype=torch.float64)
    sel = torch.rand(n, generator=g, device=device, dtype=torch.float64)
    prim = torch.bucketize(sel, cum).clamp_(max=len(areas) - 1)
    u = torch.rand(n, generator=g, device=device, dtype=torch.float64)
    v = torch.rand(n, generator=g, device=device, dtype=torch.float64)
    p = torch.zeros(n, 3, device=device, dtype=torch.float64)
    nr = torch.zeros(n, 3, device=device, dtype=torch.float64)

    def put(mask, px, py, pz, nx, ny, nz):
        p[mask, 0] = px; p[mask, 1] = py; p[mask, 2] = pz
        nr[mask, 0] = nx; nr[mask, 1] = ny; nr[mask, 2] = nz

    m = prim == 0
    put(m, u[m] * W, v[m] * D, 0.0, 0.0, 0.0, 1.0)
    m = prim == 1
    put(m, 0.0, u[m] * D, v[m] * Hh, 1.0, 0.0, 0.0)
    m = prim == 2
    put(m, W, u[m] * D, v[m] * Hh, -1.0, 0.0, 0.0)
    m = prim == 3
    put(m, u[m] * W, 0.0, v[m] * Hh, 0.0, 1.0, 0.0)
    m = prim == 4
    put(m, u[m] * W, D, v[m] * Hh, 0.0, -1.0, 0.0)
    for ci, (cx, cy, r) in enumerate(cyl):
        m = prim == 5 + ci
        ang = u[m] * (2 * math.pi)
        put(m, cx + r * torch.cos(ang), cy + r * torch.sin(ang), v[m] * Hh, torch.cos(ang), torch.sin(ang), 0.0)

    o = torch.tensor(origin, device=device, dtype=torch.float64)
    ray = p - o
    rng = ray.norm(dim=1, keepdim=True).clamp_(min=1e-9)
    ray = ray / rng
    noise = torch.randn(n, 1, generator=g, device=device, dtype=torch.float64) * sigma
    p = p + noise * ray
    # normals point towards the scanner (what flipNormalTowardsViewpoint yields in the local frame)
    flip = ((o - p) * nr).sum(dim=1, keepdim=True) < 0
    nr = torch.where(flip, -nr, nr)
    R = torch.tensor(_rot_z(yaw), device=device, dtype=torch.float64)
    xyz_local = ((p - o) @ R).to(torch.float32).contiguous()       # R^T (p - o)
    nrm_local = (nr @ R).to(torch.float32).contiguous()
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = _rot_z(yaw)
    T[:3, 3] = np.asarray(origin, dtype=np.float64)
    return xyz_local, nrm_local, T.astype(np.float32)


def make_scan_angular(n, origin, yaw, seed, sigma=0.002, device="cpu", scan_order=False):
    """The room as a REAL scanner samples it: rays in directions uniform on the sphere (equal angular steps), first hit with the
    floor, the four walls and the cylinders -- the point density falls off with cos(incidence) / range^2 instead of being
    uniform per area (make_scan).  Rays that leave through the open top are dropped.  Returns what make_scan returns.
    scan_order=True: the points in the order a terrestrial scanner records them (README.md:304-305: Faro Focus scans) -- the head
    turns about the vertical axis while the mirror sweeps vertical lines: columns of equal azimuth, bottom to top within a column
    (~sqrt(2 n) columns: equal angular steps in both directions); the default is the order the random rays were drawn in."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    W, D, Hh = _ROOM
    o = torch.tensor(origin, device=device, dtype=torch.float64)
    pts, nrms, az, el = [], [], [], []
    have = 0
    while have < n:
        m = int((n - have) * 1.7) + 1024
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64) * 2 - 1          # cos(polar): uniform on the sphere
        ph = torch.rand(m, generator=g, device=device, dtype=torch.float64) * (2 * math.pi)
        sr = torch.sqrt((1 - u * u).clamp_(min=0))
        d = torch.stack([sr * torch.cos(ph), sr * torch.sin(ph), u], 1)
        inf = torch.full((m,), float("inf"), device=device, dtype=torch.float64)
        t = inf.clone()
        nr = torch.zeros(m, 3, device=device, dtype=torch.float64)

        def plane(axis, value, normal):
            nonlocal t, nr
            tt = (value - o[axis]) / d[:, axis]
            p = o + tt[:, None] * d
            ok = (tt > 1e-9) & (tt < t) & (p[:, 0] >= -1e-9) & (p[:, 0] <= W + 1e-9) & (p[:, 1] >= -1e-9) & (p[:, 1] <= D + 1e-9) & (p[:, 2] >= -1e-9) & (p[:, 2] <= Hh + 1e-9)
            t = torch.where(ok, tt, t)
            nr[ok] = torch.tensor(normal, device=device, dtype=torch.float64)
        plane(2, 0.0, (0.0, 0.0, 1.0)); plane(0, 0.0, (1.0, 0.0, 0.0)); plane(0, W, (-1.0, 0.0, 0.0))
        plane(1, 0.0, (0.0, 1.0, 0.0)); plane(1, D, (0.0, -1.0, 0.0))
        a = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]
        for (cx, cy, r) in _CYL:
            fx, fy = o[0] - cx, o[1] - cy
            b = 2.0 * (fx * d[:, 0] + fy * d[:, 1])
            c = fx * fx + fy * fy - r * r
            disc = b * b - 4.0 * a * c
            tt = (-b - torch.sqrt(disc.clamp(min=0.0))) / (2.0 * a).clamp(min=1e-30)
            z = o[2] + tt * d[:, 2]
            ok = (disc > 0) & (tt > 1e-9) & (tt < t) & (z >= 0) & (z <= Hh)
            t = torch.where(ok, tt, t)
            p = o + tt[:, None] * d
            nc = torch.stack([(p[:, 0] - cx) / r, (p[:, 1] - cy) / r, torch.zeros_like(tt)], 1)
            nr = torch.where(ok[:, None], nc, nr)
        hit = torch.isfinite(t)
        noise = torch.randn(m, generator=g, device=device, dtype=torch.float64) * sigma
        p = o + (t + noise)[:, None] * d
        pts.append(p[hit]); nrms.append(nr[hit])
        if scan_order:
            az.append(ph[hit]); el.append(u[hit])
        have += int(hit.sum())
        del d, t, nr, p, u, ph, sr, noise, hit
    p = torch.cat(pts)[:n]; nr = torch.cat(nrms)[:n]
    del pts, nrms
    if scan_order:
        ncol = max(1, int(math.sqrt(2.0 * n)))
        col = torch.clamp((torch.cat(az)[:n] * (ncol / (2 * math.pi))).to(torch.int64), max=ncol - 1)
        key = col.to(torch.float64) * 4.0 + (torch.cat(el)[:n] + 1.0)          # column, then elevation (cos(polar) + 1 in [0, 2])
        order = torch.argsort(key)
        p = p[order]; nr = nr[order]
        del az, el, col, key, order
    flip = ((o - p) * nr).sum(dim=1, keepdim=True) < 0
    nr = torch.where(flip, -nr, nr)
    R = torch.tensor(_rot_z(yaw), device=device, dtype=torch.float64)
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = _rot_z(yaw)
    T[:3, 3] = np.asarray(origin, dtype=np.float64)
    return ((p - o) @ R).to(torch.float32).contiguous(), (nr @ R).to(torch.float32).contiguous(), T.astype(np.float32)


def _surface_samples(m, g, device):
    """m points sampled uniformly by area on the surfaces of the partial-overlap room: exact positions + outward normals (f64)."""
    W, D, Hh = _ROOM
    x0, x1, y0, y1 = _PART
    areas = [W * D, D * Hh, D * Hh, W * Hh, W * Hh] + [2 * math.pi * r * Hh for (_, _, r) in _CYL] + [(y1 - y0) * Hh, (y1 - y0) * Hh, (x1 - x0) * Hh]
    cum = torch.tensor(np.cumsum(areas) / np.sum(areas), device=device, dtype=torch.float64)
    prim = torch.bucketize(torch.rand(m, generator=g, device=device, dtype=torch.float64), cum).clamp_(max=len(areas) - 1)
    u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
    v = torch.rand(m, generator=g, device=device, dtype=torch.float64)
    p = torch.zeros(m, 3, device=device, dtype=torch.float64)
    nr = torch.zeros(m, 3, device=device, dtype=torch.float64)

    def put(mask, px, py, pz, nx, ny, nz):
        p[mask, 0] = px; p[mask, 1] = py; p[mask, 2] = pz
        nr[mask, 0] = nx; nr[mask, 1] = ny; nr[mask, 2] = nz

    k = prim == 0; put(k, u[k] * W, v[k] * D, 0.0, 0.0, 0.0, 1.0)
    k = prim == 1; put(k, 0.0, u[k] * D, v[k] * Hh, 1.0, 0.0, 0.0)
    k = prim == 2; put(k, W, u[k] * D, v[k] * Hh, -1.0, 0.0, 0.0)
    k = prim == 3; put(k, u[k] * W, 0.0, v[k] * Hh, 0.0, 1.0, 0.0)
    k = prim == 4; put(k, u[k] * W, D, v[k] * Hh, 0.0, -1.0, 0.0)
    for ci, (cx, cy, r) in enumerate(_CYL):
        k = prim == 5 + ci
        ang = u[k] * (2 * math.pi)
        put(k, cx + r * torch.cos(ang), cy + r * torch.sin(ang), v[k] * Hh, torch.cos(ang), torch.sin(ang), 0.0)
    k = prim == 8; put(k, x0, y0 + u[k] * (y1 - y0), v[k] * Hh, -1.0, 0.0, 0.0)
    k = prim == 9; put(k, x1, y0 + u[k] * (y1 - y0), v[k] * Hh, 1.0, 0.0, 0.0)
    k = prim == 10; put(k, x0 + u[k] * (x1 - x0), y1, v[k] * Hh, 0.0, 1.0, 0.0)
    # the floor under the partition is not a visible surface
    under = (prim == 0) & (p[:, 0] > x0) & (p[:, 0] < x1) & (p[:, 1] < y1)
    return p, nr, ~under


def _make_scan_partial(n, origin, yaw, seed, sigma, device):
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    x0, x1, y0, y1 = _PART
    if x0 - 0.3 < origin[0] < x1 + 0.3 and origin[1] < y1 + 0.3:      # a scanner inside the partition: step aside
        origin = (origin[0] + 0.6, origin[1], origin[2])
    ps, ns, have = [], [], 0
    chunk = max(4096, int(1.6 * n))
    while have < n:                                        # deterministic: the chunk size is a function of n only
        p, nr, ok = _surface_samples(chunk, g, device)
        keep = ok & visible_from(p, origin, _CYL, _PART, PARTIAL_MAX_RANGE)
        ps.append(p[keep]); ns.append(nr[keep]); have += int(keep.sum())
    p = torch.cat(ps)[:n]; nr = torch.cat(ns)[:n]
    o = torch.tensor(origin, device=device, dtype=torch.float64)
    ray = p - o
    ray = ray / ray.norm(dim=1, keepdim=True).clamp_(min=1e-9)
    p = p + torch.randn(n, 1, generator=g, device=device, dtype=torch.float64) * sigma * ray
    flip = ((o - p) * nr).sum(dim=1, keepdim=True) < 0
    nr = torch.where(flip, -nr, nr)
    R = torch.tensor(_rot_z(yaw), device=device, dtype=torch.float64)
    xyz_local = ((p - o) @ R).to(torch.float32).contiguous()
    nrm_local = (nr @ R).to(torch.float32).contiguous()
    T = np.eye(4, dtype=np.float64)
    T[:3, :3] = _rot_z(yaw)
    T[:3, 3] = np.asarray(origin, dtype=np.float64)
    return xyz_local, nrm_local, T.astype(np.float32)


SCAN_POSES = [((4.0, 5.0, 1.5), 0.3), ((6.0, 5.5, 1.4), -0.4), ((2.5, 2.5, 1.6), 1.1), ((7.5, 7.5, 1.3), 2.0),
              ((5.0, 2.0, 1.5), -1.3), ((2.0, 7.0, 1.45), 0.8), ((8.0, 3.0, 1.55), -2.2), ((5.0, 8.0, 1.35), 2.9),
              ((3.5, 6.0, 1.5), 0.1), ((6.5, 2.5, 1.4), -0.9), ((1.5, 4.5, 1.6), 1.7), ((8.5, 6.0, 1.3), -1.9),
              ((4.5, 3.5, 1.5), 2.4), ((5.5, 6.5, 1.45), -2.8), ((7.0, 8.5, 1.55), 0.6), ((3.0, 8.5, 1.35), -0.2)]


def perturbation(index, angle_scale=1.0, scale=1.0):
    """Initial misalignment of scan `index`: 1 degree about (1,1,1)/sqrt(3) and (2,-1,1) cm for scan 1,
    smaller seeded variations of the same size for further scans; scan 0 is unperturbed.  angle_scale shrinks the rotation
    (a stretched room keeps the same displacement at its walls); scale multiplies rotation angle and translation alike (a
    worse initial alignment: more outer iterations before the run converges)."""
    if index == 0:
        return np.eye(4, dtype=np.float64)
    rs = np.random.RandomState(1234 + index)
    axis = np.array([1.0, 1.0, 1.0]) if index == 1 else rs.normal(size=3)
    ang = math.radians(1.0) if index == 1 else math.radians(rs.uniform(0.5, 1.0))
    t = np.array([0.02, -0.01, 0.01]) if index == 1 else rs.uniform(-0.02, 0.02, size=3)
    P = np.eye(4)
    P[:3, :3] = rot_axis_angle(axis, ang * angle_scale * scale)
    P[:3, 3] = t * scale
    return P


def make_scene(n_scans, n_points, seed=1234, sigma=0.002, device="cpu", room_scale=1.0, partial=False, perturb=1.0, scanner=False):
    """List of dicts {xyz, normals, T_true, T_init} (T_init = perturbation * T_true applied about the scan origin).
    partial=True: the partial-overlap room (partition wall, occlusion, maximum range; room_scale must be 1).
    perturb: scale of the initial misalignment (perturbation()).
    scanner=True: every scan as a scanner records it (make_scan_angular: density ~ cos / range^2, points in scan order)."""
    scans = []
    for i in range(n_scans):
        origin, yaw = SCAN_POSES[i % len(SCAN_POSES)]
        if scanner:
            xyz, nrm, T = make_scan_angular(n_points, origin, yaw, seed + 17 * i, sigma, device, scan_order=True)
        else:
            xyz, nrm, T = make_scan(n_points, origin, yaw, seed + 17 * i, sigma, device, room_scale, partial)
        P = perturbation(i, 1.0 / room_scale, perturb)
        Ti = T.astype(np.float64).copy()
        Ti[:3, :3] = P[:3, :3] @ Ti[:3, :3]
        Ti[:3, 3] = Ti[:3, 3] + P[:3, 3]
        scans.append({"xyz": xyz, "normals": nrm, "T_true": T, "T_init": Ti.astype(np.float32)})
    return scans


# ---- path (B): synthetic ImageRegistrator workload -------------------------------------------------------------------------
def image_pyramid_u8(img, n_levels):
    """Image::BuildImagePyramid (src/opt/image.cc:106-131): successive half-size INTER_AREA reductions of a u8 image =
    2x2 box mean, round half up; odd trailing rows / columns are dropped."""
    out = [np.ascontiguousarray(img, np.uint8)]
    for _ in range(1, n_levels):
        a = out[-1]
        h, w = (a.shape[0] // 2) * 2, (a.shape[1] // 2) * 2
        a = a[:h, :w].astype(np.uint16)
        out.append(((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    return out


REG_DISTORTION = {0: [], 1: [-0.101082, 0.0703954, 0.000438661, -0.000680887],
                  2: [0.0221184, 0.0128597, 0.000531602, -0.000388873, 0.00623079, 0.0020419, -0.000805024, 4.07704e-05]}


def _synthetic_image(i, width, height, device=None):
    """The band-limited test texture of image i as u8 [height, width]; on `device` with torch when given (a 24 MP image takes
    ~2 s in numpy and ~20 ms on the GPU), bit-identical values are not needed between the two (synthetic input)."""
    if device is None:
        yy, xx = np.mgrid[0:height, 0:width]
        return (120 + 60 * np.sin((xx + 37 * i) / 11.0) * np.cos(yy / 9.0) + 40 * np.sin((xx + 2 * yy) / 31.0)).clip(0, 250).astype(np.uint8)
    yy = torch.arange(height, device=device, dtype=torch.float32)[:, None]
    xx = torch.arange(width, device=device, dtype=torch.float32)[None, :]
    img = 120 + 60 * torch.sin((xx + 37 * i) / 11.0) * torch.cos(yy / 9.0) + 40 * torch.sin((xx + 2 * yy) / 31.0)
    return img.clamp_(0, 250).to(torch.uint8)


def _pyramid_torch(img, n_levels):
    """image_pyramid_u8 on a torch u8 tensor (same integer arithmetic); returns numpy levels."""
    out = [img]
    for _ in range(1, n_levels):
        a = out[-1]
        h, w = (a.shape[0] // 2) * 2, (a.shape[1] // 2) * 2
        a = a[:h, :w].to(torch.int32)
        out.append(((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).to(torch.uint8))
    return [l.cpu().numpy() for l in out]


def make_reg_workload(n_points=4_000_000, width=3840, height=2160, n_levels=6, K=5, n_images=4, model=0, seed=0, device=None):
    """A textured wall (y = 3) sampled on a jittered lattice with lattice neighbours, seen by `n_images` cameras placed on an arc;
    one point scale whose radius makes every observation land between pyramid levels 0 and 1.  Shapes follow BASELINE.json
    configs[4] (4K images, 6 levels); everything else (texture, poses) is synthetic."""
    rng = np.random.RandomState(seed)
    side = int(np.sqrt(n_points)); n = side * side
    uu, vv = np.meshgrid(np.linspace(-1.6, 1.6, side), np.linspace(-0.9, 0.9, side), indexing="ij")
    uu = uu + rng.uniform(-0.2, 0.2, uu.shape) * (3.2 / side); vv = vv + rng.uniform(-0.2, 0.2, vv.shape) * (1.8 / side)
    pts = np.stack([uu.ravel(), np.full(n, 3.0), vv.ravel()], 1).astype(np.float32)
    idx = np.arange(n).reshape(side, side)

    def sh(dx, dy):
        return np.roll(np.roll(idx, dx, 0), dy, 1).ravel()
    nbr = np.stack([sh(1, 0), sh(-1, 0), sh(0, 1), sh(0, -1), sh(1, 1), sh(-1, -1), sh(1, -1), sh(-1, 1)][:K], 1).astype(np.uint32)
    tex = 120 + 55 * np.sin(7.0 * pts[:, 0]) * np.cos(5.0 * pts[:, 2]) + 35 * np.sin(3.0 * pts[:, 0] + 4.0 * pts[:, 2])
    fixed = (tex[nbr] - tex[:, None]).astype(np.float32)
    params = np.array([0.55 * width, 0.55 * width, width / 2 - 0.5, height / 2 - 0.5] + REG_DISTORTION[model], np.float32)
    images = []
    for i in range(n_images):
        img = _synthetic_image(i, width, height, device)
        a = min(0.04, 1.0 / max(n_images, 1)) * (i - 0.5 * (n_images - 1))       # the arc spans at most +- 0.5 rad whatever the image count
        eye = np.array([3.0 * np.sin(a), 3.0 - 3.0 * np.cos(a), min(0.01, 0.3 / max(n_images, 1)) * i])    # (the cameras rise at most 0.3 m)
        z = np.array([0.0, 3.0, 0.0]) - eye; z /= np.linalg.norm(z)
        x = np.cross(z, [0.0, 0.0, 1.0]); x /= np.linalg.norm(x)
        y = np.cross(z, x)
        R = np.stack([x, y, z])
        t = -R @ eye
        w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
        q = np.array([w, (R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w)])
        pyr = image_pyramid_u8(img, n_levels) if device is None else _pyramid_torch(img, n_levels)
        images.append(dict(pyr=pyr, q=(q / np.linalg.norm(q)).astype(np.float32), t=t.astype(np.float32)))
    return dict(pts=pts, nbr=nbr, K=K, fixed_desc=fixed, params=params, width=width, height=height, n_levels=n_levels,
                images=images, point_radius=float(3.2 / side * 0.7), model=model)
