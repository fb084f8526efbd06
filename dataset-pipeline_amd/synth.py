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


def make_scan(n, origin, yaw, seed, sigma=0.002, device="cpu"):
    """Returns (xyz_local[n,3] f32, normals_local[n,3] f32, T_true[4,4] f32 numpy)."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    W, D, Hh = _ROOM
    areas = [W * D, D * Hh, D * Hh, W * Hh, W * Hh] + [2 * math.pi * r * Hh for (_, _, r) in _CYL]
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
    for ci, (cx, cy, r) in enumerate(_CYL):
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


SCAN_POSES = [((4.0, 5.0, 1.5), 0.3), ((6.0, 5.5, 1.4), -0.4), ((2.5, 2.5, 1.6), 1.1), ((7.5, 7.5, 1.3), 2.0),
              ((5.0, 2.0, 1.5), -1.3), ((2.0, 7.0, 1.45), 0.8), ((8.0, 3.0, 1.55), -2.2), ((5.0, 8.0, 1.35), 2.9),
              ((3.5, 6.0, 1.5), 0.1), ((6.5, 2.5, 1.4), -0.9), ((1.5, 4.5, 1.6), 1.7), ((8.5, 6.0, 1.3), -1.9),
              ((4.5, 3.5, 1.5), 2.4), ((5.5, 6.5, 1.45), -2.8), ((7.0, 8.5, 1.55), 0.6), ((3.0, 8.5, 1.35), -0.2)]


def perturbation(index):
    """Initial misalignment of scan `index`: 1 degree about (1,1,1)/sqrt(3) and (2,-1,1) cm for scan 1,
    smaller seeded variations of the same size for further scans; scan 0 is unperturbed."""
    if index == 0:
        return np.eye(4, dtype=np.float64)
    rs = np.random.RandomState(1234 + index)
    axis = np.array([1.0, 1.0, 1.0]) if index == 1 else rs.normal(size=3)
    ang = math.radians(1.0) if index == 1 else math.radians(rs.uniform(0.5, 1.0))
    t = np.array([0.02, -0.01, 0.01]) if index == 1 else rs.uniform(-0.02, 0.02, size=3)
    P = np.eye(4)
    P[:3, :3] = rot_axis_angle(axis, ang)
    P[:3, 3] = t
    return P


def make_scene(n_scans, n_points, seed=1234, sigma=0.002, device="cpu"):
    """List of dicts {xyz, normals, T_true, T_init} (T_init = perturbation * T_true applied about the scan origin)."""
    scans = []
    for i in range(n_scans):
        origin, yaw = SCAN_POSES[i % len(SCAN_POSES)]
        xyz, nrm, T = make_scan(n_points, origin, yaw, seed + 17 * i, sigma, device)
        P = perturbation(i)
        Ti = T.astype(np.float64).copy()
        Ti[:3, :3] = P[:3, :3] @ Ti[:3, :3]
        Ti[:3, 3] = Ti[:3, 3] + P[:3, 3]
        scans.append({"xyz": xyz, "normals": nrm, "T_true": T, "T_init": Ti.astype(np.float32)})
    return scans
