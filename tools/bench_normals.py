"""(A') NormalEstimator measurement (normals/s; algorithmic 12 k + 28 B per point): one synthetic room scan resident
in HBM, e3d_normals_knn end to end (grid levels, sort, table, search + covariance + eigenvector, outputs left on the device), and
the CPU restatement (all host cores: the reference's NormalEstimationTwoPassOMP is OpenMP over points) on a same-density slab.

    python tools/bench_normals.py [--points 20000000] [--k 32] [--cpu-points 300000]
Prints one JSON line.  Not a bench.py line (bench.py keeps BASELINE.json's metric); numbers go to DESIGN.md section 5."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20_000_000)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--cpu-points", type=int, default=1_000_000)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--angular", action="store_true", help="sample the room as a scanner does (uniform in angle: density ~ cos / range^2) instead of uniformly per area")
    a = ap.parse_args()
    e3d = importlib.import_module("dataset-pipeline_amd")
    synth = importlib.import_module("dataset-pipeline_amd.synth")
    capi = importlib.import_module("dataset-pipeline_amd.capi")
    dev = torch.device("cuda:0")
    origin, yaw = synth.SCAN_POSES[0]
    xyz, _, _ = (synth.make_scan_angular if a.angular else synth.make_scan)(a.points, origin, yaw, 1234, device=dev)
    xyz = xyz.contiguous()
    n = int(xyz.shape[0])
    on = torch.empty((n, 3), dtype=torch.float32, device=dev)
    oc = torch.empty(n, dtype=torch.float32, device=dev)
    vp = np.zeros(3, np.float32)
    torch.cuda.synchronize()

    def call():
        r = capi.lib().e3d_normals_knn(C.c_void_p(xyz.data_ptr()), n, a.k, C.c_void_p(vp.ctypes.data), C.c_void_p(on.data_ptr()),
                                       C.c_void_p(oc.data_ptr()), None)
        assert r == 0, capi.lib().e3d_last_error()
    call()                                                   # cold: first launches
    t0 = time.perf_counter()
    for _ in range(a.repeat):
        call()
    dt = (time.perf_counter() - t0) / a.repeat
    finite = float(torch.isfinite(on).all(dim=1).float().mean())
    unit = float(((on * on).sum(1).sqrt() - 1).abs()[torch.isfinite(on).all(dim=1)].max())
    out = {"metric": "normals/s", "value": n / dt, "points": n, "k": a.k, "ms_per_call": dt * 1e3, "sampling": "angular (scanner)" if a.angular else "uniform per area",
           "algorithmic_bytes_per_point": 12 * a.k + 28, "algorithmic_GBs": n * (12 * a.k + 28) / dt / 1e9,
           "finite_fraction": finite, "max_abs_norm_minus_1": unit}
    if not a.no_cpu:
        from oracle import binding as ob
        # same density: a slab of the scan holding about cpu-points points
        x = xyz[:, 0]
        xs = torch.sort(x[torch.randperm(n, device=dev)[:min(n, 2_000_000)]]).values
        i0 = int(0.4 * len(xs))
        lo, hi = float(xs[i0]), float(xs[min(len(xs) - 1, i0 + max(1, int(len(xs) * a.cpu_points / n)))])
        sub = xyz[(x >= lo) & (x < hi)].cpu().numpy()
        t0 = time.perf_counter()
        cn, cc = ob.normals(sub, k=a.k)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sub) / dtc, "unit": "normals/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "%d points (slab x in [%.2f, %.2f) of the same scan), kd-tree build + k-search + two-pass "
                                         "covariance, OpenMP over points, %.1f s" % (len(sub), lo, hi, dtc)}
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
