"""Throughput of MergeClosePoints (e3d_merge_close_points) on scan-line ordered points.  usage: python tools/bench_merge.py [n]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
e3d = importlib.import_module("dataset-pipeline_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
side = int(np.sqrt(n)); n = side * side
rng = np.random.RandomState(0)
sp = 0.002
u, v = np.meshgrid(np.arange(side, dtype=np.float32) * sp, np.arange(side, dtype=np.float32) * sp, indexing="ij")
P = np.stack([u.ravel(), v.ravel(), np.zeros(n, np.float32)], 1) + rng.normal(0, sp * 0.15, (n, 3)).astype(np.float32)
P = P.astype(np.float32)
col = rng.uniform(0, 255, n).astype(np.float32); sidx = (rng.uniform(0, 1, n) < 0.5).astype(np.uint8); mxr = np.full(n, 0.1, np.float32)
for dist in (0.003, 0.008):
    e3d.merge_close_points(dist, 2, P[:1000], col[:1000], sidx[:1000], mxr[:1000])
    t = time.perf_counter()
    out = e3d.merge_close_points(dist, 2, P, col, sidx, mxr)
    dt = time.perf_counter() - t
    print("n %d merge_distance %.3f -> %d points in %.3f s (%.1f M points/s incl. host copies)" % (n, dist, len(out[0]), dt, n / dt / 1e6))
