"""BASELINE.json configs[2] shape: N scans x M points, all movable, all directed pairs.  usage: python tools/bench_c3.py [scans] [points] [iters] [d]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000_000
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = float(sys.argv[4]) if len(sys.argv) > 4 else 0.01
dev = torch.device("cuda", 0)
scans = synth.make_scene(ns, n, seed=1234, sigma=0.002, device=dev)
icp = e3d.PointToPlaneICP(device=0)
for s in scans:
    icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
del scans
torch.cuda.empty_cache()
icp.run(d, 0, 1, 1e-10, False)
icp.clear_records()
torch.cuda.synchronize(); t = time.perf_counter()
for it in range(1, 1 + iters):
    icp.run(d, it, 1, 1e-10, False)
torch.cuda.synchronize(); dt = time.perf_counter() - t
r = icp.iter_records()
corr = sum(x["correspondences"] for x in r); q = sum(x["queries"] for x in r)
print("per iteration: nn_ms", [round(x["t_nn_ms"], 1) for x in r], "lm_ms", [round(x["t_lm_ms"], 1) for x in r], "corr", [x["correspondences"] for x in r])
print("c3: %d scans x %d points, %d directed pairs: %.1f ms/iteration, %.3g correspondences/s, %.3g NN queries/s, LM %.1f ms/iter (%.1f passes), HBM in use %.1f GB"
      % (ns, n, ns * (ns - 1), dt / iters * 1e3, corr / dt, q / dt, sum(x["t_lm_ms"] for x in r) / iters,
         sum(x["full_passes"] + x["cost_passes"] + x["multi_cost_passes"] for x in r) / iters, torch.cuda.mem_get_info()[1] / 1e9 - torch.cuda.mem_get_info()[0] / 1e9))
