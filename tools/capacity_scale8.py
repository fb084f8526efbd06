"""Capacity check for the 8-GPU weak-scaling workload on ONE GPU: the clouds every rank holds at N = 8 (2 x 400 M points on the
stretched room), grids, one outer iteration over a 1/8 slice of the queries (rank 0 of a world of 8 with a no-op all-reduce)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
t0 = time.time()
scans = synth.make_scene(2, 50_000_000 * N, seed=1234, sigma=0.002, device=dev, room_scale=float(np.sqrt(N)))
torch.cuda.synchronize()
print("generated in %.1f s, torch peak %.1f GB" % (time.time() - t0, torch.cuda.max_memory_allocated() / 1e9), flush=True)
icp = e3d.PointToPlaneICP(device=0)
for s in scans:
    icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
del scans
torch.cuda.empty_cache()
icp.set_shard(0, N, lambda arr: None)
for it in range(3):
    t0 = time.time()
    icp.run(0.01, it, 1, 1e-10, False)
    r = icp.iter_records()[-1]
    free, total = torch.cuda.mem_get_info()
    print("iteration %d: %.1f ms wall, transform %.2f nn %.2f (kernels %.2f) lm %.2f ms, local corr %d of %d queries, HBM in use %.1f GB" %
          (it, (time.time() - t0) * 1e3, r["t_transform_ms"], r["t_nn_ms"], r["t_nn_query_ms"], r["t_lm_ms"], r["correspondences"], r["queries"],
           (total - free) / 1e9), flush=True)
