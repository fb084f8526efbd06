"""Profiling helper (not part of the product): a few ICP iterations on a synthetic scene with a forced NN kernel.
usage: python tools/prof_nn.py <points_per_scan> <nn_mode> [iters]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
n = int(sys.argv[1]); mode = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
e3d.lib().e3d_init(0)
e3d.lib().e3d_set_nn_mode(mode)
scans = synth.make_scene(2, n, seed=1234, device="cuda")
torch.cuda.synchronize()
icp = e3d.PointToPlaneICP(device=0)
for s in scans:
    icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
icp.set_max_inner_iterations(int(os.environ.get("E3D_PROF_INNER", "2")))
t0 = time.time()
for it in range(iters):
    icp.run(0.01, it, 1, 1e-10, False)
r = icp.iter_records()
print("mode", mode, "n", n, "nn_query_ms per iter", [round(x["t_nn_query_ms"], 2) for x in r], "corr", [x["correspondences"] for x in r])
