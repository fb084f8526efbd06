"""Per-iteration NN-phase time of the row kernels (mode 3 = k_nn_rows, 4 = k_nn_mfma, the MFMA-filtered variant) over a whole ICP run.
usage: python tools/nn_trend.py [points_per_scan] [iterations]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
scans = synth.make_scene(2, n, seed=1234, sigma=0.002, device=dev)
for mode in (3, 4):
    e3d.lib().e3d_set_nn_mode(mode)
    icp = e3d.PointToPlaneICP(device=0)
    for s in scans:
        icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
    for it in range(iters):
        icp.run(0.01, it, 1, 1e-10, False)
    r = icp.iter_records()
    print("mode", mode, "nn_query_ms:", " ".join("%.1f" % x["t_nn_query_ms"] for x in r))
    print("mode", mode, "corr (M):   ", " ".join("%.1f" % (x["correspondences"] / 1e6) for x in r))
    del icp
