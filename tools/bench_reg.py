"""Throughput probe of the path-(B) kernels on one (image, point scale): residuals/s of the accumulate pass, cost pass,
observation refresh.  usage: python tools/bench_reg.py [n_points] [width] [height]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from reg_util import pyramid_u8, look_at_pose, quat_from_R, texture
e3d = importlib.import_module("dataset-pipeline_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 3840
H = int(sys.argv[3]) if len(sys.argv) > 3 else 2160
K, L = 5, 6
rng = np.random.RandomState(0)
side = int(np.sqrt(n)); n = side * side
uu, vv = np.meshgrid(np.linspace(-1.6, 1.6, side), np.linspace(-0.9, 0.9, side), indexing="ij")
uu = uu + rng.uniform(-0.2, 0.2, uu.shape) * (3.2 / side); vv = vv + rng.uniform(-0.2, 0.2, vv.shape) * (1.8 / side)
pts = np.stack([uu.ravel(), np.full(n, 3.0), vv.ravel()], 1).astype(np.float32)
idx = np.arange(n).reshape(side, side)
def sh(dx, dy): return np.roll(np.roll(idx, dx, 0), dy, 1).ravel()
nbr = np.stack([sh(1, 0), sh(-1, 0), sh(0, 1), sh(0, -1), sh(1, 1)], 1).astype(np.uint32)   # lattice neighbours (cheap to build)
tex = texture(pts[:, 0].astype(np.float64), pts[:, 2].astype(np.float64))
fixed = (tex[nbr] - tex[:, None]).astype(np.float32)
params = np.array([0.55 * W, 0.55 * W, W / 2 - 0.5, H / 2 - 0.5], np.float32)
R0, t0 = look_at_pose((0.0, 0.0, 0.0), (0, 3, 0)); q = quat_from_R(R0)
yy, xx = np.mgrid[0:H, 0:W]
img = (120 + 60 * np.sin(xx / 11.0) * np.cos(yy / 9.0) + 40 * np.sin((xx + 2 * yy) / 31.0)).clip(0, 250).astype(np.uint8)
pyr = pyramid_u8(img, L)
P = e3d.RegProblem(e3d.default_reg_params(image_scale_count=L, point_neighbor_count=K))
P.set_intrinsics(0, W, H, params, 0, L); P.set_image(0, 0, pyr); P.set_image_pose(0, q, t0)
P.set_point_scale(0, pts, 3.2 / side * 0.7, nbr, fixed); P.set_splat_points(pts)
def timed(f, reps=5):
    f(); t = time.perf_counter()
    for _ in range(reps): f()
    return (time.perf_counter() - t) / reps
t_depth = timed(lambda: P.render_depth(0, 0))
nobs = P.observe(0, 0, 0, 1)
t_obs = timed(lambda: P.observe(0, 0, 0, 1))
_, _, sums, counts = P.accumulate(0, 0)
t_acc = timed(lambda: P.accumulate(0, 0))
t_cost = timed(lambda: P.cost(0, 0))
res = int(counts[0] + counts[1])
print("points %d image %dx%d observations %d residuals %d" % (n, W, H, nobs, res))
print("render_depth %.3f ms | observe %.3f ms (%.1f M pts/s) | accumulate (pass1+pass2) %.3f ms (%.1f M residuals/s) | cost %.3f ms (%.1f M residuals/s)" %
      (t_depth * 1e3, t_obs * 1e3, n / t_obs / 1e6, t_acc * 1e3, res / t_acc / 1e6, t_cost * 1e3, res / t_cost / 1e6))
# f4: GroundTruthCreator visibility counting / ground-truth depth of the same points as "scan" (incl. the occlusion rendering)
P.set_scan_points(pts)
t_cnt = timed(lambda: P.count_scan_observations(0))
t_gt = timed(lambda: P.ground_truth_depth(0, W, H, min_count=1))
print("scan visibility count %.3f ms (%.1f M scan points/s incl. occlusion depth) | ground-truth depth + read-back %.3f ms" %
      (t_cnt * 1e3, n / t_cnt / 1e6, t_gt * 1e3))
