#!/usr/bin/env python
"""bench.py -- ICPScanAligner hot path on MI355X (BASELINE.json metric: ICP correspondences/sec + ms/iter).

    python bench.py --gpus N --steps K --warmup W

Workload (N=1): BASELINE.json configs[1] -- ICPScanAligner on 2 scans of ~50 M points each,
`-d 0.01 --max_iterations 100` -- on seeded synthetic scans of that shape generated directly in HBM (no
network / no terrace data here).  A "step" is one outer ICP iteration, i.e. one
PointToPlaneICP::Run(d, it, /*max_num_iterations*/1, thr) exactly as the tool's loop calls it
(src/exe/icp_scan_aligner.cc:342-343): transform + bbox, exact 1-NN correspondence search for both directed
pairs, and the full inner Levenberg-Marquardt solve (<= 150 iterations x (1 + <= 10 tries)).
Nothing is skipped or cached inside the timed region; inputs are resident in HBM before it starts.

N>1 (one process per GPU, torch.distributed / RCCL): weak scaling -- every scan has 50 M x N points, every rank
holds both scans and handles 1/N of each directed pair's queries and correspondences; the 6x6 normal
equations + cost are all-reduced once per LM pass.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_CORR_PASS = 56   # 2 x i32 + 4 x vec3 f32 per correspondence per pass
ALG_BYTES_PER_QUERY = 32       # 12 in + 8 out + 12 amortised target


def crop_world(scan, lo, hi):
    """Points of a scan whose true world x lies in [lo, hi) (numpy, host)."""
    import torch
    T = scan["T_true"]
    x = scan["xyz"]
    # elementwise on purpose: torch's gemv path returned wrong values for 50 M-row operands on this ROCm build
    wx = x[:, 0] * float(T[0, 0]) + x[:, 1] * float(T[0, 1]) + x[:, 2] * float(T[0, 2]) + float(T[0, 3])
    m = (wx >= lo) & (wx < hi)
    return scan["xyz"][m].cpu().numpy(), scan["normals"][m].cpu().numpy()


def cpu_baseline(scans, d, thr, slab):
    """Reference-faithful CPU path (the oracle, "kind": "port") on a bounded sample of the same workload."""
    from oracle import binding as ob
    ob.lib()
    o = ob.OracleICP()
    n = []
    for s in scans:
        xyz, nrm = crop_world(s, slab[0], slab[1])
        n.append(int(xyz.shape[0]))
        o.add_point_cloud(xyz, nrm, s["T_init"], False)
    iters = 3
    t0 = time.perf_counter()
    for it in range(iters):
        o.run(d, it, 1, thr, False)
    dt = time.perf_counter() - t0
    recs = o.iter_records()
    corr = sum(r["correspondences"] for r in recs)
    # the same sample with the NN phase spread over every host core (queries are independent; results and their
    # order are unchanged); the inner LM stays on one thread as in the reference, so this is an upper bound for what more cores
    # buy the reference's structure, not a different algorithm
    o2 = ob.OracleICP()
    o2.set_all_core(True)
    for s in scans:
        xyz, nrm = crop_world(s, slab[0], slab[1])
        o2.add_point_cloud(xyz, nrm, s["T_init"], False)
    t0 = time.perf_counter()
    for it in range(iters):
        o2.run(d, it, 1, thr, False)
    dt2 = time.perf_counter() - t0
    recs2 = o2.iter_records()
    corr2 = sum(r["correspondences"] for r in recs2)
    all_core = {"value": corr2 / dt2, "unit": "correspondences/s", "cores": os.cpu_count(), "ms_per_iter": dt2 / iters * 1e3,
                "t_nn_s": sum(r["t_nn_s"] for r in recs2) / iters, "t_lm_s": sum(r["t_lm_s"] for r in recs2) / iters,
                "same_correspondences": bool(corr2 == corr),
                "note": "NN queries on all host cores (OpenMP), inner LM single-threaded; same slab and iterations"}
    return {
        "all_core": all_core,
        "value": corr / dt, "unit": "correspondences/s", "cores": 2, "kind": "port",
        "sample": "%d outer iterations on the world-x slab [%.2f, %.2f) m of both scans (%d + %d points, same density "
                  "and flags, %.1f s of CPU work); NN phase on 2 threads (one per directed pair, as icp_point_to_plane.cc:208), "
                  "inner LM single-threaded" % (iters, slab[0], slab[1], n[0], n[1], dt),
        "ms_per_iter": dt / iters * 1e3, "correspondences": int(corr / iters),
        "t_nn_s": sum(r["t_nn_s"] for r in recs) / iters, "t_lm_s": sum(r["t_lm_s"] for r in recs) / iters,
        "lm_passes": int(sum(r["accumulate_passes"] + r["cost_passes"] for r in recs) / iters), "host_cores_available": os.cpu_count(),
    }


def reg_traffic(model, n_images):
    """Measured HBM bytes of pass 1 + pass 2 for the leg's n_images launches (rocprofv3 PMC of this bench, per launch, from
    profiles/round1_traffic.json: same 4K images and 4 M points)."""
    path = os.path.join(ROOT, "profiles", "round1_traffic.json")
    if not os.path.exists(path):
        return None
    k = json.load(open(path))["kernels"]
    names = {0: ("k_reg_pass1<0, false>", "k_reg_pass2<8, 10, 0, 10, true>"), 2: ("k_reg_pass1<2, false>", "k_reg_pass2_mfma<5, 18>")}.get(model)
    if not names or any(n not in k for n in names):
        return None
    return n_images * sum(k[n]["hbm_bytes_per_launch"] for n in names)


def image_registrator_leg(e3d, synth, cpu=True):
    """Second BASELINE.json metric: ImageRegistrator residuals/s = (#fixed + #variable colour residuals) / wall time of the
    accumulate pass of IntrinsicsAndPoseOptimizer::Apply (src/opt/intrinsics_and_pose_optimizer.cc:87-92,624-837) over all
    images of one GPU; also ms per full RunOnCurrentScale iteration.  4 synthetic 3840x2160 images (6 pyramid levels,
    configs[4] shape), 4 M points, K = 5, for the 4-parameter PINHOLE and the 12-parameter THIN_PRISM_FISHEYE model."""
    out = {}
    for model, name in ((0, "PINHOLE"), (2, "THIN_PRISM_FISHEYE")):
        Wl = synth.make_reg_workload(n_points=4_000_000, n_images=4, model=model)
        P = e3d.RegProblem(e3d.default_reg_params(image_scale_count=Wl["n_levels"], point_neighbor_count=Wl["K"]))
        P.set_intrinsics(0, Wl["width"], Wl["height"], Wl["params"], 0, Wl["n_levels"], camera_type=model)
        P.set_point_scale(0, Wl["pts"], Wl["point_radius"], Wl["nbr"], Wl["fixed_desc"])
        P.set_splat_points(Wl["pts"])
        ids = list(range(len(Wl["images"])))
        for i, im in enumerate(Wl["images"]):
            P.set_image(i, 0, im["pyr"]); P.set_image_pose(i, im["q"], im["t"])
        P.update_observations(1)                                   # cold call: buffer growth, first launches
        t0 = time.perf_counter(); P.update_observations(1); t_obs = time.perf_counter() - t0
        # ObservationsCache path (the reference's mode after the first image scale): indexed re-projection, no depth rendering
        P.determine_observed_indices(); P.set_cache_observations(True); P.update_observations(1)
        t0 = time.perf_counter(); P.update_observations(1); t_obs_cached = time.perf_counter() - t0
        P.set_cache_observations(False); P.update_observations(1)
        t0 = time.perf_counter(); P.color_update(); t_col = time.perf_counter() - t0
        for i in ids:
            P.accumulate(i, 0)
        reps = 3
        t0 = time.perf_counter()
        res = 0
        for _ in range(reps):
            for i in ids:
                _, _, _, c = P.accumulate(i, 0)
                res += int(c[0] + c[1])
        t_acc = (time.perf_counter() - t0) / reps
        res //= reps
        t0 = time.perf_counter(); P.compute_cost(); t_cost = time.perf_counter() - t0
        t0 = time.perf_counter(); _, _, its = P.run_on_current_scale(2, 0.0, 15, False); t_run = time.perf_counter() - t0
        I = len(Wl["params"]); r4 = (I + 10) // 4; K = Wl["K"]
        obs = res // 2
        alg = obs * (16 * r4 + K * (4 + 16 * r4) + 8 * K + 9)          # own row + K x (row slot, neighbour row) + descriptors + idx/flag/count
        out[name] = {"residuals_per_s": res / t_acc, "accumulate_ms": t_acc * 1e3, "images": len(ids), "points": len(Wl["pts"]),
                     "residuals": res, "unknowns_per_image_block": I + 6,
                     "observation_refresh_ms": t_obs * 1e3, "cached_observation_refresh_ms": t_obs_cached * 1e3, "colour_update_ms": t_col * 1e3, "cost_ms": t_cost * 1e3,
                     "ms_per_run_iteration": t_run / max(its, 1) * 1e3,
                     "roofline": {"bound": "hbm", "achieved": alg / t_acc / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": alg / t_acc / 1e9 / HBM_PEAK_GBS, "traffic": reg_traffic(model, len(ids)),
                                  "kernel": "k_reg_pass1 + %s (+ reduce, read-back) of e3d_reg_accumulate" % ("k_reg_pass2" if I + 6 == 10 else "k_reg_pass2_mfma (v_mfma_f64_16x16x4_f64)"),
                                  "algorithmic_bytes": alg}}
        del P
    if cpu:
        from oracle import reg_binding as rb
        from oracle.reg_driver import OracleRegProblem
        Wl = synth.make_reg_workload(n_points=4_000_000, n_images=1, model=0)           # one image of the GPU workload
        O = OracleRegProblem(K=Wl["K"], image_scale_count=Wl["n_levels"])
        O.set_intrinsics(0, Wl["width"], Wl["height"], Wl["params"], 0, Wl["n_levels"])
        O.set_point_scale(0, Wl["pts"], Wl["point_radius"], Wl["nbr"], Wl["fixed_desc"])
        O.set_splat_points(Wl["pts"])
        O.set_image(0, 0, Wl["images"][0]["pyr"]); O.set_image_pose(0, Wl["images"][0]["q"], Wl["images"][0]["t"])
        O.update_observations(1); O.color_update()
        S = O.scales[0]; im = O.images[0]; I0 = O.intr[0]; o = O.obs[(0, 0)]
        reps = 8
        t0 = time.perf_counter()
        for _ in range(reps):
            _, _, _, c = rb.accumulate(S["pts"], float(S["radius"]), S["nbr"], O.K, S["fixed"], S["var"], S["counts"], I0["levels"][0], I0["min"],
                                       im["pyr"], O._R(im), im["t"], o[:4], o[4], O.robust_type, O.robust_param, O.fixed_weight, O.var_weight)
        tc = (time.perf_counter() - t0) / reps
        out["cpu_baseline"] = {"value": float(c[0] + c[1]) / tc, "unit": "residuals/s", "cores": 1, "kind": "port",
                               "sample": "%d accumulate passes (oracle_reg_accumulate, single thread like the reference) of one 3840x2160 "
                                         "PINHOLE image, 4 M points, K = 5: %d residuals in %.2f s each" % (reps, int(c[0] + c[1]), tc)}
        out["speedup_vs_cpu"] = out["PINHOLE"]["residuals_per_s"] / out["cpu_baseline"]["value"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--points", type=int, default=0, help="points per scan (default 50 M x gpus)")
    ap.add_argument("--distance", type=float, default=0.01)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reg", action="store_true", help="skip the ImageRegistrator leg (N=1 only)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    # E3D_BENCH_SHARE_GPU=1: all ranks on GPU 0 with the gloo backend -- only for smoke-testing the multi-rank code
    # path on a 1-GPU box; real runs use one GPU per rank and RCCL ("nccl").
    share_gpu = os.environ.get("E3D_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    e3d = importlib.import_module("dataset-pipeline_amd")
    synth = importlib.import_module("dataset-pipeline_amd.synth")
    if e3d.lib().e3d_init(local_rank) < 1:
        raise SystemExit("libe3dhip: no device")

    n_points = args.points if args.points > 0 else 50_000_000 * world
    d = float(args.distance)
    thr = 1e-10     # README.md:101 recommended flags; never converges within the bench's few iterations

    # weak scaling: N times the points on N times the floor area (the room stretched by sqrt(N) in x and y), i.e. the point density
    # -- and with it the candidates per query -- of the 1-GPU workload; every rank then searches its 1/N of the queries
    room_scale = float(np.sqrt(n_points / 50_000_000.0)) if (world > 1 and args.points == 0) else 1.0
    scans = synth.make_scene(2, n_points, seed=1234, sigma=0.002, device=dev, room_scale=room_scale)
    torch.cuda.synchronize()
    icp = e3d.PointToPlaneICP(device=local_rank)
    for s in scans:
        icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)

    base = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # slab width chosen for ~4 M points per scan at this density (floor + two walls = 16 m^2 per metre of x): about
        # 10 s of CPU work for the 3 timed iterations (kd-tree builds + searches + inner LM)
        width = min(10.0, 4.0e6 / (n_points / 242.6 * 16.0))
        base = cpu_baseline(scans, d, thr, (4.0, 4.0 + width))
    for s in scans:
        del s["xyz"], s["normals"]
    torch.cuda.empty_cache()

    if world > 1:
        importlib.import_module("dataset-pipeline_amd.dist").attach(icp, device=dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        icp.run(d, it, 1, thr, False)
    warm_nn_ms = [r["t_nn_query_ms"] for r in icp.iter_records()]      # reported so that the rocprof per-launch average can be reconciled
    icp.clear_records()
    barrier()
    t0 = time.perf_counter()
    for it in range(args.warmup, args.warmup + args.steps):
        icp.run(d, it, 1, thr, False)
    barrier()
    dt = time.perf_counter() - t0
    recs = icp.iter_records()
    local = np.array([
        dt,
        sum(r["correspondences"] for r in recs), sum(r["queries"] for r in recs),
        sum(r["t_lm_kernel_ms"] for r in recs), sum(r["t_nn_query_ms"] for r in recs),
        sum(r["full_passes"] + r["cost_passes"] + r["multi_cost_passes"] for r in recs),
        sum(r["t_transform_ms"] for r in recs), sum(r["t_nn_ms"] for r in recs), sum(r["t_lm_ms"] for r in recs),
    ], dtype=np.float64)
    if world > 1:
        cdev = torch.device("cpu") if share_gpu else dev
        tmax = torch.tensor([local[0]], device=cdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = torch.from_numpy(local).to(cdev)
        dist.all_reduce(tsum)
        tot = tsum.cpu().numpy()
        dt = float(tmax.item())
    else:
        tot = local
    if rank == 0:
        reg_leg = None
        if world == 1 and not args.no_reg:
            del icp
            torch.cuda.empty_cache()
            reg_leg = image_registrator_leg(e3d, synth, cpu=not args.no_cpu_baseline)
        K = args.steps
        corr, queries = tot[1], tot[2]
        lm_ms, nn_ms, passes = tot[3] / world, tot[4] / world, tot[5] / world
        n_nn_launch = 2 * K
        # dominant kernel = the one with the larger summed duration in the timed region (per rank)
        if lm_ms >= nn_ms:
            per_launch_bytes = ALG_BYTES_PER_CORR_PASS * (corr / world / K)     # local correspondences of one iteration
            avg_ms = lm_ms / max(passes, 1)
            kernel = "k_lm_pass (fused cost + Gramian pass, a7/a8)"
            kernel_key = "k_lm_pass<1>"
        else:
            per_launch_bytes = ALG_BYTES_PER_QUERY * (queries / world / n_nn_launch)
            avg_ms = nn_ms / n_nn_launch
            kernel = "k_nn_rows (exact 1-NN within radius over LDS-staged cell rows, a5)"
            kernel_key = "k_nn_rows"
        achieved = per_launch_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM traffic per launch of that kernel: rocprofv3 PMC (FETCH_SIZE / WRITE_SIZE, separate passes, gfx950
        # corrections applied) of this same workload, committed under profiles/ (bench.py cannot run the profiler itself)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "round1_traffic.json")
        if world == 1 and n_points == 50_000_000 and os.path.exists(tpath):
            tk = json.load(open(tpath))["kernels"].get(kernel_key)
            if tk:
                traffic, traffic_src = tk["hbm_bytes_per_launch"], "profiles/round1_traffic.json"
        out = {
            "metric": "ICP correspondences/sec", "value": corr / dt, "unit": "correspondences/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (f64 accumulation)",
            "data": "synthetic",
            "config": {"workload": "ICPScanAligner 2 scans (BASELINE.json configs[1]), -d %g, one outer iteration per step"
                                   % d,
                       "points_per_scan": n_points, "scans": 2, "directed_pairs": 2, "room_scale": room_scale,
                       "parallelism": "dp%d over source-point slices, all-reduce of 6x6 normal equations" % world},
            "ms_per_iter": dt / K * 1e3,
            "nn_queries_per_s": queries / dt,
            "lm_passes_per_iter": passes / K,
            "breakdown_ms_per_iter": {"transform_bbox": tot[6] / world / K, "nn_search_and_compaction": tot[7] / world / K,
                                      "lm_total": tot[8] / world / K, "lm_pass_kernels": lm_ms / K, "nn_query_kernels": nn_ms / K,
                                      "warmup_nn_query_kernels": warm_nn_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "kernel": kernel,
                         "note": "the NN kernel is VALU-issue bound, not HBM bound (rocprofv3 PMC: SQ_ACTIVE_INST_VALU ~ 86 % of its "
                                 "wave cycles, profiles/round1_nn_rows_pmc_sq_*.txt); the LM pass kernel streams at the HBM roofline "
                                 "(see other.k_lm_pass_GBs)",
                         "algorithmic_bytes_per_launch": per_launch_bytes, "avg_launch_ms": avg_ms,
                         "other": {"k_lm_pass_GBs": (ALG_BYTES_PER_CORR_PASS * corr / world / K) / (lm_ms / max(passes, 1) * 1e-3) / 1e9 if lm_ms > 0 else None,
                                   "k_nn_query_GBs": (ALG_BYTES_PER_QUERY * queries / world / n_nn_launch) / (nn_ms / n_nn_launch * 1e-3) / 1e9 if nn_ms > 0 else None}},
        }
        if world == 1 and not args.no_reg:
            out["image_registrator"] = reg_leg
        if base is not None:
            out["cpu_baseline"] = base
            out["speedup_vs_cpu_iteration_rate"] = (base["ms_per_iter"] / base["correspondences"]) / ((dt / K * 1e3) / (corr / K))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
