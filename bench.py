#!/usr/bin/env python
"""bench.py -- the ETH3D scan-alignment / image-registration hot paths on MI355X
(BASELINE.json metric: ICP correspondences/sec + ms/iter; ImageRegistrator residuals/sec, 1/2/4/8 GPU).

    python bench.py --gpus N --steps K --warmup W

Headline (`value`): BASELINE.json configs[1] -- ICPScanAligner on 2 scans of ~50 M points each, `-d 0.01` -- on seeded synthetic
scans of that shape generated directly in HBM (no network / no terrace data here).  A "step" is one outer ICP iteration, i.e. one
PointToPlaneICP::Run(d, it, /*max_num_iterations*/1, thr) exactly as the tool's loop calls it (src/exe/icp_scan_aligner.cc:342-343):
transform + bbox, exact 1-NN correspondence search for both directed pairs, and the full inner Levenberg-Marquardt solve
(<= 150 iterations x (1 + <= 10 tries)).  Inputs are resident in HBM first.  What a step carries over from the step before it is what
the library's handle carries between the outer iterations of any Run(): the static grids, the per-query certificates of the NN search
(a query whose old partner is provably still its unique nearest neighbour is settled by one gather instead of a search) and the
resident correspondence rows (only rows whose partner changed are rewritten).  Both are exact -- every step's counts, distances and
poses are those of a search from scratch (tests/test_gpu_icp.py, tests/test_gpu_at_size.py) -- and both are part of the product,
not of the bench; `ms_per_step_settling` / `ms_per_step_steady` separate the steps in which most certificates still break from
the settled ones, and E3D_NN_CERT=0 / E3D_ICP_RESIDENT=0 time the job without either.  No result is reused across steps.

N > 1: one process per GPU.  Launched by torchrun (RANK / WORLD_SIZE in the environment) or, if not, bench.py spawns the N ranks
itself.  The ranks talk through the library's own RCCL communicator (e3d_comm_*: per-pair normal-equation blocks all-reduced in HBM
on the library's stream); torch.distributed (gloo) only carries the 128-byte communicator id, the barriers and the max-over-ranks
time.  Headline at N > 1 = the same 2-scan job, weak scaling (N x the points on N x the floor area, every rank 1/N of each pair's
queries) -- at N = 1 it is the BENCH workload.

Output: the LAST line of stdout is one compact JSON object (< 4 KB: headline keys, `config`, `dtype`, the dominant kernel's `roofline`,
`cpu_baseline`, `whole_run`, one summary per secondary leg under `legs`); the full result -- every kernel group, per-step arrays, the
prose -- goes to bench_detail.json next to this file (and to gpurun_out/ when that directory exists).  The timed loop is the tool's:
it ends with the iteration whose Run() reports convergence (`converged_at_iteration`; `steps` = the steps actually timed).

Further legs (detail file; summaries in the line):
  partial_overlap    (N = 1) the headline job on the partial-overlap room (30 - 60 % of the points find a partner): same
                     per-kernel table, so the no-partner branch of FindCorrespondencesFast is part of a measured steady state
  allpairs           BASELINE.json north_star's scaling target / configs[2] shape: 16 scans all-pairs (240 directed pairs, 90-unknown
                     LM system), STRONG scaling (same job at every N, every rank a slice of every pair) -- compare `allpairs.value`
                     across N
  image_registrator  (N = 1) configs[3] shape: 23 images of 6048 x 4032, THIN_PRISM_FISHEYE, ~10 M points, K = 5: residuals/s of the
                     accumulate pass, per-kernel roofline from HIP events, ms per RunOnCurrentScale iteration
  normal_estimation  (N = 1) SURVEY 8(d) (A'): normals/s for 20 M points, k = 32 and k = 8
  cpu_baseline       the CPU restatement (oracle/, kind "port") on bounded samples of the same workloads, on this box's cores
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
ALG_BYTES_PER_CORR_PASS = 56   # 2 x i32 + 4 x vec3 f32 per correspondence per pass
ALG_BYTES_PER_QUERY = 32       # 12 in + 8 out + 12 amortised target
TRAFFIC_JSON = next((p for p in (os.path.join(ROOT, "profiles", "round%d_traffic.json" % r) for r in (6, 5, 4, 3, 2)) if os.path.exists(p)),
                    os.path.join(ROOT, "profiles", "round6_traffic.json"))


def load_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this same command (bench.py cannot run the
    profiler itself): profiles/round<N>_traffic.json (the newest), written by tools/make_traffic_json.py from tools/prof_round5.sh's passes."""
    if not os.path.exists(TRAFFIC_JSON):
        return None, None
    k = json.load(open(TRAFFIC_JSON)).get("kernels", {}).get(kernel_key)
    return (k["hbm_bytes_per_launch"], "profiles/" + os.path.basename(TRAFFIC_JSON)) if k else (None, None)


def crop_world(scan, lo, hi):
    """Points of a scan whose true world x lies in [lo, hi) (numpy, host)."""
    T = scan["T_true"]
    x = scan["xyz"]
    # elementwise on purpose: torch's gemv path returned wrong values for 50 M-row operands on this ROCm build
    wx = x[:, 0] * float(T[0, 0]) + x[:, 1] * float(T[0, 1]) + x[:, 2] * float(T[0, 2]) + float(T[0, 3])
    m = (wx >= lo) & (wx < hi)
    return scan["xyz"][m].cpu().numpy(), scan["normals"][m].cpu().numpy()


def cpu_baseline_icp(scans, d, thr, slab, n_points, poses=None, first_iteration=0):
    """Reference-faithful CPU path (the oracle, "kind": "port") on a bounded sample of the same workload: SURVEY 8(d): >= 3
    outer iterations, median.  `poses`: global_T_cloud per scan to start from -- the GPU run's poses after its warm-up iterations, so
    that both sides are timed in the same regime of the alignment (the reference's cost per iteration falls with the number of LM
    passes as the poses settle, exactly like the GPU's)."""
    from oracle import binding as ob
    ob.lib()
    cores = ob.set_num_threads(ob.effective_cores())      # (the affinity mask capped by the cgroup quota, not os.cpu_count())
    clouds, n = [], []
    for i, s in enumerate(scans):
        xyz, nrm = crop_world(s, slab[0], slab[1])
        n.append(int(xyz.shape[0]))
        clouds.append((xyz, nrm, s["T_init"] if poses is None else poses[i]))

    def run(all_core):
        o = ob.OracleICP()
        if all_core:
            o.set_all_core(True)
        for (xyz, nrm, T) in clouds:
            o.add_point_cloud(xyz, nrm, T, False)
        times = []
        for it in range(first_iteration, first_iteration + 3):
            t0 = time.perf_counter()
            o.run(d, it, 1, thr, False)
            times.append(time.perf_counter() - t0)
        return o.iter_records(), times

    recs, times = run(False)
    med = int(np.argsort(times)[1])
    # the same sample with every phase on every host core -- kd-tree builds, queries (independent; results and their
    # order unchanged) and the reductions of the LM passes (oracle_icp.c: accumulate_pass_par / cost_pass_par)
    recs2, times2 = run(True)
    med2 = int(np.argsort(times2)[1])
    frac = float(n[0] + n[1]) / float(2 * n_points)
    return {
        "value": recs[med]["correspondences"] / times[med], "unit": "correspondences/s", "cores": 2, "kind": "port",
        "start": "the GPU run's poses after its %d warm-up iterations (the regime the GPU steps are timed in)" % first_iteration if poses is not None else "the initial poses",
        "sample": "median of 3 outer iterations on the world-x slab [%.2f, %.2f) m of both scans: %d + %d points = %.1f %% of the "
                  "2 x %d of configs[1], same density and flags, %.1f s of CPU work; NN phase on 2 threads (one per directed pair, as "
                  "icp_point_to_plane.cc:208), inner LM single-threaded" % (slab[0], slab[1], n[0], n[1], 100 * frac, n_points, sum(times)),
        "ms_per_iter": times[med] * 1e3, "ms_per_iter_all": [t * 1e3 for t in times], "correspondences": int(recs[med]["correspondences"]),
        "sample_fraction_of_configs1": frac,
        "t_nn_s": recs[med]["t_nn_s"], "t_lm_s": recs[med]["t_lm_s"],
        "lm_passes": int(recs[med]["accumulate_passes"] + recs[med]["cost_passes"]), "host_cores_available": cores, "host_cpu_count": os.cpu_count(),
        "all_core": {"value": recs2[med2]["correspondences"] / times2[med2], "unit": "correspondences/s", "cores": cores,
                     "ms_per_iter": times2[med2] * 1e3, "t_nn_s": recs2[med2]["t_nn_s"], "t_lm_s": recs2[med2]["t_lm_s"],
                     "same_correspondences": bool([r["correspondences"] for r in recs2] == [r["correspondences"] for r in recs]),
                     "note": "SURVEY 8(d)(ii): kd-tree builds (OpenMP tasks), NN queries and both LM passes (per-thread H / b / cost partials added in thread order) on all host cores; "
                             "same slab and iterations; the sums' order differs from the sequential pass, so counts may differ in the last digits"},
    }


def candidates_per_query(scans, poses, cell, device, n_sample=200_000):
    """What an exact search of one query has to look at: the target points in the 27 cells (cell = the search radius, as the
    library's grids) around a query of scan 0, scans at the given global_T_cloud.  Sampled queries, torch on the GPU; the library's
    grid has the same cell size in the target's local frame (another origin: the same statistics, not the same cells)."""
    import torch

    def to_global(x, T):
        T = np.asarray(T, np.float64)
        return torch.stack([x[:, 0] * float(T[r, 0]) + x[:, 1] * float(T[r, 1]) + x[:, 2] * float(T[r, 2]) + float(T[r, 3]) for r in range(3)], 1)
    n = scans[0]["xyz"].shape[0]
    g = torch.Generator(device=device); g.manual_seed(7)
    sample = torch.randint(0, n, (min(n_sample, n),), generator=g, device=device)
    gs = to_global(scans[0]["xyz"][sample].to(torch.float64), poses[0])
    gt = to_global(scans[1]["xyz"].to(torch.float64), poses[1])
    origin = gt.min(0).values - 2.0 * cell
    ct = ((gt - origin) / cell).floor().to(torch.int64)
    D = ct.max(0).values + 4
    uk, cnt = torch.unique((ct[:, 2] * D[1] + ct[:, 1]) * D[0] + ct[:, 0], return_counts=True)
    del gt, ct
    cs = ((gs - origin) / cell).floor().to(torch.int64)
    ok = ((cs >= 1) & (cs < D - 1)).all(1)
    tot = torch.zeros(len(sample), dtype=torch.int64, device=device)
    own = torch.zeros_like(tot)
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                k = ((cs[:, 2] + dz) * D[1] + cs[:, 1] + dy) * D[0] + cs[:, 0] + dx
                pos = torch.searchsorted(uk, k).clamp_(max=len(uk) - 1)
                c = torch.where(ok & (uk[pos] == k), cnt[pos], torch.zeros_like(tot))
                tot += c
                if dx == 0 and dy == 0 and dz == 0:
                    own = c
    has = tot > 0
    t = tot[has].to(torch.float64)
    q = lambda v, f: float(torch.quantile(v, f)) if len(v) else 0.0
    return {"sampled_queries": int(len(sample)), "with_candidates_fraction": float(has.double().mean()), "mean": float(t.mean()) if len(t) else 0.0,
            "median": q(t, 0.5), "p99": q(t, 0.99), "max": float(t.max()) if len(t) else 0.0,
            "target_points_per_occupied_cell_mean": float(cnt.double().mean()), "target_points_per_cell_max": int(cnt.max()),
            "own_cell_max": int(own.max()), "cell_m": cell,
            "note": "target points in the 27 cells around a query (queries with at least one): what k_nn_rows evaluates per query is the "
                    "union of its wave segment's rows, a superset of this"}


def sum_records(recs):
    return np.array([
        sum(r["correspondences"] for r in recs), sum(r["queries"] for r in recs),
        sum(r["t_lm_kernel_ms"] for r in recs), sum(r["t_nn_query_ms"] for r in recs),
        sum(r["full_passes"] + r["cost_passes"] + r["multi_cost_passes"] for r in recs),
        sum(r["t_transform_ms"] for r in recs), sum(r["t_nn_ms"] for r in recs), sum(r["t_lm_ms"] for r in recs),
        sum(r["t_lm_full_kernel_ms"] for r in recs), sum(r["full_passes"] for r in recs),
        sum(r["t_nn_certify_ms"] for r in recs), sum(r["nn_certify_launches"] for r in recs), sum(r["nn_certify_queries"] for r in recs),
        sum(r["t_nn_bounded_ms"] for r in recs), sum(r["nn_bounded_launches"] for r in recs), sum(r["nn_bounded_queries"] for r in recs),
        sum(r["t_nn_search_ms"] for r in recs), sum(r["nn_search_launches"] for r in recs), sum(r["nn_search_queries"] for r in recs),
        sum(r["t_nn_sort_ms"] for r in recs), sum(r["t_nn_scan_ms"] for r in recs), sum(r["t_nn_compact_ms"] for r in recs),
        sum(r["multi_cost_passes"] for r in recs),
        sum(r["corr_rows_rewritten"] for r in recs), sum(r["corr_rows_walked"] for r in recs),
        sum(r["multi_cost_poses"] for r in recs), sum(r["lm_passes_skipped"] for r in recs),
        sum(r["nn_update_launches"] for r in recs), sum(r["nn_kernel_launches"] for r in recs),
    ], dtype=np.float64)


class Ranks:
    """Control plane between the ranks (gloo): barriers, max / sum of small host vectors; data plane = the library's RCCL comm."""

    def __init__(self, rank, world, local_rank, e3d):
        import torch
        self.rank, self.world, self.local_rank = rank, world, local_rank
        self.dist = None
        self.comm = None
        if world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            self.dist = dist
            uid = torch.zeros(128, dtype=torch.uint8)
            if rank == 0:
                uid = torch.frombuffer(bytearray(e3d.Comm.unique_id()), dtype=torch.uint8).clone()
            dist.broadcast(uid, src=0)
            self.comm = e3d.Comm(bytes(uid.numpy().tobytes()), rank, world, local_rank)     # ncclCommInitRank
            assert self.comm.world_size == world

    def barrier(self):
        import torch
        torch.cuda.synchronize()
        if self.dist:
            self.dist.barrier()
        torch.cuda.synchronize()

    def reduce(self, vec, op="sum"):
        import torch
        if not self.dist:
            return np.asarray(vec, np.float64)
        t = torch.from_numpy(np.asarray(vec, np.float64).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return t.numpy()

    def close(self):
        if self.comm:
            self.comm.destroy()
        if self.dist:
            self.dist.destroy_process_group()


def run_icp(e3d, R, icp, d, thr, warmup, steps, warmed=False, first_warm_converged=None):
    """W untimed + K timed outer iterations, barrier + synchronize on both sides, time = max over ranks.
    warmed: the caller has already run the W warm-up iterations on this handle.
    The loop is the tool's (src/exe/icp_scan_aligner.cc:342-370): an iteration whose Run() reports convergence is the last one --
    the timed region ends with it (`steps_run` < K then; `converged_at` = that iteration).  Every rank sees the same poses, hence the
    same return value."""
    converged_at = first_warm_converged
    if not warmed:
        for it in range(warmup):
            if icp.run(d, it, 1, thr, False):
                converged_at = it
                break
    warm = icp.iter_records()
    icp.clear_records()
    R.barrier()
    t0 = time.perf_counter()
    each = []
    if converged_at is None:
        for it in range(warmup, warmup + steps):
            t1 = time.perf_counter()
            conv = icp.run(d, it, 1, thr, False)                  # returns after the library has synchronised its stream (poses are host data)
            each.append((time.perf_counter() - t1) * 1e3)
            if conv:
                converged_at = it
                break
    R.barrier()
    dt = float(R.reduce([time.perf_counter() - t0], "max")[0])
    recs = icp.iter_records()
    for r, ms in zip(recs, each):
        r["wall_ms"] = ms
    tot = R.reduce(sum_records(recs))
    n = max(len(each), 1)
    per_rank = {"nn_kernel_ms_per_iter": sum(r["t_nn_query_ms"] for r in recs) / n, "lm_kernel_ms_per_iter": sum(r["t_lm_kernel_ms"] for r in recs) / n,
                "nn_ms_per_iter": sum(r["t_nn_ms"] for r in recs) / n, "lm_ms_per_iter": sum(r["t_lm_ms"] for r in recs) / n}
    return dt, tot, warm, recs, per_rank, len(each), converged_at


def whole_run(e3d, scans, d, thr, device, max_iterations=100):
    """What the tool does with two scans already in memory (src/exe/icp_scan_aligner.cc:280-372 without the file I/O): AddPointCloud
    for every scan, then Run(d, it, 1, thr) from iteration 0 until it reports convergence or --max_iterations (100, BASELINE.json
    configs[1]) -- grid builds, first-touch transforms and the expensive first iterations included.  A handle of its own."""
    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    icp = e3d.PointToPlaneICP(device=device)
    for s in scans:
        icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
    t_add = time.perf_counter() - t0
    each, conv = [], None
    for it in range(max_iterations):
        t1 = time.perf_counter()
        c = icp.run(d, it, 1, thr, False)
        each.append((time.perf_counter() - t1) * 1e3)
        if c:
            conv = it
            break
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    recs = icp.iter_records()
    corr = sum(r["correspondences"] for r in recs)
    out = {"wall_s": wall, "add_cloud_s": t_add, "iterations_run": len(each), "converged_at_iteration": conv, "max_iterations": max_iterations,
           "ms_per_iteration_mean": float(np.mean(each)), "ms_per_iteration_each": each, "correspondences_per_s": corr / wall,
           "first_iteration_ms": each[0], "note": "add_cloud -> convergence on a fresh handle: grid builds, first-touch transforms and iterations 0.. included; "
                                                  "inputs resident in HBM (the binary's PLY read / write: tools/bench_tool_icp.py, DESIGN.md section 5)"}
    del icp
    torch.cuda.empty_cache()
    return out


def step_breakdown(r):
    """Where one outer iteration's time goes (HIP events of this rank; the last timed step = the settled regime)."""
    keys = ("t_transform_ms", "t_nn_certify_ms", "t_nn_bounded_ms", "t_nn_search_ms", "t_nn_sort_ms", "t_nn_scan_ms", "t_nn_compact_ms",
            "t_lm_kernel_ms", "t_lm_full_kernel_ms")
    out = {k[2:]: r[k] for k in keys}
    out["kernels_sum_ms"] = sum(r[k] for k in keys if k != "t_lm_full_kernel_ms")
    out["nn_phase_ms"], out["lm_phase_ms"] = r.get("t_nn_ms"), r.get("t_lm_ms")      # events around the phases: kernels + host work in between
    out.update({"wall_ms": r.get("wall_ms"), "full_passes": r["full_passes"], "cost_passes": r["cost_passes"], "multi_cost_passes": r["multi_cost_passes"],
                "multi_cost_poses": r["multi_cost_poses"], "inner_iterations": r["inner_iterations"], "rows_rewritten": r["corr_rows_rewritten"]})
    return out


def comm_report(R, steps, per_rank):
    """What the first multi-GPU run has to explain itself with: this rank's collectives by HIP events (enqueue -> completion on the
    library's stream, so waiting for the slowest rank is inside), their number and payload, against the kernels they follow."""
    ms, calls, nbytes = R.comm.stats()
    mx = R.reduce([ms, per_rank["lm_kernel_ms_per_iter"] * steps, per_rank["nn_ms_per_iter"] * steps], "max")
    mn = -R.reduce([-ms, -per_rank["lm_kernel_ms_per_iter"] * steps, -per_rank["nn_ms_per_iter"] * steps], "max")
    return {"allreduce_calls_per_iter": calls / steps, "allreduce_bytes_per_call": nbytes / max(calls, 1),
            "allreduce_ms_per_iter_rank0": ms / steps, "allreduce_ms_per_iter_max_over_ranks": mx[0] / steps,
            "allreduce_ms_per_iter_min_over_ranks": mn[0] / steps,
            "lm_kernel_ms_per_iter_max_min_over_ranks": [mx[1] / steps, mn[1] / steps],
            "nn_phase_ms_per_iter_max_min_over_ranks": [mx[2] / steps, mn[2] / steps],
            "note": "HIP events around every ncclAllReduce on the library's stream: the time includes waiting for the slowest rank's "
                    "kernels in front of its all-reduce; the minimum over the ranks is close to the collective itself"}


def icp_kernel_table(tot, world, K, pairs, moved_points, lm_kernel_name="k_lm_pass<1>"):
    """Per-kernel-group table of an ICP leg from the summed iteration records (sum_records): HIP-event time per launch against
    SURVEY 8(d)'s algorithmic bytes.  pairs = directed pairs per outer iteration, moved_points = points transformed per iteration."""
    corr, queries = tot[0], tot[1]
    lm_ms, nn_ms, passes = tot[2] / world, tot[3] / world, tot[4] / world
    n_nn_launch = pairs * K
    lm_bytes = ALG_BYTES_PER_CORR_PASS * (corr / world / K)                  # one pass over this rank's correspondences
    nn_bytes = ALG_BYTES_PER_QUERY * (queries / world / n_nn_launch)         # one directed pair's queries
    lm_full_ms, full_passes = tot[8] / world, tot[9] / world
    lm_avg, nn_avg = lm_full_ms / max(full_passes, 1), nn_ms / n_nn_launch
    multi_ms, multi_passes = lm_ms - lm_full_ms, tot[22] / world
    rows_rewritten, rows_walked, multi_poses, passes_skipped = tot[23] / world, tot[24] / world, tot[25] / world, tot[26] / world
    upd_launches = max(tot[27] / world, 1.0)          # row-update launches (one per batch of pairs, or per pair: E3D_ICP_BATCH)
    lm_moved = 48.0 * rows_walked / K       # what a pass READS: three float4 per row it walks (resident rows: incl. the zero rows of listed groups)
    kernels = {
        "k_lm_pass": {"what": "%s: fused cost + Gramian pass over the correspondence rows (a7/a8); %.2f launches per iteration; "
                              "`algorithmic` = SURVEY 8(d)'s 56 B per correspondence, `moved` = the 48 B per row the pass reads" % (lm_kernel_name, full_passes / K),
                      "algorithmic_bytes_per_launch": lm_bytes, "avg_launch_ms": lm_avg, "GBs": lm_bytes / (lm_avg * 1e-3) / 1e9 if lm_avg > 0 else None,
                      "moved_bytes_per_launch": lm_moved, "GBs_moved": lm_moved / (lm_avg * 1e-3) / 1e9 if lm_avg > 0 else None,
                      "frac_moved": lm_moved / (lm_avg * 1e-3) / 1e9 / HBM_PEAK_GBS if lm_avg > 0 else None,
                      "summed_ms_per_iter": lm_full_ms / K},
        "k_lm_cost_multi": {"what": "the costs of the DISTINCT new poses among LM tries 1..9 in one pass over the rows (a8); %.2f launches per iteration, "
                                    "%.2f poses per launch; %.2f passes per iteration not launched at all (every pose asked for already evaluated)"
                                    % (multi_passes / K, multi_poses / max(multi_passes, 1), passes_skipped / K),
                            "algorithmic_bytes_per_launch": lm_bytes, "avg_launch_ms": multi_ms / multi_passes if multi_passes else None,
                            "GBs": lm_bytes / (multi_ms / multi_passes * 1e-3) / 1e9 if multi_passes and multi_ms > 0 else None,
                            "summed_ms_per_iter": multi_ms / K},
    }

    def nn_kernel(what, t_ms, launches, n_queries):
        # per launch: 32 B per query the launch covered; averages over this rank's launches in the timed region
        if launches <= 0:
            return {"what": what, "launches_per_iter": 0.0, "summed_ms_per_iter": 0.0, "avg_launch_ms": None, "algorithmic_bytes_per_launch": None, "GBs": None}
        avg, by = t_ms / launches, ALG_BYTES_PER_QUERY * n_queries / launches
        return {"what": what, "launches_per_iter": launches / K, "summed_ms_per_iter": t_ms / K, "avg_launch_ms": avg,
                "algorithmic_bytes_per_launch": by, "GBs": by / (avg * 1e-3) / 1e9 if avg > 0 else None}
    kernels["k_nn_certify"] = nn_kernel("partner of the last search still the unique nearest neighbour? (one gather per query, a5)", tot[10] / world, tot[11] / world, tot[12] / world)
    kernels["k_nn_bounded"] = nn_kernel("k_nn_bounded_half: exact search inside the ball of the old partner's distance over the half-cell directory, one thread per "
                                        "listed query (a5); bound by the latency and issue rate of its candidate gathers, not by its 32 algorithmic bytes",
                                        tot[13] / world, tot[14] / world, tot[15] / world)
    kernels["k_nn_rows"] = nn_kernel("exact search of the remaining queries, sorted by target cell, LDS-staged candidate rows (a5)", tot[16] / world, tot[17] / world, tot[18] / world)

    def stream_kernel(what, t_ms, launches, bytes_per_launch):
        avg = t_ms / launches if launches else None
        return {"what": what, "launches_per_iter": launches / K, "summed_ms_per_iter": t_ms / K, "avg_launch_ms": avg,
                "algorithmic_bytes_per_launch": bytes_per_launch, "GBs": bytes_per_launch / (avg * 1e-3) / 1e9 if (avg and bytes_per_launch) else None}
    kernels["k_transform_bbox"] = stream_kernel("a3: the cloud whose pose changed into the global frame + bounding box, 32 B per point moved (impl cloud 0 never "
                                                "moves and is not transformed again)", tot[5] / world, K, 32.0 * moved_points)
    kernels["query_keys_and_sort"] = stream_kernel("cell keys + rocPRIM radix sort of the queries the row kernel searches (a5 prep)", tot[19] / world, max(tot[17] / world, 1), None)
    kernels["match_scan"] = stream_kernel("totals of the per-block match counts / distance sums + the lists of active 64-row groups (3 small kernels per batch of pairs)",
                                          tot[20] / world, upd_launches, None)
    kernels["k_corr_update"] = stream_kernel("resident correspondence rows brought up to date: match, encoded partner and distance of every query (12 B) + "
                                             "116 B per row whose partner changed (%.3g rows of %.3g per launch); with E3D_ICP_RESIDENT=0: k_compact_corr, every row"
                                             % (rows_rewritten / upd_launches, queries / world / upd_launches),
                                             tot[21] / world, upd_launches, (12.0 * queries / world + 116.0 * rows_rewritten) / upd_launches)
    accounted = sum((v["summed_ms_per_iter"] or 0.0) for v in kernels.values())
    kernels["nn_search_per_pair"] = {"what": "certify + bounded + rows per directed pair: 32 B per query of the pair", "algorithmic_bytes_per_launch": nn_bytes, "avg_launch_ms": nn_avg,
                                     "GBs": nn_bytes / (nn_avg * 1e-3) / 1e9 if nn_avg > 0 else None, "summed_ms_per_iter": nn_ms / K}
    return kernels, accounted, dict(corr=corr, queries=queries, lm_ms=lm_ms, nn_ms=nn_ms, passes=passes, nn_bytes=nn_bytes, nn_avg=nn_avg,
                                    rows_rewritten=rows_rewritten, rows_walked=rows_walked)


def leg_terrace(e3d, synth, R, args, dev, partial=False, variant=None):
    """configs[1]: 2 scans, both movable; N > 1: weak scaling on a stretched room (same point density).
    partial=True: the same job on the partial-overlap room (30 - 60 % of the points find a partner).
    variant "regression": the headline job from round 4's start (--perturb 1.0: 1 degree, 2.4 cm; converges in iteration 18) -- the
    number that stays comparable from round to round; "scanner": both scans as a terrestrial scanner records them (density
    ~ cos / range^2, points in scan order: synth.make_scan_angular) from the headline's start."""
    import torch
    world = R.world
    secondary = partial or variant is not None
    perturb = 1.0 if variant == "regression" else args.perturb
    n_points = args.points if args.points > 0 else 50_000_000 * (1 if secondary else world)
    d, thr = float(args.distance), 1e-10     # README.md:101 recommended flags (--convergence_threshold 1e-10)
    room_scale = float(np.sqrt(n_points / 50_000_000.0)) if (world > 1 and args.points == 0 and not secondary) else 1.0
    if args.room_scale > 0 and not secondary:
        room_scale = float(args.room_scale)      # (one GPU at the sizes of an N-GPU weak-scaling run: --points 400000000 --room-scale 2.8284)
    scans = synth.make_scene(2, n_points, seed=1234, sigma=0.002, device=dev, room_scale=room_scale, partial=partial, perturb=perturb, scanner=(variant == "scanner"))
    torch.cuda.synchronize()
    whole = None
    if world == 1 and not secondary and not args.no_whole_run:
        whole = whole_run(e3d, scans, d, thr, R.local_rank)
    icp = e3d.PointToPlaneICP(device=R.local_rank)
    for s in scans:
        icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
    if R.comm:
        icp.set_comm(R.comm)
    warm_conv = None
    for it in range(args.warmup):                            # untimed warm-up steps (the timed region follows below)
        if icp.run(d, it, 1, thr, False):
            warm_conv = it
            break
    cand = None
    if world == 1 and (variant == "scanner" or not secondary):
        cand = candidates_per_query(scans, [icp.get_result_global_T_cloud(i) for i in range(2)], d, dev)
        torch.cuda.empty_cache()
    base = None
    if R.rank == 0 and world == 1 and not args.no_cpu_baseline and not secondary:
        # slab width chosen for ~4 M points per scan at this density (floor + two walls = 16 m^2 per metre of x); started from the
        # poses the GPU run has reached after its warm-up, so both sides are timed in the same regime
        width = min(10.0, 4.0e6 / (n_points / 242.6 * 16.0))
        base = cpu_baseline_icp(scans, d, thr, (4.0, 4.0 + width), n_points, poses=[icp.get_result_global_T_cloud(i) for i in range(2)],
                                first_iteration=args.warmup)
    del scans
    torch.cuda.empty_cache()
    if R.comm:
        R.comm.stats(reset=True)
    dt, tot, warm, recs, per_rank, K, converged_at = run_icp(e3d, R, icp, d, thr, args.warmup, args.steps, warmed=True, first_warm_converged=warm_conv)
    if K == 0:
        raise SystemExit("bench.py: the run converged inside the %d warm-up iterations (iteration %d): nothing to time" % (args.warmup, warm_conv))
    kernels, accounted, m = icp_kernel_table(tot, world, K, 2, n_points)      # (every rank transforms the whole clouds)
    corr, queries, lm_ms, nn_ms, passes = m["corr"], m["queries"], m["lm_ms"], m["nn_ms"], m["passes"]
    rows_rewritten, rows_walked = m["rows_rewritten"], m["rows_walked"]
    dom = max(("k_lm_pass", "k_lm_cost_multi", "k_nn_certify", "k_nn_bounded", "k_nn_rows", "k_corr_update"), key=lambda k: kernels[k]["summed_ms_per_iter"] or 0.0)
    traffic, traffic_src = load_traffic({"k_lm_pass": "k_lm_pass<1>"}.get(dom, dom)) if (world == 1 and n_points == 50_000_000 and not secondary) else (None, None)
    ach = kernels[dom]["GBs"] or 0.0
    # settling / steady split: the first timed steps still re-search most queries (the poses move); "steady" = the steps whose NN
    # kernels cost at most 1.25 x the last step's
    nnq = [r["t_nn_query_ms"] for r in recs]
    first_steady = next((i for i, v in enumerate(nnq) if v <= 1.25 * nnq[-1]), len(nnq) - 1)
    wall = [r["wall_ms"] for r in recs]
    out = {
        "metric": "ICP correspondences/sec", "value": corr / dt, "unit": "correspondences/s",
        "n_gpus": R.comm.world_size if R.comm else 1, "steps": K, "warmup": args.warmup, "ms_per_step": dt / K * 1e3,
        "steps_requested": args.steps, "converged_at_iteration": converged_at,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32 (f64 accumulation)",
        "data": "synthetic (no terrace scans in this image: seeded room scans of the configs[1] shape, generated in HBM)",
        "config": {"workload": "ICPScanAligner 2 scans (BASELINE.json configs[1]), -d %g, one outer iteration per step%s" %
                               (d, "; PARTIAL-OVERLAP room (partition wall, occlusion, 6.5 m range): %.0f %% of the queries find a partner" % (100.0 * corr / max(queries, 1)) if partial else
                                   "; SCANNER-SAMPLED scans (rays uniform in angle: density ~ cos / range^2, points in scan order)" if variant == "scanner" else
                                   "; REGRESSION scene (round 4's start: 1 degree, 2.4 cm)" if variant == "regression" else ""),
                   "points_per_scan": n_points, "scans": 2, "directed_pairs": 2, "room_scale": room_scale, "initial_misalignment_scale": perturb,
                   "matched_fraction": corr / max(queries, 1),
                   "parallelism": "dp%d over source-point slices, RCCL all-reduce of the 6x6 normal-equation blocks" % world},
        "ms_per_iter": dt / K * 1e3, "nn_queries_per_s": queries / dt, "lm_passes_per_iter": passes / K,
        "corr_rows_rewritten_per_iter": rows_rewritten / K, "corr_rows_walked_per_pass": rows_walked / K,
        "ms_per_step_each": wall,
        "ms_per_step_settling": float(np.mean(wall[:first_steady])) if first_steady > 0 else None,
        "ms_per_step_steady": float(np.mean(wall[first_steady:])), "steady_from_timed_step": first_steady,
        "last_step_ms": step_breakdown(recs[-1]),
        "breakdown_ms_per_iter": {"transform_bbox": tot[5] / world / K, "nn_search_and_compaction": tot[6] / world / K,
                                  "lm_total": tot[7] / world / K, "lm_pass_kernels": lm_ms / K, "nn_query_kernels": nn_ms / K,
                                  "kernels_accounted": accounted, "host_sync_and_small_kernels": dt / K * 1e3 - accounted,
                                  "nn_query_kernels_per_timed_iteration": nnq,
                                  "warmup_nn_query_kernels": [r["t_nn_query_ms"] for r in warm]},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "frac_of_bytes_moved": kernels[dom].get("frac_moved"),
                     "traffic": traffic, "traffic_source": traffic_src, "kernel": dom + ": " + kernels[dom]["what"],
                     "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"], "avg_launch_ms": kernels[dom]["avg_launch_ms"],
                     "note": "dominant kernel = largest summed HIP-event duration in the timed region; `kernels` lists every kernel group of the step "
                             "(their summed_ms_per_iter add up to breakdown_ms_per_iter.kernels_accounted; the rest of ms_per_step is host work and synchronisation)",
                     "kernels": kernels},
    }
    if world > 1:
        out["per_rank_rank0"] = per_rank
    if R.comm:
        out["comm"] = comm_report(R, K, per_rank)
    if cand is not None:
        out["candidates_per_query"] = cand
    if whole is not None:
        out["whole_run"] = whole
    if base is not None:
        out["cpu_baseline"] = base
        out["speedup_vs_cpu_iteration_rate"] = (base["ms_per_iter"] / base["correspondences"]) / ((dt / K * 1e3) / (corr / K))
    del icp
    torch.cuda.empty_cache()
    return out


def leg_allpairs(e3d, synth, R, args, dev):
    """north_star scaling target: S scans all-pairs, all movable, STRONG scaling (the same job at every N)."""
    import torch
    S, n, d, thr = args.allpairs_scans, args.allpairs_points, float(args.allpairs_distance), 1e-10
    scans = synth.make_scene(S, n, seed=4321, sigma=0.002, device=dev)
    torch.cuda.synchronize()
    icp = e3d.PointToPlaneICP(device=R.local_rank)
    for s in scans:
        icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
    if R.comm:
        icp.set_comm(R.comm)
    # two untimed iterations (grids, first full searches), then ten timed ones: the first of them still search most queries
    # again (the poses move by centimetres), the last ones run on certificates and resident rows -- both regimes are reported,
    # `value` covers all timed iterations.  --steps below 10 shortens the timed part (profiling).
    warmup, steps = 2, max(1, min(args.steps if args.steps != 20 else 10, 10))
    if R.comm:
        R.comm.stats(reset=True)
    dt, tot, warm, recs, per_rank, steps_run, converged_at = run_icp(e3d, R, icp, d, thr, warmup, steps)
    if steps_run == 0:
        raise SystemExit("bench.py: the all-pairs job converged inside its warm-up iterations")
    steps_requested, steps = steps, steps_run
    free, total = torch.cuda.mem_get_info(R.local_rank)
    world = R.world
    kernels, accounted, m = icp_kernel_table(tot, world, steps, S * (S - 1), n * (S - 1), lm_kernel_name="k_lm_pass<2> / <3> (two-sided pairs; <1> for the pairs of cloud 0)")
    wall = [r["wall_ms"] for r in recs]
    nnq = [r["t_nn_ms"] for r in recs]
    first_steady = next((i for i, v in enumerate(wall) if v <= 1.25 * wall[-1]), len(wall) - 1)
    dom = max(("k_lm_pass", "k_lm_cost_multi", "k_nn_certify", "k_nn_bounded", "k_nn_rows", "k_corr_update", "query_keys_and_sort"), key=lambda k: kernels[k]["summed_ms_per_iter"] or 0.0)
    out = {"metric": "ICP correspondences/sec", "value": tot[0] / dt, "unit": "correspondences/s", "scaling": "strong",
           "n_gpus": R.comm.world_size if R.comm else 1, "steps": steps, "warmup": warmup, "ms_per_iter": dt / steps * 1e3,
           "steps_requested": steps_requested, "converged_at_iteration": converged_at,
           "ms_per_iter_each": wall,
           "ms_per_iter_settling": float(np.mean(wall[:first_steady])) if first_steady > 0 else None,
           "ms_per_iter_steady": float(np.mean(wall[first_steady:])), "steady_from_timed_iteration": first_steady,
           "nn_ms_per_iter_each": nnq,
           "config": {"workload": "%d synthetic scans x %d points, all movable, all %d directed pairs, -d %g (north_star: 16-scan all-pairs ICP; "
                                  "configs[2] shape with --allpairs-scans 8 --allpairs-points 20000000)" % (S, n, S * (S - 1), d),
                      "unknowns": 6 * (S - 1), "parallelism": "every rank one slice of every directed pair's queries; one RCCL all-reduce of the "
                                                              "per-pair blocks per LM pass"},
           "correspondences_per_iter": tot[0] / steps, "queries_per_iter": tot[1] / steps, "lm_passes_per_iter": tot[4] / R.world / steps,
           "rank0_ms_per_iter": per_rank, "hbm_in_use_GB_rank0": (total - free) / 1e9,
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": dom + ": " + kernels[dom]["what"],
                        "achieved": kernels[dom]["GBs"], "frac": (kernels[dom]["GBs"] or 0.0) / HBM_PEAK_GBS,
                        "frac_of_bytes_moved": kernels[dom].get("frac_moved"), "traffic": None,
                        "algorithmic_bytes_per_launch": kernels[dom]["algorithmic_bytes_per_launch"], "avg_launch_ms": kernels[dom]["avg_launch_ms"],
                        "kernels": kernels, "kernels_accounted_ms_per_iter": accounted}}
    if R.comm:
        out["comm"] = comm_report(R, steps, per_rank)
    # launches of the NN phase per iteration (search, row update, totals; the record counts launches of the timed kernels)
    out["nn_launches_per_iter"] = {"certify": tot[11] / R.world / steps, "bounded": tot[14] / R.world / steps, "rows": tot[17] / R.world / steps,
                                   "row_update": sum(r["nn_update_launches"] for r in recs) / steps, "all_kernels": sum(r["nn_kernel_launches"] for r in recs) / steps,
                                   "batches": sum(r["nn_batches"] for r in recs) / steps, "radix_sorts": sum(r["nn_sort_calls"] for r in recs) / steps}
    del icp
    torch.cuda.empty_cache()
    if R.world == 1 and not args.no_scale_model:
        # What one rank of an 8-GPU run does, MEASURED on this GPU.  (1) The job once more on one handle with a tap on its reductions
        # (e3d_icp_set_shard with world 1: every buffer the library would all-reduce -- the per-pair counts of an iteration, the per-set
        # sums of every LM pass -- passes through the callback; recorded).  (2) A handle as rank 0 of a world of 8 -- its eighth of
        # every directed pair's queries -- whose all-reduce callback returns the recorded sums: it takes the decisions and reaches the
        # poses of the real job bit for bit (checked), so its launches, host round trips, searches and LM passes are those of rank 0
        # of a real 8-GPU run; only the collectives are missing.  t1 / t8 bounds the 8-GPU speed-up from above; (8 t8 - t1) / 7 is
        # the part of an iteration that does not divide by the number of GPUs.
        W8 = args.scale_model_world
        tape = []
        icpA = e3d.PointToPlaneICP(device=R.local_rank)
        for s in scans:
            icpA.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
        icpA.set_shard(0, 1, lambda arr: tape.append(arr.copy()))
        for it in range(warmup + steps):
            icpA.run(d, it, 1, thr, False)
        poses_a = [icpA.get_result_global_T_cloud(i) for i in range(S)]
        del icpA
        torch.cuda.empty_cache()
        pos = [0]

        def replay(arr):
            a = tape[pos[0]]
            if a.shape != arr.shape:
                raise RuntimeError("scale model: reduction %d has %d values, the recorded one %d" % (pos[0], arr.size, a.size))
            arr[:] = a
            pos[0] += 1
        icp8 = e3d.PointToPlaneICP(device=R.local_rank)
        for s in scans:
            icp8.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
        icp8.set_shard(0, W8, replay)
        for it in range(warmup):
            icp8.run(d, it, 1, thr, False)
        icp8.clear_records()
        torch.cuda.synchronize()
        each8 = []
        # (the replay callback is Python: one generation-2 garbage collection inside it cost 35 - 47 ms of one iteration in three of
        # five sessions -- E3D_LM_PROFILE=1 showed a single callback of that length; a real rank's all-reduce is RCCL, not Python)
        import gc
        gc.collect()
        gc.disable()
        try:
            for it in range(warmup, warmup + steps):
                t1 = time.perf_counter()
                icp8.run(d, it, 1, thr, False)
                each8.append((time.perf_counter() - t1) * 1e3)
        finally:
            gc.enable()
        same = all(np.array_equal(icp8.get_result_global_T_cloud(i), poses_a[i]) for i in range(S)) and pos[0] == len(tape)
        rec8 = icp8.iter_records()
        t1_ms, t8_ms = float(np.mean(wall)), float(np.mean(each8))
        s1, s8 = float(np.mean(wall[first_steady:])), float(np.mean(each8[first_steady:]))
        out["scale_model"] = {"world": W8, "ms_per_iter_n1": t1_ms, "ms_per_iter_as_rank0_of_world": t8_ms, "ms_per_iter_each_as_rank0": each8,
                              "modelled_speedup": t1_ms / t8_ms, "non_dividing_ms_per_iter": (W8 * t8_ms - t1_ms) / (W8 - 1),
                              "steady": {"ms_per_iter_n1": s1, "ms_per_iter_as_rank0_of_world": s8, "modelled_speedup": s1 / s8,
                                         "non_dividing_ms_per_iter": (W8 * s8 - s1) / (W8 - 1)},
                              "poses_equal_single_gpu_run": bool(same),
                              "last_iteration_as_rank0": step_breakdown(dict(rec8[-1], wall_ms=each8[-1])),
                              "last_iteration_n1": step_breakdown(recs[-1]),
                              "iterations_as_rank0": [step_breakdown(dict(r, wall_ms=w)) for r, w in zip(rec8, each8)],
                              "iterations_n1": [step_breakdown(r) for r in recs],
                              "nn_kernel_launches_per_iter_as_rank0": float(np.mean([r["nn_kernel_launches"] for r in rec8])),
                              "note": "rank 0 of a world of %d on this GPU, fed the recorded reductions of the single-GPU run (same decisions, same poses): "
                                      "an upper bound of the speed-up (no collective time, no skew between ranks)" % W8}
        del icp8
        torch.cuda.empty_cache()
    del scans
    return out


def leg_image_registrator(e3d, synth, args, dev):
    """configs[3] shape: 23 images of 6048 x 4032 (6 pyramid levels), THIN_PRISM_FISHEYE, ~10 M points, K = 5, 150 unknowns.
    residuals/s = (#fixed + #variable colour residuals) / wall time of the accumulate pass of IntrinsicsAndPoseOptimizer::Apply
    (src/opt/intrinsics_and_pose_optimizer.cc:87-92,624-837) over all images; per-kernel roofline from HIP events with SURVEY 8(d)'s
    bytes (pass 1: 24 + 12 + 8 + 4 (I + 7) B per observation; pass 2: 8 K + 4 K + 4 (K + 1)(I + 7) B per residual pair)."""
    import torch
    t0 = time.perf_counter()
    Wl = synth.make_reg_workload(n_points=args.reg_points, width=6048, height=4032, n_images=args.reg_images, model=2, device=dev)
    t_gen = time.perf_counter() - t0
    K, I = Wl["K"], len(Wl["params"])
    P = e3d.RegProblem(e3d.default_reg_params(image_scale_count=Wl["n_levels"], point_neighbor_count=K))
    P.set_intrinsics(0, Wl["width"], Wl["height"], Wl["params"], 0, Wl["n_levels"], camera_type=2)
    P.set_point_scale(0, Wl["pts"], Wl["point_radius"], Wl["nbr"], Wl["fixed_desc"])
    P.set_splat_points(Wl["pts"])
    ids = list(range(len(Wl["images"])))
    for i, im in enumerate(Wl["images"]):
        P.set_image(i, 0, im["pyr"]); P.set_image_pose(i, im["q"], im["t"])
    P.update_observations(1)
    t0 = time.perf_counter(); P.update_observations(1); t_obs = time.perf_counter() - t0
    P.color_update()
    for i in ids:
        P.accumulate(i, 0)
    P.kernel_times(reset=True)
    reps = 3
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = 0
    for _ in range(reps):
        for i in ids:
            _, _, _, c = P.accumulate(i, 0)
            res += int(c[0] + c[1])
    t_acc = (time.perf_counter() - t0) / reps
    res //= reps
    p1_ms, p2_ms, n_obs, calls = P.kernel_times(reset=True)
    p1_ms /= calls; p2_ms /= calls; n_obs /= calls                         # per launch (= per image)
    res_per_launch = res / len(ids)
    b1 = (24 + 12 + 8 + 4 * (I + 7)) * n_obs
    b2 = (8 * K + 4 * K + 4 * (K + 1) * (I + 7)) * (res_per_launch / 2)    # both residual kinds share the gathers: counted once
    state0 = [P.get_image_pose(i) for i in ids]
    t0 = time.perf_counter(); _, cost, its = P.run_on_current_scale(3, 0.0, 15, False); t_run = time.perf_counter() - t0
    # the same three iterations again from the same state with the phase profile on (the library then synchronises at every phase
    # boundary: slower, but the split adds up): where the ~80 ms of an iteration go
    for i, (q, t) in zip(ids, state0):
        P.set_image_pose(i, q, t)
    try:
        P.set_intrinsics(0, Wl["width"], Wl["height"], Wl["params"], 0, Wl["n_levels"], camera_type=2)
    except Exception:
        pass                                  # (the optimised intrinsics stay: the phase split does not depend on them)
    P.profile(True)
    t0 = time.perf_counter(); _, _, its_p = P.run_on_current_scale(3, 0.0, 15, False); t_prof = time.perf_counter() - t0
    phases = P.profile(False)
    # every kernel group of those iterations: HIP-event time per launch against the bytes the group has to move at least
    # (per unit, stated here; K neighbours, I intrinsics parameters).  Several of them are not HBM streams at all -- an atomic
    # z-buffer, a 21 x 21 window, f64 elementary functions per projected point -- so `frac` says how far from the HBM roof the
    # group runs, `bound` what holds it there.
    group_bytes = {
        "depth.zbuffer_clear": (4.0, "padded pixel: one 4 B store", "hbm"),
        "depth.splat_bin": (20.0, "splat point: 16 B read + one 4 B atomicMin into the z-buffer", "L2 atomics at random pixels (one per point)"),
        "depth.small_splat_tiles": (32.0, "(tile, small splat) pair: 8 B key/value through the radix sort + the 16 B rectangle", "hbm"),
        "depth.min_filter": (16.0, "output pixel: 4 B in + 4 B out in each of the two separable 21-tap passes", "LDS window + issue"),
        "depth.mesh_raster": (28.0, "triangle: 3 vertex indices + its share of the projected vertices", "issue (edge functions in f64)"),
        "obs.eval_all_points": (36.0, "point: 16 B position + 4 B depth lookup + 16 B (valid, x, y, scale) out", "issue: projection with f64 atan / polynomial distortion"),
        "obs.eval_listed_points": (36.0, "listed point: 4 B index + 16 B position + 16 B out", "issue: the same projection"),
        "obs.scan": (12.0, "candidate: flag + zeroed distance slot read, slot cleared", "hbm"),
        "obs.compact": (32.0, "candidate: 16 B in, 16 B out per kept observation", "hbm"),
        "obs.neighbour_flags": (13.0 + 12.0 * K, "observation: index + K neighbour indices + K row gathers + K rows out + flag; 8 B per point of row map", "4 B gathers at random points"),
        "intensity.sample": (24.0, "observation: 12 B position + 8 texels + 4 B out", "texel gathers"),
        "intensity.scatter": (16.0, "observation: 4 B index + 4 B value + 4 B scattered store + 4 B of the cleared point array", "hbm (scatter)"),
        "cost": (13.0 + 16.0 * K, "observation: index, flag, intensity, count + K x (neighbour index, neighbour intensity gather, fixed + variable descriptor)", "4 B gathers at random points"),
        "color.accumulate": (17.0 + 16.0 * K, "observation: as cost, with the K variable descriptors added in place (read-modify-write; a point occurs once per image, "
                                              "no atomics)", "4 B gathers at random points + read-modify-write"),
        "color.finish": (4.0 + 8.0 * K, "point: K descriptors divided by the count, in place", "hbm"),
        "color.clear": (4.0 + 4.0 * K, "point: descriptors + count cleared", "hbm"),
        "accumulate.pass1": (24.0 + 12.0 + 8.0 + 4.0 * (I + 7), "observation: SURVEY 8(d)", "issue: Jacobians of the projection (f64 elementary functions)"),
        "accumulate.pass2": ((8.0 * K + 4.0 * K + 4.0 * (K + 1) * (I + 7)) / 2.0 * (res_per_launch / max(n_obs, 1.0)),
                             "observation: SURVEY 8(d)'s bytes per residual pair x pairs per observation", "matrix-core issue; rows of the K neighbours are L2 hits"),
    }
    # the kernels behind a group in the committed counter passes (profiles/round<N>_traffic.json: HBM bytes per launch; a 4-image run
    # of this same leg, so per-launch figures carry over)
    group_kernels = {"depth.zbuffer_clear": [], "depth.splat_bin": ["k_splat_bin<2>"], "depth.min_filter": ["k_min_filter_tile"],
                     "obs.eval_all_points": ["k_obs_eval<2>"], "obs.eval_listed_points": ["k_obs_eval<2>"], "obs.compact": ["k_obs_compact"],
                     "obs.neighbour_flags": ["k_obs_flags", "k_obs_mark"], "intensity.sample": ["k_reg_intensity"], "cost": ["k_reg_cost<5>"],
                     "color.accumulate": ["k_color_accumulate"], "color.finish": ["k_color_finish"],
                     "accumulate.pass1": ["k_reg_pass1<2, false>"], "accumulate.pass2": ["k_reg_pass2_mfma<5, 18>"]}
    groups = {}
    for name, (ms, calls, units) in sorted(P.kernel_groups.items()):
        bpu, what, bound = group_bytes.get(name, (None, "", ""))
        trs = [load_traffic(kn)[0] for kn in group_kernels.get(name, [])]
        traffic = sum(trs) if trs and all(t is not None for t in trs) else None
        per = ms / max(calls, 1)
        by = bpu * units / max(calls, 1) if bpu else None
        groups[name] = {"ms_per_iteration": ms / max(its_p, 1), "launches_per_iteration": calls / max(its_p, 1), "avg_launch_ms": per,
                        "units_per_launch": units / max(calls, 1), "bytes_per_unit": bpu, "unit": what,
                        "algorithmic_bytes_per_launch": by, "GBs": by / (per * 1e-3) / 1e9 if (by and per > 0) else None,
                        "frac": by / (per * 1e-3) / 1e9 / HBM_PEAK_GBS if (by and per > 0) else None, "bound": bound,
                        "traffic": traffic, "traffic_over_algorithmic": (traffic / by) if (traffic and by) else None}
    free, total = torch.cuda.mem_get_info(0)
    tr1, src1 = load_traffic("k_reg_pass1<2, false>")
    p2_variant = os.environ.get("E3D_REG_PASS2", "")
    p2_f32 = p2_variant in ("tile32", "mfma32")
    p2_kernel = "k_reg_pass2_tile32" if p2_f32 else "k_reg_pass2_mfma"
    tr2, src2 = load_traffic("k_reg_pass2_tile32<5, 18>" if p2_f32 else "k_reg_pass2_mfma<5, 18>")
    out = {"metric": "ImageRegistrator residuals/sec", "value": res / t_acc, "unit": "residuals/s",
           "dtype": ("f32 rows; H, b: f32 fma chains of 32 (b: <= 20) residual pairs added into f64 -- NARROWER than the reference (E3D_REG_PASS2=%s, opt-in)" % p2_variant) if p2_f32 else
                    "f32 rows; H, b: every product formed exactly in f64 and summed in f64 (v_mfma_f64_16x16x4_f64 / v_fma_f64) -- the reference's sum of "
                    "single products in f64 (intrinsics_and_pose_optimizer.cc:1246-1247), each term at least as accurate as its fl32(fl32(w J_i) J_j)",
           "config": {"workload": "%d images 6048x4032 (6 levels) THIN_PRISM_FISHEYE, %d points, K = %d (BASELINE.json configs[3] shape)"
                                  % (len(ids), len(Wl["pts"]), K), "unknowns": I + 6 * len(ids)},
           "residuals": res, "accumulate_ms_all_images": t_acc * 1e3, "observation_refresh_ms_all_images": t_obs * 1e3,
           "ms_per_run_iteration": t_run / max(its, 1) * 1e3, "run_iterations": its, "hbm_in_use_GB": (total - free) / 1e9,
           "run_phase_profile": {"note": "e3d_reg_profile: wall clock per phase of the same %d iterations with a stream synchronisation at every phase boundary "
                                         "(%.1f ms per iteration that way, %.1f ms without); iteration 1 has no Apply, so apply.* covers %d calls" %
                                         (its_p, t_prof / max(its_p, 1) * 1e3, t_run / max(its, 1) * 1e3, max(its_p - 1, 0)),
                                 "ms_total": phases, "iterations": its_p,
                                 "kernel_groups": groups,
                                 "kernel_groups_ms_per_iteration": sum(g["ms_per_iteration"] for g in groups.values())},
           "input_generation_s": t_gen,
           "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "k_reg_pass1": {"algorithmic_bytes_per_launch": b1, "avg_launch_ms": p1_ms, "achieved": b1 / (p1_ms * 1e-3) / 1e9,
                                        "frac": b1 / (p1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr1, "traffic_source": src1},
                        p2_kernel: {"algorithmic_bytes_per_launch": b2, "avg_launch_ms": p2_ms, "achieved": b2 / (p2_ms * 1e-3) / 1e9,
                                    "frac_of_survey_bytes": b2 / (p2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "frac": (tr2 / (p2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr2 else None,
                                    "frac_basis": "HBM bytes the counters saw (traffic) / launch time / 8 TB/s: SURVEY 8(d)'s 516 B per residual pair "
                                                  "assume every neighbour row comes from HBM, but the rows of the K neighbours are L2 hits "
                                                  "(frac_of_survey_bytes is that figure and does not describe the kernel)",
                                    "traffic": tr2, "traffic_source": src2,
                                    "note": "gathers of neighbour rows are served by L2; the kernel is bound by the issue rate of the matrix "
                                            "instruction (v_mfma_f64_16x16x4_f64: 16 per neighbour slot of 64 observations, DESIGN.md section 10.1), not by HBM"}}}
    if not args.no_cpu_baseline:
        from oracle import reg_binding as rb
        from oracle.reg_driver import OracleRegProblem
        n1 = 2_000_000
        W1 = synth.make_reg_workload(n_points=n1, width=6048, height=4032, n_images=1, model=2, device=dev)
        O = OracleRegProblem(K=W1["K"], image_scale_count=W1["n_levels"])
        O.set_intrinsics(0, W1["width"], W1["height"], W1["params"], 0, W1["n_levels"], model=2)
        O.set_point_scale(0, W1["pts"], W1["point_radius"], W1["nbr"], W1["fixed_desc"])
        O.set_splat_points(W1["pts"])
        O.set_image(0, 0, W1["images"][0]["pyr"]); O.set_image_pose(0, W1["images"][0]["q"], W1["images"][0]["t"])
        O.update_observations(1); O.color_update()
        S = O.scales[0]; im = O.images[0]; I0 = O.intr[0]; o = O.obs[(0, 0)]
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            _, _, _, c = rb.accumulate(S["pts"], float(S["radius"]), S["nbr"], O.K, S["fixed"], S["var"], S["counts"], I0["levels"][0], I0["min"],
                                       im["pyr"], O._R(im), im["t"], o[:4], o[4], O.robust_type, O.robust_param, O.fixed_weight, O.var_weight)
            times.append(time.perf_counter() - t0)
        tc = float(np.median(times))
        out["cpu_baseline"] = {"value": float(c[0] + c[1]) / tc, "unit": "residuals/s", "cores": 1, "kind": "port",
                               "sample": "median of 3 accumulate passes (oracle_reg_accumulate, single thread like the reference) of ONE "
                                         "6048x4032 THIN_PRISM_FISHEYE image with %d points, K = 5: %d residuals in %.2f s" % (len(W1["pts"]), int(c[0] + c[1]), tc)}
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    del P
    torch.cuda.empty_cache()
    return out


def leg_normals(e3d, synth, args, dev):
    """(A') NormalEstimationTwoPassOMP: one synthetic room scan of 20 M points resident in HBM, e3d_normals_knn end to end (grid,
    sorts, search + covariance + eigenvector); algorithmic bytes 12 k + 28 per point."""
    import ctypes as C
    import torch
    capi = importlib.import_module("dataset-pipeline_amd.capi")
    origin, yaw = synth.SCAN_POSES[0]
    n = args.normals_points
    xyz, _, _ = synth.make_scan(n, origin, yaw, 1234, device=dev)
    xyz = xyz.contiguous()
    on = torch.empty((n, 3), dtype=torch.float32, device=dev)
    oc = torch.empty(n, dtype=torch.float32, device=dev)
    vp = np.zeros(3, np.float32)
    torch.cuda.synchronize()
    out = {"metric": "normals/s", "points": n, "unit": "normals/s"}
    for k in (32, 8):
        def call():
            r = capi.lib().e3d_normals_knn(C.c_void_p(xyz.data_ptr()), n, k, C.c_void_p(vp.ctypes.data), C.c_void_p(on.data_ptr()), C.c_void_p(oc.data_ptr()), None)
            assert r == 0, capi.lib().e3d_last_error()
        call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        alg = n * (12 * k + 28)
        tr, src = load_traffic("k_knn_normals_k%d" % k)
        out["k%d" % k] = {"value": n / dt, "ms_per_call": dt * 1e3,
                          "roofline": {"bound": "hbm", "achieved": alg / dt / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / dt / 1e9 / HBM_PEAK_GBS,
                                       "traffic": tr, "traffic_source": src, "algorithmic_bytes_per_launch": alg,
                                       "kernel": "e3d_normals_knn whole call (grid build + k-NN + covariance + eigenvector)"}}
    out["value"] = out["k32"]["value"]
    # the same call on a scan sampled as a scanner samples (rays uniform in angle: density ~ cos / range^2; synth.make_scan_angular) --
    # what real scans look like; the figure above is the uniform-per-area scan of the earlier rounds
    xyz_a, _, _ = synth.make_scan_angular(n, origin, yaw, 1234, device=dev)
    xyz_a = xyz_a.contiguous()
    torch.cuda.synchronize()
    out["scanner_sampled"] = {"note": "density ~ cos(incidence) / range^2: %.0f %% of the points within 2 m of the scanner" % (100.0 * float((xyz_a.norm(dim=1) < 2.0).float().mean()))}
    for k in (32, 8):
        def call_a():
            r = capi.lib().e3d_normals_knn(C.c_void_p(xyz_a.data_ptr()), n, k, C.c_void_p(vp.ctypes.data), C.c_void_p(on.data_ptr()), C.c_void_p(oc.data_ptr()), None)
            assert r == 0, capi.lib().e3d_last_error()
        call_a()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            call_a()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        out["scanner_sampled"]["k%d" % k] = {"value": n / dt, "ms_per_call": dt * 1e3, "frac": n * (12 * k + 28) / dt / 1e9 / HBM_PEAK_GBS}
    del xyz_a
    if not args.no_cpu_baseline:
        from oracle import binding as ob
        x = xyz[:, 0]
        xs = torch.sort(x[torch.randperm(n, device=dev)[:min(n, 2_000_000)]]).values
        i0 = int(0.4 * len(xs))
        lo, hi = float(xs[i0]), float(xs[min(len(xs) - 1, i0 + max(1, int(len(xs) * 1_000_000 / n)))])
        sub = xyz[(x >= lo) & (x < hi)].cpu().numpy()
        cores = ob.set_num_threads(ob.effective_cores())
        t0 = time.perf_counter()
        ob.normals(sub, k=32)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": len(sub) / dtc, "unit": "normals/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
                               "sample": "k = 32 on %d points (slab x in [%.2f, %.2f) of the same scan): kd-tree build + k-search + two-pass "
                                         "covariance, OpenMP over points like NormalEstimationTwoPassOMP, %.1f s" % (len(sub), lo, hi, dtc)}
        out["speedup_vs_cpu"] = out["k32"]["value"] / out["cpu_baseline"]["value"]
    del xyz, on, oc
    torch.cuda.empty_cache()
    return out


def spawn(args):
    """--gpus N without a launcher: start the N ranks ourselves (same command line) and relay rank 0's JSON line."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=0, help="points per scan of the headline leg (default 50 M x gpus)")
    ap.add_argument("--distance", type=float, default=0.01)
    ap.add_argument("--room-scale", type=float, default=0.0, help="stretch of the headline room's floor plan (default: sqrt(points / 50 M) at N > 1, else 1)")
    ap.add_argument("--perturb", type=float, default=2.5, help="scale of the headline scene's initial misalignment (synth.perturbation): 2.5 = 2.5 degrees and "
                    "(5, -2.5, 2.5) cm, the smallest start from which the run still needs the 25 outer iterations of --warmup 5 --steps 20 "
                    "(it converges in iteration 25; tools/icp_converge.py: 2.2 -> 23, 2.4 -> 24, 2.5 -> 25, 3.0 -> 29)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-whole-run", action="store_true", help="skip the add_cloud -> convergence run of the headline leg (N = 1)")
    ap.add_argument("--no-scale-model", action="store_true", help="skip the rank-0-of-8 run of the all-pairs leg (N = 1)")
    ap.add_argument("--scale-model-world", type=int, default=8)
    ap.add_argument("--detail", default=None, help="where the full result goes (default: bench_detail.json next to bench.py, and gpurun_out/ if it exists)")
    ap.add_argument("--no-reg", action="store_true", help="skip the ImageRegistrator leg (N = 1 only)")
    ap.add_argument("--no-normals", action="store_true", help="skip the normal-estimation leg (N = 1 only)")
    ap.add_argument("--no-allpairs", action="store_true", help="skip the all-pairs scaling leg")
    ap.add_argument("--no-partial", action="store_true", help="skip the partial-overlap ICP leg (N = 1 only)")
    ap.add_argument("--partial-only", action="store_true", help="profiling: the headline leg itself on the partial-overlap room (N = 1)")
    ap.add_argument("--no-regression", action="store_true", help="skip the regression leg (round 4's scene; N = 1 only)")
    ap.add_argument("--no-scanner", action="store_true", help="skip the scanner-sampled ICP leg (N = 1 only)")
    ap.add_argument("--variant-only", choices=["regression", "scanner"], default=None, help="profiling: one of the headline's variant legs alone (N = 1)")
    ap.add_argument("--only", choices=["reg", "normals", "allpairs"], default=None,
                    help="profiling / development: run one of the secondary legs alone and print its JSON (N = 1; allpairs also N > 1)")
    ap.add_argument("--allpairs-scans", type=int, default=16)
    ap.add_argument("--allpairs-points", type=int, default=10_000_000)
    ap.add_argument("--allpairs-distance", type=float, default=0.02)
    ap.add_argument("--reg-images", type=int, default=23)
    ap.add_argument("--reg-points", type=int, default=10_000_000)
    ap.add_argument("--normals-points", type=int, default=20_000_000)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn(args))                           # never a silent 1-GPU run
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible)")
    if world > torch.cuda.device_count():
        raise SystemExit("--gpus %d: only %d HIP device(s) visible (one rank per GPU)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    e3d = importlib.import_module("dataset-pipeline_amd")
    synth = importlib.import_module("dataset-pipeline_amd.synth")
    if e3d.lib().e3d_init(local_rank) < 1:
        raise SystemExit("libe3dhip: no device")
    R = Ranks(rank, world, local_rank, e3d)

    if args.only:
        if args.only == "allpairs":
            res = leg_allpairs(e3d, synth, R, args, dev)
        elif rank == 0 and world == 1:
            res = leg_image_registrator(e3d, synth, args, dev) if args.only == "reg" else leg_normals(e3d, synth, args, dev)
        else:
            res = None
        if rank == 0:
            print(json.dumps(res))
        R.close()
        return
    if args.partial_only and world == 1:
        if rank == 0:
            print(json.dumps(leg_terrace(e3d, synth, R, args, dev, partial=True)))
        R.close()
        return
    if args.variant_only and world == 1:
        if rank == 0:
            print(json.dumps(leg_terrace(e3d, synth, R, args, dev, variant=args.variant_only)))
        R.close()
        return
    out = leg_terrace(e3d, synth, R, args, dev)
    if world == 1 and not args.no_regression:
        out["regression"] = leg_terrace(e3d, synth, R, args, dev, variant="regression")
    if world == 1 and not args.no_scanner:
        out["scanner_sampled"] = leg_terrace(e3d, synth, R, args, dev, variant="scanner")
    if world == 1 and not args.no_partial:
        out["partial_overlap"] = leg_terrace(e3d, synth, R, args, dev, partial=True)
    if not args.no_allpairs:
        out["allpairs"] = leg_allpairs(e3d, synth, R, args, dev)
    if rank == 0 and world == 1:
        if not args.no_reg:
            out["image_registrator"] = leg_image_registrator(e3d, synth, args, dev)
        if not args.no_normals:
            out["normal_estimation"] = leg_normals(e3d, synth, args, dev)
    if rank == 0:
        emit(out, args)
    R.close()


def _r(v, digits=4):
    """float -> `digits` significant digits (the compact line carries summaries, the detail file the full values)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    try:
        return float("%.*g" % (digits, float(v)))
    except (TypeError, ValueError):
        return v


def compact_line(d, detail_path="bench_detail.json"):
    """The ONE line the driver parses (last line of stdout): the contract's headline keys, `config`, `dtype`, the dominant kernel's
    `roofline`, `cpu_baseline`, and one summary object per secondary leg.  Everything else -- kernel tables, per-step arrays, prose --
    stays in the detail file.  Bounded: tests/test_bench_line.py asserts < 4096 bytes on a recorded run."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")
    o = {k: (_r(d[k], 6) if k in ("value", "ms_per_step") else d[k]) for k in keep if k in d}
    o["dtype"] = "f32 (f64 accumulation)" if "f32" in str(d.get("dtype", "")) else d.get("dtype")
    o["data"] = "synthetic"
    c = d.get("config", {})
    o["config"] = {"workload": "ICPScanAligner 2 scans x %s points, -d %s (BASELINE.json configs[1] shape), one outer iteration per step"
                               % (c.get("points_per_scan"), str(c.get("workload", "")).split("-d ")[-1].split(",")[0] if "-d " in str(c.get("workload", "")) else "?"),
                   "points_per_scan": c.get("points_per_scan"), "scans": c.get("scans"), "parallelism": str(c.get("parallelism", "")).split(" ")[0],
                   "initial_misalignment_scale": c.get("initial_misalignment_scale")}
    for k in ("steps_requested", "converged_at_iteration"):
        o[k] = d.get(k)
    for k in ("ms_per_step_settling", "ms_per_step_steady", "lm_passes_per_iter"):
        o[k] = _r(d.get(k))
    rf = d.get("roofline", {})
    o["roofline"] = {"bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"), "frac": _r(rf.get("frac")),
                     "frac_of_bytes_moved": _r(rf.get("frac_of_bytes_moved")), "traffic": _r(rf.get("traffic")), "traffic_source": rf.get("traffic_source"),
                     "kernel": str(rf.get("kernel", "")).split(":")[0], "algorithmic_bytes_per_launch": _r(rf.get("algorithmic_bytes_per_launch")),
                     "avg_launch_ms": _r(rf.get("avg_launch_ms"))}
    # the runner-up beside the dominant kernel: on the default scene the exact search of the first iterations (k_nn_rows) and the fused
    # cost + Gramian pass (k_lm_pass) are within a few per cent of each other in summed time, and which one leads changes from run to run
    ks = rf.get("kernels", {})
    groups = [k for k in ("k_lm_pass", "k_lm_cost_multi", "k_nn_certify", "k_nn_bounded", "k_nn_rows", "k_corr_update") if k in ks]
    dom = str(rf.get("kernel", "")).split(":")[0]
    if dom in ks:
        o["roofline"]["summed_ms_per_iter"] = _r(ks[dom].get("summed_ms_per_iter"))
    rest = sorted((k for k in groups if k != dom), key=lambda k: -(ks[k].get("summed_ms_per_iter") or 0.0))
    if rest:
        k2 = ks[rest[0]]
        o["roofline_runner_up"] = {"kernel": rest[0], "summed_ms_per_iter": _r(k2.get("summed_ms_per_iter")), "achieved": _r(k2.get("GBs")),
                                   "frac": _r((k2.get("GBs") or 0.0) / HBM_PEAK_GBS), "frac_of_bytes_moved": _r(k2.get("frac_moved")),
                                   "avg_launch_ms": _r(k2.get("avg_launch_ms"))}
    cb = d.get("cpu_baseline")
    if cb:
        o["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                             "sample": "median of 3 iterations, slab of both scans (%.1f %% of the points), from the GPU run's poses after warm-up"
                                       % (100.0 * cb.get("sample_fraction_of_configs1", 0.0)),
                             "ms_per_iter": _r(cb.get("ms_per_iter")), "all_cores_value": _r(cb.get("all_core", {}).get("value")), "host_cores_available": cb.get("host_cores_available")}
        o["speedup_vs_cpu_iteration_rate"] = _r(d.get("speedup_vs_cpu_iteration_rate"))
    w = d.get("whole_run")
    if w:
        o["whole_run"] = {"wall_s": _r(w["wall_s"]), "iterations_run": w["iterations_run"], "converged_at_iteration": w["converged_at_iteration"],
                          "add_cloud_s": _r(w["add_cloud_s"]), "first_iteration_ms": _r(w["first_iteration_ms"]), "correspondences_per_s": _r(w["correspondences_per_s"])}
    if "comm" in d:
        cm = d["comm"]
        o["comm"] = {k: _r(cm[k]) for k in ("allreduce_calls_per_iter", "allreduce_bytes_per_call", "allreduce_ms_per_iter_max_over_ranks", "allreduce_ms_per_iter_min_over_ranks") if k in cm}
    legs = {}
    p = d.get("partial_overlap")
    if p:
        legs["partial_overlap"] = {"value": _r(p["value"]), "ms_per_step": _r(p["ms_per_step"]), "steps": p.get("steps"), "converged_at_iteration": p.get("converged_at_iteration"),
                                   "matched_fraction": _r(p.get("config", {}).get("matched_fraction")), "ms_per_step_steady": _r(p.get("ms_per_step_steady")),
                                   "roofline_kernel": str(p.get("roofline", {}).get("kernel", "")).split(":")[0], "roofline_frac": _r(p.get("roofline", {}).get("frac"))}
    rg = d.get("regression")
    if rg:
        legs["regression"] = {"scene": "round 4's start (perturb 1.0)", "ms_per_step": _r(rg["ms_per_step"]), "steps": rg.get("steps"), "converged_at_iteration": rg.get("converged_at_iteration"),
                              "round4_ms_per_step": 8.59}
    sc = d.get("scanner_sampled")
    if sc:
        cq, cq0 = sc.get("candidates_per_query") or {}, d.get("candidates_per_query") or {}
        legs["scanner_sampled"] = {"ms_per_step": _r(sc["ms_per_step"]), "value": _r(sc["value"]), "steps": sc.get("steps"), "converged_at_iteration": sc.get("converged_at_iteration"),
                                   "vs_uniform_headline": _r(sc["ms_per_step"] / d["ms_per_step"]) if d.get("ms_per_step") else None,
                                   "roofline_kernel": str(sc.get("roofline", {}).get("kernel", "")).split(":")[0], "roofline_frac": _r(sc.get("roofline", {}).get("frac")),
                                   "candidates_per_query": {"mean": _r(cq.get("mean")), "p99": _r(cq.get("p99")), "uniform_scan_mean": _r(cq0.get("mean"))}}
    a = d.get("allpairs")
    if a:
        la = {"value": _r(a["value"]), "unit": a["unit"], "scaling": a["scaling"], "n_gpus": a.get("n_gpus"), "steps": a["steps"], "ms_per_iter": _r(a["ms_per_iter"]),
              "ms_per_iter_settling": _r(a.get("ms_per_iter_settling")), "ms_per_iter_steady": _r(a.get("ms_per_iter_steady")),
              "workload": str(a.get("config", {}).get("workload", "")).split(" points")[0].replace(" synthetic", "") + " points, all pairs",
              "roofline_kernel": str(a.get("roofline", {}).get("kernel", "")).split(":")[0], "roofline_frac": _r(a.get("roofline", {}).get("frac")),
              "roofline_frac_of_bytes_moved": _r(a.get("roofline", {}).get("frac_of_bytes_moved"))}
        if "nn_launches_per_iter" in a:
            la["nn_launches_per_iter"] = {k: _r(v) for k, v in a["nn_launches_per_iter"].items() if k in ("rows", "radix_sorts", "all_kernels", "batches")}
        sm = a.get("scale_model")
        if sm:
            la["scale_model"] = {"world": sm["world"], "ms_per_iter_n1": _r(sm["ms_per_iter_n1"]), "ms_per_iter_as_rank0_of_world": _r(sm["ms_per_iter_as_rank0_of_world"]),
                                 "modelled_speedup": _r(sm["modelled_speedup"]), "non_dividing_ms_per_iter": _r(sm["non_dividing_ms_per_iter"]),
                                 "steady_modelled_speedup": _r(sm["steady"]["modelled_speedup"]), "steady_non_dividing_ms_per_iter": _r(sm["steady"]["non_dividing_ms_per_iter"]),
                                 "poses_equal_single_gpu_run": sm.get("poses_equal_single_gpu_run")}
        if "comm" in a:
            la["comm"] = {k: _r(a["comm"][k]) for k in ("allreduce_calls_per_iter", "allreduce_ms_per_iter_max_over_ranks", "allreduce_ms_per_iter_min_over_ranks") if k in a["comm"]}
        legs["allpairs"] = la
    g = d.get("image_registrator")
    if g:
        rf2 = g.get("roofline", {})
        p2 = next((v for k, v in rf2.items() if k.startswith("k_reg_pass2")), {})
        narrow = "NARROWER" in str(g.get("dtype", ""))
        lg = {"metric": g.get("metric"), "value": _r(g.get("value")), "unit": g.get("unit"),
              "dtype": "f32 rows; H, b: f32 chains added into f64 (opt-in, narrower)" if narrow else "f32 rows; H, b: exact products, f64 sums",
              "workload": str(g.get("config", {}).get("workload", "")).split(" (BASELINE")[0], "accumulate_ms_all_images": _r(g.get("accumulate_ms_all_images")),
              "ms_per_run_iteration": _r(g.get("ms_per_run_iteration")), "pass1_frac": _r(rf2.get("k_reg_pass1", {}).get("frac")),
              "pass2_kernel": next((k for k in rf2 if k.startswith("k_reg_pass2")), None), "pass2_avg_launch_ms": _r(p2.get("avg_launch_ms")), "pass2_frac": _r(p2.get("frac"))}
        if "cpu_baseline" in g:
            lg["cpu_baseline"] = {k: _r(g["cpu_baseline"][k]) for k in ("value", "unit", "cores", "kind")}
        legs["image_registrator"] = lg
    nl = d.get("normal_estimation")
    if nl:
        ln = {"points": nl["points"], "unit": nl["unit"]}
        for k in ("k32", "k8"):
            if k in nl:
                ln[k] = {"value": _r(nl[k]["value"]), "ms_per_call": _r(nl[k]["ms_per_call"]), "frac": _r(nl[k]["roofline"]["frac"]),
                         "traffic_over_algorithmic": _r(nl[k]["roofline"]["traffic"] / nl[k]["roofline"]["algorithmic_bytes_per_launch"]) if nl[k]["roofline"].get("traffic") else None}
        ss = nl.get("scanner_sampled", {})
        ln["scanner_sampled_ms"] = {k: _r(ss[k]["ms_per_call"]) for k in ("k32", "k8") if k in ss}
        if "cpu_baseline" in nl:
            ln["cpu_baseline"] = {k: _r(nl["cpu_baseline"][k]) for k in ("value", "unit", "cores", "kind")}
        legs["normal_estimation"] = ln
    if legs:
        o["legs"] = legs
    o["detail"] = detail_path
    return o


def emit(out, args):
    """The full result -> the detail file(s); the compact line -> the LAST line of stdout."""
    paths = [args.detail] if args.detail else [os.path.join(ROOT, "bench_detail.json")]
    if not args.detail and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(out, f)
        except OSError as e:                                       # a read-only tree must not cost the line
            print("bench.py: could not write %s: %s" % (pth, e), file=sys.stderr)
    sys.stdout.flush()
    print(json.dumps(compact_line(out, os.path.relpath(paths[0], ROOT)), separators=(",", ":")), flush=True)


if __name__ == "__main__":
    main()
