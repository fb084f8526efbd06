"""The library's alternative data flows give the same bits.  Their switches are environment variables read once per process, so
each setting runs in its own interpreter: the certificate search of several directed pairs per host round trip against pair by
pair or one launch per pair (E3D_ICP_BATCH = 0 / 1; default 2: one launch per kernel and batch), the far lists of a batch keyed, sorted and searched in one launch each against pair by pair (E3D_NN_FAR_BATCH), the LM step's damped solves on host threads (E3D_LM_SOLVE_THREADS), the certificates' motion bound per query against the clouds' global one (E3D_NN_PERQUERY), the key kernel that settles queries with an empty 27-cell block against sorting them all (E3D_NN_PRUNE), certificates tested in every outer iteration against skipped while none holds (E3D_NN_CERT_SKIP) and against no certificates at all (E3D_NN_CERT=0: every query searched in every iteration), resident against compacted correspondence rows (E3D_ICP_RESIDENT), the speculative last LM step
(E3D_LM_SPECULATE), the row update's per-block results written by the certificate kernel for the blocks it settles whole against the update
computing them all (E3D_NN_FUSE_UPDATE = 0; E3D_NN_FUSE_GATE = 1: for every certified pair, not only the nearly settled ones), far-list queries that start the bounded search from a probe of their own half cell against sort + row kernel for all of them (E3D_NN_SEED = 0; E3D_NN_SEED_FRAC / _NEAR / _FRESH: from the first search on, other seed distances); the kNN estimator's single scan with sampled thresholds against the two-pass kernels (E3D_KNN_SINGLE), with
and without the lists the 125-cell pass starts from (E3D_KNN_SEED), the wave-per-query form of that pass (E3D_KNN_WIDE_WAVE), the sampled thresholds from the block population against the distance histogram, and deliberately poor ones (E3D_KNN_EST, E3D_KNN_EST_SCALE); (B): an iteration's cost taken from the next Apply's accumulation against the separate cost pass (E3D_REG_FUSE_COST = 0)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ICP_CODE = r"""
import importlib, json, sys
sys.path.insert(0, %r)
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
import torch
# dense enough for the certificate search, resident rows and pair batches (>= 4 points per 2 cm cell), and a search radius below the
# initial misalignment: most queries start without a partner (far lists through the row kernel, short ones through the bounded search)
scans = synth.make_scene(4, 3000000, seed=33, device=torch.device("cuda", 0))
icp = e3d.PointToPlaneICP()
ids = [icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], i == 3) for i, s in enumerate(scans)]
icp.run(0.02, 0, 8, 1e-9, False)
pairs = [[int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(r[4]).hex()] for r in icp.pair_records()]
poses = [[float(v).hex() for v in icp.get_result_global_T_cloud(i).ravel()] for i in ids if i >= 0]
rec = icp.iter_records()
print("RESULT" + json.dumps({"pairs": pairs, "poses": poses, "batches": sum(r["nn_batches"] for r in rec), "launches": sum(r["nn_kernel_launches"] for r in rec),
                             "certified": sum(r["nn_certify_queries"] for r in rec), "searched": sum(r["nn_search_queries"] + r["nn_bounded_queries"] for r in rec), "rows_queries": sum(r["nn_search_queries"] for r in rec)}))
""" % ROOT

# the seeded far lists against the ORACLE directly: every far-list query of every search probes for a seed (from the first search on,
# whatever the last search matched), small enough for the oracle's kd-tree ICP to finish in seconds
SEED_ORACLE_CODE = r"""
import importlib, json, sys
import numpy as np
sys.path.insert(0, %r)
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
from oracle import binding as ob
scans = synth.make_scene(3, 300000, seed=91, sigma=0.002, room_scale=0.25)      # ~6 points per 4 cm cell: the certificate path with its dense directory
g = e3d.PointToPlaneICP(); o = ob.OracleICP()
for i, s in enumerate(scans):
    xyz, nrm = np.asarray(s["xyz"]), np.asarray(s["normals"])
    a = g.add_point_cloud(xyz, nrm, s["T_init"], False); b = o.add_point_cloud(xyz, nrm, s["T_init"], False)
    assert a == b
cg = g.run(0.04, 0, 5, 1e-9, False); co = o.run(0.04, 0, 5, 1e-9, False)
rec = g.iter_records()
pg = [[int(r[0]), int(r[1]), int(r[2]), int(r[3])] for r in g.pair_records()]
po = [[int(r[0]), int(r[1]), int(r[2]), int(r[3])] for r in o.pair_records()]
err = 0.0
for i in range(3):
    err = max(err, float(np.abs(np.asarray(g.get_result_global_T_cloud(i), np.float64) - np.asarray(o.get_result_global_T_cloud(i), np.float64)).max()))
print("RESULT" + json.dumps({"same_return": bool(cg == co), "pairs_equal": pg == po, "n_pairs": len(pg), "pose_err": err,
                             "bounded": sum(r["nn_bounded_queries"] for r in rec), "rows": sum(r["nn_search_queries"] for r in rec)}))
""" % ROOT

KNN_CODE = r"""
import importlib, json, sys, hashlib
import numpy as np
sys.path.insert(0, %r)
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
origin, yaw = synth.SCAN_POSES[0]
xyz, _, _ = synth.make_scan_angular(300000, origin, yaw, 5)
out = {}
for k in (8, 16, 32):
    n, c, knn = e3d.normals_knn(xyz.numpy(), k, (0, 0, 0), return_knn=True)
    out[str(k)] = [hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() for a in (n, c, knn)]
print("RESULT" + json.dumps(out))
""" % ROOT


def _run(code, env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1]
    return json.loads(line[len("RESULT"):])


@pytest.mark.timeout(600)
def test_icp_data_flows_agree():
    base = _run(ICP_CODE, {})
    assert len(base["pairs"]) > 0
    assert base["batches"] > 0 and base["certified"] > 0             # the default ran batches of pairs through the one-launch kernels
    strip = lambda r: {k: r[k] for k in ("pairs", "poses")}
    noseed = _run(ICP_CODE, {"E3D_NN_SEED": "0"})
    for env in ({"E3D_ICP_BATCH": "0"}, {"E3D_ICP_BATCH": "1"}, {"E3D_LM_SPECULATE": "0"}, {"E3D_NN_PERQUERY": "0"}, {"E3D_NN_PERQUERY": "0", "E3D_ICP_BATCH": "0"},
                {"E3D_NN_PRUNE": "0"}, {"E3D_NN_PRUNE": "0", "E3D_ICP_BATCH": "0"}, {"E3D_NN_PRUNE_MIN": "1", "E3D_ICP_BATCH": "1"},
                {"E3D_NN_CERT_SKIP": "0"}, {"E3D_NN_CERT_SKIP": "0", "E3D_ICP_BATCH": "0"}, {"E3D_NN_CERT": "0"}, {"E3D_NN_CERT": "0", "E3D_NN_PRUNE": "0"},
                {"E3D_NN_FAR_BATCH": "0"}, {"E3D_NN_FAR_BATCH": "0", "E3D_NN_PRUNE": "0"}, {"E3D_NN_PRUNE_MIN": "1"}, {"E3D_LM_SOLVE_THREADS": "0"},
                {"E3D_NN_FUSE_UPDATE": "0"}, {"E3D_NN_FUSE_GATE": "1.0"}, {"E3D_NN_FUSE_GATE": "1.0", "E3D_NN_CERT_SKIP": "0"},
                {"E3D_NN_SEED": "0"}, {"E3D_NN_SEED_FRAC": "0", "E3D_NN_SEED_FRESH": "1"}, {"E3D_NN_SEED_FRAC": "0", "E3D_NN_SEED_NEAR": "1.0"},
                {"E3D_NN_SEED_FRAC": "0", "E3D_NN_PRUNE": "0"}, {"E3D_NN_SEED_FRAC": "0", "E3D_NN_CERT": "0", "E3D_NN_SEED_NEAR": "0.7"},
                {"E3D_NN_SEED_FRAC": "0", "E3D_NN_CERT_SKIP": "0"}):
        other = _run(ICP_CODE, env)
        assert strip(other) == strip(base), env                      # same kernel bodies, same sums: bit for bit
        if env == {"E3D_NN_PERQUERY": "0"}:
            # the bound per query certifies at least what the clouds' global bound certifies (rigid poses): fewer searches
            assert other["searched"] >= base["searched"], (other["searched"], base["searched"])
            print("queries searched: %d with the motion bound per query, %d with the clouds' global bound" % (base["searched"], other["searched"]))
        if env == {"E3D_NN_PRUNE": "0"}:
            # without the occupancy bits every far-list query is sorted and visited by the row kernel
            assert other["rows_queries"] > base["rows_queries"], (other["rows_queries"], base["rows_queries"])
            print("queries the row kernel visited: %d with the key kernel settling empty blocks, %d without" % (base["rows_queries"], other["rows_queries"]))
        if env == {"E3D_NN_FAR_BATCH": "0"}:
            # round 6: the far lists of a batch of pairs keyed, sorted and searched in one launch each instead of pair by pair
            # (the pair-by-pair path has no seeds: its row kernel visits what the batch's visits without them)
            assert other["launches"] > base["launches"] and other["rows_queries"] == noseed["rows_queries"], (other["launches"], base["launches"])
            print("kernel launches of the NN phase: %d with the far lists of a batch in one launch, %d pair by pair" % (base["launches"], other["launches"]))
        if env == {"E3D_NN_SEED": "0"}:
            # round 6: far-list queries with a target point close by start the bounded search from it instead of going through the sort and the row kernel
            assert other["rows_queries"] > base["rows_queries"], (other["rows_queries"], base["rows_queries"])
            print("queries the row kernel visited: %d with seeds, %d without" % (base["rows_queries"], other["rows_queries"]))
        if list(env) == ["E3D_ICP_BATCH"]:
            assert other["batches"] == 0 and other["launches"] > base["launches"], (env, other["launches"], base["launches"])
    # resident vs compacted rows: the pair records (counts, distance sums) are those of the same searches; the LM passes add the
    # same f32 terms in a different order of f64 sums (include/e3d_hip.h), so the poses agree to the tolerances of
    # test_gpu_icp.py::test_resident_rows_equal_compacted_rows, not necessarily bit for bit
    other = _run(ICP_CODE, {"E3D_ICP_RESIDENT": "0"})
    assert [p[:4] for p in other["pairs"]] == [p[:4] for p in base["pairs"]]
    for a, b in zip(other["pairs"], base["pairs"]):
        assert abs(float.fromhex(a[4]) - float.fromhex(b[4])) <= 1e-9 * max(1.0, abs(float.fromhex(b[4])))
    for pa, pb in zip(other["poses"], base["poses"]):
        for a, b in zip(pa, pb):
            assert abs(float.fromhex(a) - float.fromhex(b)) <= 2e-6


@pytest.mark.timeout(600)
def test_seeded_far_lists_match_the_oracle():
    """Far-list queries that start the bounded search from a probe of their own half cell (k_query_seed_multi), forced on for every
    search: per-pair correspondence counts of every iteration identical to the oracle's kd-tree ICP, poses within the north star's bars."""
    seeded = _run(SEED_ORACLE_CODE, {"E3D_NN_SEED_FRAC": "0", "E3D_NN_SEED_FRESH": "1"})
    plain = _run(SEED_ORACLE_CODE, {"E3D_NN_SEED": "0"})
    for r in (seeded, plain):
        assert r["same_return"] and r["pairs_equal"] and r["n_pairs"] > 0, r
        assert r["pose_err"] <= 1e-5, r
    assert seeded["bounded"] > plain["bounded"] and seeded["rows"] < plain["rows"], (seeded, plain)      # the seeds were used
    print("seeded: %d queries through the bounded search, %d through the row kernel; without seeds %d / %d"
          % (seeded["bounded"], seeded["rows"], plain["bounded"], plain["rows"]))


@pytest.mark.timeout(600)
def test_knn_scan_variants_agree():
    base = _run(KNN_CODE, {})
    for env in ({"E3D_KNN_SINGLE": "0"}, {"E3D_KNN_SEED": "0"}, {"E3D_KNN_WIDE_SPREAD": "1"}, {"E3D_KNN_WIDE_WAVE": "0"}, {"E3D_KNN_XCD": "0"},
                {"E3D_KNN_REP_STRIDE": "32", "E3D_KNN_REP_AVG": "1"}, {"E3D_KNN_EST": "0"}, {"E3D_KNN_EST_SCALE": "0.5"}, {"E3D_KNN_EST_SCALE": "3"}):
        assert _run(KNN_CODE, env) == base, env


LONG_RUN_CODE = r"""
import importlib, json, sys
sys.path.insert(0, %r)
e3d = importlib.import_module("dataset-pipeline_amd")
synth = importlib.import_module("dataset-pipeline_amd.synth")
import torch
scans = synth.make_scene(2, 3000000, seed=5, device=torch.device("cuda", 0))       # (dense enough for the certificate search: >= 4 points per 2 cm cell)
icp = e3d.PointToPlaneICP()
for s in scans:
    icp.add_point_cloud(s["xyz"], s["normals"], s["T_init"], False)
for it in range(130):
    icp.run(0.02, it, 1, 0.0, False)          # threshold 0: never converged, the poses are re-composed 130 times
rec = icp.iter_records()
print("RESULT" + json.dumps({"certified_last": rec[-1]["nn_certify_queries"], "searched_last": rec[-1]["nn_search_queries"] + rec[-1]["nn_bounded_queries"],
                             "queries_last": rec[-1]["queries"]}))
""" % ROOT


@pytest.mark.timeout(600)
def test_per_query_motion_bound_survives_a_long_run():
    """poses are re-composed in f32 every outer iteration (Tn = R * T) and never re-orthonormalised; with a fixed 4e-6
    test of L^T L = I the per-query motion bound of the certificates silently fell back to the clouds' global bound after some dozen
    iterations.  Now near-rigid poses scale the bound (pose_ortho_dev): over 130 iterations no pose update goes into the global bound
    (E3D_NN_STATS reports every one that does), and in the last iteration nearly every query is still settled by its certificate."""
    e = dict(os.environ)
    e["E3D_NN_STATS"] = "1"
    p = subprocess.run([sys.executable, "-c", LONG_RUN_CODE], env=e, capture_output=True, text=True, timeout=500)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "global motion bound" not in p.stderr, [ln for ln in p.stderr.splitlines() if "global motion bound" in ln][:3]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][-1][len("RESULT"):])
    assert r["certified_last"] == r["queries_last"] and r["searched_last"] < 0.05 * r["queries_last"], r


REG_CODE = r"""
import importlib, json, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
e3d = importlib.import_module("dataset-pipeline_amd")
from reg_util import make_multi_image_scene, plane_depth_pyramid
out = {}
for name, model, var_weight, depth_weight, iters, without in (("thin_prism", 2, 1.0, 0.0, 7, 15), ("pinhole_fixed_only", 0, 0.0, 0.0, 5, 15),
                                                              ("opencv_depth", 1, 1.0, 0.5, 5, 15), ("counter_at_its_limit", 2, 1.0, 0.0, 6, 2)):
    M = make_multi_image_scene(n_points=6000, n_images=3, seed=6, perturb=0.006, model=model)
    prm = e3d.default_reg_params(image_scale_count=M["n_levels"], point_neighbor_count=M["K"], variable_residuals_weight=var_weight,
                                 depth_residuals_weight=depth_weight)
    G = e3d.RegProblem(prm)
    G.set_intrinsics(0, M["width"], M["height"], M["params"], 0, M["n_levels"], camera_type=M["model"])
    G.set_point_scale(0, M["pts"], M["point_radius"], M["nbr"], M["fixed_desc"]); G.set_splat_points(M["pts"])
    for i, im in enumerate(M["images"]):
        G.set_image(i, 0, im["pyr"]); G.set_image_pose(i, im["q_init"], im["t_init"])
        if depth_weight > 0: G.set_depth_maps(i, plane_depth_pyramid(M, im))
    G.profile(True)
    conv, cost, its = G.run_on_current_scale(iters, 0.0, without, False)
    G.profile(False)
    poses = [[float(v).hex() for v in np.concatenate([np.ravel(a) for a in G.get_image_pose(i)])] for i in range(3)]
    w, h, pg, _ = G.intrinsics_level(0, 0)
    out[name] = {"converged": conv, "iterations": its, "cost": cost, "poses": poses, "intrinsics": [float(v).hex() for v in pg],
                 "cost_launches": G.kernel_groups.get("cost", (0, 0, 0))[1], "pass2_launches": G.kernel_groups.get("accumulate.pass2", (0, 0, 0))[1]}
print("RESULT" + json.dumps(out))
""" % (ROOT, os.path.join(ROOT, "tests"))


@pytest.mark.timeout(600)
def test_reg_cost_from_the_next_accumulation():
    """RunOnCurrentScale: an iteration's "Cost (considering occlusions)" and the next Apply's "Initial residual" are the same sums over
    the same observations at the same state (bit-equal in the oracle's run; optimizer.cc:140-147, intrinsics_and_pose_optimizer.cc:176-185).
    The default takes the cost from the next Apply's accumulation whenever the loop is certain to go on; E3D_REG_FUSE_COST = 0 runs the
    separate cost pass every iteration as the reference does.  Same iterations, same poses and intrinsics bit for bit, same optimum to
    1e-12; fewer cost launches, no more accumulations."""
    fused = _run(REG_CODE, {})
    plain = _run(REG_CODE, {"E3D_REG_FUSE_COST": "0"})
    for name in fused:
        f, p = fused[name], plain[name]
        assert (f["converged"], f["iterations"]) == (p["converged"], p["iterations"]), name
        assert f["poses"] == p["poses"] and f["intrinsics"] == p["intrinsics"], name
        assert abs(f["cost"] - p["cost"]) <= 1e-12 * abs(p["cost"]), (name, f["cost"], p["cost"])
        assert f["cost_launches"] < p["cost_launches"], (name, f["cost_launches"], p["cost_launches"])
        assert f["pass2_launches"] <= p["pass2_launches"], (name, f["pass2_launches"], p["pass2_launches"])
        print("%s: %d iterations, cost launches %d (separate pass: %d), accumulations %d (%d)"
              % (name, f["iterations"], f["cost_launches"], p["cost_launches"], f["pass2_launches"], p["pass2_launches"]))


@pytest.mark.timeout(900)
def test_randomised_jobs_match_the_oracle():
    """tests/fuzz_icp_vs_oracle.py: jobs drawn at random (2 - 4 scans, dense rooms through the certificate search and sparse ones through
    the hash-table search, scanner-shaped sampling, a fixed cloud or none, 3 - 6 outer iterations) -- per-pair correspondence counts of
    every iteration identical to the oracle's kd-tree ICP, poses within 1e-5, under the default switches, with every far-list query
    seeded, with the certificate / row-update fusion forced on and pair by pair without seeds."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_icp_vs_oracle.py"), "--trials", "6", "--seed", "7"],
                       capture_output=True, text=True, timeout=850)
    assert p.returncode == 0 and "FUZZ OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    print("\n".join(ln for ln in p.stdout.splitlines() if ln.startswith("switches")))


@pytest.mark.timeout(900)
def test_randomised_clouds_match_the_oracle_normals():
    """tests/fuzz_normals_vs_oracle.py: clouds drawn at random (room and scanner-sampled scans, clusters with far outliers, lattices
    with every distance tied, duplicated points, a thin line; k in 3 .. 70; scaled and shifted by up to 250 m) -- neighbour lists
    equal, normals and curvatures bit for bit, under the default switches and three alternative data flows of the estimator."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_normals_vs_oracle.py"), "--trials", "12", "--seed", "3"],
                       capture_output=True, text=True, timeout=850)
    assert p.returncode == 0 and "FUZZ OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
    print("\n".join(ln for ln in p.stdout.splitlines() if ln.startswith("switches")))


@pytest.mark.timeout(600)
def test_randomised_registration_problems_match_the_oracle():
    """tests/fuzz_reg_vs_oracle.py: (B) problems drawn at random over the thirteen camera classes (points, images, image size, pose error,
    colour weight) -- observation lists equal, colour update, cost and one Apply (accepted / lambda, poses) against the oracle's driver."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_reg_vs_oracle.py"), "--trials", "12", "--seed", "5"],
                       capture_output=True, text=True, timeout=550)
    assert p.returncode == 0 and "FUZZ OK" in p.stdout, p.stdout[-3000:] + p.stderr[-2000:]
