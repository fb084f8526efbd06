"""Two PROCESSES, one torch.distributed group (gloo), the HIP library on every rank: the process-per-GPU layout of bench.py
and the tools, with both ranks on the one GPU of the test box (RCCL refuses two ranks on one device, so the exchange is gloo;
dist.attach / dist.attach_reg, the slices, the exchanges and the decisions taken on every rank are the ones of an N-GPU run).
The thread-sharded tests of test_gpu_distributed.py drive the same library code from one process with an in-process all-reduce;
here nothing is shared between the ranks but the process group."""
import importlib
import os
import sys

import numpy as np
import pytest

from conftest import pose_error

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ICP_ARGS = dict(d=0.12, iters=4, thr=1e-9)


def _icp_clouds(synth):
    scans = synth.make_scene(3, 30000, seed=21)
    return [(s["xyz"].numpy(), s["normals"].numpy(), s["T_init"], i == 2) for i, s in enumerate(scans)]


def _icp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        e3d = importlib.import_module("dataset-pipeline_amd")
        d = importlib.import_module("dataset-pipeline_amd.dist")
        synth = importlib.import_module("dataset-pipeline_amd.synth")
        icp = e3d.PointToPlaneICP(device=0)
        ids = [icp.add_point_cloud(*c) for c in _icp_clouds(synth)]          # every rank holds every cloud; the queries are sliced
        d.attach(icp)
        conv = icp.run(ICP_ARGS["d"], 0, ICP_ARGS["iters"], ICP_ARGS["thr"], False)
        q.put((rank, bool(conv), [tuple(int(v) for v in r[:4]) for r in icp.pair_records()],
               [icp.get_result_global_T_cloud(i) for i in ids if i >= 0], int(icp.iter_records()[0]["queries"])))
    except Exception as ex:  # noqa: BLE001
        q.put((rank, "error: %r" % (ex,), None, None, None))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, port):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=280) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(300)
def test_two_processes_icp_equals_single_rank(e3d, synth):
    ref = e3d.PointToPlaneICP(device=0)
    ids = [ref.add_point_cloud(*c) for c in _icp_clouds(synth)]
    conv_ref = ref.run(ICP_ARGS["d"], 0, ICP_ARGS["iters"], ICP_ARGS["thr"], False)
    ref_counts = [tuple(int(v) for v in r[:4]) for r in ref.pair_records()]
    ref_T = [ref.get_result_global_T_cloud(i) for i in ids if i >= 0]
    res = _spawn(_icp_worker, 2, 31500 + os.getpid() % 1000)
    assert all(not isinstance(r[1], str) for r in res), res
    for rank, conv, counts, Ts, _ in res:
        assert conv == conv_ref and counts == ref_counts                          # global counts on every rank
        for T, Tr in zip(Ts, ref_T):
            ang, tr = pose_error(T, Tr)
            assert ang <= 1e-5 and tr <= 1e-4
    for Ta, Tb in zip(res[0][3], res[1][3]):
        assert np.array_equal(Ta, Tb)                                             # the ranks end bit-identical
    q0 = ref.iter_records()[0]["queries"]
    assert res[0][4] + res[1][4] == q0 and abs(res[0][4] - res[1][4]) <= 8        # each rank searched its half of the queries


def _reg_scene():
    from reg_util import make_multi_image_scene
    return make_multi_image_scene(n_points=6000, n_images=4, seed=8, perturb=0.006, model=2)


def _reg_build(e3d, M, shard=None):
    P = e3d.RegProblem(e3d.default_reg_params(image_scale_count=M["n_levels"], point_neighbor_count=M["K"]))
    rank, world = (None, 1)
    if shard is not None:
        d, rank, world = shard
        d.attach_reg(P)
    P.set_intrinsics(0, M["width"], M["height"], M["params"], 0, M["n_levels"], camera_type=2)
    P.set_point_scale(0, M["pts"], M["point_radius"], M["nbr"], M["fixed_desc"])
    P.set_splat_points(M["pts"])
    for i, im in enumerate(M["images"]):
        owned = rank is None or (i % world) == rank
        P.set_image(i, 0, im["pyr"] if owned else None)
        P.set_image_pose(i, im["q_init"], im["t_init"])
    return P


def _reg_worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        e3d = importlib.import_module("dataset-pipeline_amd")
        d = importlib.import_module("dataset-pipeline_amd.dist")
        M = _reg_scene()
        P = _reg_build(e3d, M, (d, rank, world))
        r = P.run_on_current_scale(5, 0.0, 15, False)
        q.put((rank, (bool(r[0]), float(r[1]), int(r[2])), [tuple(np.asarray(v) for v in P.get_image_pose(i)) for i in range(len(M["images"]))],
               np.asarray(P.intrinsics_level(0, 0)[2])))
    except Exception as ex:  # noqa: BLE001
        q.put((rank, "error: %r" % (ex,), None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_processes_image_registration_equals_single_rank(e3d):
    from test_gpu_reg import _pose_delta
    M = _reg_scene()
    ref = _reg_build(e3d, M)
    r_ref = ref.run_on_current_scale(5, 0.0, 15, False)
    res = _spawn(_reg_worker, 2, 32500 + os.getpid() % 1000)
    assert all(not isinstance(r[1], str) for r in res), res
    for rank, r, poses, intr in res:
        assert r[0] == bool(r_ref[0]) and r[2] == r_ref[2]                        # same convergence flag and iteration count
        assert abs(r[1] - r_ref[1]) <= 1e-6 * r_ref[1]
        for i, (qv, tv) in enumerate(poses):
            ang, tr = _pose_delta(qv, tv, *ref.get_image_pose(i))
            assert ang <= 1e-5 and tr <= 1e-5
    for (qa, ta), (qb, tb) in zip(res[0][2], res[1][2]):
        assert np.array_equal(qa, qb) and np.array_equal(ta, tb)                  # bit-identical state on both ranks
    assert np.array_equal(res[0][3], res[1][3])
