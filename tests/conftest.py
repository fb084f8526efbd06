import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _gpu_count():
    try:
        e3d = importlib.import_module("dataset-pipeline_amd")
        return max(e3d.lib().e3d_init(0), 0)
    except Exception:  # noqa: BLE001
        return 0


def pytest_collection_modifyitems(config, items):
    if _gpu_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def e3d():
    """The product package (HIP library behind the reference's class surface)."""
    return importlib.import_module("dataset-pipeline_amd")


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def rb():
    """The CPU oracle binding of path (B) (test infrastructure)."""
    from oracle import reg_binding
    reg_binding.lib()
    return reg_binding


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("dataset-pipeline_amd.synth")


def plane_case():
    """Inputs of PointToPlaneICP.PlaneCaseSuccess (src/opt/test/test_icp.cc:111-172)."""
    xs, ys = np.meshgrid(np.arange(50), np.arange(50), indexing="ij")
    xyz = np.stack([xs.ravel(), ys.ravel(), np.zeros(2500)], 1).astype(np.float32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (2500, 1))
    xyz = np.vstack([xyz, [[0, 0, 20]]]).astype(np.float32)
    n1 = np.array([1, 0, 1], np.float32)
    n1 = n1 / np.sqrt(np.float32(2))
    nrm = np.vstack([nrm, [n1]]).astype(np.float32)
    T0 = np.eye(4, dtype=np.float32)
    T1 = np.eye(4, dtype=np.float32)
    T1[0, 3] = 1
    return xyz, nrm, T0, T1


def identical_cloud_case(seed=0, n_clouds=20, n_points=50):
    """PointToPlaneICP.IdenticalCloudAlignment (src/opt/test/test_icp.cc:39-109) with an in-repo seeded generator
    (the reference's random stream is libstdc++/Eigen/libc specific; its assertion is generator independent)."""
    rng = np.random.RandomState(seed)
    P = rng.uniform(-1, 1, (n_points, 3)).astype(np.float32)
    N = rng.normal(size=(n_points, 3))
    N = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float32)
    Ts = []
    for _ in range(n_clouds):
        ax = rng.uniform(-0.05, 0.05, 3)
        while np.linalg.norm(ax) < 1e-4:
            ax = rng.uniform(-0.05, 0.05, 3)
        ax = ax / np.linalg.norm(ax)
        ang = rng.uniform(-np.pi / 18, np.pi / 18)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = R.astype(np.float32)
        T[:3, 3] = rng.uniform(-0.05, 0.05, 3).astype(np.float32)
        Ts.append(T)
    return P, N, Ts


def pose_error(Ta, Tb):
    """(rotation angle [rad], translation distance [m]) between two 4x4 poses."""
    Ra, Rb = Ta[:3, :3].astype(np.float64), Tb[:3, :3].astype(np.float64)
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    ang = float(np.arccos(np.clip(c, -1.0, 1.0)))
    # arccos loses precision near 0: use the skew part there
    S = Ra.T @ Rb
    s = 0.5 * np.linalg.norm([S[2, 1] - S[1, 2], S[0, 2] - S[2, 0], S[1, 0] - S[0, 1]])
    if s < 1e-3:
        ang = float(s)
    return ang, float(np.linalg.norm(Ta[:3, 3].astype(np.float64) - Tb[:3, 3].astype(np.float64)))
