"""Data-parallel semantics on the real kernels: two ranks (two processes on one card, gloo — RCCL refuses two
ranks per device) each taking half of a global batch must produce, after the exchanges of wseg_amd/loss_hip.py (prototype
candidates, hard-pixel records) and the averaged gradient all-reduce, the SAME loss scalars and the SAME gradient as one
process running the whole batch — the reference computes its loss on the gathered batch (contrast_train.py:108)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b):
    for name, x, y in zip(a["names"], a["scalars"], b["scalars"]):
        assert abs(float(x) - float(y)) <= 2e-5 * max(1.0, abs(float(x))), (str(name), float(x), float(y))
    scale = np.abs(a["grad"]).max()
    assert np.abs(a["grad"] - b["grad"]).max() <= 2e-4 * scale, (np.abs(a["grad"] - b["grad"]).max(), scale)
    assert abs(float(a["gnorm"]) - float(b["gnorm"])) <= 1e-4 * float(a["gnorm"])


def test_rccl_exchange_path_one_rank(tmp_path):
    """The exchanges through the real backend: a one-rank RCCL group with WSEG_FORCE_DIST=1 runs both all-gathers and the
    bucketed asynchronous gradient all-reduce on RCCL's streams; the step must equal the local path's."""
    env = dict(os.environ, WSEG_INTRA_KEY_SEED="5", PYTHONPATH=ROOT)
    worker = os.path.join(ROOT, "tests", "ddp_worker.py")
    one, forced = str(tmp_path / "one.npz"), str(tmp_path / "forced.npz")
    subprocess.run([sys.executable, worker, one, "2", "128"], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, worker, forced, "2", "128"], check=True, env=dict(env, WSEG_FORCE_DIST="1"), timeout=600)
    _same(np.load(one), np.load(forced))


def test_two_ranks_equal_one_global_batch(tmp_path):
    env = dict(os.environ, WSEG_INTRA_KEY_SEED="5", WSEG_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    worker = os.path.join(ROOT, "tests", "ddp_worker.py")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    n_global, size = 4, 128
    subprocess.run([sys.executable, worker, one, str(n_global), str(size)], check=True, env=env, timeout=600)
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", "29533", worker, two, str(n_global), str(size)], check=True, env=env, timeout=600)
    _same(np.load(one), np.load(two))
