"""Multi-rank protocol on CPU ranks (gloo, world_size 2 and 4) — SURVEY.md §8e.

What shards and what is exchanged:
  (1) gradients: every rank's flat buffer is all-reduced (SUM) and the fused SGD scales by 1/world;
  (2) prototypes: the reference takes the top-32 per class over the WHOLE batch (contrast_train.py:202-203, k is not
      scaled by N), so ranks all-gather their local top-32 candidates (value + feature row) and run the same merge.
These tests prove, with the CPU oracle as the single-process reference, that the exchange reproduces the global-batch
result exactly.  The candidate / merge functions below mirror the semantics of csrc/loss.hip's proto_candidates /
proto_merge kernels (lowest index first on ties, fully tied rows take the tie table, rank-0 first) — the kernels
themselves are checked against the same semantics on the GPU in tests/test_gpu_loss.py.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F


def local_candidates(ncam, feat, K, tie_idx):
    """ncam [n,21,npix], feat [n*npix,128] -> (val [21,K], fea [21,K,128], const [21])."""
    n, c, npix = ncam.shape
    rows = ncam.transpose(0, 1).reshape(c, -1)
    const = rows.max(1)[0] == rows.min(1)[0]
    # stable descending sort = highest value first, lowest index first on ties
    idx = torch.argsort(rows, dim=1, descending=True, stable=True)[:, :K]
    idx[const] = tie_idx
    return torch.gather(rows, 1, idx), feat[idx], const


def merge(vals, feas, consts):
    """vals [W,21,K], feas [W,21,K,128], consts [W,21] -> prototypes [21,128]."""
    W, c, K = vals.shape
    v = vals.permute(1, 0, 2).reshape(c, W * K)
    f = feas.permute(1, 0, 2, 3).reshape(c, W * K, -1)
    order = torch.argsort(v, dim=1, descending=True, stable=True)[:, :K]
    all_const = consts.all(0)
    order[all_const] = torch.arange(K)                      # fully tied everywhere: rank 0's set
    tv = torch.gather(v, 1, order)
    tf = torch.gather(f, 1, order.unsqueeze(-1).expand(-1, -1, f.shape[-1]))
    return F.normalize((tv.unsqueeze(-1) * tf).sum(1) / tv.sum(1, keepdim=True), dim=-1)


def _worker(rank, world, port, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(123)
        n_total, npix, K = 4, 256, 32
        ncam = torch.rand(n_total, 21, npix, generator=g)
        ncam[:, 0] = 0.2                                     # constant bg row (Q5)
        ncam[:, 7] = -1.0                                    # a class that never wins the CAM gate: constant row
        ncam[2:, 9] = -1.0                                   # constant on rank 1 only: must lose the merge by value
        feat = torch.randn(n_total * npix, 128, generator=g)
        tie = torch.arange(K)
        # single-process global-batch reference (what the reference's DataParallel loop computes on device 0)
        gv, gf, gc = local_candidates(ncam, feat, K, tie)
        ref = merge(gv[None], gf[None], gc[None])
        # this rank's shard (images shard contiguously: rank r owns images [r*n/W, (r+1)*n/W))
        per = n_total // world
        sl = slice(rank * per, (rank + 1) * per)
        lv, lf, lc = local_candidates(ncam[sl], feat[rank * per * npix:(rank + 1) * per * npix], K, tie)
        av = [torch.empty_like(lv) for _ in range(world)]
        af = [torch.empty_like(lf) for _ in range(world)]
        ac = [torch.empty_like(lc) for _ in range(world)]
        dist.all_gather(av, lv); dist.all_gather(af, lf); dist.all_gather(ac, lc)
        got = merge(torch.stack(av), torch.stack(af), torch.stack(ac))
        np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-6, atol=1e-7)
        # gradient protocol: SUM all-reduce + 1/world inside the optimizer == gradient of the global-batch mean
        w = torch.randn(1000, generator=torch.Generator().manual_seed(5))
        x = torch.randn(n_total, 1000, generator=torch.Generator().manual_seed(6))
        wl = w.clone().requires_grad_(True)
        ((x[sl] @ wl) ** 2).mean().backward()                 # per-rank mean over its shard
        flat = wl.grad.clone()
        dist.all_reduce(flat)
        wg = w.clone().requires_grad_(True)
        ((x @ wg) ** 2).mean().backward()
        np.testing.assert_allclose((flat / world).numpy(), wg.grad.numpy(), rtol=1e-4, atol=1e-4)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_prototype_and_gradient_exchange(tmp_path, world):
    """world 2 and world 4 (one image per rank: the class that is constant on the last two ranks only must lose the merge by value)"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
