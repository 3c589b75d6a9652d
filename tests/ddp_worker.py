"""Worker of tests/test_gpu_ddp_equivalence.py: one data-parallel rank (gloo, all ranks on cuda:0) running the fused HIP
training step on its slice of a global batch; rank 0 saves the averaged gradient slices and the rank-averaged scalars."""
import contextlib
import io
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(out_path, n_global, size, world, rank):
    from wseg_amd import synth
    from wseg_amd.optim import PolyOptimizer
    from wseg_amd.resnet38_contrast import Net
    from wseg_amd.train import Trainer
    n = n_global // world
    sl = slice(rank * n, (rank + 1) * n)
    model = Net(precision="fp32")
    with contextlib.redirect_stdout(io.StringIO()):
        groups = model.get_parameter_groups()
    opt = PolyOptimizer([{'params': groups[0], 'lr': 0.0, 'weight_decay': 5e-4}, {'params': groups[1], 'lr': 0.0, 'weight_decay': 0},
                         {'params': groups[2], 'lr': 0.0, 'weight_decay': 5e-4}, {'params': groups[3], 'lr': 0.0, 'weight_decay': 0}],
                        lr=0.0, weight_decay=5e-4, max_step=100)
    model.load_state_dict(synth.procedural_state_dict(0))
    model.cuda()
    model.train()
    masks = [synth.synthetic_dropout_masks(n_global, 40), synth.synthetic_dropout_masks(n_global, 41)]
    model.set_dropout_masks([{k: v[sl].clone() for k, v in m.items()} for m in masks])
    tr = Trainer(model, opt, 0.20, random.Random(0), rng_parity=False,
                 bg_topk_idx=torch.arange(32, dtype=torch.int32))
    img = synth.synthetic_images(n_global, size, 9)[sl].cuda()
    lab = synth.synthetic_labels(n_global, 9)[sl].cuda()
    got = tr.step(img, lab)                              # (world > 1: all-gathers + bucketed all-reduce inside)
    eng = model._engine
    grad = eng.flat_g * (1.0 / world)                    # what the fused SGD applies (grad_scale)
    scal = torch.stack([got[k].float() for k in sorted(got)])
    if world > 1:
        dist.all_reduce(scal)
        scal /= world
    if rank == 0:
        step = max(1, grad.numel() // 200000)
        np.savez(out_path, grad=grad[::step].cpu().numpy(), gnorm=float(grad.double().norm()),
                 scalars=scal.cpu().numpy(), names=np.array(sorted(got)))


if __name__ == "__main__":
    out_path, n_global, size = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo")
    elif os.environ.get("WSEG_FORCE_DIST", "0") == "1":     # one-rank RCCL group: the real backend's call path on one GPU
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29534"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    run(out_path, n_global, size, world, rank)
    if dist.is_initialized():
        dist.destroy_process_group()
