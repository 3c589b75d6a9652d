"""CLI of the reference's contrast_train.py (same flags and defaults, :37-54) on the MI355X path.

    python -m wseg_amd.contrast_train --weights <ckpt.pth|procedural> [--synthetic N] ...
    torchrun --nproc-per-node 8 --master-addr 127.0.0.1 -m wseg_amd.contrast_train ...   (one process per GPU, RCCL)

`--batch_size` keeps the reference's meaning — the GLOBAL batch (nn.DataParallel splits it over the GPUs, contrast_train.py:80-90,108):
every rank takes batch_size / world images per step and max_step = (len(dataset) // batch_size) * max_epoches (:88).
Additive flags only: --labels (path of cls_labels.npy/.npz), --synthetic N (N procedural images instead of
VOC), --precision, --rng_parity, --device_augment (the transform chain of :64-75 on the GPU, wseg_amd/augment.py: DataLoader workers only
decode the JPEGs and draw the random parameters), --seed (rank r seeds torch with seed + r: Dropout2d masks and hard-pixel keys differ per rank).  Logging keeps the reference's line format and its
`imps` definition (images, not views, per second; :413-420); tensorboardX is not available offline.
"""
import argparse
import importlib
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist

from . import data as wdata
from . import synth
from .optim import PolyOptimizer
from .train import Trainer

KEYS = ['loss', 'loss_cls', 'loss_er', 'loss_ecr', 'loss_nce', 'loss_intra_nce', 'loss_cross_nce', 'loss_cross_nce2']


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--batch_size", default=8, type=int)
    parser.add_argument("--max_epoches", default=8, type=int)
    parser.add_argument("--network", default="wseg_amd.resnet38_contrast", type=str)
    parser.add_argument("--lr", default=0.01, type=float)
    parser.add_argument("--num_workers", default=8, type=int)
    parser.add_argument("--wt_dec", default=5e-4, type=float)
    parser.add_argument("--train_list", default="voc12/train_aug.txt", type=str)
    parser.add_argument("--val_list", default="voc12/val.txt", type=str)
    parser.add_argument("--session_name", default="resnet38", type=str)
    parser.add_argument("--crop_size", default=448, type=int)
    parser.add_argument("--weights", required=True, type=str)
    parser.add_argument("--voc12_root", default='VOC2012', type=str)
    parser.add_argument("--tblog_dir", default='./tblog', type=str)
    parser.add_argument("--bg_threshold", default=0.20, type=float)
    parser.add_argument("--labels", default="voc12/cls_labels.npy", type=str)
    parser.add_argument("--synthetic", default=0, type=int)
    parser.add_argument("--precision", default=None, choices=[None, "bf16", "fp32", "bf16x3"])
    parser.add_argument("--rng_parity", action="store_true")
    parser.add_argument("--seed", default=0, type=int)
    parser.add_argument("--device_augment", action="store_true",
                        help="training augmentation on the GPU (bit-exact against this package's PIL restatement; parity with the reference's torchvision transforms is unpinned: not importable offline)")
    args = parser.parse_args(argv)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    if args.batch_size % world != 0:
        raise SystemExit(f"--batch_size {args.batch_size} is the global batch (reference semantics) and must divide by the {world} ranks")
    local_batch = args.batch_size // world
    torch.manual_seed(args.seed + rank)                     # torch's default seed is a constant: without this every rank would draw the
    np.random.seed(args.seed + rank)                        # same Dropout2d masks and hard-pixel keys for its different images
    os.makedirs(os.path.join('result', args.session_name), exist_ok=True)
    if rank == 0:
        print(vars(args))

    Net = getattr(importlib.import_module(args.network), 'Net')
    model = Net(precision=args.precision) if args.precision else Net()

    aug = None
    if args.synthetic:
        n_img = args.synthetic
        loader = None
    elif args.device_augment:
        from . import augment as waug
        ds = waug.VOC12ClsDatasetRaw(args.train_list, args.voc12_root, args.labels, args.crop_size)
        n_img = len(ds)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True) if world > 1 else None
        loader = torch.utils.data.DataLoader(ds, batch_size=local_batch, shuffle=sampler is None, sampler=sampler, collate_fn=waug.collate,
                                             num_workers=args.num_workers, pin_memory=True, drop_last=True,
                                             worker_init_fn=lambda wid: (np.random.seed(args.seed + 1 + wid + 1000 * rank), random.seed(args.seed + 1 + wid + 1000 * rank)))
        aug = waug.DeviceAugment(dev, args.crop_size)
    else:
        ds = wdata.VOC12ClsDataset(args.train_list, args.voc12_root, args.labels, wdata.train_transform(model, args.crop_size))
        n_img = len(ds)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True) if world > 1 else None
        loader = torch.utils.data.DataLoader(ds, batch_size=local_batch, shuffle=sampler is None, sampler=sampler,
                                             num_workers=args.num_workers, pin_memory=True, drop_last=True,
                                             worker_init_fn=lambda wid: np.random.seed(args.seed + 1 + wid + 1000 * rank))
    steps_per_epoch = n_img // args.batch_size
    max_step = steps_per_epoch * args.max_epoches

    param_groups = model.get_parameter_groups()
    optimizer = PolyOptimizer([
        {'params': param_groups[0], 'lr': args.lr, 'weight_decay': args.wt_dec},
        {'params': param_groups[1], 'lr': 2 * args.lr, 'weight_decay': 0},
        {'params': param_groups[2], 'lr': 10 * args.lr, 'weight_decay': args.wt_dec},
        {'params': param_groups[3], 'lr': 20 * args.lr, 'weight_decay': 0}
    ], lr=args.lr, weight_decay=args.wt_dec, max_step=max_step)

    if args.weights == "procedural":
        weights_dict = synth.procedural_state_dict(0, device=dev)
    elif args.weights[-7:] == '.params':
        raise SystemExit("mxnet .params conversion needs mxnet (absent offline): convert to a .pth state_dict first")
    else:
        weights_dict = torch.load(args.weights, map_location="cpu", weights_only=True)
    model.load_state_dict(weights_dict, strict=False)
    model.cuda(dev)
    model.train()
    trainer = Trainer(model, optimizer, args.bg_threshold, random.Random(args.seed * 1000003 + rank), args.rng_parity)

    sums = {k: 0.0 for k in KEYS}
    cnt = 0
    t_start = stage_start = time.time()
    for ep in range(args.max_epoches):
        if loader is not None and world > 1:
            loader.sampler.set_epoch(ep)
        it = iter(loader) if loader is not None else None
        if aug is not None:
            it = aug.batches(it)                            # (decode workers -> device augmentation on the caller's stream, in front of the step; overlap=True: one batch ahead on a side stream)
        def fetch(itn):
            if itn >= steps_per_epoch:
                return None
            if it is None:
                sd = (ep * steps_per_epoch + itn) * world + rank
                return synth.synthetic_images(local_batch, args.crop_size, seed=sd, device=dev), synth.synthetic_labels(local_batch, seed=sd, device=dev)
            if aug is not None:
                return next(it)
            pack = next(it)
            return pack[1].cuda(dev, non_blocking=True), pack[2].cuda(dev, non_blocking=True)

        cur = fetch(0)
        for itn in range(steps_per_epoch):
            # one batch ahead: the step is told the NEXT images, whose weight-independent prefix (conv1a, the frozen b2 blocks) it computes inside
            # its own loss phase (Trainer.step next_img1)
            nxt = fetch(itn + 1)
            img, lab = cur
            losses = trainer.step(img, lab, next_img1=nxt[0] if nxt is not None else None)
            cur = nxt
            cnt += 1
            for k in KEYS:                                  # device-side accumulation: no per-step host sync
                sums[k] = sums[k] + losses[k]
            if (optimizer.global_step - 1) % 50 == 0 and rank == 0:
                elapsed = time.time() - t_start
                est_finish = t_start + elapsed / (optimizer.global_step / max_step)
                vals = tuple(float(sums[k]) / cnt for k in KEYS)
                print('Iter:%5d/%5d | ' % (optimizer.global_step - 1, max_step),
                      'loss: %.4f | loss_cls: %.4f | loss_er: %.4f | loss_ecr: %.4f | '
                      'loss_nce: %.4f | loss_intra_nce: %.4f | loss_cross_nce: %.4f | loss_cross_nce2: %.4f' % vals,
                      'imps:%.1f | ' % ((itn + 1) * args.batch_size / (time.time() - stage_start)),
                      'Fin:%s | ' % time.ctime(int(est_finish)),
                      'lr: %.4f' % (optimizer.param_groups[0]['lr']), flush=True)
                sums = {k: 0.0 for k in KEYS}
                cnt = 0
        if rank == 0:
            print('')
        stage_start = time.time()
    if rank == 0:
        print(args.session_name)
        torch.save(model.state_dict(), os.path.join('result', args.session_name, 'contrast.pth'))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
