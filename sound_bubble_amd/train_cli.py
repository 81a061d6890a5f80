#!/usr/bin/env python3
"""Training driver with the reference's CLI and experiment-JSON surface
(src/train_pt.py:37-209, src/training/tain_val.py:24-88), one process per GPU.

  python -m sound_bubble_amd.train_cli --config experiments/bubble_small_synthetic.json --run_dir runs/x
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m sound_bubble_amd.train_cli ...

Reference JSONs load unchanged: dotted class paths resolve through the `src.*` alias modules / the harness alias
table.  `--synthetic` swaps the (host-side, out-of-scope) dataset for SyntheticBubbleDataset when the data is absent.
"""
import argparse
import json
import os
import random
import shutil
import time

import numpy as np
import torch

from .harness import import_attr


def seed_all(seed):                      # src/utils.py:161-167
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def to_device(batch, device):            # tain_val.py:11-22
    if torch.is_tensor(batch):
        return batch.to(device, non_blocking=True)
    if isinstance(batch, dict):
        return {k: to_device(v, device) for k, v in batch.items()}
    if isinstance(batch, (list, tuple)):
        return [to_device(x, device) for x in batch]
    return batch


def train_epoch(hl, loader, device):     # tain_val.py:51-88 (loss.item() deferred to one sync per step)
    hl.train()
    tot, n = 0.0, 0
    for idx, batch in enumerate(loader):
        batch = to_device(batch, device)
        hl.reset_grad()
        loss, B = hl.training_step(batch, idx)
        loss.backward()
        hl.backprop()
        tot += float(loss.detach()) * B
        n += B
    return tot / max(n, 1)


def test_epoch(hl, loader, device):      # tain_val.py:24-49
    hl.eval()
    tot, n = 0.0, 0
    with torch.no_grad():
        for idx, batch in enumerate(loader):
            batch = to_device(batch, device)
            loss, B = hl.validation_step(batch, idx)
            tot += float(loss) * B
            n += B
    return tot / max(n, 1)


def make_dataset(params, key, split, synthetic, with_dis):
    if not synthetic:
        try:
            return import_attr(params[f"{key}_dataset"])(**params[f"{key}_data_args"], split=split)
        except Exception as e:                                     # missing module / data dir
            print(f"[train_cli] dataset {params.get(key + '_dataset')} unavailable ({type(e).__name__}: {e}); "
                  "falling back to SyntheticBubbleDataset")
    from .data import SyntheticBubbleDataset
    args = params.get(f"{key}_data_args", {})
    return SyntheticBubbleDataset(n_items=args.get("n_items", 64), n_samples=args.get("n_samples", 120000),
                                  with_dis_embed=with_dis, split=split)


def per_rank_batch(json_batch, world, batch_per_gpu=False):
    """The reference's nn.DataParallel SPLITS the JSON batch_size over the GPUs (hl_module:34-35), so the JSON value is
    the global batch: each of `world` ranks takes batch_size / world (default; unchanged JSONs keep their optimisation
    behaviour).  --batch_per_gpu reads it as the per-GPU batch instead (weak scaling, as bench.py measures)."""
    if batch_per_gpu or world == 1:
        return json_batch
    if json_batch % world:
        raise ValueError(f"batch_size {json_batch} is not divisible by {world} ranks (or pass --batch_per_gpu)")
    return json_batch // world


def make_loaders(data_train, data_val, params, world, rank, batch_per_gpu=False):
    """Train: DistributedSampler (equal shard sizes by padding, reshuffled every epoch through set_epoch), drop_last so
    that every rank steps the same number of equal batches (mean-of-local-means == global mean).
    Val: rank r takes items r, r + world, ... with NO padding and NO dropping -- shards may be ragged or empty; the
    harness merges (sum, count) pairs over the ranks (PLModule.sync_epoch_metrics), so the epoch means are exact."""
    nw = min(os.cpu_count() or 1, params.get("num_workers", 0), 8)
    bs = per_rank_batch(params["batch_size"], world, batch_per_gpu)
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(data_train, world, rank, shuffle=True)
        train_loader = torch.utils.data.DataLoader(data_train, batch_size=bs, sampler=sampler, num_workers=nw,
                                                   pin_memory=True, drop_last=True)
        data_val = torch.utils.data.Subset(data_val, list(range(rank, len(data_val), world)))
    else:
        train_loader = torch.utils.data.DataLoader(data_train, batch_size=bs, shuffle=True, num_workers=nw,
                                                   pin_memory=True)
    test_loader = torch.utils.data.DataLoader(data_val, batch_size=params["eval_batch_size"], num_workers=nw,
                                              pin_memory=True)
    return train_loader, test_loader


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True)
    ap.add_argument("--run_dir", required=True)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--use_nondeterministic_cudnn", action="store_true")     # accepted, no-op
    ap.add_argument("--project_name", default="AcousticBubble")
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--epochs", type=int, default=None, help="override params['epochs']")
    ap.add_argument("--batch_per_gpu", action="store_true",
                    help="read the JSON batch_size as the per-GPU batch (default: the global batch, split over the ranks "
                         "like the reference's nn.DataParallel does)")
    ap.add_argument("--wandb", action="store_true", help="log to wandb if importable (off by default: no network)")
    args = ap.parse_args(argv)

    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)
    seed_all(args.seed)
    with open(args.config, "rb") as f:
        params = json.load(f)
    with_dis = "dis_embd3" in params["pl_module_args"]["model"]
    data_train = make_dataset(params, "train", "train", args.synthetic, with_dis)
    data_val = make_dataset(params, "val", "val", args.synthetic, with_dis)

    train_loader, test_loader = make_loaders(data_train, data_val, params, world, rank, args.batch_per_gpu)
    hl = import_attr(params["pl_module"])(**params["pl_module_args"])
    ckdir = os.path.join(args.run_dir, "checkpoints")
    if rank == 0:
        os.makedirs(ckdir, exist_ok=True)
        if not os.path.exists(os.path.join(args.run_dir, "config.json")):
            shutil.copyfile(args.config, os.path.join(args.run_dir, "config.json"))
    best_path, state_path = os.path.join(ckdir, "best.pt"), os.path.join(ckdir, "last.pt")
    # resume: rank 0 decides (only it writes last.pt, and run directories may be node-local) and every rank follows --
    # load_state ends in collectives, so a per-rank os.path.exists() could leave ranks waiting for each other
    resume = [os.path.exists(state_path) if rank == 0 else False]
    if world > 1:
        dist.broadcast_object_list(resume, src=0)
    if resume[0]:
        hl.load_state(state_path)
    wandb_run = None
    if args.wandb and rank == 0:
        try:
            import wandb
            wandb_run = wandb.init(project=params.get("project_name", args.project_name),
                                   name=os.path.basename(args.run_dir.rstrip("/")))
        except Exception as e:
            print("[train_cli] wandb unavailable:", e)
    n_epochs = args.epochs if args.epochs is not None else params["epochs"]
    for epoch in range(hl.epoch, n_epochs):
        # data-side RNGs (crop offsets, perturbations drawn by the loader workers) differ per rank, as the reference's single
        # process draws independently per sample; model-side randomness does not exist (no dropout), and the sampler keeps
        # its own (seed, epoch) generator, so the sharding is unaffected
        seed_all(args.seed + epoch + 100003 * rank)
        if hasattr(train_loader.sampler, "set_epoch"):
            train_loader.sampler.set_epoch(epoch)                     # a different shuffle every epoch
        hl.on_epoch_start()
        print("CURRENT learning rate: {:0.08f}".format(hl.get_current_lr()))
        t1 = time.time()
        tl = train_epoch(hl, train_loader, device)
        print(f"Train epoch time: {time.time() - t1:02f}s\nTrain set: Average Loss: {tl:.4f}")
        seed_all(0)
        vl = test_epoch(hl, test_loader, device)
        print(f"Test set: Average Loss: {vl:.4f}")
        if rank == 0:
            hl.on_epoch_end(best_path, wandb_run)
            hl.dump_state(state_path)
        else:
            hl.on_epoch_end(os.devnull, None)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
