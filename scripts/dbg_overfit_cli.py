#!/usr/bin/env python3
"""The overfit run as train_cli drives it (DataLoader with pinned memory + shuffle, one validation pass and on_epoch_end --
watchdog check, side-stream reprobe, scheduler -- per epoch), with the watchdog word read after EVERY step: which phase trips
it first?  usage: dbg_overfit_cli.py [epochs] [--no-reprobe]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sound_bubble_amd import ops                                 # noqa: E402
from sound_bubble_amd.harness import import_attr                 # noqa: E402
from sound_bubble_amd.train_cli import seed_all, to_device, make_loaders       # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
no_reprobe = "--no-reprobe" in sys.argv
params = json.load(open(os.path.join(ROOT, "experiments", "overfit_test_samples.json")))
seed_all(0)
mk = lambda key, split: import_attr(params[f"{key}_dataset"])(**params[f"{key}_data_args"], split=split)
train_loader, test_loader = make_loaders(mk("train", "train"), mk("val", "val"), params, 1, 0)
hl = import_attr(params["pl_module"])(**params["pl_module_args"])
dev = torch.device("cuda")
if no_reprobe:
    ops.overlap_reprobe = lambda: True


def check(where):
    torch.cuda.synchronize()
    bad = ops.read_sched_status()
    if bad:
        print("WATCHDOG tripped:", where, "overlap log tail", ops.OVERLAP_LOG[-3:], "counts", ops.SCHED_COUNTS, flush=True)
        sys.exit(3)


for epoch in range(epochs):
    seed_all(epoch)
    hl.train()
    t0 = time.time()
    for idx, batch in enumerate(train_loader):
        batch = to_device(batch, dev)
        hl.reset_grad()
        loss, B = hl.training_step(batch, idx)
        loss.backward()
        check(f"epoch {epoch} train step {idx}: after backward")
        hl.backprop()
        l = float(loss.detach())
        if l != l:
            print("NaN loss at epoch", epoch, "step", idx, flush=True)
            sys.exit(4)
        check(f"epoch {epoch} train step {idx}: after the optimiser")
    hl.eval()
    with torch.no_grad():
        for idx, batch in enumerate(test_loader):
            vl, _ = hl.validation_step(to_device(batch, dev), idx)
            check(f"epoch {epoch} validation step {idx}")
    hl.on_epoch_end(os.devnull, None)
    check(f"epoch {epoch} on_epoch_end")
    print(epoch, f"loss {l:.4f} val {float(vl):.4f} {time.time() - t0:.2f}s", ops.OVERLAP_LOG[-1][0] if ops.OVERLAP_LOG else "", flush=True)
print("no anomaly in", epochs, "epochs")
