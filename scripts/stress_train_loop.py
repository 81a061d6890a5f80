#!/usr/bin/env python3
"""Stress the train_cli epoch loop (train steps with their one D2H each, a validation pass, on_epoch_end) on CACHED batches of the
overfit experiment, counting watchdog trips (ops.LAST_TRIPS says which bounded wait gave up) and non-finite losses.
usage: stress_train_loop.py [--epochs N] [--sleep MS] [--batch B] [--loader]
  --sleep MS   random host sleep in [0, MS] ms in front of every step (the GPU idles between steps as it does behind a slow loader)
  --loader     the real DataLoader of train_cli instead of cached batches"""
import argparse
import json
import os
import random
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sound_bubble_amd import ops, _lib as L                      # noqa: E402
from sound_bubble_amd.harness import import_attr                 # noqa: E402
from sound_bubble_amd.train_cli import seed_all, to_device, make_loaders       # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=200)
ap.add_argument("--sleep", type=float, default=0.0)
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--loader", action="store_true")
ap.add_argument("--no-val", action="store_true")
ap.add_argument("--config", default="overfit_test_samples.json", help="experiment JSON under experiments/ (bubble_small_synthetic.json: the 0.3 M config, time-segmented passes at --batch 32)")
args = ap.parse_args()
params = json.load(open(os.path.join(ROOT, "experiments", args.config)))
if args.batch:
    params["batch_size"] = params["eval_batch_size"] = args.batch
seed_all(0)
mk = lambda key, split: import_attr(params[f"{key}_dataset"])(**params[f"{key}_data_args"], split=split)
train_loader, test_loader = make_loaders(mk("train", "train"), mk("val", "val"), params, 1, 0)
hl = import_attr(params["pl_module"])(**params["pl_module_args"])
dev = torch.device("cuda")
if not args.loader:
    train_batches = [b for b in train_loader]
    val_batches = [b for b in test_loader]
# the last overlapped forwards' flag arrays, read back after a trip: did memory ever hold the full counts the poller never saw?
import collections
RECENT = collections.deque(maxlen=96)
_init = ops.FwdOverlap.__init__


def _patched(self, *a, **k):
    _init(self, *a, **k)
    RECENT.append(self)


ops.FwdOverlap.__init__ = _patched


def dump_recent():
    torch.cuda.synchronize()
    nbad = nzero = 0
    for i, o in enumerate(RECENT):
        f = o.flags.cpu().tolist()
        nsl = (o.need.max().item() + 1) if hasattr(o, "need") else len(f) - 4
        T_sl = f[524:524 + nsl]               # (round 5: 4 control words + 520 ints of hand-back block in front of the slab flags)
        if all(v == 0 for v in T_sl):
            nzero += 1
        elif any(v != o.producer_tiles for v in T_sl) or f[0] != o.producer_tiles:
            nbad += 1
            print(f"  overlap #{i} of {len(RECENT)} (oldest first): started {f[0]} counters {f[1:4]} tiles {o.producer_tiles} slabs {T_sl}", flush=True)
    print(f"  of {len(RECENT)} recent flag arrays: {nbad} hold partial counts IN MEMORY, {nzero} are all zero (never produced)", flush=True)


def kfd_evicted_ms():
    """KFD's per-process queue-eviction clock (ms this process' queues spent evicted: waves context-saved, e.g. for a page-table
    update behind an MMU notifier).  Hypothesis under test: the producer's 'freeze' is such an eviction, after
    which the pollers get their CUs back first and the producer's saved waves find no room until the pollers leave.  None where
    the sysfs node is absent."""
    import glob
    tot, seen = 0, False
    # (every process directory: inside the GPU box's container os.getpid() is not the pid KFD knows this process by -- checked
    #  at the end of round 5 -- and the box is single-tenant)
    for f in glob.glob("/sys/class/kfd/kfd/proc/*/stats_*/evicted_ms"):
        try:
            tot += int(open(f).read().strip() or 0)
            seen = True
        except (OSError, ValueError):
            pass
    return tot if seen else None


ev0 = ev_last = kfd_evicted_ms()
gu_last = 0
print("kfd evicted_ms at start:", ev0, flush=True)
trips, nans, steps = 0, 0, 0
ep_med, t_ep0 = 1.0, time.time()
t00 = time.time()
for epoch in range(args.epochs):
    seed_all(epoch)
    hl.train()
    it = train_loader if args.loader else random.sample(train_batches, len(train_batches))
    for idx, batch in enumerate(it):
        if args.sleep:
            time.sleep(random.random() * args.sleep * 1e-3)
        batch = to_device(batch, dev)
        hl.reset_grad()
        loss, B = hl.training_step(batch, idx)
        loss.backward()
        hl.backprop()
        l = float(loss.detach())
        steps += 1
        if l != l:
            nans += 1
            if nans == 1 or nans % 100 == 0:
                print(f"epoch {epoch} step {idx}: non-finite loss (#{nans})", flush=True)
    hl.eval()
    if not args.no_val:
        with torch.no_grad():
            for idx, batch in enumerate(test_loader if args.loader else val_batches):
                vl, _ = hl.validation_step(to_device(batch, dev), idx)
    try:
        hl.on_epoch_end(os.devnull, None)
    except L.SoundBubbleHipError as e:
        trips += 1
        print(f"epoch {epoch}: TRIP {ops.LAST_TRIPS[-1:]} counts {ops.SCHED_COUNTS} loss {l}", flush=True)
        dump_recent()
        seed_all(1000 + epoch)
        hl = import_attr(params["pl_module"])(**params["pl_module_args"])           # (the parameters are garbage now: start over)
    gu, ev = ops.read_giveups(), kfd_evicted_ms()
    if gu != gu_last or ev != ev_last:          # a hand-back event and / or a queue eviction in this epoch: do they coincide?
        print(f"epoch {epoch} t={time.time() - t00:.1f}s: give-ups +{gu - gu_last} (total {gu}), kfd evicted_ms "
              f"{'n/a' if ev is None else f'+{ev - (ev_last or 0)} (total {ev})'}", flush=True)
        gu_last, ev_last = gu, ev
    t_ep = time.time() - t_ep0 if epoch else 0.0
    if epoch > 3 and t_ep > 3 * ep_med:
        print(f"epoch {epoch}: SLOW {t_ep:.2f}s (typical {ep_med:.2f}s)", flush=True)
    elif epoch > 0:
        ep_med = 0.9 * ep_med + 0.1 * t_ep if epoch > 1 else t_ep
    t_ep0 = time.time()
    if epoch % 50 == 0:
        print(epoch, f"loss {l:.4f} steps {steps} trips {trips} nan-steps {nans} {time.time() - t00:.0f}s", flush=True)
print("forward-consumer give-ups (harmless):", ops.read_giveups(), "| kfd evicted_ms start / end:", ev0, "/", kfd_evicted_ms(), flush=True)
# which order the inter-frame passes really took (a re-probe that finds the side stream serialised switches to the plain order)
rp = [e for e in ops.OVERLAP_LOG if e[0] == "reprobe"]
print("schedule counts:", ops.SCHED_COUNTS, "| re-probes:", len(rp), "failed:", [(round(e[3], 3), round(e[4], 3)) for e in rp if e[2] != 1],
      "| slowest passing pair (ms):", max([e[4] for e in rp if e[2] == 1], default=None), flush=True)
try:
    print("watchdog debug words at the end (a -DSB_TRIP_DEBUG library: [52] = longest wait in polls):", ops.flag_arena(0).status_debug(), flush=True)
except Exception as e:
    print("no debug words:", e)
print(f"DONE epochs {args.epochs} steps {steps} trips {trips} nan-steps {nans} {time.time() - t00:.0f}s args {vars(args)}")
