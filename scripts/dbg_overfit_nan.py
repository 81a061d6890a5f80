#!/usr/bin/env python3
"""Find the first bad step of the overfit run (experiments/overfit_test_samples.json): per step, is the estimate finite, did a
watchdog word trip (ops.read_sched_status), are the gradients / parameters finite, how long did the step take.  On the first
anomaly: report, then replay the same step from the pre-step snapshot (parameters + Adam moments) a few times, in the default
schedules and with the overlapped / segmented schedules off, and save the snapshot.  usage: dbg_overfit_nan.py [max_steps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sound_bubble_amd import ops                                 # noqa: E402
from sound_bubble_amd.harness import import_attr                 # noqa: E402
from sound_bubble_amd.train_cli import seed_all, to_device       # noqa: E402

max_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
params = json.load(open(os.path.join(ROOT, "experiments", "overfit_test_samples.json")))
seed_all(0)
ds = import_attr(params["train_dataset"])(**params["train_data_args"], split="train")
items = [ds[i] for i in range(9)]
coll = torch.utils.data.default_collate
hl = import_attr(params["pl_module"])(**params["pl_module_args"])
hl.train()
dev = torch.device("cuda")
g = torch.Generator().manual_seed(0)


def snapshot():
    return dict(flat=hl.bucket.flat.clone(), m=hl.optimizer.m.clone(), v=hl.optimizer.v.clone(), step=hl.optimizer.step_count)


def restore(s):
    from sound_bubble_amd.forms import bump_weight_epoch
    hl.bucket.flat.copy_(s["flat"]); hl.optimizer.m.copy_(s["m"]); hl.optimizer.v.copy_(s["v"])
    hl.optimizer.step_count = s["step"]
    bump_weight_epoch()


def one_step(batch, apply=True):
    t0 = time.perf_counter()
    hl.reset_grad()
    loss, B = hl.training_step(batch, 0)
    est_ok = bool(torch.isfinite(loss.detach()).item())
    loss.backward()
    ops.deferred_join()
    torch.cuda.synchronize()
    bad = ops.read_sched_status()
    gfin = bool(torch.isfinite(hl.bucket.grad).all().item())
    gmax = float(hl.bucket.grad.abs().max().item())
    if apply:
        hl.backprop()
    pfin = bool(torch.isfinite(hl.bucket.flat).all().item())
    return dict(loss=float(loss.detach()), loss_finite=est_ok, watchdog=bad, grad_finite=gfin, grad_absmax=gmax,
                param_finite=pfin, ms=(time.perf_counter() - t0) * 1e3)


log = []
for step in range(max_steps):
    idx = torch.randperm(9, generator=g).tolist()
    batch = to_device(coll([items[i] for i in idx]), dev)
    snap = snapshot()
    r = one_step(batch)
    log.append(r["loss"])
    if step % 8 == 7 and not r["watchdog"]:        # the epoch's validation pass (eval-mode forward, same nine scenes)
        hl.eval()
        with torch.no_grad():
            vl, _ = hl.validation_step(to_device(coll(items), dev), 0)
        torch.cuda.synchronize()
        r["val_loss"], r["val_watchdog"] = float(vl), ops.read_sched_status()
        hl.train()
        if r["val_watchdog"] or r["val_loss"] != r["val_loss"]:
            print("ANOMALY in the validation pass after step", step, json.dumps(r), flush=True)
            r["watchdog"] = r["val_watchdog"] or [-1]
    if step % 25 == 0:
        print(step, json.dumps(r), flush=True)
    if r["watchdog"] or not (r["loss_finite"] and r["grad_finite"] and r["param_finite"]) or (step > 3 and r["ms"] > 400):
        print("ANOMALY at step", step, json.dumps(r), "order", idx, flush=True)
        torch.save(dict(snap={k: (v.cpu() if torch.is_tensor(v) else v) for k, v in snap.items()}, order=idx, step=step),
                   os.path.join(ROOT, "gpurun_out", "overfit_nan_snapshot.pt"))
        for name, env in (("default", {}), ("default again", {}), ("no cross overlap", dict(BWD_CROSS_OVERLAP=False)),
                          ("no bwd overlap", dict(BWD_CROSS_OVERLAP=False, BWD_OVERLAP=False)),
                          ("no fwd/bwd overlap, no segments", dict(BWD_CROSS_OVERLAP=False, BWD_OVERLAP=False, FWD_OVERLAP=False,
                                                                  TIME_SEGMENTS=False))):
            old = {k: getattr(ops, k) for k in env}
            for k, v in env.items():
                setattr(ops, k, v)
            restore(snap)
            rr = one_step(batch, apply=False)
            print("  replay [%s]" % name, json.dumps(rr), flush=True)
            for k, v in old.items():
                setattr(ops, k, v)
        break
else:
    print("no anomaly in", max_steps, "steps; last losses", log[-5:])
