#!/usr/bin/env python3
"""After `train_cli --config experiments/overfit_test_samples.json --run_dir D`: the per-epoch loss curve out of D's last.pt
(metric_values, the reference's checkpoint layout) and the demo evaluation (eval_samples, src/test_samples.py's counterpart)
of D's best.pt over the three bundled scene sets, as one JSON file.  usage: overfit_report.py RUN_DIR OUT.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sound_bubble_amd.eval_samples import evaluate_dir          # noqa: E402
from sound_bubble_amd.harness import import_attr                # noqa: E402

run_dir, out = sys.argv[1], sys.argv[2]
last = torch.load(os.path.join(run_dir, "checkpoints", "last.pt"), map_location="cpu", weights_only=False)
mv = last["metric_values"]
curve = {k: [mv[e][k]["epoch"] / mv[e][k]["num_elements"] if k in mv[e] else None for e in sorted(mv)]
         for k in ("train/loss", "val/loss", "train/si_sdr_i", "val/si_sdr_i", "val/decay")}
params = json.load(open(os.path.join(run_dir, "config.json")))
hl = import_attr(params["pl_module"])(**dict(params["pl_module_args"], init_ckpt=None, use_dp=False))
hl.load_state(os.path.join(run_dir, "checkpoints", "best.pt"))
hl.eval()
rows = {}
for sset, thr in (("syn_1m", 1.0), ("syn_1_5m", 1.5), ("syn_2m", 2.0)):
    rows[sset] = evaluate_dir(hl.model, os.path.join(ROOT, "tests", "golden", "test_samples_full", sset), thr)
    for r in rows[sset]:
        print(sset, json.dumps(r))
json.dump({"epochs": len(mv), "best_epoch": hl.epoch - 1, "curve": curve, "eval_best": rows}, open(out, "w"), indent=1)
