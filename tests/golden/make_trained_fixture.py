#!/usr/bin/env python3
"""Parity at an operating point that means something: a checkpoint TRAINED by this repo's HIP path (train_cli on
experiments/overfit_test_samples.json, the nine bundled demo scenes) is loaded -- `load_state_dict(strict=True)` -- into the
IMPORTED REFERENCE network (src/models/tfgridnet_realtime_clean_dis_embd3/net.py:20, through make_goldens.py's three
stand-ins for absent third-party packages), which then separates the nine scenes as src/test_samples.py:90-112 does (one-hot
of the scene set's radius, ground truth = the mic00 voices inside the bubble) and scores them with the reference's own NumPy
metrics (helpers/eval_utils.py).  Build container only (needs /root/reference); emits data:

  tests/golden/trained_overfit.npz   reference outputs (fp32) of one scene per radius, SI-SDR / input SI-SDR / SNR of all nine

The checkpoint itself (tests/golden/trained_overfit_best.pt, in the reference's own best.pt layout, written by
sound_bubble_amd.harness.PLModule.dump_state on the GPU box) is an input of this script and is committed beside its output;
tests/test_gpu_trained.py loads the same file into the HIP model and must land on these outputs and scores.

usage: python tests/golden/make_trained_fixture.py [--ckpt tests/golden/trained_overfit_best.pt] [--slim]
  --slim   rewrite the checkpoint without its optimizer moments first (2 MB instead of 6.8 MB; the keys stay, `optimizer`
           becomes a fresh Adam state: the reference's loader accepts it)"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import wave

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_goldens as MG                                       # noqa: E402  (shims + REF)

SETS = (("syn_1m", 1.0, [0.0, 0.0, 1.0]), ("syn_1_5m", 1.5, [0.0, 1.0, 0.0]), ("syn_2m", 2.0, [1.0, 0.0, 0.0]))


def read_scene(d, radius):
    meta = json.load(open(os.path.join(d, "metadata.json")))
    with wave.open(os.path.join(d, "mixture.wav"), "rb") as w:
        mix = np.frombuffer(w.readframes(w.getnframes()), "<i2").reshape(-1, 6).T.astype(np.float32) / 32768.0
    gt = np.zeros((1, mix.shape[1]), np.float32)
    ntg = 0
    for spk in sorted(k for k in meta if k.startswith("voice")):
        if meta[spk]["dis"] <= radius:
            ntg += 1
            with wave.open(os.path.join(d, f"mic00_{spk}.wav"), "rb") as w:
                gt[0] += np.frombuffer(w.readframes(w.getnframes()), "<i2").astype(np.float32) / 32768.0
    return mix, gt, ntg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", default=os.path.join(HERE, "trained_overfit_best.pt"))
    ap.add_argument("--config", default=os.path.join(ROOT, "experiments", "overfit_test_samples.json"))
    ap.add_argument("--out", default=os.path.join(HERE, "trained_overfit.npz"))
    ap.add_argument("--slim", action="store_true")
    args = ap.parse_args()
    assert os.path.isdir(MG.REF), "reference tree not present: this fixture can only be made in the build container"
    shim = tempfile.mkdtemp(prefix="sb_oracle_shims_")
    MG._write_shims(shim)
    sys.dont_write_bytecode = True
    sys.path[:0] = [shim, MG.REF, os.path.join(MG.REF, "helpers")]
    import torch
    import eval_utils                                            # reference helpers/eval_utils.py
    torch.set_num_threads(8)
    NetBig = importlib.import_module("src.models.tfgridnet_realtime_clean_dis_embd3.net").Net
    params = json.load(open(args.config))["pl_module_args"]["model_params"]
    ck = torch.load(args.ckpt, map_location="cpu", weights_only=False)
    if args.slim and ck.get("optimizer", {}).get("state"):
        ck["optimizer"] = dict(ck["optimizer"], state={})
        torch.save(ck, args.ckpt)
        print("slimmed", args.ckpt, os.path.getsize(args.ckpt), "bytes")
    model = NetBig(**params).eval()
    missing = model.load_state_dict(ck["model"], strict=True)   # strict: every key of the HIP model's state_dict is the reference's
    print("loaded", args.ckpt, "epoch", ck.get("current_epoch"), missing)
    rec = {"meta::params": np.array(repr(sorted(params.items()))), "meta::ckpt_epoch": np.int64(ck.get("current_epoch", -1))}
    for sset, radius, onehot in SETS:
        for si, scene in enumerate(("00000", "00001", "00002")):
            d = os.path.join(HERE, "test_samples_full", sset, scene)
            mix, gt, ntg = read_scene(d, radius)
            with torch.no_grad():
                out = model({"mixture": torch.from_numpy(mix)[None], "dis_embed": torch.tensor([onehot])})["output"][0].numpy()
            key = f"{sset}/{scene}"
            o64, g64, m64 = out[0].astype(np.float64), gt[0].astype(np.float64), mix[0].astype(np.float64)
            rec[key + "::n_targets"] = np.int64(ntg)
            if ntg:
                rec[key + "::si_sdr"] = np.float64(eval_utils.si_sdr(o64, g64))
                rec[key + "::input_si_sdr"] = np.float64(eval_utils.si_sdr(m64, g64))
                rec[key + "::snr"] = np.float64(eval_utils.snr(o64, g64))
            else:                                                  # empty bubble: the reference reports the energy decay (test_samples.py)
                rec[key + "::decay_db"] = np.float64(10 * np.log10((m64 ** 2).sum() / max((o64 ** 2).sum(), 1e-20)))
            if si == 1:                                            # one scene per radius travels with its full output
                rec[key + "::output"] = out.astype(np.float32)
            print(key, "targets", ntg, {k.split("::")[1]: round(float(v), 3) for k, v in rec.items()
                                        if k.startswith(key) and k.split("::")[1] in ("si_sdr", "input_si_sdr", "decay_db")})
    np.savez_compressed(args.out, **rec)
    print("->", args.out, os.path.getsize(args.out), "bytes")


if __name__ == "__main__":
    main()
