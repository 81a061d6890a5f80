#!/usr/bin/env python3
"""Demo evaluation over `test_samples/<set>/<scene>/` folders -- the counterpart of src/test_samples.py:35-112,114-246.

Scene layout: mixture.wav (6 ch, 24 kHz, PCM16), mic00_voiceNN.wav (mono ground truth per speaker), metadata.json
(voiceNN.dis in metres, or centimetres when metadata['real']).  Ground truth = sum of the mic00 recordings of the
speakers with dis <= threshold; the distance one-hot is [0,0,1] / [0,1,0] / [1,0,0] for 1 / 1.5 / 2 m
(test_samples.py:96-104).  Metrics: SNR / SI-SDR per helpers/eval_utils.py:4-23 (NumPy), decay when the bubble is empty.

  python -m sound_bubble_amd.eval_samples test_samples/syn_1m RUN_DIR --distance_threshold 1
"""
import argparse
import glob
import json
import math
import os
import wave

import numpy as np
import torch

ONE_HOT = {1.0: [0.0, 0.0, 1.0], 1.5: [0.0, 1.0, 0.0], 2.0: [1.0, 0.0, 0.0]}


def read_wav(path):
    """PCM16 WAV -> float32 [channels, samples] in [-1, 1) (x / 32768, as librosa/soundfile do)."""
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2 and w.getcomptype() == "NONE", "PCM16 only"
        n, ch, sr = w.getnframes(), w.getnchannels(), w.getframerate()
        data = np.frombuffer(w.readframes(n), dtype="<i2").reshape(n, ch).T
    return (data.astype(np.float32) / 32768.0), sr


def write_wav(path, data, sr):
    data = np.atleast_2d(data)
    pcm = np.clip(np.round(data.T * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(data.shape[0])
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes(pcm.tobytes())


def snr_np(est, gt, scale_invariant=False):
    """helpers/eval_utils.py:4-23."""
    est, gt = np.asarray(est, np.float64), np.asarray(gt, np.float64)
    a = np.dot(est, gt) / np.dot(gt, gt) if scale_invariant else 1.0
    e_sig = a * gt
    e_noise = e_sig - est
    return 10 * math.log10((e_sig ** 2).sum() / ((e_noise ** 2).sum() + 1e-9))


def si_sdr_np(est, gt):
    return snr_np(est, gt, True)


def load_testcase(sample_dir, distance_threshold, sr=24000):
    """-> metadata, mixture [M, N] float32, gt [1, N], list of in-bubble speakers  (test_samples.py:35-88)"""
    with open(os.path.join(sample_dir, "metadata.json"), "rb") as f:
        meta = json.load(f)
    mixture, fs = read_wav(os.path.join(sample_dir, "mixture.wav"))
    assert fs == sr, f"expected {sr} Hz"
    gt = np.zeros((1, mixture.shape[-1]), np.float32)
    targets = []
    for spk in sorted(k for k in meta if k.startswith("voice")):
        dis = meta[spk]["dis"] / 100 if meta.get("real") else meta[spk]["dis"]
        if dis <= distance_threshold:
            solo, _ = read_wav(os.path.join(sample_dir, f"mic00_{spk}.wav"))
            gt += solo[:1, : gt.shape[-1]]
            targets.append(meta[spk])
    return meta, mixture, gt, targets


@torch.no_grad()
def run_testcase(model, mixture, distance_threshold, device="cuda"):
    """test_samples.py:90-112 -> output [1, N] numpy"""
    if float(distance_threshold) not in ONE_HOT:
        raise ValueError("Invalid distance threshold")
    x = torch.from_numpy(np.ascontiguousarray(mixture)).to(device).unsqueeze(0)
    inputs = {"mixture": x, "dis_embed": torch.tensor([ONE_HOT[float(distance_threshold)]], device=device)}
    return model(inputs)["output"].squeeze(0).cpu().numpy()


def evaluate_dir(model, test_dir, distance_threshold, device="cuda", out_dir=None):
    rows = []
    for d in sorted(glob.glob(os.path.join(test_dir, "*"))):
        if not os.path.isdir(d):
            continue
        meta, mix, gt, tg = load_testcase(d, distance_threshold)
        out = run_testcase(model, mix, distance_threshold, device)
        row = {"sample": os.path.basename(d), "n_targets": len(tg)}
        if len(tg) == 0:
            row["decay"] = 10 * math.log10((mix[0] ** 2).sum()) - 10 * math.log10((out[0] ** 2).sum() + 1e-30)
        else:
            row.update(input_si_sdr=si_sdr_np(mix[0], gt[0]), si_sdr=si_sdr_np(out[0], gt[0]),
                       input_snr=snr_np(mix[0], gt[0]), snr=snr_np(out[0], gt[0]))
            row["si_sdr_i"] = row["si_sdr"] - row["input_si_sdr"]
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
            write_wav(os.path.join(out_dir, f"{row['sample']}_output.wav"), out, 24000)
        rows.append(row)
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("test_dir")
    ap.add_argument("run_dir", help="directory with config.json and checkpoints/best.pt (utils.load_torch_pretrained)")
    ap.add_argument("--distance_threshold", type=float, default=1.0)
    ap.add_argument("--save_dir", default=None)
    args = ap.parse_args(argv)
    from .harness import import_attr
    with open(os.path.join(args.run_dir, "config.json")) as f:
        params = json.load(f)
    pa = dict(params["pl_module_args"], init_ckpt=None, use_dp=False)
    hl = import_attr(params["pl_module"])(**pa)
    hl.load_state(os.path.join(args.run_dir, "checkpoints", "best.pt"))
    hl.eval()
    for r in evaluate_dir(hl.model, args.test_dir, args.distance_threshold, out_dir=args.save_dir):
        print(json.dumps(r))


if __name__ == "__main__":
    main()
