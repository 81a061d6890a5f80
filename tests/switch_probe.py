"""Child process of tests/test_gpu_switch_matrix.py (not collected by pytest): the environment it is started with carries
ONE documented dispatch switch (README "Switches"); every switch is read at import, hence a process per switch.

Two stages, printed as one JSON line:
  golden -- tiny_big / tiny_small forward + every parameter gradient against the REFERENCE goldens (tests/golden/*.npz);
  medium -- seeded random weights at a geometry where the schedules the switches steer really engage (big family: 2
            blocks, B = 8, 1.6 s -> 73 inter-frame tiles x 200 steps: overlapped forward / backward eligible; small family:
            B = 32 -> 290 tiles: fused + time-segmented), output and gradients saved by the switch-free run (--save) and
            compared by every other run (--compare).
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conftest import load_golden, golden_state_dict, rel_l2      # noqa: E402

def medium():
    import bench                                   # the BASELINE model families, two blocks each
    W = bench.WORKLOADS
    return {"big": (W["big"][0], dict(W["big"][1], B=2), 8), "small": (W["small"][0], dict(W["small"][1], B=2), 32)}


N_SAMPLES = 38400     # 1.6 s at 24 kHz -> 200 frames


def grads_vs(m, want):
    worst = ("", 0.0)
    for k, p in m.named_parameters():
        g = want(k)
        assert p.grad is not None, k
        e = rel_l2(p.grad.cpu().numpy(), g) if np.abs(g).max() > 0 else float(p.grad.abs().max())
        if e > worst[1]:
            worst = (k, e)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--compare")
    a = ap.parse_args()
    import torch
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    from sound_bubble_amd.functional import SnrlpLossFn
    out = {"golden": {}, "medium": {}, "bptt": ops.BPTT}

    for name, cls in (("tiny_big", "NetDisEmbd3"), ("tiny_small", "NetOptim")):
        rec, params, _ = load_golden(name)
        m = getattr(sb, cls)(**params)
        m.load_state_dict(golden_state_dict(rec, torch), strict=True)
        m = m.cuda().train()
        inp = {"mixture": torch.from_numpy(rec["mixture"]).cuda()}
        if "dis_embed" in rec:
            inp["dis_embed"] = torch.from_numpy(rec["dis_embed"]).cuda()
        est = m(inp)["output"]
        loss, lv = SnrlpLossFn.apply(est, torch.from_numpy(rec["target"]).cuda(), 100.0)
        loss.backward()
        ops.check_sched_status()
        k, e = grads_vs(m, lambda k_: rec["grad::" + k_])
        with torch.no_grad():                            # the inference dispatch (few-sequence vector kernel, workspaces, ...)
            ev = m(inp)["output"] if os.environ.get("PROBE_NO_EVAL") != "1" else est.detach()
        out["golden"][name] = {"fwd": rel_l2(est.detach().cpu().numpy(), rec["output"]), "grad": e, "worst": k,
                               "fwd_eval": rel_l2(ev.cpu().numpy(), rec["output"])}

    if os.environ.get("PROBE_POISON"):                   # debugging aid: every byte the medium stage allocates starts as NaN
        del m
        torch.cuda.empty_cache()
        t = torch.full((int(float(os.environ["PROBE_POISON"]) * (1 << 28)),), float("nan"), device="cuda")
        del t
    saved = np.load(a.compare) if a.compare else None
    keep = {}
    for wl, (cls, params, B) in medium().items():
        torch.manual_seed(7)
        m = getattr(sb, cls)(**params).cuda().train()
        g = torch.Generator().manual_seed(11)
        mix = torch.randn(B, 6, N_SAMPLES, generator=g) * 0.1
        tgt = torch.randn(B, 1, N_SAMPLES, generator=g) * 0.1
        tgt[1::4] = 0                                    # silent targets: the loss's negative branch
        inp = {"mixture": mix.cuda()}
        if cls == "NetDisEmbd3":
            inp["dis_embed"] = torch.eye(3)[torch.arange(B) % 3].cuda()
        ops.PROFILE = {}
        est = m(inp)["output"]
        loss, lv = SnrlpLossFn.apply(est, tgt.cuda(), 100.0)
        loss.backward()
        torch.cuda.synchronize()
        labels = sorted(ops.PROFILE)
        ops.PROFILE = None
        ops.check_sched_status()
        with torch.no_grad():                            # the inference dispatch at this geometry (overlapped forward in inference)
            ev = m(inp)["output"]
        ops.check_sched_status()
        e_eval = rel_l2(ev.cpu().numpy(), est.detach().cpu().numpy())
        if saved is None:
            keep[wl + "::est"] = est.detach().cpu().numpy()
            for k, p in m.named_parameters():
                keep[wl + "::" + k] = p.grad.cpu().numpy()
            out["medium"][wl] = {"labels": labels, "fwd_eval": e_eval}
        else:
            k, e = grads_vs(m, lambda k_: saved[wl + "::" + k_])
            out["medium"][wl] = {"fwd": rel_l2(est.detach().cpu().numpy(), saved[wl + "::est"]), "grad": e, "worst": k,
                                 "labels": labels, "fwd_eval": e_eval}
    if a.save:
        np.savez(a.save, **keep)
    out["overlap"] = bool(ops.overlap_available())
    print("SWITCH_PROBE " + json.dumps(out))


if __name__ == "__main__":
    main()
