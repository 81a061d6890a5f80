"""Training converges, and parity holds where SI-SDR means something (VERDICT r4 missing #1 / next #3).

* A checkpoint TRAINED by this repo's HIP path (train_cli, experiments/overfit_test_samples.json, the nine bundled demo scenes)
  sits in tests/golden/trained_overfit_best.pt in the reference's own best.pt layout.  tests/golden/make_trained_fixture.py
  loaded it -- strict -- into the IMPORTED REFERENCE network in the build container and stored what that network outputs and
  scores on the nine scenes (src/test_samples.py:90-112, helpers/eval_utils.py).  The HIP model with the same file must land
  on the reference's outputs (rel-L2 <= 1e-3, the north star's bar) and SI-SDR (+-0.05 dB) at POSITIVE SI-SDR.
* A short run of the reference's epoch loop (src/train_pt.py:124-177 / tain_val.py:51-88) on those scenes must bring the
  loss down monotone-ish, with the overlapped schedules' watchdog silent."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
SETS = (("syn_1m", 1.0), ("syn_1_5m", 1.5), ("syn_2m", 2.0))


def _params():
    return json.load(open(os.path.join(ROOT, "experiments", "overfit_test_samples.json")))


def _trained_module():
    from sound_bubble_amd.harness import import_attr
    p = _params()
    hl = import_attr(p["pl_module"])(**dict(p["pl_module_args"], init_ckpt=None, use_dp=False))
    hl.load_state(os.path.join(GOLD, "trained_overfit_best.pt"))
    hl.eval()
    return hl


def test_trained_checkpoint_matches_the_imported_reference_at_positive_si_sdr():
    from sound_bubble_amd import eval_samples as ES
    ref = np.load(os.path.join(GOLD, "trained_overfit.npz"))
    hl = _trained_module()
    n_scored, n_positive, worst_db, worst_l2 = 0, 0, 0.0, 0.0
    for sset, radius in SETS:
        for scene in ("00000", "00001", "00002"):
            d = os.path.join(GOLD, "test_samples_full", sset, scene)
            meta, mix, gt, tg = ES.load_testcase(d, radius)
            out = ES.run_testcase(hl.model, mix, radius)
            key = f"{sset}/{scene}"
            assert len(tg) == int(ref[key + "::n_targets"])
            if key + "::output" in ref:
                r = ref[key + "::output"]
                l2 = float(np.linalg.norm(out - r) / np.linalg.norm(r))
                worst_l2 = max(worst_l2, l2)
                assert l2 <= 1e-3, (key, l2)
            if len(tg):
                s = ES.si_sdr_np(out[0], gt[0])
                want = float(ref[key + "::si_sdr"])
                worst_db = max(worst_db, abs(s - want))
                assert abs(s - want) <= 0.05, (key, s, want)
                assert abs(ES.si_sdr_np(mix[0], gt[0]) - float(ref[key + "::input_si_sdr"])) <= 1e-3
                n_scored += 1
                n_positive += want > 0.0
            else:
                dec = 10 * np.log10((mix[0].astype(np.float64) ** 2).sum() / max((out[0].astype(np.float64) ** 2).sum(), 1e-20))
                assert abs(dec - float(ref[key + "::decay_db"])) <= 0.05, (key, dec)
    print(f"trained checkpoint: {n_scored} scored scenes, {n_positive} at SI-SDR > 0 dB in the reference; worst |dSI-SDR| "
          f"{worst_db:.2e} dB, worst output rel-L2 {worst_l2:.2e}")
    assert n_scored == 6 and n_positive >= 5        # an operating point, not -50 dB plumbing


def test_loss_comes_down_over_a_short_run_with_a_silent_watchdog():
    """~100 optimiser steps of the harness loop on cached batches of the nine scenes (B = 9: ragged 82-tile inter-frame passes,
    every overlapped schedule engaged): the epoch-mean loss falls by more than 10x, stays finite, never rises by more than a
    third epoch over epoch after the first two, and no bounded wait of the overlapped schedules gives up."""
    from sound_bubble_amd import ops
    from sound_bubble_amd.harness import import_attr
    from sound_bubble_amd.train_cli import make_loaders, seed_all, to_device
    p = _params()
    seed_all(0)
    mk = lambda key, split: import_attr(p[f"{key}_dataset"])(**p[f"{key}_data_args"], split=split)
    loader, _ = make_loaders(mk("train", "train"), mk("val", "val"), p, 1, 0)
    batches = [b for b in loader]
    hl = import_attr(p["pl_module"])(**p["pl_module_args"])
    dev = torch.device("cuda")
    hl.train()
    means = []
    for epoch in range(13):
        tot = 0.0
        for idx, batch in enumerate(batches):
            batch = to_device(batch, dev)
            hl.reset_grad()
            loss, B = hl.training_step(batch, idx)
            loss.backward()
            hl.backprop()
            tot += float(loss.detach())
        means.append(tot / len(batches))
        assert np.isfinite(means[-1]), means
    assert ops.read_sched_status() == [], ops.LAST_TRIPS[-1:]
    print("epoch-mean loss:", [round(m, 3) for m in means])
    assert means[-1] < 0.1 * means[0], means
    assert all(b <= a + abs(a) / 3 + 0.2 for a, b in zip(means[2:], means[3:])), means


def test_b9_stress_1000_train_steps_no_trip_no_skipped_update_finite_loss():
    """>= 1 000 B = 9 train steps of the harness loop (ragged 82-tile inter-frame passes: the one geometry at
    which the overlapped forward's producer was ever seen to stand still; every overlapped schedule engaged), cached batches, one
    loss read per 8 epochs: no bounded wait gives up, the guarded optimiser skipped nothing, every loss read is finite, and the
    overlapped orders really were the ones that ran."""
    import time
    from sound_bubble_amd import ops
    from sound_bubble_amd.harness import import_attr
    from sound_bubble_amd.train_cli import make_loaders, seed_all, to_device
    p = _params()
    seed_all(0)
    mk = lambda key, split: import_attr(p[f"{key}_dataset"])(**p[f"{key}_data_args"], split=split)
    loader, _ = make_loaders(mk("train", "train"), mk("val", "val"), p, 1, 0)
    dev = torch.device("cuda")
    batches = [to_device(b, dev) for b in loader]
    assert batches[0][0]["mixture"].shape[0] == 9          # (inputs, targets) pairs, as tain_val.py:66-70 takes them
    hl = import_attr(p["pl_module"])(**p["pl_module_args"])
    hl.train()
    ops.sched_counts_reset()
    g0 = ops.read_giveups() if ops._OVERLAP_OK else 0
    t0 = time.time()
    steps = 0
    while steps < 1000:
        for idx, batch in enumerate(batches):
            hl.reset_grad()
            loss, B = hl.training_step(batch, idx)
            loss.backward()
            hl.backprop()                         # (checks the watchdog word every 50th step and raises on a trip)
            steps += 1
        if (steps // len(batches)) % 8 == 0:
            assert np.isfinite(float(loss.detach())), steps
    torch.cuda.synchronize()
    dt = time.time() - t0
    assert np.isfinite(float(loss.detach()))
    assert ops.read_sched_status() == [], ops.LAST_TRIPS[-1:]
    assert int(hl.optimizer.skipped.item()) == 0
    counts = dict(ops.SCHED_COUNTS)
    print(f"{steps} B = 9 train steps in {dt:.1f} s ({1e3 * dt / steps:.1f} ms / step): schedules {counts}, "
          f"hand-back events {(ops.read_giveups() - g0) if ops._OVERLAP_OK else 'n/a'}")
    if ops.overlap_available() and not ops._OVERLAP_LOST:
        assert counts["fwd_overlapped"] > 0 and counts["bwd_overlapped"] > 0, counts


def test_guarded_adam_skips_the_update_when_the_watchdog_word_is_set():
    """sb_adam_step_guarded: with *guard != 0 parameters and moments stay bit-identical and *skipped counts the step; with
    *guard == 0 (and with guard == NULL, the legacy entry) the update is torch.optim.Adam's."""
    import ctypes as C
    from sound_bubble_amd import _lib as L
    lib = L.load()
    torch.manual_seed(0)
    n = 10007
    p0, g = torch.randn(n, device="cuda"), torch.randn(n, device="cuda")
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-2)
    ref.grad = g.clone()
    opt.step()
    vp = lambda t: C.c_void_p(t.data_ptr())
    for guard_val in (1, 0, None):
        p, m, v = p0.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
        guard = torch.full((1,), guard_val, dtype=torch.int32, device="cuda") if guard_val is not None else None
        skipped = torch.zeros(1, dtype=torch.int32, device="cuda")
        if guard is None:
            rc = lib.sb_adam_step(vp(p), vp(g), vp(m), vp(v), n, 1e-2, 0.9, 0.999, 1e-8, 1, 1.0, 0.0, None, None)
        else:
            rc = lib.sb_adam_step_guarded(vp(p), vp(g), vp(m), vp(v), n, 1e-2, 0.9, 0.999, 1e-8, 1, 1.0, 0.0, None, vp(guard),
                                          vp(skipped), None)
        assert rc == 0
        torch.cuda.synchronize()
        if guard_val == 1:
            assert torch.equal(p, p0) and not m.any() and not v.any() and int(skipped) == 1
        else:
            assert int(skipped) == 0
            assert torch.allclose(p, ref.detach(), rtol=1e-5, atol=1e-7)
