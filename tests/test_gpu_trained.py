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
