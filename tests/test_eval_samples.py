"""BASELINE config 1 (plumbing + numerics): test_samples/syn_1m scenes (trimmed reference fixtures) through the
WAV/metadata reader, the model with shared weights, and the SI-SDR formula.  Parity bar from the north star:
output <= 1e-3 relative L2, SI-SDR within +-0.05 dB of the reference's value."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_golden, golden_state_dict, rel_l2

SAMPLES = os.path.join(GOLDEN, "test_samples", "syn_1m")


def _golden():
    z = np.load(os.path.join(GOLDEN, "samples_syn_1m.npz"))
    return {k: z[k] for k in z.files}


def test_reader_assembles_ground_truth_like_the_reference():
    from sound_bubble_amd.eval_samples import load_testcase
    g = _golden()
    counts = {}
    for scene in ("00000", "00001", "00002"):
        meta, mix, gt, tg = load_testcase(os.path.join(SAMPLES, scene), 1.0)
        assert mix.shape == (6, 36000) and mix.dtype == np.float32 and np.abs(mix).max() <= 1.0
        np.testing.assert_array_equal(gt, g[scene + "::gt"])
        counts[scene] = len(tg)
    assert counts == {"00000": 0, "00001": 1, "00002": 2}          # one scene each with 0 / 1 / 2 in-bubble speakers


def test_wav_roundtrip(tmp_path):
    from sound_bubble_amd.eval_samples import read_wav, write_wav
    x, sr = read_wav(os.path.join(SAMPLES, "00001", "mixture.wav"))
    write_wav(str(tmp_path / "a.wav"), x, sr)
    y, _ = read_wav(str(tmp_path / "a.wav"))
    np.testing.assert_array_equal(x, y)


def test_oracle_on_samples_matches_reference(torch_mod):
    torch = torch_mod
    from oracle.tfgridnet_oracle import OracleNet
    from sound_bubble_amd.eval_samples import load_testcase, si_sdr_np, snr_np
    rec, params, flavour = load_golden("tiny_big")
    m = OracleNet(flavour, **params).eval()
    m.load_state_dict(golden_state_dict(rec, torch))
    g = _golden()
    for scene in ("00001", "00002"):
        _, mix, gt, _ = load_testcase(os.path.join(SAMPLES, scene), 1.0)
        with torch.no_grad():
            out = m({"mixture": torch.from_numpy(mix)[None], "dis_embed": torch.tensor([[0.0, 0.0, 1.0]])})["output"][0].numpy()
        assert rel_l2(out, g[scene + "::output"]) < 5e-6
        assert abs(si_sdr_np(out[0], gt[0]) - float(g[scene + "::si_sdr"])) < 0.05
        assert abs(si_sdr_np(mix[0], gt[0]) - float(g[scene + "::input_si_sdr"])) < 1e-6     # formula pinned
        assert abs(snr_np(out[0], gt[0]) - float(g[scene + "::snr"])) < 0.05


@pytest.mark.gpu
def test_hip_model_on_samples_si_sdr_parity():
    import torch
    import sound_bubble_amd as sb
    from sound_bubble_amd.eval_samples import evaluate_dir, load_testcase, run_testcase
    rec, params, _ = load_golden("tiny_big")
    m = sb.NetDisEmbd3(**params)
    m.load_state_dict(golden_state_dict(rec, torch))
    m = m.cuda().eval()
    g = _golden()
    rows = {r["sample"]: r for r in evaluate_dir(m, SAMPLES, 1.0)}
    assert rows["00000"]["n_targets"] == 0 and "decay" in rows["00000"]
    for scene in ("00001", "00002"):
        _, mix, gt, _ = load_testcase(os.path.join(SAMPLES, scene), 1.0)
        out = run_testcase(m, mix, 1.0)
        assert rel_l2(out, g[scene + "::output"]) < 2e-4                    # north-star bar: 1e-3 (187 frames, random weights)
        assert abs(rows[scene]["si_sdr"] - float(g[scene + "::si_sdr"])) < 0.05
    with pytest.raises(ValueError):
        run_testcase(m, np.zeros((6, 960), np.float32), 1.2)


# ---- the other two radii of test_samples/ and one full-length scene (round 3; src/test_samples.py:35-112) ----
MORE = [("syn_1_5m", "00001", 1.5, 36000, 1), ("syn_2m", "00002", 2.0, 120000, 2)]


def _golden_more():
    z = np.load(os.path.join(GOLDEN, "samples_more.npz"))
    return {k: z[k] for k in z.files}


def test_reader_and_oracle_on_the_other_radii(torch_mod):
    torch = torch_mod
    from oracle.tfgridnet_oracle import OracleNet
    from sound_bubble_amd.eval_samples import ONE_HOT, load_testcase, si_sdr_np
    assert ONE_HOT[1.0] == [0.0, 0.0, 1.0] and ONE_HOT[1.5] == [0.0, 1.0, 0.0] and ONE_HOT[2.0] == [1.0, 0.0, 0.0]
    rec, params, flavour = load_golden("tiny_big")
    m = OracleNet(flavour, **params).eval()
    m.load_state_dict(golden_state_dict(rec, torch))
    g = _golden_more()
    for sset, scene, thr, n, ntg in MORE:
        _, mix, gt, tg = load_testcase(os.path.join(GOLDEN, "test_samples", sset, scene), thr)
        assert mix.shape == (6, n) and len(tg) == ntg == int(g[f"{sset}/{scene}::n_targets"])
        np.testing.assert_array_equal(gt, g[f"{sset}/{scene}::gt"])
        with torch.no_grad():
            out = m({"mixture": torch.from_numpy(mix)[None], "dis_embed": torch.tensor([ONE_HOT[thr]])})["output"][0].numpy()
        assert rel_l2(out, g[f"{sset}/{scene}::output"]) < 5e-6
        assert abs(si_sdr_np(out[0], gt[0]) - float(g[f"{sset}/{scene}::si_sdr"])) < 0.05
        assert abs(si_sdr_np(mix[0], gt[0]) - float(g[f"{sset}/{scene}::input_si_sdr"])) < 1e-6


@pytest.mark.gpu
def test_hip_model_on_the_other_radii_and_a_full_length_scene():
    """evaluate_dir with thresholds 1.5 / 2.0 (one-hots [0, 1, 0] / [1, 0, 0]); syn_2m/00002 is a full 5 s scene (625 frames,
    the BASELINE clip length) -- rel-L2 <= 2e-4 against the REFERENCE model's output, SI-SDR within 0.05 dB."""
    import torch
    import sound_bubble_amd as sb
    from sound_bubble_amd.eval_samples import evaluate_dir
    rec, params, _ = load_golden("tiny_big")
    m = sb.NetDisEmbd3(**params)
    m.load_state_dict(golden_state_dict(rec, torch))
    m = m.cuda().eval()
    g = _golden_more()
    for sset, scene, thr, n, ntg in MORE:
        out_dir = os.path.join("/tmp", f"sb_eval_{sset}")
        rows = {r["sample"]: r for r in evaluate_dir(m, os.path.join(GOLDEN, "test_samples", sset), thr, out_dir=out_dir)}
        row = rows[scene]
        assert row["n_targets"] == ntg
        from sound_bubble_amd.eval_samples import read_wav
        assert abs(row["si_sdr"] - float(g[f"{sset}/{scene}::si_sdr"])) < 0.05
        assert abs(row["input_si_sdr"] - float(g[f"{sset}/{scene}::input_si_sdr"])) < 1e-4
        from sound_bubble_amd.eval_samples import load_testcase, run_testcase
        _, mix, _, _ = load_testcase(os.path.join(GOLDEN, "test_samples", sset, scene), thr)
        out = run_testcase(m, mix, thr)
        assert out.shape == (1, n) and rel_l2(out, g[f"{sset}/{scene}::output"]) < 2e-4
        wav, sr = read_wav(os.path.join(out_dir, f"{scene}_output.wav"))       # the written demo output round-trips
        assert sr == 24000 and wav.shape[-1] == n


# ---- BASELINE configs[0] at its OWN size (round 4): the 6-block 0.5 M network of
# syn_experiments/pretrain_stage.json:8-27 on syn_1m/00001 at its full 5 s, as src/test_samples.py:90-112 runs it ----
FULL = os.path.join(GOLDEN, "test_samples_full", "syn_1m")


def _golden_6block():
    rec, params, flavour = load_golden("samples_6block")
    assert params["B"] == 6 and params["D"] == 32 and flavour == "dis_embd3"
    return rec, params, flavour


def test_oracle_on_the_full_scene_through_the_six_block_model(torch_mod):
    torch = torch_mod
    from oracle.tfgridnet_oracle import OracleNet
    from sound_bubble_amd.eval_samples import load_testcase, si_sdr_np
    rec, params, flavour = _golden_6block()
    m = OracleNet(flavour, **params).eval()
    m.load_state_dict(golden_state_dict(rec, torch))
    assert sum(p.numel() for p in m.parameters()) == 501398
    _, mix, gt, tg = load_testcase(os.path.join(FULL, "00001"), 1.0)
    assert mix.shape == (6, 120000) and len(tg) == int(rec["n_targets"]) == 1
    np.testing.assert_array_equal(gt, rec["gt"])
    with torch.no_grad():
        out = m({"mixture": torch.from_numpy(mix)[None], "dis_embed": torch.tensor([[0.0, 0.0, 1.0]])})["output"][0].numpy()
    assert rel_l2(out, rec["output"]) < 5e-6
    assert abs(si_sdr_np(out[0], gt[0]) - float(rec["si_sdr"])) < 0.05
    assert abs(si_sdr_np(mix[0], gt[0]) - float(rec["input_si_sdr"])) < 1e-6


@pytest.mark.gpu
def test_hip_six_block_model_on_the_full_scene():
    """evaluate_dir over the untrimmed scene with the 6-block model: rel-L2 <= 2e-4 against the REFERENCE model's output
    (north-star bar 1e-3), SI-SDR within 0.05 dB of the reference's value"""
    import torch
    import sound_bubble_amd as sb
    from sound_bubble_amd.eval_samples import evaluate_dir, load_testcase, run_testcase
    rec, params, _ = _golden_6block()
    m = sb.NetDisEmbd3(**params)
    m.load_state_dict(golden_state_dict(rec, torch))
    m = m.cuda().eval()
    rows = {r["sample"]: r for r in evaluate_dir(m, FULL, 1.0)}
    row = rows["00001"]
    assert row["n_targets"] == 1
    assert abs(row["si_sdr"] - float(rec["si_sdr"])) < 0.05 and abs(row["input_si_sdr"] - float(rec["input_si_sdr"])) < 1e-4
    _, mix, _, _ = load_testcase(os.path.join(FULL, "00001"), 1.0)
    out = run_testcase(m, mix, 1.0)
    assert out.shape == (1, 120000) and rel_l2(out, rec["output"]) < 2e-4
