"""GPU tests of the training plumbing: fused clip+Adam vs torch.optim.Adam on the oracle, the PLModule
protocol driven like train_pt.py, checkpoints, on-device metrics."""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, load_golden, golden_state_dict, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("clip", [None, 1.0])
def test_two_train_steps_match_oracle_adam(torch_gpu, clip, monkeypatch):
    """Optimizer arithmetic parity (fused clip + Adam == clip_grad_norm_ + torch.optim.Adam).  Runs with the exact
    fp32 BPTT records: Adam's first steps are ~lr*sign(g), which turns the 1e-4-level gradient rounding of the
    compact fp16 records into visible parameter differences on near-zero-gradient entries."""
    torch = torch_gpu
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    monkeypatch.setattr(ops, "BPTT", "wide")
    from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step
    from oracle.tfgridnet_oracle import OracleNet, snrlp_loss
    rec, params, flavour = load_golden("tiny_small")
    sd = golden_state_dict(rec, torch)
    m = sb.NetOptim(**params)
    m.load_state_dict(sd)
    m = m.cuda().train()
    bucket = FlatBucket(m)
    opt = FusedAdam(bucket, lr=2e-3)
    o = OracleNet(flavour, **params).train()
    o.load_state_dict(sd)
    oopt = torch.optim.Adam(o.parameters(), lr=2e-3)
    mix, tgt = torch.from_numpy(rec["mixture"]), torch.from_numpy(rec["target"])
    for _ in range(2):
        loss = train_step(m, bucket, opt, {"mixture": mix.cuda()}, tgt.cuda(), 100.0, grad_clip=clip)
        oopt.zero_grad()
        ol = snrlp_loss(o({"mixture": mix})["output"], tgt, 100.0).mean()
        ol.backward()
        if clip:
            torch.nn.utils.clip_grad_norm_(o.parameters(), clip)
        oopt.step()
        assert abs(float(loss) - float(ol.detach())) < 1e-3 * max(1.0, abs(float(ol.detach())))
    worst = 0.0
    od = dict(o.named_parameters())
    for k, p in m.named_parameters():
        worst = max(worst, rel_l2(p.detach().cpu().numpy(), od[k].detach().numpy()))
    assert worst < 2e-4, worst


def test_training_trajectory_matches_oracle_in_default_mode(torch_gpu):
    """What a user of the reference cares about: with the DEFAULT kernel set (fp16x3 forward, compact fp16 BPTT
    records, scaled fp16 dgates) a short training run follows the oracle's loss curve.  12 clip+Adam steps on a
    fixed batch; per-step loss within 2e-3 relative (the curve drops by far more than that, so it is really the
    same optimisation), final SI-SDR-style quantity (the loss itself) within 0.05 dB."""
    torch = torch_gpu
    import sound_bubble_amd as sb
    from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step
    from oracle.tfgridnet_oracle import OracleNet, snrlp_loss
    rec, params, flavour = load_golden("tiny_small")
    sd = golden_state_dict(rec, torch)
    m = sb.NetOptim(**params)
    m.load_state_dict(sd)
    m = m.cuda().train()
    bucket = FlatBucket(m)
    opt = FusedAdam(bucket, lr=1e-3)
    o = OracleNet(flavour, **params).train()
    o.load_state_dict(sd)
    oopt = torch.optim.Adam(o.parameters(), lr=1e-3)
    mix, tgt = torch.from_numpy(rec["mixture"]), torch.from_numpy(rec["target"])
    ours, ref = [], []
    for _ in range(12):
        ours.append(float(train_step(m, bucket, opt, {"mixture": mix.cuda()}, tgt.cuda(), 100.0, grad_clip=1.0)))
        oopt.zero_grad()
        ol = snrlp_loss(o({"mixture": mix})["output"], tgt, 100.0).mean()
        ol.backward()
        torch.nn.utils.clip_grad_norm_(o.parameters(), 1.0)
        oopt.step()
        ref.append(float(ol.detach()))
    ours, ref = np.array(ours), np.array(ref)
    assert abs(ref[0] - ref[-1]) > 20 * 2e-3 * abs(ref[0]), "the run must actually optimise something"
    np.testing.assert_allclose(ours, ref, rtol=2e-3, atol=0.05)


def test_batch_metrics_match_definitions(torch_gpu):
    torch = torch_gpu
    from sound_bubble_amd.metrics import batch_metrics
    from oracle.tfgridnet_oracle import si_sdr_np
    g = torch.Generator().manual_seed(0)
    B, N = 3, 24000
    gt = torch.randn(B, 1, N, generator=g) * 0.1
    est = gt * 0.7 + 0.05 * torch.randn(B, 1, N, generator=g) + 0.01
    mix = torch.randn(B, 6, N, generator=g) * 0.1 + gt
    m = batch_metrics(est.cuda(), gt.cuda(), mix.cuda()[:, 0], ("snr", "si_sdr", "si_snr", "si_sdr_i"))
    for b in range(B):
        e, t, x = est[b, 0].double().numpy(), gt[b, 0].double().numpy(), mix[b, 0].double().numpy()
        assert abs(m["si_sdr"][b] - si_sdr_np(e, t)) < 2e-3                 # helpers/eval_utils.py formula
        assert abs(m["snr"][b] - si_sdr_np(e, t, scale_invariant=False)) < 2e-3
        assert abs(m["si_sdr_i"][b] - (si_sdr_np(e, t) - si_sdr_np(x, t))) < 4e-3
        assert abs(m["si_snr"][b] - si_sdr_np(e - e.mean(), t - t.mean())) < 2e-3
        assert abs(m["decay"][b] - (10 * np.log10((x ** 2).sum()) - 10 * np.log10((e ** 2).sum()))) < 2e-3


def test_plmodule_protocol_and_checkpoint(torch_gpu, tmp_path):
    """Drive the harness exactly as src/train_pt.py does, from a reference-style experiment JSON."""
    torch = torch_gpu
    from sound_bubble_amd.harness import import_attr
    from sound_bubble_amd.train_cli import train_epoch, test_epoch
    from sound_bubble_amd.data import SyntheticBubbleDataset
    params = json.load(open(os.path.join(ROOT, "experiments", "bubble_small_synthetic.json")))
    params["pl_module_args"]["model_params"]["B"] = 1
    hl = import_attr(params["pl_module"])(**params["pl_module_args"])
    assert abs(hl.get_current_lr() - 2e-4) < 1e-9                     # LinearLR start_factor 0.1 * 2e-3
    ds = SyntheticBubbleDataset(n_items=8, n_samples=4800, with_dis_embed=False, silent_every=4)
    loader = torch.utils.data.DataLoader(ds, batch_size=4)
    w0 = hl.model.tfgridnet.deconv.weight.detach().clone()
    hl.on_epoch_start()
    tl = train_epoch(hl, loader, "cuda")
    vl = test_epoch(hl, loader, "cuda")
    assert np.isfinite(tl) and np.isfinite(vl)
    assert not torch.equal(w0, hl.model.tfgridnet.deconv.weight)
    best, last = str(tmp_path / "best.pt"), str(tmp_path / "last.pt")
    hl.on_epoch_end(best, None)
    hl.dump_state(last)
    assert os.path.exists(best) and hl.epoch == 1 and hl.get_current_lr() > 2e-4
    for k in ("train/loss", "val/loss", "val/si_sdr_i", "val/decay"):
        assert k in hl.metric_values[0], k
    state = torch.load(last, weights_only=False)
    assert set(state) >= {"model", "optimizer", "current_epoch", "metric_values", "statistics", "scheduler"}
    assert "tfgridnet.enc.filterbank._filters" in state["model"]          # reference key names
    hl2 = import_attr(params["pl_module"])(**params["pl_module_args"])
    hl2.load_state(last)
    assert hl2.epoch == 1 and abs(hl2.get_current_lr() - hl.get_current_lr()) < 1e-12
    assert torch.equal(hl2.model.tfgridnet.deconv.weight, hl.model.tfgridnet.deconv.weight)
    assert hl2.optimizer.step_count == hl.optimizer.step_count


@pytest.mark.parametrize("exact", [True, False], ids=["wide-bptt", "compact-bptt"])
def test_resume_from_reference_format_checkpoint_takes_the_reference_third_step(torch_gpu, exact, monkeypatch):
    """Row f2: a last.pt in the REFERENCE's layout (torch.optim.Adam state_dict under 'optimizer', SequentialLR state
    under 'scheduler'; written by the reference Net + torch Adam after two steps, tests/golden/make_goldens.py) is
    loaded by PLModule.load_state and the third clip+Adam step on the HIP path lands where the reference's third step
    does: the parameter UPDATE (not just the parameters) and an Adam moment are compared.  hl_module:115-156."""
    torch = torch_gpu
    from sound_bubble_amd import ops
    from sound_bubble_amd.harness import PLModule
    from conftest import GOLDEN
    monkeypatch.setattr(ops, "BPTT", "wide" if exact else "compact")
    rec, params, _ = load_golden("tiny_small")
    gold = np.load(os.path.join(GOLDEN, "ckpt_resume_tiny_small.npz"))
    path = os.path.join(GOLDEN, "ref_format_last_tiny_small.pt")
    sched = [{"name": "torch.optim.lr_scheduler.LinearLR", "params": {"start_factor": 0.1, "total_iters": 10}, "epochs": 10},
             {"name": "torch.optim.lr_scheduler.ConstantLR", "params": {"factor": 1}, "epochs": 20},
             {"name": "torch.optim.lr_scheduler.StepLR", "params": {"step_size": 2, "gamma": 0.95}, "epochs": 120}]
    hl = PLModule(model="src.models.tfgridnet_realtime_clean_optim.net.Net", model_params=params, sr=24000,
                  optimizer="torch.optim.Adam", optimizer_params={"lr": 2e-3}, scheduler="sequential",
                  scheduler_params=sched, loss="src.losses.SNRLP.SNRLPLoss",
                  loss_params={"snr_loss_name": "snr", "neg_weight": 50}, metrics=["si_sdr_i"], grad_clip=1)
    hl.load_state(path)
    assert abs(hl.get_current_lr() - float(gold["lr_at_step3"])) < 1e-12 and hl.optimizer.step_count == 2
    before = {k: p.detach().clone() for k, p in hl.model.named_parameters()}
    hl.train()
    batch = ({"mixture": torch.from_numpy(rec["mixture"]).cuda()},
             {"target": torch.from_numpy(rec["target"]).cuda(), "num_target_speakers": torch.tensor([1, 0]),
              "num_interfering_speakers": torch.tensor([0, 0]), "num_noises": torch.tensor([1, 1])})
    hl.reset_grad()
    loss, _ = hl.training_step(batch, 0)
    loss.backward()
    hl.backprop()
    assert abs(float(loss.detach()) - float(gold["loss"][2])) < 2e-3 * abs(float(gold["loss"][2]))
    num = den = 0.0
    for k, p in hl.model.named_parameters():
        want = torch.from_numpy(gold["param_after3::" + k]).cuda()
        num += float(((p.detach() - before[k]) - (want - before[k])).double().square().sum())
        den += float((want - before[k]).double().square().sum())
    err = (num / den) ** 0.5
    assert err < (2e-3 if exact else 2e-2), err                 # relative error of the whole parameter UPDATE
    names = [k for k, _ in hl.model.named_parameters()]
    for k in gold.files:
        if k.startswith("exp_avg_after3::"):
            i = names.index(k.split("::", 1)[1])
            p, o = hl.bucket.params[i], hl.bucket.offsets[i]
            got = hl.optimizer.m[o:o + p.numel()].view(p.shape).cpu().numpy()
            assert rel_l2(got, gold[k]) < (2e-4 if exact else 2e-3), k


def test_weight_forms_follow_parameter_updates(torch_gpu):
    """forms.WeightForms: the kernel-layout copies of the conv weights are refreshed when a parameter changes -- through
    torch (version counters), through the fused Adam kernel (weight epoch), after FlatBucket re-points the parameters --
    and NOT re-gathered in an inference loop."""
    torch = torch_gpu
    import sound_bubble_amd as sb
    from sound_bubble_amd import forms
    from sound_bubble_amd.train import FlatBucket, FusedAdam
    rec, params, _ = load_golden("tiny_small")
    m = sb.NetOptim(**params)
    m.load_state_dict(golden_state_dict(rec, torch))
    m = m.cuda().eval()
    x = {"mixture": torch.from_numpy(rec["mixture"]).cuda()}
    with torch.no_grad():
        y0 = m(x)["output"].clone()
        wf = m._weight_forms()
        src0 = wf.source_key()
        y1 = m(x)["output"]
        assert wf.source_key() == src0 and torch.equal(y0, y1)        # nothing changed: same job table, same result
        m.tfgridnet.blocks[0].conv.weight.mul_(1.5)                    # torch in-place update
        y2 = m(x)["output"].clone()
        assert wf.source_key() == src0 and not torch.equal(y2, y0)     # same addresses, fresh forms (refresh at every forward)
        bucket = FlatBucket(m)                                         # parameters move into the flat bucket
        y3 = m(x)["output"]
        assert torch.equal(y3, y2)
        opt = FusedAdam(bucket, lr=1e-2)
        bucket.grad.fill_(1e-3)
        e0 = forms.WEIGHT_EPOCH
        opt.step()                                                     # the HIP kernel writes behind torch's back
        assert forms.WEIGHT_EPOCH == e0 + 1
        y4 = m(x)["output"]
        assert not torch.equal(y4, y2)
        ref = sb.NetOptim(**params).cuda().eval()                      # a fresh model with the updated weights agrees
        ref.load_state_dict(m.state_dict())
        assert torch.equal(ref(x)["output"], y4)
