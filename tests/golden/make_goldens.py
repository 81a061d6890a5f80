#!/usr/bin/env python3
"""Generate golden input/output vectors from the *reference* Sound_Bubble model.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  It imports the reference `Net` classes

    src/models/tfgridnet_realtime_clean_dis_embd3/net.py:20   ("big" family)
    src/models/tfgridnet_realtime_clean_optim/net.py:20       ("small" family)

with three tiny stand-ins for third-party packages that are absent from this
image (espnet2.get_layer, espnet2.AbsSeparator, asteroid_filterbanks.make_enc_dec
-- see SURVEY.md Appendix B).  The stand-ins are written to a temp dir at run
time and are never imported by the product or by the tests.

What it emits (tests/golden/*.npz): seeded weights (the reference state_dict),
seeded inputs, the reference outputs, per-stage intermediates captured with
forward hooks, next_state, a 3-chunk streaming trace, and parameter gradients
of the SNRLP pre-train loss.  Fixtures are data only; no reference source text
is stored.

PARITY CAVEAT: the STFT filter bank comes from the third-party
`asteroid_filterbanks` (unpinned in requirements2.txt:15) which is not
installed here.  The filter values in the fixtures come from the stand-in,
which restates asteroid's published STFTFB formula.  The reference stores the
filters as state_dict buffers (tfgridnet.{enc,dec}.filterbank._filters), so a
real checkpoint overrides them; everything downstream of the filters is stock
torch.nn and therefore a faithful oracle.

Usage:  python tests/golden/make_goldens.py  [--out tests/golden]
"""
import argparse
import importlib
import os
import sys
import tempfile
import textwrap

import numpy as np

REF = "/root/reference"


def _write_shims(root):
    def w(rel, body=""):
        p = os.path.join(root, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(body))

    for pkg in ("espnet2", "espnet2/torch_utils", "espnet2/enh", "espnet2/enh/separator"):
        w(pkg + "/__init__.py")
    w("espnet2/torch_utils/get_layer_from_string.py", """
        import torch
        def get_layer(l_name, library=torch.nn):
            for k in dir(library):
                if k.lower() == l_name.lower():
                    return getattr(library, k)
            raise NotImplementedError(l_name)
        """)
    w("espnet2/enh/separator/abs_separator.py", """
        import torch
        class AbsSeparator(torch.nn.Module):
            pass
        """)
    # asteroid_filterbanks stand-in: STFTFB + Encoder + Decoder (published formula)
    w("asteroid_filterbanks/__init__.py", """
        import numpy as np, torch, torch.nn as nn, torch.nn.functional as F
        class STFTFB(nn.Module):
            def __init__(self, n_filters, kernel_size, stride=None, window=None,
                         sample_rate=8000.0, **kwargs):
                super().__init__()
                self.n_filters, self.kernel_size = n_filters, kernel_size
                self.stride = stride if stride else kernel_size // 2
                self.cutoff = int(n_filters / 2 + 1)
                self.n_feats_out = 2 * self.cutoff
                if window is None:
                    window = np.hanning(kernel_size + 1)[:-1] ** 0.5
                lpad = int((n_filters - kernel_size) // 2)
                rpad = int(n_filters - kernel_size - lpad)
                self.window = np.concatenate([np.zeros((lpad,)), window, np.zeros((rpad,))])
                filters = np.fft.fft(np.eye(n_filters))
                filters /= 0.5 * np.sqrt(kernel_size * n_filters / self.stride)
                filters = np.vstack([np.real(filters[: self.cutoff, :]),
                                     np.imag(filters[: self.cutoff, :])])
                filters[0, :] /= np.sqrt(2)
                filters[n_filters // 2, :] /= np.sqrt(2)
                filters = torch.from_numpy(filters * self.window).unsqueeze(1).float()
                self.register_buffer("_filters", filters)
                self.register_buffer("_sample_rate", torch.zeros(1) + sample_rate)
            def filters(self):
                return self._filters
        class Encoder(nn.Module):
            def __init__(self, filterbank, padding=0):
                super().__init__()
                self.filterbank, self.stride, self.padding = filterbank, filterbank.stride, padding
            def forward(self, x):
                shp = x.shape
                y = F.conv1d(x.reshape(-1, 1, shp[-1]), self.filterbank.filters(),
                             stride=self.stride, padding=self.padding)
                return y.view(shp[:-1] + y.shape[-2:])
        class Decoder(nn.Module):
            def __init__(self, filterbank, padding=0, output_padding=0):
                super().__init__()
                self.filterbank, self.stride = filterbank, filterbank.stride
                self.padding, self.output_padding = padding, output_padding
            def forward(self, spec):
                shp = spec.shape
                y = F.conv_transpose1d(spec.reshape(-1, shp[-2], shp[-1]), self.filterbank.filters(),
                                       stride=self.stride, padding=self.padding,
                                       output_padding=self.output_padding)
                return y.view(shp[:-2] + (-1,))
        def make_enc_dec(fb_name, n_filters, kernel_size, stride=None, sample_rate=8000.0,
                         who_is_pinv=None, padding=0, output_padding=0, **kwargs):
            assert fb_name == "stft"
            kwargs.pop("window_type", None)   # upstream swallows it via **kwargs
            enc = Encoder(STFTFB(n_filters, kernel_size, stride=stride, sample_rate=sample_rate), padding=padding)
            dec = Decoder(STFTFB(n_filters, kernel_size, stride=stride, sample_rate=sample_rate),
                          padding=padding, output_padding=output_padding)
            return enc, dec
        """)


def _flatten_state(d, prefix=""):
    out = {}
    for k in sorted(d.keys()):
        v = d[k]
        if isinstance(v, dict):
            out.update(_flatten_state(v, prefix + k + "::"))
        else:
            out[prefix + k] = v.detach().cpu().numpy().copy()
    return out


def snrlp_loss(torch, est, gt, neg_weight):
    """SNRLPLoss restated for the generator (src/losses/SNRLP.py:17-42 with
    asteroid SingleSrcNegSDR('snr'): zero-mean, EPS=1e-8).  The harness the
    reference uses (PLModule) cannot be imported here (wandb/torchmetrics/...)."""
    B = est.shape[0]
    comp = torch.zeros(B, dtype=est.dtype)
    mask = gt.abs().amax(dim=(1, 2)) == 0
    if mask.any():
        comp[mask] = (est[mask] - gt[mask]).abs().mean() * neg_weight
    if (~mask).any():
        e = est[~mask].reshape(-1, est.shape[-1])
        t = gt[~mask].reshape(-1, gt.shape[-1])
        e = e - e.mean(dim=1, keepdim=True)
        t = t - t.mean(dim=1, keepdim=True)
        ratio = (t ** 2).sum(1) / (((e - t) ** 2).sum(1) + 1e-8)
        comp[~mask] = -10 * torch.log10(ratio + 1e-8)
    return comp


def make_case(torch, Net, name, params, B, n_frames, seed, needs_dis, out_dir,
              with_grads=True, with_stream=True, with_stages=True):
    torch.manual_seed(seed)
    model = Net(**params).eval()
    chunk, pad = params["stft_chunk_size"], params["stft_pad_size"]
    N = n_frames * chunk - 37          # exercise mod_pad (not a multiple of 192)
    g = torch.Generator().manual_seed(seed + 1)
    base = 0.1 * torch.randn(B, 1, N + 8, generator=g)
    mix = torch.cat([base[..., 4 - min(m, 4):4 - min(m, 4) + N] for m in range(params["num_ch"])], 1)
    mix = (mix + 0.02 * torch.randn(B, params["num_ch"], N, generator=g)).clamp(-1, 1)
    inputs = {"mixture": mix}
    if needs_dis:
        dis = torch.zeros(B, 3)
        for b in range(B):
            dis[b, (b + 1) % 3] = 1.0
        inputs["dis_embed"] = dis

    stages = {}
    hooks = []
    if with_stages:
        tg = model.tfgridnet
        hooks.append(tg.enc.register_forward_hook(lambda m, i, o: stages.__setitem__("stft", o.detach().numpy().copy())))
        hooks.append(tg.conv.register_forward_hook(lambda m, i, o: stages.__setitem__("conv_ln", o.detach().numpy().copy())))
        for bi, blk in enumerate(tg.blocks):
            hooks.append(blk.register_forward_hook(
                lambda m, i, o, bi=bi: stages.__setitem__(f"block{bi}", o[0].detach().numpy().copy())))
        hooks.append(tg.deconv.register_forward_hook(lambda m, i, o: stages.__setitem__("deconv", o.detach().numpy().copy())))

    with torch.no_grad():
        res = model(dict(inputs))
    for h in hooks:
        h.remove()
    out = res["output"].numpy().copy()
    rec = {"output": out, "mixture": mix.numpy()}
    if needs_dis:
        rec["dis_embed"] = inputs["dis_embed"].numpy()
    for k, v in stages.items():
        rec["stage::" + k] = v
    for k, v in _flatten_state(res["next_state"]).items():
        rec["next_state::" + k] = v
    for k, v in model.state_dict().items():
        if k.endswith("filterbank._filters"):
            continue                      # stored once in stft_filters.npz (enc == dec)
        rec["param::" + k] = v.numpy().copy()

    if with_grads:
        model.train()
        g2 = torch.Generator().manual_seed(seed + 2)
        tgt = 0.05 * torch.randn(B, 1, N, generator=g2)
        tgt[B - 1] = 0.0                                   # one silent-target sample
        model.zero_grad()
        est = model(dict(inputs))["output"]
        loss_vec = snrlp_loss(torch, est, tgt, neg_weight=100.0)
        loss = loss_vec.mean()
        loss.backward()
        rec["target"] = tgt.numpy()
        rec["loss_vec"] = loss_vec.detach().numpy()
        rec["loss"] = np.float32(loss.item())
        for k, p in model.named_parameters():
            rec["grad::" + k] = p.grad.numpy().copy()
        model.eval()

    if with_stream:
        # causal_infer-style: 3 chunks of [B, M, chunk+pad], pad=False, carried state
        x = mix[..., : 3 * chunk + pad]
        state = model.init_buffers(B, "cpu")
        outs = []
        with torch.no_grad():
            for c in range(3):
                fr = {"mixture": x[..., c * chunk: c * chunk + chunk + pad]}
                if needs_dis:
                    fr["dis_embed"] = inputs["dis_embed"]
                r = model(fr, state, pad=False)
                state = r["next_state"]
                outs.append(r["output"].numpy().copy())
                if c == 2:
                    for k, v in _flatten_state(state).items():
                        rec["stream::state::" + k] = v
        rec["stream::output"] = np.concatenate(outs, -1)
        rec["stream::input"] = x.numpy().copy()

    rec["meta::params"] = np.array(repr(sorted(params.items())))
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **rec)
    nparam = sum(p.numel() for p in model.parameters())
    print(f"{name}: params={nparam} out_rms={np.sqrt((out**2).mean()):.4e} -> {path} "
          f"({os.path.getsize(path)/1e6:.2f} MB)")


def grad_fingerprint(np_grad, key, nproj=16):
    """L2 norm and `nproj` seeded +-1 projections of a gradient tensor, in float64: what the 6-block default-constructor fixtures
    keep for the tensors whose full gradient they do not store (a random +-1 projection of an error vector e has magnitude
    ~|e|, so |delta projection| / |g| estimates the relative L2 error)."""
    import zlib
    g = np.asarray(np_grad, np.float64).ravel()
    rng = np.random.default_rng(zlib.crc32(key.encode()))
    signs = rng.integers(0, 2, size=(nproj, g.size), dtype=np.int8) * 2 - 1
    return np.concatenate([[np.sqrt((g * g).sum())], signs @ g])


def make_default_ctor(torch, Net, name, seed, needs_dis, out_dir, B=2, n_frames=6):
    """The reference constructor's own defaults (net.py:21-26: stft 160 / 120 -> n_fft 280, F 141; num_ch 2, D 64, H 128,
    six blocks, conv-LSTM intra path, merge_method "None", no first LayerNorm) -- every default but ONE: Net() itself raises
    ZeroDivisionError in the reference (L = 0 heads -> `emb_dim // n_head`, tfgridnet_causal.py:596 / optim :484; see
    make_ctor_behaviour), so L = 4, the value of every shipped JSON (unused without attention).  ~2.4 M parameters: the weights are NOT stored
    -- the test rebuilds them from the seed (same initialisers in the same order; per-tensor sums / norms are stored as the
    check) -- and full gradients are stored for everything outside the blocks and for the first and the last block; every
    tensor (those included) also gets a fingerprint (grad_fingerprint)."""
    torch.manual_seed(seed)
    model = Net(L=4).eval()
    chunk, pad, num_ch = 160, 120, 2
    N = n_frames * chunk - 37
    g = torch.Generator().manual_seed(seed + 1)
    base = 0.1 * torch.randn(B, 1, N + 8, generator=g)
    mix = torch.cat([base[..., 4 - min(m, 4):4 - min(m, 4) + N] for m in range(num_ch)], 1)
    mix = (mix + 0.02 * torch.randn(B, num_ch, N, generator=g)).clamp(-1, 1)
    inputs = {"mixture": mix}
    if needs_dis:
        dis = torch.zeros(B, 3)
        for b in range(B):
            dis[b, (b + 1) % 3] = 1.0
        inputs["dis_embed"] = dis
    with torch.no_grad():
        res = model(dict(inputs))
    rec = {"output": res["output"].numpy().copy(), "mixture": mix.numpy(), "meta::seed": np.int64(seed)}
    if needs_dis:
        rec["dis_embed"] = inputs["dis_embed"].numpy()
    for k, v in _flatten_state(res["next_state"]).items():
        rec["next_state::" + k] = v
    for k, v in model.state_dict().items():
        if k.endswith("filterbank._filters") or k.endswith("_sample_rate"):
            continue
        w = v.double().numpy()
        rec["wsum::" + k] = np.array([w.sum(), np.sqrt((w * w).sum())])
    rec["filters"] = model.tfgridnet.enc.filterbank._filters.numpy().copy()
    model.train()
    g2 = torch.Generator().manual_seed(seed + 2)
    tgt = 0.05 * torch.randn(B, 1, N, generator=g2)
    tgt[B - 1] = 0.0
    model.zero_grad()
    est = model(dict(inputs))["output"]
    loss_vec = snrlp_loss(torch, est, tgt, neg_weight=100.0)
    loss_vec.mean().backward()
    rec["target"] = tgt.numpy()
    rec["loss_vec"] = loss_vec.detach().numpy()
    n_blocks = len(model.tfgridnet.blocks)
    for k, p in model.named_parameters():
        gnp = p.grad.numpy()
        rec["gfp::" + k] = grad_fingerprint(gnp, k)
        inner = k.startswith("tfgridnet.blocks.") and int(k.split(".")[2]) not in (0, n_blocks - 1)
        if not inner:
            rec["grad::" + k] = gnp.copy()
    model.eval()
    x = mix[..., : 3 * chunk + pad]
    state = model.init_buffers(B, "cpu")
    outs = []
    with torch.no_grad():
        for c in range(3):
            fr = {"mixture": x[..., c * chunk: c * chunk + chunk + pad]}
            if needs_dis:
                fr["dis_embed"] = inputs["dis_embed"]
            r = model(fr, state, pad=False)
            state = r["next_state"]
            outs.append(r["output"].numpy().copy())
        for k, v in _flatten_state(state).items():
            rec["stream::state::" + k] = v
    rec["stream::output"] = np.concatenate(outs, -1)
    rec["stream::input"] = x.numpy().copy()
    rec["meta::params"] = np.array(repr([("L", 4)]))
    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, **rec)
    nparam = sum(p.numel() for p in model.parameters())
    print(f"{name}: params={nparam} out_rms={np.sqrt((rec['output']**2).mean()):.4e} -> {path} "
          f"({os.path.getsize(path)/1e6:.2f} MB)")


def make_ctor_behaviour(nets, out_dir):
    """What the reference's constructors do when called with NO arguments, and with L alone given: the exception type and
    text, or the parameter count.  (Both raise ZeroDivisionError at their own defaults: L = 0.)"""
    import json
    out = {}
    for tag, Net in nets.items():
        for label, kw in (("no_arguments", {}), ("L=4", {"L": 4})):
            try:
                m = Net(**kw)
                out[f"{tag}::{label}"] = {"constructs": True, "parameters": sum(p.numel() for p in m.parameters()),
                                          "n_freqs": int(m.tfgridnet.n_freqs) if hasattr(m.tfgridnet, "n_freqs") else None}
            except Exception as e:       # noqa: BLE001 -- the point is to record whatever the reference does
                out[f"{tag}::{label}"] = {"constructs": False, "exception": type(e).__name__, "message": str(e)}
    with open(os.path.join(out_dir, "ctor_behaviour.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("ctor_behaviour.json:", out)


def make_samples(torch, NetBig, out_dir, n_keep=36000):
    """test_samples/syn_1m scenes (reference fixtures, MIT licence), trimmed to 1.5 s, pushed through the REFERENCE
    model (weights of tiny_big.npz) with the reference's own NumPy metrics (helpers/eval_utils.py)."""
    import json
    import wave
    import shutil
    sys.path.insert(0, os.path.join(REF, "helpers"))
    import eval_utils                                            # reference helpers/eval_utils.py
    z = np.load(os.path.join(out_dir, "tiny_big.npz"))
    params = dict(eval(str(z["meta::params"])))
    model = NetBig(**params).eval()
    sd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param::")}
    filt = torch.from_numpy(np.load(os.path.join(out_dir, "stft_filters.npz"))["filters"])
    sd["tfgridnet.enc.filterbank._filters"] = filt
    sd["tfgridnet.dec.filterbank._filters"] = filt.clone()
    model.load_state_dict(sd)
    rec = {}
    for scene in ("00000", "00001", "00002"):
        src = os.path.join(REF, "test_samples", "syn_1m", scene)
        dst = os.path.join(out_dir, "test_samples", "syn_1m", scene)
        os.makedirs(dst, exist_ok=True)
        shutil.copyfile(os.path.join(src, "metadata.json"), os.path.join(dst, "metadata.json"))
        for fn in sorted(os.listdir(src)):
            if fn.endswith(".wav"):
                with wave.open(os.path.join(src, fn), "rb") as w:
                    par, data = w.getparams(), w.readframes(n_keep)
                with wave.open(os.path.join(dst, fn), "wb") as w:
                    w.setparams(par)
                    w.writeframes(data)
        meta = json.load(open(os.path.join(dst, "metadata.json")))
        with wave.open(os.path.join(dst, "mixture.wav"), "rb") as w:
            mix = np.frombuffer(w.readframes(n_keep), "<i2").reshape(-1, 6).T.astype(np.float32) / 32768.0
        gt = np.zeros((1, mix.shape[1]), np.float32)
        for spk in sorted(k for k in meta if k.startswith("voice")):
            if meta[spk]["dis"] <= 1.0:
                with wave.open(os.path.join(dst, f"mic00_{spk}.wav"), "rb") as w:
                    gt[0] += np.frombuffer(w.readframes(n_keep), "<i2").astype(np.float32) / 32768.0
        with torch.no_grad():
            out = model({"mixture": torch.from_numpy(mix)[None], "dis_embed": torch.tensor([[0.0, 0.0, 1.0]])})["output"][0].numpy()
        rec[scene + "::output"] = out
        rec[scene + "::gt"] = gt
        if np.abs(gt).max() > 0:
            rec[scene + "::si_sdr"] = np.float64(eval_utils.si_sdr(out[0].astype(np.float64), gt[0].astype(np.float64)))
            rec[scene + "::input_si_sdr"] = np.float64(eval_utils.si_sdr(mix[0].astype(np.float64), gt[0].astype(np.float64)))
            rec[scene + "::snr"] = np.float64(eval_utils.snr(out[0].astype(np.float64), gt[0].astype(np.float64)))
            print(scene, "SI-SDR", rec[scene + "::si_sdr"], "input", rec[scene + "::input_si_sdr"])
    np.savez_compressed(os.path.join(out_dir, "samples_syn_1m.npz"), **rec)


def make_samples_more(torch, NetBig, out_dir):
    """The other two radii of test_samples/ (src/test_samples.py:96-104 one-hots: 1.5 m -> [0, 1, 0], 2 m -> [1, 0, 0]):
    syn_1_5m/00001 trimmed to 1.5 s like the syn_1m scenes, and syn_2m/00002 at its FULL 5 s length (625 frames), both
    through the REFERENCE model (weights of tiny_big.npz) with the reference's own NumPy metrics."""
    import json
    import wave
    import shutil
    sys.path.insert(0, os.path.join(REF, "helpers"))
    import eval_utils                                            # reference helpers/eval_utils.py
    z = np.load(os.path.join(out_dir, "tiny_big.npz"))
    params = dict(eval(str(z["meta::params"])))
    model = NetBig(**params).eval()
    sd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param::")}
    filt = torch.from_numpy(np.load(os.path.join(out_dir, "stft_filters.npz"))["filters"])
    sd["tfgridnet.enc.filterbank._filters"] = filt
    sd["tfgridnet.dec.filterbank._filters"] = filt.clone()
    model.load_state_dict(sd)
    rec = {}
    for sset, scene, thr, onehot, n_keep in (("syn_1_5m", "00001", 1.5, [0.0, 1.0, 0.0], 36000),
                                             ("syn_2m", "00002", 2.0, [1.0, 0.0, 0.0], 120000)):
        src = os.path.join(REF, "test_samples", sset, scene)
        dst = os.path.join(out_dir, "test_samples", sset, scene)
        os.makedirs(dst, exist_ok=True)
        shutil.copyfile(os.path.join(src, "metadata.json"), os.path.join(dst, "metadata.json"))
        for fn in sorted(os.listdir(src)):
            if fn.endswith(".wav"):
                with wave.open(os.path.join(src, fn), "rb") as w:
                    par, data = w.getparams(), w.readframes(n_keep)
                with wave.open(os.path.join(dst, fn), "wb") as w:
                    w.setparams(par)
                    w.writeframes(data)
        meta = json.load(open(os.path.join(dst, "metadata.json")))
        with wave.open(os.path.join(dst, "mixture.wav"), "rb") as w:
            mix = np.frombuffer(w.readframes(n_keep), "<i2").reshape(-1, 6).T.astype(np.float32) / 32768.0
        gt = np.zeros((1, mix.shape[1]), np.float32)
        ntg = 0
        for spk in sorted(k for k in meta if k.startswith("voice")):
            if meta[spk]["dis"] <= thr:
                ntg += 1
                with wave.open(os.path.join(dst, f"mic00_{spk}.wav"), "rb") as w:
                    gt[0] += np.frombuffer(w.readframes(n_keep), "<i2").astype(np.float32) / 32768.0
        with torch.no_grad():
            out = model({"mixture": torch.from_numpy(mix)[None], "dis_embed": torch.tensor([onehot])})["output"][0].numpy()
        key = f"{sset}/{scene}"
        rec[key + "::output"] = out
        rec[key + "::gt"] = gt
        rec[key + "::n_targets"] = np.int64(ntg)
        rec[key + "::si_sdr"] = np.float64(eval_utils.si_sdr(out[0].astype(np.float64), gt[0].astype(np.float64)))
        rec[key + "::input_si_sdr"] = np.float64(eval_utils.si_sdr(mix[0].astype(np.float64), gt[0].astype(np.float64)))
        rec[key + "::snr"] = np.float64(eval_utils.snr(out[0].astype(np.float64), gt[0].astype(np.float64)))
        print(key, mix.shape, "targets", ntg, "SI-SDR", rec[key + "::si_sdr"], "input", rec[key + "::input_si_sdr"])
    np.savez_compressed(os.path.join(out_dir, "samples_more.npz"), **rec)


def make_samples_6block(torch, NetBig, big_params, out_dir):
    """BASELINE configs[0] at its stated model: test_samples/syn_1m/00001 at its FULL 5 s (120 000 samples, 625 frames)
    through the REFERENCE 6-block 0.5 M-parameter network of syn_experiments/pretrain_stage.json:8-27 -- seeded default-init
    weights (no trained checkpoint exists offline), stored with the fixture -- as src/test_samples.py:90-112 runs it (one-hot
    [0, 0, 1] for the 1 m bubble, GT = sum of the mic00 voices within 1 m), with the reference's own NumPy metrics."""
    import json
    import wave
    import shutil
    sys.path.insert(0, os.path.join(REF, "helpers"))
    import eval_utils                                            # reference helpers/eval_utils.py
    torch.manual_seed(20260930)
    model = NetBig(**big_params).eval()
    assert sum(p.numel() for p in model.parameters()) == 501398
    src = os.path.join(REF, "test_samples", "syn_1m", "00001")
    dst = os.path.join(out_dir, "test_samples_full", "syn_1m", "00001")
    os.makedirs(dst, exist_ok=True)
    for fn in sorted(os.listdir(src)):                           # the scene's own files (MIT-licensed fixtures), untrimmed
        shutil.copyfile(os.path.join(src, fn), os.path.join(dst, fn))
    meta = json.load(open(os.path.join(dst, "metadata.json")))
    with wave.open(os.path.join(dst, "mixture.wav"), "rb") as w:
        mix = np.frombuffer(w.readframes(w.getnframes()), "<i2").reshape(-1, 6).T.astype(np.float32) / 32768.0
    gt = np.zeros((1, mix.shape[1]), np.float32)
    ntg = 0
    for spk in sorted(k for k in meta if k.startswith("voice")):
        if meta[spk]["dis"] <= 1.0:
            ntg += 1
            with wave.open(os.path.join(dst, f"mic00_{spk}.wav"), "rb") as w:
                gt[0] += np.frombuffer(w.readframes(w.getnframes()), "<i2").astype(np.float32) / 32768.0
    with torch.no_grad():
        out = model({"mixture": torch.from_numpy(mix)[None], "dis_embed": torch.tensor([[0.0, 0.0, 1.0]])})["output"][0].numpy()
    rec = {"meta::params": np.array(repr(sorted(big_params.items()))), "output": out, "gt": gt, "n_targets": np.int64(ntg),
           "si_sdr": np.float64(eval_utils.si_sdr(out[0].astype(np.float64), gt[0].astype(np.float64))),
           "input_si_sdr": np.float64(eval_utils.si_sdr(mix[0].astype(np.float64), gt[0].astype(np.float64))),
           "snr": np.float64(eval_utils.snr(out[0].astype(np.float64), gt[0].astype(np.float64)))}
    for k, p in model.state_dict().items():                       # as make_case: everything but the (shared) STFT filters
        if not k.endswith("_filters"):
            rec["param::" + k] = p.detach().numpy().copy()
    path = os.path.join(out_dir, "samples_6block.npz")
    np.savez_compressed(path, **rec)
    print("samples_6block:", mix.shape, "targets", ntg, "SI-SDR", rec["si_sdr"], "input", rec["input_si_sdr"],
          f"-> {path} ({os.path.getsize(path) / 1e6:.2f} MB)")


def copy_all_samples_full(out_dir):
    """All nine bundled demo scenes (test_samples/syn_{1m,1_5m,2m}/0000{0,1,2}: the reference's MIT-licensed fixtures) at
    their full 5 s under test_samples_full/ -- the scene folders sound_bubble_amd.data.BubbleFolderDataset trains on
    (experiments/overfit_test_samples.json) and eval_samples evaluates.  Data only."""
    import shutil
    for sset in ("syn_1m", "syn_1_5m", "syn_2m"):
        for scene in ("00000", "00001", "00002"):
            src = os.path.join(REF, "test_samples", sset, scene)
            dst = os.path.join(out_dir, "test_samples_full", sset, scene)
            os.makedirs(dst, exist_ok=True)
            for fn in sorted(os.listdir(src)):
                shutil.copyfile(os.path.join(src, fn), os.path.join(dst, fn))
    print("test_samples_full: 9 scenes copied")


def make_state_io(torch, nets, out_dir):
    """edge/flatbuf.py:8-25 name order: the reference's own flatten_state_buffers over init_buffers of each family
    (+ attention buffers), and the reference's named_parameters() order (= torch.optim.Adam state indices)."""
    import json
    sys.path.insert(0, os.path.join(REF, "edge"))
    flatbuf = importlib.import_module("flatbuf")                   # reference edge/flatbuf.py
    rec = {}
    for name, (Net, params) in nets.items():
        model = Net(**params).eval()
        names, bufs = flatbuf.flatten_state_buffers(model.init_buffers(1, "cpu"))
        back = flatbuf.unflatten_state_buffers(names, bufs)
        n2, _ = flatbuf.flatten_state_buffers(back)
        assert n2 == names
        rec[name] = {"params": repr(sorted(params.items())), "state_names": names,
                     "state_shapes": [list(b.shape) for b in bufs],
                     "parameter_order": [k for k, _ in model.named_parameters()]}
    with open(os.path.join(out_dir, "state_io.json"), "w") as f:
        json.dump(rec, f, indent=1)
    print("state_io.json:", {k: len(v["state_names"]) for k, v in rec.items()})


def make_ckpt(torch, NetSmall, out_dir):
    """A checkpoint in the REFERENCE's format (PLModule.dump_state, hl_module:141-156: {'model', 'optimizer' =
    torch.optim.Adam.state_dict(), 'current_epoch', 'metric_values', 'statistics', 'scheduler'}) written after two
    optimiser steps / scheduler epochs of the reference Net (tiny_small weights and batch, raspberrypi_model_pretrain.json's
    Adam + sequential scheduler + grad_clip 1), and the parameters the reference reaches with a THIRD step from it.
    (PLModule itself cannot be imported here -- wandb / torchmetrics / asteroid -- so its dump_state dict is restated.)"""
    z = np.load(os.path.join(out_dir, "tiny_small.npz"))
    params = dict(eval(str(z["meta::params"])))
    model = NetSmall(**params).train()
    sd = {k[7:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param::")}
    filt = torch.from_numpy(np.load(os.path.join(out_dir, "stft_filters.npz"))["filters"])
    sd["tfgridnet.enc.filterbank._filters"] = filt
    sd["tfgridnet.dec.filterbank._filters"] = filt.clone()
    model.load_state_dict(sd)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    L = torch.optim.lr_scheduler
    sched = L.SequentialLR(opt, [L.LinearLR(opt, start_factor=0.1, total_iters=10), L.ConstantLR(opt, factor=1),
                                 L.StepLR(opt, step_size=2, gamma=0.95)], [10, 30])     # hl_module:460-481
    mix, tgt = torch.from_numpy(z["mixture"]), torch.from_numpy(z["target"])

    def step():
        opt.zero_grad()
        loss = snrlp_loss(torch, model({"mixture": mix})["output"], tgt, 50.0).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        return float(loss.detach())

    losses = []
    for epoch in range(2):
        losses.append(step())
        sched.step()
    state = dict(model=model.state_dict(), optimizer=opt.state_dict(), current_epoch=2,
                 metric_values={0: {"val/loss": {"step": None, "epoch": 3.0, "num_elements": 2}},
                                1: {"val/loss": {"step": None, "epoch": 2.5, "num_elements": 2}}},
                 statistics={}, scheduler=sched.state_dict())
    torch.save(state, os.path.join(out_dir, "ref_format_last_tiny_small.pt"))
    lr3 = opt.param_groups[0]["lr"]
    losses.append(step())
    rec = {"loss": np.array(losses, np.float64), "lr_at_step3": np.float64(lr3)}
    for k, p in model.named_parameters():
        rec["param_after3::" + k] = p.detach().numpy().copy()
    for i, (k, p) in enumerate(model.named_parameters()):
        if k.endswith("inter_rnn.weight_hh_l0"):
            rec["exp_avg_after3::" + k] = opt.state[p]["exp_avg"].numpy().copy()
    np.savez_compressed(os.path.join(out_dir, "ckpt_resume_tiny_small.npz"), **rec)
    print("ref_format_last_tiny_small.pt + ckpt_resume_tiny_small.npz: losses", losses, "lr", lr3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--only", default="", help="comma-separated case names (default: regenerate everything)")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    def case(torch, cls, name, *a, **k):
        if not only or name in only:
            make_case(torch, cls, name, *a, **k)
    assert os.path.isdir(REF), "reference tree not present: goldens can only be made in the build container"
    shim = tempfile.mkdtemp(prefix="sb_oracle_shims_")
    _write_shims(shim)
    sys.dont_write_bytecode = True
    sys.path[:0] = [shim, REF]
    import torch
    torch.set_num_threads(8)
    NetBig = importlib.import_module("src.models.tfgridnet_realtime_clean_dis_embd3.net").Net
    NetSmall = importlib.import_module("src.models.tfgridnet_realtime_clean_optim.net").Net

    common = dict(stft_chunk_size=192, stft_pad_size=96, num_ch=6, L=4, I=1, J=1, H=64, E=2,
                  use_attn=False, lookahead=True, chunk_causal=True, use_first_ln=True,
                  merge_method="early_cat")
    # parameter-count sanity: 501398 / 231125 / 498050
    big = dict(common, D=32, B=6, local_atten_len=100, conv_lstm=False, dis_type="conv3")
    small = dict(common, D=16, B=3, local_atten_len=50, conv_lstm=True, lstm_down=5)
    orange = dict(common, D=32, B=6, local_atten_len=50, conv_lstm=False, lstm_down=5)
    for nm, N_, p, want in (("big", NetBig, big, 501398), ("small", NetSmall, small, 231125),
                            ("orange", NetSmall, orange, 498050)):
        n = sum(q.numel() for q in N_(**p).parameters())
        assert n == want, (nm, n, want)
    print("parameter counts match SURVEY: 501398 / 231125 / 498050")

    m0 = NetBig(**big)
    fe = m0.tfgridnet.enc.filterbank._filters.numpy()
    fd = m0.tfgridnet.dec.filterbank._filters.numpy()
    assert np.array_equal(fe, fd)
    np.savez_compressed(os.path.join(args.out, "stft_filters.npz"), filters=fe)

    # tiny cases (2 blocks, 7 frames) -- full per-stage + grads + streaming
    case(torch, NetBig, "tiny_big", dict(big, B=2), B=2, n_frames=7, seed=11, needs_dis=True, out_dir=args.out)
    case(torch, NetSmall, "tiny_small", dict(small, B=2), B=2, n_frames=7, seed=12, needs_dis=False, out_dir=args.out)
    case(torch, NetSmall, "tiny_orange", dict(orange, B=2), B=2, n_frames=7, seed=13, needs_dis=False, out_dir=args.out)
    # dis_embd3 flavour of the conv-LSTM intra path (pads by 3 and crops)
    case(torch, NetBig, "tiny_big_convlstm", dict(big, B=2, D=16, conv_lstm=True), B=2, n_frames=7, seed=14,
              needs_dis=True, out_dir=args.out, with_stream=False)
    # full-band local self-attention ON (off in every shipped JSON; tfgridnet_causal.py:639-684,856-898):
    # window 100 (zero-filled, unmasked history dominates) and window 4 (windows inside the data)
    case(torch, NetBig, "tiny_big_attn100", dict(big, B=2, use_attn=True), B=2, n_frames=7, seed=15,
              needs_dis=True, out_dir=args.out)
    case(torch, NetSmall, "tiny_orange_attn4", dict(orange, B=2, use_attn=True, local_atten_len=4), B=2,
              n_frames=7, seed=16, needs_dis=False, out_dir=args.out)
    # other microphone counts (every shipped JSON has num_ch=6; the reference's constructor default is 2): feature stack of
    # 5 M - 3 channels -- 7 for M = 2, 17 for M = 4
    case(torch, NetBig, "tiny_big_2ch", dict(big, B=2, num_ch=2), B=2, n_frames=7, seed=17, needs_dis=True,
              out_dir=args.out)
    case(torch, NetSmall, "tiny_small_4ch", dict(small, B=2, num_ch=4), B=2, n_frames=7, seed=18, needs_dis=False,
              out_dir=args.out, with_stream=False)
    # merge_method "None" -- the reference constructor's default (tfgridnet_causal.py:341-342,486-493): the 3x3 convolution sees
    # the 2 M (re, im) channels alone, no ILD / IPD features
    case(torch, NetBig, "tiny_big_nomerge", dict(big, B=2, merge_method="None"), B=2, n_frames=7, seed=19, needs_dis=True,
              out_dir=args.out)
    case(torch, NetSmall, "tiny_small_nomerge", dict(small, B=2, merge_method="None"), B=2, n_frames=7, seed=20, needs_dis=False,
              out_dir=args.out, with_stream=False)
    # the reference constructor's own widths (net.py:21-26: D = 64, H = 128), which no shipped JSON uses and the HIP kernels do
    # not cover yet: two blocks, 6 frames, so that the oracle is pinned at those shapes before kernels exist
    case(torch, NetBig, "tiny_big_h128d64", dict(big, B=2, D=64, H=128), B=2, n_frames=6, seed=22, needs_dis=True,
              out_dir=args.out, with_stream=False)
    case(torch, NetSmall, "tiny_small_h128d64", dict(small, B=2, D=64, H=128), B=2, n_frames=6, seed=23, needs_dis=False,
              out_dir=args.out, with_stream=False)
    # the reference constructors called with NO arguments (net.py:21-26): n_fft 280, 2 microphones, D = 64, H = 128, six conv-LSTM blocks
    if not only or "default_ctor_big" in only:
        make_default_ctor(torch, NetBig, "default_ctor_big", seed=24, needs_dis=True, out_dir=args.out)
    if not only or "default_ctor_small" in only:
        make_default_ctor(torch, NetSmall, "default_ctor_small", seed=25, needs_dis=False, out_dir=args.out)
    if not only or "ctor_behaviour" in only:
        make_ctor_behaviour({"dis_embd3": NetBig, "optim": NetSmall}, args.out)
    # real small config, 1 s clip (125 frames), forward only
    case(torch, NetSmall, "small_1s", small, B=1, n_frames=125, seed=21, needs_dis=False, out_dir=args.out,
              with_grads=False, with_stream=False, with_stages=False)
    if not only or "samples" in only:
        make_samples(torch, NetBig, args.out)
    if not only or "samples_more" in only:
        make_samples_more(torch, NetBig, args.out)
    if not only or "samples_6block" in only:
        make_samples_6block(torch, NetBig, big, args.out)
    if not only or "state_io" in only:
        make_state_io(torch, {"small": (NetSmall, small), "big": (NetBig, big), "orange": (NetSmall, orange),
                              "big_attn": (NetBig, dict(big, use_attn=True)),
                              "tiny_small": (NetSmall, dict(small, B=2))}, args.out)
    if not only or "ckpt" in only:
        make_ckpt(torch, NetSmall, args.out)
    if not only or "samples_full" in only:
        copy_all_samples_full(args.out)


if __name__ == "__main__":
    main()
