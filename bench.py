#!/usr/bin/env python3
"""Sound-Bubble hot-path benchmark on MI355X.

Metric (BASELINE.json): utterances/s of one optimiser step (forward + SNRLP loss + backward +
[RCCL all-reduce] + clip + Adam) on synthetic 6-ch x 24 kHz x 5 s utterances, whole job over N GPUs.
The HEADLINE (last JSON line) is the configuration the metric's 1/2/4/8-GPU series is quoted on:
BASELINE configs[2]/[3], the 0.5 M-param "big" model (syn_experiments/pretrain_stage.json model_params),
batch 16 per GPU (weak scaling; 8 GPUs = global batch 128).

Launch: `python bench.py --gpus 1` or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
 bench.py --gpus N --steps K --warmup W`.

At N = 1 the default run first measures the secondary numbers -- streaming chunk loop (configs[4], both model sizes),
forward-only utt/s (both), the small config's train step (configs[1], B = 32) -- printing one JSON line each, then the
headline, whose `secondary` object carries all of them again (so the LAST line alone holds every number of the run).
`--workload small|big|big-attn` prints that single train line only; `--headline-only` skips the secondary measurements.

The timed region is event-free; the per-kernel table behind `roofline` (the recurrent kernel with the largest total time,
HIP events on the launch stream) comes from a separate, untimed pass of the same step.  Every train line names its
BPTT-state precision (`bptt_mode`; default "wide": fp32-class records (fp32 c_prev, 24-bit fixed-point gates), two-term gradient products = the reference's own
arithmetic) and carries the sibling mode's number (`compact_bptt`: fp16 BPTT state, opt-in) with its own roofline, and, at
N = 1, `cpu_baseline` (the oracle -- a CPU port of the reference's algorithm -- on the host cores: train B = 1 / 4, eval
forward B = 1 / 4, bounded).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

COMMON = dict(stft_chunk_size=192, stft_pad_size=96, num_ch=6, L=4, I=1, J=1, H=64, E=2, use_attn=False,
              lookahead=True, chunk_causal=True, use_first_ln=True, merge_method="early_cat")
WORKLOADS = {
    # name: (class, model_params, batch/GPU, neg_weight, grad_clip, lr)
    "small": ("NetOptim", dict(COMMON, D=16, B=3, conv_lstm=True, lstm_down=5, local_atten_len=50), 32, 50.0, 1.0,
              2e-3),
    "big": ("NetDisEmbd3", dict(COMMON, D=32, B=6, conv_lstm=False, local_atten_len=100, dis_type="conv3"), 16,
            100.0, None, 1.2e-3),   # "grad_clip" sits at the JSON top level there -> PLModule does not clip (F10a)
    # extra (not a BASELINE config): the big model with the full-band attention of row a10 switched on
    "big-attn": ("NetDisEmbd3", dict(COMMON, D=32, B=6, conv_lstm=False, local_atten_len=100, dis_type="conv3",
                                     use_attn=True), 16, 100.0, None, 1.2e-3),
}
N_SAMPLES = 120000
MFMA_F32_PEAK = 157.3e12        # dense fp32-input MFMA (= fp32 vector) peak, MI355X_MICROARCH.md
MFMA_BF16_PEAK = 2500e12        # dense bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def fwd_flops_per_utt(p, T=625, F=145, M=6):
    """SURVEY.md 8(d) algorithmic FLOPs of one forward (1 MAC = 2)."""
    C, H, nb = p["D"], p["H"], p["B"]
    step = 2 * 4 * H * (C + H)
    stft = 2 * 290 * 288 * T * M
    conv = 2 * 27 * C * 9 * T * F
    if p["conv_lstm"]:
        K = F // p["lstm_down"]
        intra = 2 * C * C * 5 * K * T + 2 * step * K * T + 2 * 2 * H * C * 5 * K * T
    else:
        intra = 2 * step * F * T + 2 * 2 * H * C * F * T
    inter = step * T * F + 2 * H * C * T * F
    return stft + conv + nb * (intra + inter) + 2 * C * 2 * 9 * T * F + 2 * 290 * 288 * (T + 1)


def fwd_bytes_per_utt(p, T=625, F=145, M=6):
    """SURVEY.md 8(d) compulsory HBM bytes of one forward."""
    return 4 * (M * (N_SAMPLES + 96) + p["D"] * T * F * (2 + 4 * p["B"]) + N_SAMPLES)


def synth_batch(torch, B, seed, device, with_dis, cycle_radii=False):
    g = torch.Generator().manual_seed(seed)
    base = 0.1 * torch.randn(B, 1, N_SAMPLES + 8, generator=g)
    mix = torch.cat([base[..., 4 - min(m, 4): 4 - min(m, 4) + N_SAMPLES] for m in range(6)], 1)
    mix = (mix + 0.02 * torch.randn(B, 6, N_SAMPLES, generator=g)).clamp(-1, 1)
    tgt = 0.05 * torch.randn(B, 1, N_SAMPLES, generator=g)
    tgt[7::8] = 0.0                              # every 8th sample: silent target (negative branch)
    inputs = {"mixture": mix.to(device)}
    if with_dis:
        dis = torch.zeros(B, 3)                  # configs[2] all [0, 1, 0] (1.5 m); configs[3] cycles the radii
        dis[torch.arange(B), torch.tensor(radius_columns(B, cycle_radii))] = 1.0
        inputs["dis_embed"] = dis.to(device)
    return inputs, tgt.to(device)


def radius_columns(B, cycle_radii):
    """one-hot column per utterance: 1 = 1.5 m for all of configs[2]; configs[3] (N > 1) cycles 2 m / 1.5 m / 1 m"""
    return [(i % 3) if cycle_radii else 1 for i in range(B)]


def rank_plan(wl, rank, world, batch=0):
    """what a rank of the data-parallel bench feeds its replica: its seed, its slice of the global batch, its radii"""
    cls, _, B, _, _, _ = WORKLOADS[wl]
    B = batch or B
    return {"rank": rank, "seed": 1234 + rank, "batch_per_gpu": B, "global_batch": B * world,
            "radius_columns": radius_columns(B, world > 1) if cls != "NetOptim" else None}


def host_cores():
    """Usable host cores: min(os.cpu_count, affinity mask, cgroup cpu.max quota).  The GPU box reports 256
    CPUs but the pod is capped (cpu.max) -- oversubscribing OpenMP there is catastrophically slow."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(torch, wl, budget_s=36.0):
    """Oracle (CPU port of the reference algorithm, oracle/tfgridnet_oracle.py -- the reference's own .py cannot travel to the
    GPU box) on the host cores, the four legs of SURVEY.md 8(d): train step at B = 1 (the metric; `value`) and B = 4, eval
    forward at B = 1 and B = 4, each one warm-up + a bounded number of timed passes within the time budget."""
    from oracle.tfgridnet_oracle import OracleNet, snrlp_loss
    cls, params, _, negw, clip, lr = WORKLOADS[wl]
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = OracleNet("optim" if cls == "NetOptim" else "dis_embd3", **params).train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    t_start = time.time()

    def timed(fn, max_n, share):
        fn()                                     # warm-up
        t0, n = time.time(), 0
        while n < max_n:
            fn()
            n += 1
            if time.time() - t_start > budget_s * share:
                break
        return (time.time() - t0) / n, n

    legs = {}
    for name, B, train, max_n, share in (("train_b1", 1, True, 3, 0.30), ("eval_fwd_b1", 1, False, 3, 0.42),
                                         ("eval_fwd_b4", 4, False, 2, 0.62), ("train_b4", 4, True, 1, 1.0)):
        inputs, tgt = synth_batch(torch, B, 1234, "cpu", cls != "NetOptim")
        if time.time() - t_start > budget_s * 0.9 and legs:
            legs[name] = {"dropped": "time budget"}
            continue
        if train:
            m.train()

            def fn():
                opt.zero_grad()
                est = m(dict(inputs))["output"]
                snrlp_loss(est, tgt, negw).mean().backward()
                if clip:
                    torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
                opt.step()
        else:
            m.eval()

            def fn():
                with torch.no_grad():
                    m(dict(inputs))["output"]
        dt, n = timed(fn, max_n, share)
        legs[name] = {"utt_s": B / dt, "s_per_pass": dt, "passes": n, "batch": B}
    main = legs["train_b1"]
    return {"value": main["utt_s"], "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
            "os_cpu_count": os.cpu_count(), "cpu_model": cpu_model_string(),
            "sample": f"{main['passes']} train steps of batch 1 ({wl} config, 5 s clips), oracle CPU port, after 1 warm-up; "
                      f"other legs (eval forward B = 1 / 4, train B = 4) under `legs`, {time.time() - t_start:.0f} s in all",
            "legs": legs}


def cpu_stream_leg(torch, wl="small", n_chunks=625, budget_s=14.0):
    """BASELINE.md section 4, last bullet: the CPU side of configs[4] -- the oracle (CPU port of the reference's algorithm)
    fed 8 ms chunks [1, 6, 288] with carried state, as edge/causal_infer.py:15-26 does; chunks/s and p50 per chunk on the
    host cores, bounded (all 625 chunks of a 5 s clip when they fit the budget)."""
    import numpy as np
    from oracle.tfgridnet_oracle import OracleNet
    cls, params = WORKLOADS[wl][:2]
    torch.manual_seed(0)
    m = OracleNet("optim" if cls == "NetOptim" else "dis_embd3", **params).eval()
    g = torch.Generator().manual_seed(1234)
    frames = 0.1 * torch.randn(n_chunks, 1, 6, 288, generator=g)
    dis = None if cls == "NetOptim" else torch.tensor([[0.0, 1.0, 0.0]])
    state = m.init_buffers(1, "cpu")
    lat = []
    t_start = time.perf_counter()
    with torch.no_grad():
        for i in range(n_chunks):
            inp = {"mixture": frames[i]}
            if dis is not None:
                inp["dis_embed"] = dis
            t1 = time.perf_counter()
            state = m(inp, state, pad=False)["next_state"]
            lat.append(time.perf_counter() - t1)
            if time.perf_counter() - t_start > budget_s and i >= 20:
                break
    lat = np.array(lat[5:]) * 1e3                              # the first chunks carry allocator / thread-pool warm-up
    return {"chunks_s": 1e3 / float(lat.mean()), "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
            "chunks": int(len(lat)), "realtime_factor": 8.0 / float(np.percentile(lat, 50)),
            "sample": f"{len(lat)} chunks of [1, 6, 288] ({wl} config) through the oracle CPU port with carried state, "
                      f"{torch.get_num_threads()} threads, after 5 warm-up chunks"}


def parity_check(torch, sb, dev):
    """`parity` object of the headline (BASELINE.json metric: "...; SI-SDRi vs ref"): ONE committed scene -- the reference's
    test_samples/syn_1m/00001 at its full 5 s -- through the 6-block 0.5 M network of pretrain_stage.json with the weights of
    tests/golden/samples_6block.npz, against the output the REFERENCE model produced for it (same fixture, generated by
    tests/golden/make_goldens.py from the imported reference).  Runs outside the timed region; no oracle involved."""
    import ast
    import numpy as np
    from sound_bubble_amd.eval_samples import load_testcase, run_testcase, si_sdr_np
    gold = os.path.join(ROOT, "tests", "golden")
    path = os.path.join(gold, "samples_6block.npz")
    scene = os.path.join(gold, "test_samples_full", "syn_1m", "00001")
    if not (os.path.exists(path) and os.path.isdir(scene)):
        return {"skipped": "fixture tests/golden/samples_6block.npz not present"}
    z = np.load(path)
    params = dict(ast.literal_eval(str(z["meta::params"])))
    sd = {k[len("param::"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param::")}
    filt = torch.from_numpy(np.load(os.path.join(gold, "stft_filters.npz"))["filters"])
    sd["tfgridnet.enc.filterbank._filters"] = filt.clone()
    sd["tfgridnet.dec.filterbank._filters"] = filt.clone()
    m = sb.NetDisEmbd3(**params)
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    _, mix, gt, tg = load_testcase(scene, 1.0)
    out = run_testcase(m, mix, 1.0, device=dev)
    ref = z["output"]
    rel = float(np.sqrt(((out.astype(np.float64) - ref) ** 2).sum() / (ref.astype(np.float64) ** 2).sum()))
    ours, theirs = si_sdr_np(out[0], gt[0]), float(z["si_sdr"])
    inp = si_sdr_np(mix[0], gt[0])
    return {"scene": "test_samples/syn_1m/00001 (5 s, 1 in-bubble speaker), 6-block model of syn_experiments/pretrain_stage.json, "
                     "seeded weights (no trained checkpoint offline: SI-SDR values are plumbing-grade)",
            "fwd_rel_l2": rel, "fwd_rel_l2_bar": 1e-3,
            "si_sdr_db": ours, "si_sdr_ref_db": theirs, "si_sdr_delta_db": ours - theirs, "si_sdr_delta_bar_db": 0.05,
            "si_sdr_i_db": ours - inp, "si_sdr_i_ref_db": theirs - float(z["input_si_sdr"]),
            "reference_output": "tests/golden/samples_6block.npz (the imported reference model's own output)"}


def parity_trained(torch, dev):
    """`parity.trained` of the headline: the checkpoint this repo's HIP path TRAINED (tests/golden/trained_overfit_best.pt: train_cli on
    experiments/overfit_test_samples.json) against what the IMPORTED REFERENCE network computes with the same file
    (tests/golden/trained_overfit.npz, made by tests/golden/make_trained_fixture.py in the build container) -- SI-SDR at a
    positive operating point instead of the -50 dB of seeded weights.  Outside the timed region; no oracle involved."""
    import numpy as np
    from sound_bubble_amd.eval_samples import load_testcase, run_testcase, si_sdr_np
    from sound_bubble_amd.harness import import_attr
    gold = os.path.join(ROOT, "tests", "golden")
    ck, fx = os.path.join(gold, "trained_overfit_best.pt"), os.path.join(gold, "trained_overfit.npz")
    if not (os.path.exists(ck) and os.path.exists(fx)):
        return {"skipped": "tests/golden/trained_overfit_best.pt / trained_overfit.npz not present"}
    p = json.load(open(os.path.join(ROOT, "experiments", "overfit_test_samples.json")))
    hl = import_attr(p["pl_module"])(**dict(p["pl_module_args"], init_ckpt=None, use_dp=False))
    hl.load_state(ck)
    hl.eval()
    ref = np.load(fx)
    rows, worst_l2, worst_db = [], 0.0, 0.0
    for sset, radius in (("syn_1m", 1.0), ("syn_1_5m", 1.5), ("syn_2m", 2.0)):
        for scene in ("00001", "00002"):
            key = f"{sset}/{scene}"
            _, mix, gt, tg = load_testcase(os.path.join(gold, "test_samples_full", sset, scene), radius)
            out = run_testcase(hl.model, mix, radius, device=dev)
            s_, want = si_sdr_np(out[0], gt[0]), float(ref[key + "::si_sdr"])
            row = {"scene": key, "si_sdr_db": s_, "si_sdr_ref_db": want, "si_sdr_i_ref_db": want - float(ref[key + "::input_si_sdr"])}
            if key + "::output" in ref:
                r = ref[key + "::output"].astype(np.float64)
                row["fwd_rel_l2"] = float(np.sqrt(((out.astype(np.float64) - r) ** 2).sum() / (r ** 2).sum()))
                worst_l2 = max(worst_l2, row["fwd_rel_l2"])
            worst_db = max(worst_db, abs(s_ - want))
            rows.append(row)
    return {"checkpoint": "tests/golden/trained_overfit_best.pt (trained here on the nine bundled demo scenes: an overfit, not a "
                          "generalisation claim), epoch %d" % int(ref["meta::ckpt_epoch"]),
            "reference_output": "tests/golden/trained_overfit.npz (the imported reference network with the same checkpoint)",
            "worst_fwd_rel_l2": worst_l2, "fwd_rel_l2_bar": 1e-3, "worst_si_sdr_delta_db": worst_db, "si_sdr_delta_bar_db": 0.05,
            "mean_si_sdr_db": float(np.mean([r["si_sdr_db"] for r in rows])),
            "mean_si_sdr_i_ref_db": float(np.mean([r["si_sdr_i_ref_db"] for r in rows])), "scenes": rows}


def two_product_forward(torch, dist, sb, ops, args, dev, world, rank):
    """secondary.forward_{small,big}_2prod: the OPT-IN reduced-product forward (sb_lstm_fwd_args.products = 2, SB_LSTM_PRODUCTS=2:
    activations as ONE fp16 term against hi + lo weights in the recurrent products) -- its speed, and its distance from the
    default arithmetic and from the reference on the committed scene.  Never the headline: 11-bit activations are not fp32-class."""
    import numpy as np
    res = {}
    old = ops.LSTM_PRODUCTS
    try:
        base = {}
        for w2 in ("small", "big"):                          # distance on the bench batch itself: default vs two products
            cls, params, B = WORKLOADS[w2][0], WORKLOADS[w2][1], min(4, args.batch or WORKLOADS[w2][2])
            torch.manual_seed(0)
            model = getattr(sb, cls)(**params).to(dev).eval()
            inputs, _ = synth_batch(torch, B, 1234, dev, cls != "NetOptim")
            outs = []
            for prod in (3, 2):
                ops.LSTM_PRODUCTS = prod
                with torch.no_grad():
                    outs.append(model(inputs)["output"].double())
            base[w2] = float((outs[1] - outs[0]).norm() / outs[0].norm())
        ops.LSTM_PRODUCTS = 2
        for w2 in ("small", "big"):
            o = train_line(torch, dist, sb, ops, w2, args, dev, world, rank, forward_only=True, with_cpu=False)
            res[f"forward_{w2}_2prod"] = {
                "utt_s": o["value"], "ms_per_step": o["ms_per_step"], "hbm_frac": o["forward_roofline"]["hbm_frac"],
                "rel_l2_vs_default_arithmetic": base[w2], "batch": o["config"]["batch_per_gpu"],
                "dtype": "f32 storage/accumulate; recurrent products on the fp16 pipe with hi+lo WEIGHTS x single-term fp16 "
                         "ACTIVATIONS (2 products per MAC, 11-bit activations: NOT fp32-class; opt-in SB_LSTM_PRODUCTS=2)"}
        pc = parity_check(torch, sb, dev)
        if "fwd_rel_l2" in pc:
            res["forward_big_2prod"]["scene_vs_reference"] = {k: pc[k] for k in ("fwd_rel_l2", "si_sdr_delta_db", "fwd_rel_l2_bar")}
    finally:
        ops.LSTM_PRODUCTS = old
    return res


def vendor_gpu_baseline(torch, wl, B, dev, steps=3):
    """Second yardstick of SURVEY.md 8(d), measurement only: the same oracle restatement (stock torch.nn ops ->
    MIOpen RNN / rocBLAS / ATen kernels) moved onto the GPU -- i.e. what running the reference unmodified on
    PyTorch-ROCm gives.  Falls back to smaller batches if the library path runs out of memory."""
    from oracle.tfgridnet_oracle import OracleNet, snrlp_loss
    cls, params, _, negw, clip, lr = WORKLOADS[wl]
    torch.manual_seed(0)
    m = OracleNet("optim" if cls == "NetOptim" else "dis_embd3", **params).to(dev).train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    while B >= 1:
        try:
            inputs, tgt = synth_batch(torch, B, 1234, dev, cls != "NetOptim")

            def step():
                opt.zero_grad()
                est = m(dict(inputs))["output"]
                snrlp_loss(est, tgt, negw).mean().backward()
                if clip:
                    torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
                opt.step()

            step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            return {"value": B / dt, "unit": "utterances/s", "batch": B, "ms_per_step": dt * 1e3,
                    "kind": "oracle restatement on the GPU through stock torch.nn ops (MIOpen RNN, rocBLAS, ATen)",
                    "sample": f"{steps} train steps of batch {B} after 1 warm-up"}
        except torch.OutOfMemoryError:
            B //= 2
            torch.cuda.empty_cache()
    return None


def stream_bench(torch, sb, args, wl, cls, params, dev):
    """Config 5: B=1, 625 chunks of [1, 6, 288] (8 ms hop) through the hipGraph-captured chunk step."""
    import numpy as np
    from sound_bubble_amd.streaming import StreamingSeparator
    torch.manual_seed(0)
    model = getattr(sb, cls)(**params).to(dev).eval()
    dis = torch.tensor([[0.0, 1.0, 0.0]], device=dev) if cls != "NetOptim" else None
    sep = StreamingSeparator(model, 1, dis_embed=dis, use_graph=not args.no_graph)
    g = torch.Generator().manual_seed(1234)
    frames = (0.1 * torch.randn(625, 1, 6, 288, generator=g)).to(dev)
    for i in range(20):                                       # warm-up (includes graph capture)
        sep.feed(frames[i])
    torch.cuda.synchronize()
    lat = []
    t0 = time.perf_counter()
    for i in range(625):
        t1 = time.perf_counter()
        sep.feed(frames[i])
        torch.cuda.synchronize()                              # per-chunk latency = copy + replay + completion
        lat.append(time.perf_counter() - t1)
    dt = time.perf_counter() - t0
    lat = np.array(lat) * 1e3
    fpu = fwd_flops_per_utt(params) / 625.0
    obj = {
        "metric": "streaming chunks/sec (8 ms hop, 6 mics, B=1)", "value": 625 / dt, "unit": "chunks/s", "n_gpus": 1,
        "steps": 625, "warmup": 20, "ms_per_step": dt / 625 * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_FWD, "data": "synthetic",
        "config": {"workload": f"stream-{wl} (BASELINE configs[4]): {cls} D={params['D']} B={params['B']}, chunk [1,6,288] -> [1,1,192], "
                               f"{'eager launches' if args.no_graph else 'hipGraph replay'}"},
        "latency_ms": {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)),
                       "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
        "realtime_factor": 8.0 / float(np.percentile(lat, 50)),
        "reference_claim": "6.36 ms per 8 ms chunk on an embedded CPU (README.md:9)",
        "chunk_mflop": fpu / 1e6}
    print(json.dumps(obj), flush=True)
    return obj


DTYPE_BY_MODE = {
    "wide": "f32 storage/accumulate; BPTT state: fp32 c_prev, post-activation gates as 24-bit fixed point (|error| <= 2^-25 / 2^-24 on [0, 1] / [-1, 1]: fp32-class), fp32-sized side outputs; every matrix product "
            "on the fp16 pipe with TWO-term split operands (3 products per MAC, 22 mantissa bits: fp32-class) in forward "
            "AND backward",
    "compact": "f32 storage/accumulate; forward products fp16 hi+lo split (fp32-class); BPTT state (gate / c_prev records, "
               "dgates, LayerNorm-output and hs side outputs) fp16, single-term gradient products (SB_BPTT=compact, opt-in)",
    "legacy": "f32 storage/accumulate; forward products fp16 hi+lo split; BPTT state fp32, unfused round-1 kernels "
              "(SB_BPTT=legacy)",
}
SIBLING = {"wide": "compact", "compact": "wide", "legacy": "wide"}
DTYPE_FWD = "f32 storage/accumulate; matrix products on the fp16 pipe with hi+lo split operands (3 products per MAC, fp32-class)"

# rocprof kernel-name patterns of the labels ops.PROFILE uses (for the committed PMC traffic JSONs, whose keys are
# "<kernel name> grid=<threads>").  The two forward recurrences share one instantiation since the intra-frame pass applies
# its Linear in the kernel too: they are told apart by the grid (bidirectional = twice the workgroups of more tiles).
# the overlapped inter-frame backward: recurrence (dgates to HBM, slab flags) + stream kernel, counted in plain order
# (SB_BWD_PAIR_SERIAL=1 passes, profiles/r*_pmc_traffic_<wl>_wide_pair.json): the pair's traffic is the SUM of the two
# The backward recurrence's variant switches: thirteen positional bools up to round 3 / the first r04 profiles
# (lstm_bwd_rec_bf_kernel<FULL, REC16, FUSE_C, DG16, SEG, FST, LNB, BI, HS16B, RECOMP, SLAB, XP, ...>), one bit set since
# (lstm_bwd_rec_bf_kernel<KF, FUSE_C, FST>, bits as in csrc/sb_lstm_bf_bwd.hip) -- both spellings occur in committed summaries.
_KBITS = {"FULL": 0, "REC16": 1, "DG16": 2, "SEG": 3, "LNB": 4, "BI": 5, "HS16B": 6, "RECOMP": 7, "SLAB": 8, "XP": 9, "SPLIT": 10,
          "GREC": 11, "PROD": 12, "CONS": 13}


def _bwd_variant(name):
    import re
    m = re.search(r"lstm_bwd_rec_bf_kernel<([^<>]*)>", name)
    if not m:
        return None
    args = [x.strip() for x in m.group(1).split(",")]
    if len(args) == 3 and re.fullmatch(r"\d+[uU]?", args[0]):
        kf = int(args[0].rstrip("uU"))
        v = {k: bool(kf >> b & 1) for k, b in _KBITS.items()}
        v["FUSE_C"], v["FST"] = int(args[1]), int(args[2])
        return v
    order = ["FULL", "REC16", "FUSE_C", "DG16", "SEG", "FST", "LNB", "BI", "HS16B", "RECOMP", "SLAB", "XP", "SPLIT", "GREC", "PROD", "CONS"]
    v = {k: False for k in _KBITS}
    v["FUSE_C"] = v["FST"] = 0
    for k, x in zip(order, args):
        v[k] = int(x) if k in ("FUSE_C", "FST") else x == "true"
    return v


def _is(pred):
    return lambda name: (lambda v: v is not None and pred(v))(_bwd_variant(name))


def _re(pat):
    import re
    return lambda name: re.search(pat, name) is not None


PMC_PAIR = (_is(lambda v: v["DG16"] and not v["SEG"] and v["FST"] == 0 and v["SLAB"]), _re(r"lstm_bwd_stream_f16_kernel"))
# (label fragment, kernel-name predicate, selector among the matches).  The cross-pass variants (PROD / CONS bits) are matched
# before their plain-order siblings: a counter pass taken with SB_OVERLAP_FORCE=1 holds the shipped producer / consumer kernels,
# an older one only the siblings (the label then falls through to the sibling's entry, and `traffic_kernel` says so).
# Forward recurrences share one instantiation: bidirectional = the largest grid; the inter-frame pass = the OTHER grid with
# the most traced time (round 4 took the smallest grid, which is the B = 1 parity scene: VERDICT r4 weak #5).
PMC_PATTERNS = [
    ("cross-pass consumer", _is(lambda v: v["DG16"] and v["BI"] and v["FST"] > 0 and v["CONS"]), None),
    ("cross-pass producer", _is(lambda v: v["DG16"] and not v["BI"] and v["FST"] in (16, 32) and v["PROD"]), None),
    ("intra-frame fused BPTT", _is(lambda v: v["DG16"] and v["BI"] and v["FST"] > 0 and not v["CONS"]), None),
    ("inter-frame fused BPTT", _is(lambda v: v["DG16"] and not v["BI"] and v["FST"] in (16, 32) and not v["PROD"]), None),
    ("recurrence only", _is(lambda v: v["REC16"] and v["DG16"] and v["FST"] == 0 and not v["BI"]), None),
    ("lstm_bwd_stream", _re(r"lstm_bwd_stream_f16_kernel"), None),
    ("intra-frame (bidirectional)", _re(r"lstm_fwd_bf_kernel<"), "max-grid"),
    ("inter-frame (Linear fused)", _re(r"lstm_fwd_bf_kernel<"), "other-grid"),
    ("ln_film_bwd", _re(r"ln_film_bwd_kernel"), None),
]
F_CLK_PROFILED = 2.0e9          # shader clock under the counter passes (MI355X_MICROARCH.md: 1.89-1.95 GHz profiled, 2.02 not)
N_SIMD = 1024                   # 256 CUs x 4 SIMDs


def _pmc_pick(ks, label):
    """entries of a committed counter summary (`kernels`: "<name> grid=<threads>" -> dict) that belong to `label`
    -> (list of (key, entry), matched fragment) or ([], None)"""
    import re
    grid = lambda k: int(re.search(r"grid=(\d+)", k).group(1))
    time_of = lambda v: v.get("launches", 1) * v.get("avg_us_in_pmc_pass", 0.0)
    for frag, pat, sel in PMC_PATTERNS:
        if frag not in label:
            continue
        m = [(k, v) for k, v in ks.items() if pat(k)]
        if not m:
            continue
        if sel:
            top = max(grid(k) for k, _ in m)
            if sel == "max-grid":
                m = [(k, v) for k, v in m if grid(k) == top]
            else:
                rest = {}
                for k, v in m:
                    if grid(k) != top:
                        rest[grid(k)] = rest.get(grid(k), 0.0) + time_of(v)
                if not rest:
                    continue
                pick = max(rest, key=rest.get)
                m = [(k, v) for k, v in m if grid(k) == pick]
        return m, frag
    return [], None


def _newest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


_DIGEST = []


def profile_provenance(pf):
    """-> {file, tree, csrc_sha16, stale_profile}: the counter summary's stamp (scripts/pmc_summary.py) against the digest of the
    kernel sources this process runs (sound_bubble_amd.build.csrc_digest); an unstamped (pre-round-6) summary is stale by definition"""
    if not pf:
        return None
    if not _DIGEST:
        from sound_bubble_amd.build import csrc_digest
        _DIGEST.append(csrc_digest())
    prov = json.load(open(pf)).get("provenance") or {}
    return {"file": os.path.relpath(pf, ROOT), "tree": prov.get("tree"), "csrc_sha16": prov.get("csrc_sha16"),
            "running_csrc_sha16": _DIGEST[0], "stale_profile": prov.get("csrc_sha16") != _DIGEST[0]}


def pmc_traffic(workload, label, mode="wide"):
    """HBM bytes per launch of the kernel behind `label` from the newest committed rocprofv3 --pmc summary
    (profiles/r*_pmc_traffic_<workload>.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 corrections applied).
    -> (bytes per launch, source string, kernel symbol(s))"""
    if "inter overlapped" in label:
        pf = _newest(f"r*_pmc_traffic_{workload}_{mode}_pair.json")
        if not pf:
            return None, None, None
        ks = json.load(open(pf))["kernels"]
        parts, names = [], []
        for pat in PMC_PAIR:
            m = [(k, v) for k, v in ks.items() if pat(k)]
            if not m:
                return None, None, None
            parts.append(sum(v["hbm_bytes"] * v["launches"] for _, v in m) / sum(v["launches"] for _, v in m))
            names += [k for k, _ in m]
        return sum(parts), (os.path.relpath(pf, ROOT) + " (rocprofv3 --pmc passes with SB_BWD_PAIR_SERIAL=1: the pair's two kernels in "
                            "plain order, recurrence + stream kernel summed)"), names
    pf = _newest(f"r*_pmc_traffic_{workload}{'' if mode == 'compact' else '_' + mode}.json")
    if not pf:
        return None, None, None
    m, frag = _pmc_pick(json.load(open(pf))["kernels"], label)
    if not m:
        return None, None, None
    return (sum(v["hbm_bytes"] * v["launches"] for _, v in m) / sum(v["launches"] for _, v in m),
            os.path.relpath(pf, ROOT) + " (rocprofv3 --pmc passes of `bench.py --workload " + workload + "`, committed)",
            [k for k, _ in m])


def pmc_mfma_busy(workload, label, mode="wide"):
    """matrix-pipe busy fraction of the kernel behind `label` from the newest committed SQ counter summary:
    SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's SIMDs) / (launch duration in the same pass x shader clock x 1024
    SIMDs) -- VERDICT r4 weak #6: round 4 divided by SQ_BUSY_CYCLES, a per-SE count, and reported values of 3.5 .. 10."""
    pf = _newest(f"r*_pmc_sq_{workload}{'' if mode == 'compact' else '_' + mode}.json")
    if not pf:
        return None
    m, _ = _pmc_pick(json.load(open(pf))["kernels"], label)
    m = [v for _, v in m if v.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None and v.get("avg_us_in_pmc_pass")]
    if not m:
        return None
    n = sum(v["launches"] for v in m)
    if all(v.get("mfma_busy") is not None for v in m):          # the summary's own figure (clock from GRBM_GUI_ACTIVE when sane)
        return sum(v["mfma_busy"] * v["launches"] for v in m) / n
    cyc = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] * v["launches"] for v in m) / n
    us = sum(v["avg_us_in_pmc_pass"] * v["launches"] for v in m) / n
    return cyc / (us * 1e-6 * F_CLK_PROFILED * N_SIMD)


def run_workload(torch, dist, sb, ops, wl, args, dev, world, rank, *, forward_only=False, mode=None, steps=None,
                 warmup=None, profile=True):
    """W warm-up steps, then exactly K timed steps between barrier + synchronize; max over ranks.
    -> (seconds per step, per-kernel profile dict, batch per GPU, params, class name)"""
    from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step
    cls, params, B, negw, clip, lr = WORKLOADS[wl]
    B = args.batch or B
    steps = steps or args.steps
    warmup = args.warmup if warmup is None else warmup
    mode = mode or args.bptt or ops.BPTT                     # BPTT-state precision: wide (default) | compact | legacy
    old_mode, ops.BPTT = ops.BPTT, mode
    torch.manual_seed(0)                                     # identical replicas
    model = getattr(sb, cls)(**params).to(dev).train()
    bucket = FlatBucket(model)
    optim = FusedAdam(bucket, lr=lr)
    RUN["bucket_bytes"] = 4 * bucket.numel
    inputs, target = synth_batch(torch, B, 1234 + rank, dev, cls != "NetOptim", cycle_radii=world > 1)

    def step():
        if forward_only:
            with torch.no_grad():
                return model(inputs)["output"]
        return train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    try:
        for _ in range(warmup):
            step()
        barrier()
        ops.sched_counts_reset()
        # ONE event per step boundary on the launch stream (nothing around or inside the kernels): the per-step times behind
        # `ms_per_step_median` (SURVEY 8d asks for the median of >= 20 steps); `value` stays exactly K steps / wall clock
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):                               # the timed region carries no per-kernel profiling events
            step()
            marks[i + 1].record()
        barrier()
        dt = time.perf_counter() - t0
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        RUN["step_ms"] = {"median": per_step[len(per_step) // 2] if steps % 2 else
                          0.5 * (per_step[steps // 2 - 1] + per_step[steps // 2]), "min": per_step[0], "max": per_step[-1],
                          "n": steps}
        RUN["last_counts"] = dict(ops.SCHED_COUNTS, steps=steps)
        try:                                                 # overlapped-forward workgroups that handed their item back (harmless; cumulative per process)
            RUN["last_counts"]["fwd_giveups_total"] = int(ops.read_giveups()) & 0xFFFFF if ops._FLAG_ARENAS else 0
        except Exception:
            pass
        ops.check_sched_status()                             # a time-segmented launch that bailed out voids the run
        prof, prof_steps = {}, 0
        if profile:                                          # separate, untimed pass: HIP-event pairs (on the launch stream)
            prof_steps = max(2, min(4, steps))               # around every recurrent-kernel launch -> per-kernel table
            ops.PROFILE = {}
            for _ in range(prof_steps):
                step()
            barrier()
            prof = ops.PROFILE or {}
            ops.PROFILE = None
    finally:
        ops.PROFILE = None
        ops.BPTT = old_mode
    if world > 1:
        t = torch.tensor([dt], device="cpu" if dist.get_backend() == "gloo" else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    table = {}
    for label, evs in prof.items():
        ms = [e[0].elapsed_time(e[1]) for e in evs]
        side = [m for m, e in zip(ms, evs) if len(e) > 5 and e[5] == "side"]
        table[label] = dict(launches=len(evs), total_ms=sum(ms), flops=sum(e[2] for e in evs),
                            compulsory_bytes=sum(e[3] for e in evs), design_bytes=sum(e[4] for e in evs),
                            side_launches=len(side), side_ms=sum(side), events=[(e[0], e[1], e[5] if len(e) > 5 else "main") for e in evs])
    for t in table.values():
        t["steps"] = prof_steps
    # the cross-pass pair of a block (producer on the caller's stream || consumer: side-stream launch + a launch behind the
    # producer): wall time from the producer's start to the end of the consumer's call (which has joined the side stream)
    prods = [k for k in table if "[cross-pass producer]" in k]
    conss = [k for k in table if "[cross-pass consumer" in k]
    if len(prods) == 1 and len(conss) == 1:
        pe = [e for e in table[prods[0]]["events"] if e[2] == "main"]
        ce = [e for e in table[conss[0]]["events"] if e[2] == "main"]
        if len(pe) == len(ce) and pe:
            wall = [a[0].elapsed_time(b[1]) for a, b in zip(pe, ce)]
            table["__cross_pair__"] = dict(producer=prods[0], consumer=conss[0], passes=len(wall), wall_ms=sum(wall), steps=prof_steps)
    for t in table.values():
        t.pop("events", None)
    del model, bucket, optim, inputs, target
    torch.cuda.empty_cache()
    return dt / steps, table, B, params, cls


def roofline_of(table, wl, step_s, steps, forward_only, utt_s_per_gpu, params, mode="wide"):
    """`roofline` object: the kernel with the largest total time (the sum over ALL its launches, side-stream ones included --
    what `rocprofv3 --kernel-trace --stats` ranks by), its per-launch figures, and -- where it is half of an overlapped pair --
    the pair's pass-level figures."""
    fpu, bpu = fwd_flops_per_utt(params), fwd_bytes_per_utt(params)
    work_mult = 1.0 if forward_only else 3.0
    extra = {"step_flop_fraction_of_fp32_peak": work_mult * fpu * utt_s_per_gpu / MFMA_F32_PEAK,
             "step_flop_fraction_issued_of_fp16_peak": 3.0 * work_mult * fpu * utt_s_per_gpu / MFMA_BF16_PEAK,
             "step_hbm_fraction": work_mult * bpu * utt_s_per_gpu / HBM_PEAK}
    table = dict(table)
    pair = table.pop("__cross_pair__", None)
    if not table:
        return dict(bound="hbm", achieved=None, peak=HBM_PEAK / 1e9, unit="GB/s", frac=None, traffic=None, **extra)
    per = {}
    for label, t in table.items():
        n, sec = t["launches"], t["total_ms"] * 1e-3
        traffic, src, sym = pmc_traffic(wl, label, mode) if not forward_only else (None, None, None)
        per[label] = {
            "launches_per_step": n / t["steps"], "avg_launch_ms": t["total_ms"] / n,
            "kernel_ms_per_step": t["total_ms"] / t["steps"],
            "share_of_step": sec / (step_s * t["steps"]),
            "compulsory_gbs": t["compulsory_bytes"] / sec / 1e9, "frac_compulsory_bytes": t["compulsory_bytes"] / sec / HBM_PEAK,
            "design_gbs": t["design_bytes"] / sec / 1e9, "frac_design_bytes": t["design_bytes"] / sec / HBM_PEAK,
            "traffic_bytes_per_launch": traffic,
            "frac_hbm_counter": (traffic * n / sec / HBM_PEAK) if traffic else None,
            "algorithmic_tflops": t["flops"] / sec / 1e12,
            "frac_compute_issued": 3.0 * t["flops"] / sec / MFMA_BF16_PEAK,
            "mfma_busy": pmc_mfma_busy(wl, label, mode) if not forward_only else None,
            "traffic_source": src, "traffic_kernel": sym}
        if t.get("side_launches"):
            per[label].update(side_stream_launches_per_step=t["side_launches"] / t["steps"],
                              avg_side_launch_ms=t["side_ms"] / t["side_launches"],
                              avg_main_launch_ms=(t["total_ms"] - t["side_ms"]) / max(1, n - t["side_launches"]))
    top = max(table, key=lambda k: table[k]["total_ms"])
    t, p = table[top], per[top]
    sec = t["total_ms"] * 1e-3
    if forward_only:       # nothing but hs / y leaves the chip: the matrix pipe is the nearest roof
        roof = {"bound": "mfma", "achieved": 3.0 * t["flops"] / sec / 1e12, "peak": MFMA_BF16_PEAK / 1e12, "unit": "TFLOP/s",
                "frac": p["frac_compute_issued"],
                "note": "achieved = matrix flops ISSUED on the fp16 pipe (3 products per algorithmic MAC, hi+lo operands)"}
    else:
        roof = {"bound": "hbm", "achieved": p["compulsory_gbs"], "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": p["frac_compulsory_bytes"],
                "note": "kernel = the one with the largest total kernel time per step (all launches, side stream included: the "
                        "ranking of rocprofv3 --stats); achieved = SURVEY 8(d) compulsory bytes of its launches (4C in + 4C out per "
                        "position of the pass) / the sum of their durations, i.e. per launch = algorithmic_bytes_per_launch / "
                        "avg_launch_ms (HIP events on the stream each launch runs on; avg_launch_ms is the rocprofv3 average). "
                        "A consumer launch's duration includes its bounded waits for producer slabs: `pass_level` is the pair's "
                        "figure over wall time.  frac_design_bytes counts what this implementation moves (BPTT records, side "
                        "outputs), frac_hbm_counter the PMC-measured traffic"}
    roof.update({"kernel": top, "traffic": p["traffic_bytes_per_launch"], "traffic_unit": "bytes/launch",
                 "traffic_source": p["traffic_source"], "traffic_kernel": p["traffic_kernel"], "launches": t["launches"],
                 "avg_launch_ms": p["avg_launch_ms"], "kernel_ms_per_step": p["kernel_ms_per_step"],
                 "share_of_step": p["share_of_step"],
                 "share_note": "kernel time / step wall time: concurrent launches count separately, shares can sum past 1",
                 "algorithmic_bytes_per_launch": t["compulsory_bytes"] / t["launches"],
                 "design_bytes_per_launch": t["design_bytes"] / t["launches"],
                 "algorithmic_flops_per_launch": t["flops"] / t["launches"],
                 "frac_compulsory_bytes": p["frac_compulsory_bytes"], "frac_design_bytes": p["frac_design_bytes"],
                 "frac_hbm_counter": p["frac_hbm_counter"], "frac_compute_issued": p["frac_compute_issued"],
                 "mfma_busy": p["mfma_busy"], "algorithmic_tflops": p["algorithmic_tflops"]})
    if pair is not None:
        w = pair["wall_ms"] * 1e-3
        tp, tc = table[pair["producer"]], table[pair["consumer"]]
        cb = tp["compulsory_bytes"] + tc["compulsory_bytes"]
        tr = [per[k]["traffic_bytes_per_launch"] * table[k]["launches"] if per[k]["traffic_bytes_per_launch"] else None
              for k in (pair["producer"], pair["consumer"])]
        roof["pass_level"] = {
            "what": "one block's backward: inter-frame producer || intra-frame consumer (both launches), wall time from the "
                    "producer's start to the consumer's join, HIP events on the caller's stream",
            "kernels": [pair["producer"], pair["consumer"]], "passes": pair["passes"], "avg_wall_ms": pair["wall_ms"] / pair["passes"],
            "compulsory_bytes_per_pass": cb / pair["passes"], "achieved": cb / w / 1e9, "unit": "GB/s", "frac": cb / w / HBM_PEAK,
            "design_bytes_per_pass": (tp["design_bytes"] + tc["design_bytes"]) / pair["passes"],
            "frac_design_bytes": (tp["design_bytes"] + tc["design_bytes"]) / w / HBM_PEAK,
            "traffic_bytes_per_pass": (sum(tr) / pair["passes"]) if all(x is not None for x in tr) else None,
            "frac_hbm_counter": (sum(tr) / w / HBM_PEAK) if all(x is not None for x in tr) else None,
            "share_of_step": w / (step_s * pair["steps"])}
    roof["kernels"] = per
    roof.update(extra)
    if not forward_only:
        # where `traffic` / `mfma_busy` come from: committed rocprofv3 --pmc summaries, NOT this run -- with the digest of the
        # kernel sources they were taken on against this tree's
        sfx = "" if mode == "compact" else "_" + mode
        tp, sp = profile_provenance(_newest(f"r*_pmc_traffic_{wl}{sfx}.json")), profile_provenance(_newest(f"r*_pmc_sq_{wl}{sfx}.json"))
        roof["counter_profiles"] = {"traffic": tp, "sq": sp}
        roof["stale_profile"] = bool((tp and tp["stale_profile"]) or (sp and sp["stale_profile"]))
        roof["north_star_note"] = ("north_star's 0.30 of the HBM roof on the forward = ~7.9 k utt/s (big) = ~700 TFLOP/s sustained "
                                   "(SURVEY 8d): at this path's parity-grade arithmetic -- 3 fp16 products per MAC -- that is ~84 % "
                                   "utilisation of the 2.5 PFLOP/s matrix pipe by a 625-step serial recurrence; not reachable at "
                                   "fp32-class parity (secondary.forward_*: 0.08-0.09 measured)")
    return roof


def train_line(torch, dist, sb, ops, wl, args, dev, world, rank, *, forward_only=False, with_exact=True, with_cpu=True):
    step_s, table, B, params, cls = run_workload(torch, dist, sb, ops, wl, args, dev, world, rank, forward_only=forward_only)
    main_mode = args.bptt or ops.BPTT
    step_stats = RUN.get("step_ms")
    schedules = gather_schedules(dist, world, RUN.get("last_counts"))     # every rank: which inter-frame schedules its timed steps took
    if rank != 0:
        if with_exact and not forward_only:                  # every rank runs the sibling (collectives inside)
            run_workload(torch, dist, sb, ops, wl, args, dev, world, rank, mode=SIBLING[main_mode],
                         steps=max(3, args.steps // 2), warmup=min(2, args.warmup), profile=True)
        return None
    utt_s = world * B / step_s
    out = {
        "metric": "utterances/sec (6-ch, 24 kHz, 5 s) " + ("forward" if forward_only else "train-step"),
        "value": utt_s, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "ms_per_step_median": (step_stats or {}).get("median"),
        "value_at_median_step": (world * B / (step_stats["median"] * 1e-3)) if step_stats else None,
        "step_ms_spread": step_stats, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPE_FWD if forward_only else DTYPE_BY_MODE[main_mode], "data": "synthetic",
        "config": {"workload": f"{wl}: {cls} D={params['D']} B={params['B']} H=64 conv_lstm={params['conv_lstm']}, "
                               f"6ch x 120000 samples, {'forward only' if forward_only else 'fwd+SNRLP+bwd+clip+Adam'}"
                               + (" (BASELINE configs[2]; per-GPU workload of configs[3])" if wl == "big" else
                                  " (BASELINE configs[1])" if wl == "small" else ""),
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
        "roofline": roofline_of(table, wl, step_s, args.steps, forward_only, utt_s / world, params, main_mode),
        "rccl": RUN.get("rccl"),
        "schedules": {"per_rank": schedules,
                      "note": "inter-frame launches of the timed steps per rank: overlapped (producer || consumer, recurrence || "
                              "stream kernel on the library's side stream) or plain order; deferred_joins = backward passes "
                              "whose small launches (partial-row reductions, 3x3 weight gradients) rode on the side stream"},
    }
    if not forward_only:
        from sound_bubble_amd import train as _train
        out["rccl"] = dict(out["rccl"] or {}, allreduce_in_step=bool(world > 1 or _train.FORCE_ALLREDUCE),
                           allreduce_bytes=RUN.get("bucket_bytes"))
    if forward_only:
        # north-star target: >= 30 % of the HBM roofline on the forward.  The forward is 292 (big) / 223 (small) FLOP per
        # compulsory byte, so an exact-fp32 MFMA implementation tops out at 6.7 % / 8.8 %; on the fp16 pipe
        # with 3 products per MAC the ceiling is 2500/3 TFLOP/s -> ~35 % / ~47 % of the HBM roof.
        fpu, bpu = fwd_flops_per_utt(params), fwd_bytes_per_utt(params)
        hbm_roof_utt = HBM_PEAK / bpu
        out["forward_roofline"] = {
            "hbm_frac": utt_s / world / hbm_roof_utt, "north_star_target_hbm_frac": 0.30,
            "ceiling_fp32_mfma_hbm_frac": (MFMA_F32_PEAK / fpu) / hbm_roof_utt,
            "ceiling_fp16x3_hbm_frac": (MFMA_BF16_PEAK / 3.0 / fpu) / hbm_roof_utt,
            "frac_of_fp16x3_ceiling": (utt_s / world) / (MFMA_BF16_PEAK / 3.0 / fpu),
            "target_reachable": "no, at parity-grade arithmetic: with 3 products per MAC the matrix-pipe ceiling itself is "
                                "ceiling_fp16x3_hbm_frac of the HBM roof, so 0.30 would need ~85 % MFMA utilisation of a 625-step "
                                "serial recurrence; the opt-in two-product forward (secondary.forward_*_2prod, dtype stated there) "
                                "is the remaining lever",
            "gap": "the recurrent kernels issue ~1 instruction per 4 cycles from ONE wave per SIMD over a serial time "
                   "loop; the cell update (40 quarter-rate transcendentals + ~100 VALU ops per step) and the LDS hidden-"
                   "state exchange, not the MFMA issue, set the step time (profiles/: SQ_VALU_MFMA_BUSY vs SQ_BUSY)"}
    if with_exact and not forward_only:
        sib = SIBLING[main_mode]
        es, etab, _, _, _ = run_workload(torch, dist, sb, ops, wl, args, dev, world, rank, mode=sib,
                                         steps=max(3, args.steps // 2), warmup=min(2, args.warmup), profile=True)
        out["bptt_mode"] = main_mode
        sroof = roofline_of(etab, wl, es, max(3, args.steps // 2), False, B / es, params, sib)
        out[sib + "_bptt"] = {"value": world * B / es, "unit": "utterances/s", "ms_per_step": es * 1e3,
                              "steps": max(3, args.steps // 2), "dtype": DTYPE_BY_MODE[sib],
                              "ratio_to_headline": (world * B / es) / utt_s,
                              "roofline": {k: sroof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                                                  "avg_launch_ms", "share_of_step", "frac_design_bytes",
                                                                  "frac_hbm_counter", "frac_compute_issued") if k in sroof},
                              "kernels": {k: {"launches_per_step": v["launches_per_step"], "avg_launch_ms": v["avg_launch_ms"],
                                              "share_of_step": v["share_of_step"]} for k, v in sroof.get("kernels", {}).items()}}
    if with_cpu and world == 1:                               # reported baseline: rank 0 at N=1 only
        out["cpu_baseline"] = cpu_baseline(torch, wl)
        out["cpu_baseline"]["gpu_over_cpu"] = utt_s / out["cpu_baseline"]["value"]
        out["cpu_baseline"]["legs"]["stream_small"] = cpu_stream_leg(torch, "small")     # configs[4] on the CPU
    if not forward_only and rank == 0 and wl.startswith("big") and not args.no_parity:
        out["parity"] = parity_check(torch, sb, dev)
        try:
            out["parity"]["trained"] = parity_trained(torch, dev)
        except Exception as e:
            out["parity"]["trained"] = {"error": f"{type(e).__name__}: {e}"}
    return out


RUN = {}                        # per-process run facts folded into every train line: rccl (process group), schedules


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a torchrun environment: re-exec the same command line as N ranks under
    `torch.distributed.run` on this node (one process per GPU, rendezvous on 127.0.0.1).  Refuses -- non-zero exit, nothing
    measured -- when the node shows fewer than N GPUs (SB_FORCE_DEVICE, the several-ranks-on-one-GPU test hook, lifts that)."""
    import subprocess
    if os.environ.get("SB_FORCE_DEVICE") is None:
        import torch
        n = torch.cuda.device_count()
        if n < args.gpus:
            print(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node shows {n}: refusing to measure "
                  f"fewer ranks than asked for", file=sys.stderr, flush=True)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    return subprocess.call(cmd, env=env)


def dist_info(torch, dist, dev, backend, world):
    """`rccl` object of the bench lines: what the process group really is (backend, size, one device entry per rank)."""
    me = {"rank": int(os.environ.get("RANK", "0")), "device": str(dev), "pid": os.getpid()}
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        me.update(name=pr.name, cus=pr.multi_processor_count, pci_bus_id=getattr(pr, "pci_bus_id", None))
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world": 1, "devices": [me], "note": "single process, no process group"}
    devs = [None] * world
    dist.all_gather_object(devs, me)
    info = {"backend": dist.get_backend(), "world": world, "devices": devs}
    if backend == "nccl":
        try:
            info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
    return info


def rccl_world1_probe(torch, dist, dev, nbytes, reps=50):
    """VERDICT r4 #6b: what ONE all-reduce of the gradient bucket costs where it can be measured on a 1-GPU box -- RCCL in a
    process group of one rank (launch + kernel, no wire): HIP events around `reps` back-to-back calls and the host time per call.
    When the 8-GPU curve is taken, the 1 -> N loss can be set against this floor.  Runs after every timed region."""
    import time as _t
    own = False
    try:
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            own = True
        buf = torch.zeros(nbytes // 4, device=dev, dtype=torch.float32)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = _t.perf_counter()
        e0.record()
        for _ in range(reps):
            dist.all_reduce(buf)
        e1.record()
        host = (_t.perf_counter() - t0) / reps
        torch.cuda.synchronize()
        out = {"bytes": nbytes, "reps": reps, "gpu_us_per_call": e0.elapsed_time(e1) * 1e3 / reps, "host_us_per_call": host * 1e6,
               "backend": dist.get_backend(), "world": dist.get_world_size(),
               "note": "RCCL all-reduce of the flat gradient bucket in a world of ONE rank: launch + kernel floor, no xGMI traffic"}
        try:
            out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            pass
        return out
    except Exception as e:                                   # a box without a usable RCCL: recorded, not fatal
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        if own and dist.is_initialized():
            dist.destroy_process_group()


def serial_tail_probe(torch, sb, wl, dev, steps=6):
    """... and the serial tail of a step behind the backward pass: host time from `loss.backward()` returning to the optimiser's
    launch having been issued (side-stream join, [all-reduce], clip, Adam), and the GPU time from the end of the backward's last
    kernel to the end of the Adam kernel (events on the launch stream)"""
    import time as _t
    from sound_bubble_amd import ops
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket, FusedAdam, allreduce_grads
    cls, params, B, negw, clip, lr = WORKLOADS[wl]
    torch.manual_seed(0)
    model = getattr(sb, cls)(**params).to(dev).train()
    bucket = FlatBucket(model)
    optim = FusedAdam(bucket, lr=lr)
    inputs, target = synth_batch(torch, B, 1234, dev, cls != "NetOptim")
    host, gpu = [], []
    for i in range(steps):
        bucket.zero_grad()
        loss, _ = SnrlpLossFn.apply(model(inputs)["output"], target, negw)
        ops.absmax_hints_clear()
        loss.backward()
        t0 = _t.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.deferred_join()
        ops.absmax_hints_clear()
        world = allreduce_grads(bucket)
        optim.step(grad_clip=clip, world_size=world)
        e1.record()
        dt = _t.perf_counter() - t0
        torch.cuda.synchronize()
        if i >= 2:
            host.append(dt * 1e6)
            gpu.append(e0.elapsed_time(e1) * 1e3)
    del model, bucket, optim
    torch.cuda.empty_cache()
    med = lambda v: sorted(v)[len(v) // 2]
    return {"host_us_backward_return_to_adam_issued": med(host), "gpu_us_backward_end_to_adam_end": med(gpu), "steps": len(host),
            "note": "the part of a step no overlap hides: join of the side stream, (all-reduce when N > 1), clip + Adam launch"}


def gather_schedules(dist, world, counts):
    """per-rank schedule counts of the timed region, all-gathered: a side stream lost next to RCCL shows as plain-order counts"""
    if world > 1 and dist.is_initialized():
        out = [None] * world
        dist.all_gather_object(out, counts)
        return out
    return [counts]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="all", choices=["all"] + list(WORKLOADS),
                    help="all (default): secondary lines, then the headline (big); a name: that train line only")
    ap.add_argument("--headline-only", action="store_true", help="with --workload all: only the headline line")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the `parity` object (one B = 1 scene through the big model): counter passes use this so that the "
                         "scene's launches do not mix into the per-kernel averages")
    ap.add_argument("--no-exact", action="store_true", help="skip the sibling measurement in the other BPTT-state precision")
    ap.add_argument("--bptt", default=None, choices=["wide", "compact", "legacy"],
                    help="BPTT-state precision of the main line (default: SB_BPTT or wide)")
    ap.add_argument("--vendor-gpu-baseline", action="store_true",
                    help="extra leg: time the oracle restatement on the GPU through stock torch ops (MIOpen RNN)")
    ap.add_argument("--forward-only", action="store_true", help="single-workload mode: inference forward utt/s")
    ap.add_argument("--stream", action="store_true",
                    help="single-workload mode (BASELINE configs[4]): hipGraph-captured 8 ms chunk loop")
    ap.add_argument("--no-graph", action="store_true", help="with --stream: eager launches instead of hipGraph replay")
    ap.add_argument("--init-dist", action="store_true",
                    help="initialise the process group (RCCL) even at --gpus 1: the step's all-reduce then really runs")
    ap.add_argument("--launch-check", action="store_true",
                    help="launcher / process-group plumbing only: start the ranks, all-gather their devices, print one line")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))                          # plain `python bench.py --gpus N`: becomes an N-rank run

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to report a "
                 f"{args.gpus}-GPU number from {world} rank(s)")
    force = os.environ.get("SB_FORCE_DEVICE")               # test hook: several ranks on one GPU (with gloo); "cpu": no GPU at all
    backend = os.environ.get("SB_DIST_BACKEND", "nccl")      # "nccl" IS RCCL on ROCm
    if args.launch_check and force == "cpu":
        dev = torch.device("cpu")
    else:
        if force is not None:
            local = int(force)
        elif torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, this node shows {torch.cuda.device_count()}")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    if world > 1 or args.init_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        world = dist.get_world_size()                        # n_gpus of every line below is the process group's size
    rccl = dist_info(torch, dist, dev, backend, world)
    if args.launch_check:                                    # launcher / process-group plumbing only (tests/test_distributed_cpu.py)
        plan = rank_plan("big" if args.workload == "all" else args.workload, rank, world, args.batch)
        plans = [plan]
        if dist.is_initialized():
            plans = [None] * world
            dist.all_gather_object(plans, plan)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "rccl": rccl, "plans": plans}), flush=True)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return

    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    from sound_bubble_amd import train as _train
    RUN["rccl"] = rccl
    _train.FORCE_ALLREDUCE = bool(args.init_dist)            # world 1 with a process group: the bucket all-reduce still runs

    def emit(obj):
        if rank == 0 and obj is not None:
            print(json.dumps(obj), flush=True)

    single = args.workload != "all"
    wl = args.workload if single else "big"
    if args.stream:
        cls, params = WORKLOADS[wl if single else "small"][:2]
        stream_bench(torch, sb, args, wl if single else "small", cls, params, dev)
    elif args.forward_only:
        emit(train_line(torch, dist, sb, ops, wl if single else "big", args, dev, world, rank, forward_only=True,
                        with_cpu=False))
    else:
        secondary = None
        if not single and not args.headline_only and world == 1:
            # secondary measurements (BASELINE configs[1] and [4], forward-only): each printed as its own JSON line AND folded,
            # compactly, into the headline's `secondary` object so that the last line alone carries every number of the run
            secondary = {}
            for w2 in ("small", "big"):
                o = stream_bench(torch, sb, args, w2, WORKLOADS[w2][0], WORKLOADS[w2][1], dev)
                secondary[f"stream_{w2}"] = {"chunks_s": o["value"], "p50_ms": o["latency_ms"]["p50"], "p99_ms": o["latency_ms"]["p99"],
                                             "realtime_factor": o["realtime_factor"], "config": "BASELINE configs[4], hipGraph chunk loop, B=1"}
            for w2 in ("small", "big"):
                o = train_line(torch, dist, sb, ops, w2, args, dev, world, rank, forward_only=True, with_cpu=False)
                emit(o)
                secondary[f"forward_{w2}"] = {"utt_s": o["value"], "ms_per_step": o["ms_per_step"],
                                              "hbm_frac": o["forward_roofline"]["hbm_frac"],
                                              "frac_of_fp16x3_ceiling": o["forward_roofline"]["frac_of_fp16x3_ceiling"],
                                              "batch": o["config"]["batch_per_gpu"]}
            try:
                tp = two_product_forward(torch, dist, sb, ops, args, dev, world, rank)
                for k in ("small", "big"):                   # speed-up over the default arithmetic measured minutes earlier in this run
                    tp[f"forward_{k}_2prod"]["speedup_vs_default"] = tp[f"forward_{k}_2prod"]["utt_s"] / secondary[f"forward_{k}"]["utt_s"]
                secondary.update(tp)
            except Exception as e:                           # an opt-in side measurement must not void the run
                secondary["forward_2prod_error"] = f"{type(e).__name__}: {e}"
            o = train_line(torch, dist, sb, ops, "small", args, dev, world, rank, with_exact=not args.no_exact, with_cpu=False)
            emit(o)
            sib = SIBLING[o.get("bptt_mode", "wide")] + "_bptt"
            secondary["train_small"] = {"utt_s": o["value"], "ms_per_step": o["ms_per_step"], "batch": o["config"]["batch_per_gpu"],
                                        "config": "BASELINE configs[1]", "bptt_mode": o.get("bptt_mode"),
                                        sib: o.get(sib, {}).get("value"),
                                        "roofline_kernel": o["roofline"].get("kernel"), "roofline_frac": o["roofline"].get("frac")}
        out = train_line(torch, dist, sb, ops, wl, args, dev, world, rank, with_exact=not args.no_exact,
                         with_cpu=not args.no_cpu_baseline)
        if rank == 0 and secondary is not None:
            out["secondary"] = secondary
            leg = out.get("cpu_baseline", {}).get("legs", {}).get("stream_small")
            if leg and "stream_small" in secondary:       # the hipGraph chunk loop beside the CPU chunk loop
                secondary["stream_small"]["cpu_chunks_s"] = leg["chunks_s"]
                secondary["stream_small"]["cpu_p50_ms"] = leg["p50_ms"]
                secondary["stream_small"]["gpu_over_cpu"] = secondary["stream_small"]["chunks_s"] / leg["chunks_s"]
        if rank == 0 and world == 1 and not single and not args.headline_only and isinstance(out.get("rccl"), dict):
            out["rccl"]["serial_tail"] = serial_tail_probe(torch, sb, wl, dev)
            out["rccl"]["allreduce_world1"] = rccl_world1_probe(torch, dist, dev, RUN.get("bucket_bytes") or 2005600)
        if rank == 0 and args.vendor_gpu_baseline:
            B = args.batch or WORKLOADS[wl][2]
            out["vendor_gpu_baseline"] = vendor_gpu_baseline(torch, wl, B, dev)
            if out["vendor_gpu_baseline"]:
                out["vendor_gpu_baseline"]["ours_over_vendor"] = out["value"] / out["vendor_gpu_baseline"]["value"]
        emit(out)                                            # the headline is the LAST line
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
