#!/usr/bin/env python3
"""Sound-Bubble hot-path benchmark on MI355X.

Metric (BASELINE.json): utterances/s of one optimiser step (forward + SNRLP loss + backward +
[RCCL all-reduce] + clip + Adam) on synthetic 6-ch x 24 kHz x 5 s utterances, whole job over N GPUs.
Default workload = BASELINE configs[1]: the 0.3 M-param TFG_S model
(real_experiments/raspberrypi_model_pretrain.json model_params), batch 32 per GPU (weak scaling).
`--workload big` runs configs[2]/[3]: the 0.5 M-param model (syn_experiments/pretrain_stage.json),
batch 16 per GPU.

Launch: `python bench.py --gpus 1` or
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
 bench.py --gpus N --steps K --warmup W`.

One JSON line on rank 0; adds `roofline` (dominant kernel: the recurrent LSTM forward, timed live with
HIP events on the launch stream) and `cpu_baseline` (the oracle -- a CPU port of the reference's
algorithm -- timed on the host cores on a bounded sample).
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


def synth_batch(torch, B, seed, device, with_dis):
    g = torch.Generator().manual_seed(seed)
    base = 0.1 * torch.randn(B, 1, N_SAMPLES + 8, generator=g)
    mix = torch.cat([base[..., 4 - min(m, 4): 4 - min(m, 4) + N_SAMPLES] for m in range(6)], 1)
    mix = (mix + 0.02 * torch.randn(B, 6, N_SAMPLES, generator=g)).clamp(-1, 1)
    tgt = 0.05 * torch.randn(B, 1, N_SAMPLES, generator=g)
    tgt[7::8] = 0.0                              # every 8th sample: silent target (negative branch)
    inputs = {"mixture": mix.to(device)}
    if with_dis:
        dis = torch.zeros(B, 3)
        dis[torch.arange(B), torch.arange(B) % 3] = 1.0
        inputs["dis_embed"] = dis.to(device)
    return inputs, tgt.to(device)


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


def cpu_baseline(torch, wl, budget_s=20.0):
    """Oracle (CPU port of the reference algorithm, oracle/tfgridnet_oracle.py) train step on host cores."""
    from oracle.tfgridnet_oracle import OracleNet, snrlp_loss
    cls, params, _, negw, clip, lr = WORKLOADS[wl]
    cores = host_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = OracleNet("optim" if cls == "NetOptim" else "dis_embd3", **params).train()
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    B = 1
    inputs, tgt = synth_batch(torch, B, 1234, "cpu", cls != "NetOptim")

    def step():
        opt.zero_grad()
        est = m(dict(inputs))["output"]
        snrlp_loss(est, tgt, negw).mean().backward()
        if clip:
            torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
        opt.step()

    step()                                       # warm-up
    t0, n = time.time(), 0
    while True:
        step()
        n += 1
        if time.time() - t0 > budget_s or n >= 8:
            break
    dt = (time.time() - t0) / n
    return {"value": B / dt, "unit": "utterances/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps of batch {B} ({wl} config, 5 s clips), oracle CPU port, after 1 warm-up"}


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


def stream_bench(torch, sb, args, cls, params, dev):
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
    print(json.dumps({
        "metric": "streaming chunks/sec (8 ms hop, 6 mics, B=1)", "value": 625 / dt, "unit": "chunks/s", "n_gpus": 1,
        "steps": 625, "warmup": 20, "ms_per_step": dt / 625 * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"stream-{args.workload}: {cls} D={params['D']} B={params['B']}, chunk [1,6,288] -> [1,1,192], "
                               f"{'eager launches' if args.no_graph else 'hipGraph replay'}"},
        "latency_ms": {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)),
                       "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
        "realtime_factor": 8.0 / float(np.percentile(lat, 50)),
        "reference_claim": "6.36 ms per 8 ms chunk on an embedded CPU (README.md:9)",
        "chunk_mflop": fpu / 1e6}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="small", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--vendor-gpu-baseline", action="store_true",
                    help="extra leg: time the oracle restatement on the GPU through stock torch ops (MIOpen RNN)")
    ap.add_argument("--forward-only", action="store_true", help="extra mode: inference forward utt/s")
    ap.add_argument("--stream", action="store_true",
                    help="extra mode (BASELINE config 5): hipGraph-captured 8 ms chunk loop, chunks/s + p50 latency")
    ap.add_argument("--no-graph", action="store_true", help="with --stream: eager launches instead of hipGraph replay")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "SB_FORCE_DEVICE" in os.environ:                      # test hook: several ranks on one GPU (with gloo)
        local = int(os.environ["SB_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SB_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    cls, params, B, negw, clip, lr = WORKLOADS[args.workload]
    B = args.batch or B
    if args.stream:
        return stream_bench(torch, sb, args, cls, params, dev)
    torch.manual_seed(0)                                     # identical replicas
    model = getattr(sb, cls)(**params).to(dev).train()
    bucket = FlatBucket(model)
    optim = FusedAdam(bucket, lr=lr)
    inputs, target = synth_batch(torch, B, 1234 + rank, dev, cls != "NetOptim")

    def step():
        if args.forward_only:
            with torch.no_grad():
                return model(inputs)["output"]
        return train_step(model, bucket, optim, inputs, target, negw, grad_clip=clip)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.PROFILE_LSTM = []                                    # HIP-event pairs around the dominant kernel
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ev = ops.PROFILE_LSTM
    ops.PROFILE_LSTM = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        utt_s = world * B * args.steps / dt
        # dominant kernel: recurrent LSTM forward (all launches: intra + inter), HIP events on the launch stream
        lstm_ms = [e[0].elapsed_time(e[1]) for e in ev]
        lstm_flops = [e[2] for e in ev]
        tot_ms, tot_fl, tot_by = sum(lstm_ms), sum(lstm_flops), sum(e[3] for e in ev)
        n_launch = max(1, len(ev))
        ach = tot_fl / (tot_ms * 1e-3) if tot_ms > 0 else 0.0
        fpu, bpu = fwd_flops_per_utt(params), fwd_bytes_per_utt(params)
        work_mult = 1.0 if args.forward_only else 3.0
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", f"r01_final_pmc_traffic_{args.workload}.json")
        if os.path.exists(pmc) and not args.forward_only and B == WORKLOADS[args.workload][2]:
            ks = [v for k, v in json.load(open(pmc))["kernels"].items() if "lstm_fwd" in k]
            if ks:                                            # HBM bytes per launch (PMC, FETCH_SIZE x2 + WRITE_SIZE)
                traffic = sum(v["hbm_bytes"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks)
                traffic_src = os.path.relpath(pmc, ROOT) + " (rocprofv3 --pmc pass of this command, committed)"
        # Which roof binds the dominant kernel (recurrent LSTM forward, intra + inter launches)?
        #  * train step: it writes the BPTT records -- the intra-frame launches are HBM-write bound, the PMC traffic equals
        #    the algorithmic bytes (profiles/) -> "hbm": algorithmic bytes per launch / launch time against 8 TB/s;
        #  * forward only: nothing but hs / y leaves the chip -> "mfma": the kernel runs on the 16-bit matrix pipe with
        #    split operands (fp16 hi+lo, 3 products per fp32 MAC by default; bf16 3-way, 6 products with SB_LSTM_BF16X6=1;
        #    fp32-input MFMA with SB_LSTM_FP32=1): ISSUED matrix flops against the dense peak of that pipe.
        nprod = {0: 1.0, 1: 3.0, 2: 6.0}[ops.LSTM_MMA]
        mfma_peak = MFMA_BF16_PEAK if ops.LSTM_MMA else MFMA_F32_PEAK
        issued = ach * nprod
        gbs = tot_by / (tot_ms * 1e-3) / 1e9 if tot_ms > 0 else 0.0
        if args.forward_only:
            roof = {"bound": "mfma", "achieved": issued / 1e12, "peak": mfma_peak / 1e12, "unit": "TFLOP/s",
                    "frac": issued / mfma_peak}
        else:
            roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": gbs * 1e9 / HBM_PEAK}
        out = {
            "metric": "utterances/sec (6-ch, 24 kHz, 5 s) " + ("forward" if args.forward_only else "train-step"),
            "value": utt_s, "unit": "utterances/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cls} D={params['D']} B={params['B']} H=64 "
                                   f"conv_lstm={params['conv_lstm']}, 6ch x 120000 samples, "
                                   f"{'forward only' if args.forward_only else 'fwd+SNRLP+bwd+clip+Adam'}",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "roofline": dict(roof, **{
                "kernel": ({1: "lstm_fwd_bf_kernel (fp16 MFMA, hi+lo operand split, 3 products per fp32 MAC)",
                            2: "lstm_fwd_bf_kernel (bf16 MFMA, 3-way operand split, 6 products per fp32 MAC)",
                            0: "lstm_fwd_kernel (fp32-input MFMA)"}[ops.LSTM_MMA]) + ", intra+inter launches",
                "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                "launches": len(ev), "avg_launch_ms": tot_ms / n_launch,
                "algorithmic_bytes_per_launch": tot_by / n_launch,
                "algorithmic_flops_per_launch": tot_fl / n_launch,
                "algorithmic_tflops": ach / 1e12, "algorithmic_frac_of_fp32_mfma_peak": ach / MFMA_F32_PEAK,
                "issued_matrix_tflops": issued / 1e12, "issued_frac_of_matrix_peak": issued / mfma_peak,
                "hbm_gbs": gbs, "hbm_frac": gbs * 1e9 / HBM_PEAK,
                "step_flop_fraction": work_mult * fpu * utt_s / world / MFMA_F32_PEAK,
                "step_hbm_fraction": work_mult * bpu * utt_s / world / HBM_PEAK}),
        }
        if not args.no_cpu_baseline and world == 1:          # reported baseline: rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(torch, args.workload)
            out["cpu_baseline"]["gpu_over_cpu"] = utt_s / out["cpu_baseline"]["value"]
        if args.vendor_gpu_baseline and not args.forward_only:
            del model, bucket, optim
            torch.cuda.empty_cache()
            out["vendor_gpu_baseline"] = vendor_gpu_baseline(torch, args.workload, B, dev)
            if out["vendor_gpu_baseline"]:
                out["vendor_gpu_baseline"]["ours_over_vendor"] = utt_s / out["vendor_gpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
