"""The HIP train step under torch.distributed (run on the MI355X box with -m gpu): 2 ranks share GPU 0 (gloo carries the
collectives through the host; on an 8-GPU node the same code runs one rank per GPU over RCCL), so the data-parallel
path of BASELINE configs[3] is exercised with the real kernels:

 (i)  the flat gradient bucket after the all-reduce (/ world) equals the single-process global-batch gradient of
      mean_b SNRLP -- exact-BPTT arithmetic to 1e-5 (only the summation order differs), default compact arithmetic to
      the fp16-record tolerance (the dgates scale is derived per shard);
 (ii) replicas stay bit-identical: after 3 optimiser steps that include a ReduceLROnPlateau decision on val losses that
      DIFFER per rank (rank 0's shard improves, rank 1's gets worse; only the merged mean shows the plateau), every
      rank holds the same learning rate and the same parameter bits.
Reference behaviour matched: nn.DataParallel's global-batch gradient and single-process epoch bookkeeping,
src/hl_modules/distance_based_hl_module.py:34-35,174-202,430-441.
"""
import os
import socket

import numpy as np
import pytest

from conftest import load_golden, golden_state_dict, rel_l2

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _grad_worker(rank, world, port, mode, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket, allreduce_grads
    ops.BPTT = mode
    rec, params, _ = load_golden("tiny_small")
    m = sb.NetOptim(**params)
    m.load_state_dict(golden_state_dict(rec, torch))
    m = m.cuda().train()
    bucket = FlatBucket(m)
    bucket.zero_grad()
    sl = slice(rank, rank + 1)                        # shard by utterance; sample 1 has an all-zero target
    est = m({"mixture": torch.from_numpy(rec["mixture"][sl]).cuda()})["output"]
    loss, _ = SnrlpLossFn.apply(est, torch.from_numpy(rec["target"][sl]).cuda(), 100.0)
    loss.backward()
    w = allreduce_grads(bucket)
    assert w == world
    if rank == 0:
        q.put((bucket.grad / w).cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["wide", "legacy", "compact"])
def test_two_ranks_on_one_gpu_allreduce_matches_global_batch_gradient(mode):
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    g_dp = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket
    old = ops.BPTT
    ops.BPTT = mode
    try:
        rec, params, _ = load_golden("tiny_small")
        m = sb.NetOptim(**params)
        m.load_state_dict(golden_state_dict(rec, torch))
        m = m.cuda().train()
        bucket = FlatBucket(m)
        bucket.zero_grad()
        est = m({"mixture": torch.from_numpy(rec["mixture"]).cuda()})["output"]
        loss, _ = SnrlpLossFn.apply(est, torch.from_numpy(rec["target"]).cuda(), 100.0)
        loss.backward()
        g_ref = bucket.grad.cpu().numpy()
    finally:
        ops.BPTT = old
    err = rel_l2(g_dp, g_ref)
    assert err < (2e-3 if mode == "compact" else 1e-5), err


def _replica_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sound_bubble_amd.harness import PLModule
    rec, params, _ = load_golden("tiny_small")
    torch.manual_seed(100 + rank)                     # DIFFERENT initial weights per rank: the constructor must sync them
    hl = PLModule(model="src.models.tfgridnet_realtime_clean_optim.net.Net", model_params=params, sr=24000,
                  optimizer="torch.optim.Adam", optimizer_params={"lr": 2e-3},
                  scheduler="torch.optim.lr_scheduler.ReduceLROnPlateau",
                  scheduler_params={"mode": "min", "factor": 0.5, "patience": 0},
                  loss="src.losses.SNRLP.SNRLPLoss", loss_params={"snr_loss_name": "snr", "neg_weight": 50},
                  metrics=["si_sdr_i"], grad_clip=1.0)
    g = torch.Generator().manual_seed(7 + rank)       # each rank trains on its own utterances
    n = rec["mixture"].shape[-1]
    lrs = []
    # per-rank val losses: rank 0 keeps improving, rank 1 gets worse; the merged means are 1.0 -> 1.75 -> 2.5
    val = {0: [1.0, 0.5, 0.25], 1: [1.0, 3.0, 4.75]}[rank]
    for epoch in range(3):
        hl.train()
        mix = 0.1 * torch.randn(2, 6, n, generator=g)
        tgt = 0.05 * torch.randn(2, 1, n, generator=g)
        batch = ({"mixture": mix.cuda()}, {"target": tgt.cuda(), "num_target_speakers": torch.tensor([1, 2]),
                                           "num_interfering_speakers": torch.tensor([0, 1]), "num_noises": torch.tensor([1, 1])})
        hl.reset_grad()
        loss, B = hl.training_step(batch, 0)
        loss.backward()
        hl.backprop()
        hl.log_metric("val/loss", val[epoch], batch_size=4 if rank == 0 else 4)
        if rank == 1 and epoch == 2:
            hl.log_metric("val/only_on_rank1", 1.0, 1)          # ragged metric names must merge too
        hl.on_epoch_end(os.devnull, None)
        lrs.append(hl.get_current_lr())
    flat = hl.bucket.flat.detach().cpu()
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    merged_val = [hl.get_avg_metric_at_epoch("val/loss", e) for e in range(3)]
    if rank == 0:
        q.put((lrs, [t.numpy() for t in gathered], merged_val, "val/only_on_rank1" in hl.metric_values[2]))
    all_lrs = [None] * world
    dist.all_gather_object(all_lrs, lrs)
    assert all_lrs[0] == all_lrs[1], all_lrs
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_stay_bit_identical_through_plateau_scheduler():
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    lrs, flats, merged_val, ragged = q.get()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert np.array_equal(flats[0], flats[1])                  # bit-identical replicas after 3 steps
    assert np.isfinite(flats[0]).all()
    np.testing.assert_allclose(merged_val, [1.0, 1.75, 2.5])   # every rank sees the merged epoch mean
    assert lrs == [2e-3, 1e-3, 5e-4], lrs                      # plateau seen on the MERGED loss (rank 0 alone improves)
    assert ragged


def test_bench_contract_under_torchrun_two_ranks_one_gpu():
    """The driver's multi-GPU launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`)
    with 2 ranks on GPU 0 (SB_FORCE_DEVICE / SB_DIST_BACKEND=gloo hooks): rank 0 prints ONE JSON line, whole-job value,
    weak scaling, max-over-ranks timing, the all-reduce inside the timed step."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, SB_FORCE_DEVICE="0", SB_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "small", "--batch", "4", "--no-exact"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8 and d["value"] > 0
    assert abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and d["roofline"]["kernel"]


def _overlap_worker(rank, world, port, mode, q):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket, allreduce_grads
    ops.BPTT = mode
    ops.OVERLAP_MIN_FILL = 0.0                        # 19 inter-frame tiles at this size: force the overlapped schedules
    rec, params, _ = load_golden("tiny_big")
    torch.manual_seed(3)                              # BEFORE construction: the default initialisers draw too
    m = sb.NetDisEmbd3(**dict(params, B=6))           # the BIG model's depth (six blocks), seeded weights
    for p in m.parameters():
        torch.nn.init.uniform_(p, -0.2, 0.2) if p.dim() > 1 else None
    m = m.cuda().train()
    bucket = FlatBucket(m)
    g = torch.Generator().manual_seed(11)
    mix = 0.1 * torch.randn(2 * world, 6, 192 * 150 + 96, generator=g)
    tgt = 0.05 * torch.randn(2 * world, 1, 192 * 150, generator=g)
    dis = torch.eye(3)[torch.arange(2 * world) % 3]
    sl = slice(2 * rank, 2 * rank + 2)
    avail = []
    for it in range(3):                               # the other rank's kernels contend for the CUs all along
        bucket.zero_grad()
        est = m({"mixture": mix[sl].cuda(), "dis_embed": dis[sl].cuda()}, pad=False)["output"]
        loss, _ = SnrlpLossFn.apply(est, tgt[sl].cuda(), 100.0)
        loss.backward()
        w = allreduce_grads(bucket)
        ops.check_sched_status_all_ranks()            # no watchdog trip (raises on EVERY rank if one aborted)
        avail.append(ops.overlap_available())
    still = ops.overlap_reprobe() if avail[-1] else False
    out = [None] * world
    dist.all_gather_object(out, (avail, still, [e[2:] for e in ops.OVERLAP_LOG]))
    if rank == 0:
        q.put(((bucket.grad / w).cpu().numpy(), out, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["wide", "compact"])
def test_two_ranks_contending_for_one_gpu_with_overlapped_schedules(mode):
    """8-GPU readiness that one GPU can check: the six-block big model's train step in two processes on GPU 0
    with the overlapped schedules forced on -- the side-stream probe, the guarded spin-waits and the watchdog run with a
    FOREIGN process's kernels contending for the CUs (what RCCL's kernels do on a node).  Whatever the probe decides per
    rank (concurrent side stream -> overlapped launches; none -> plain order, never a slow overlapped one: the entry
    points refuse without a side stream), no watchdog trips and the all-reduced gradient equals the single-process
    global-batch gradient."""
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    g_dp, info, sd = q.get()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    print(f"[{mode}] per-rank (overlap available per step, still concurrent after, probe log):", info)
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket
    rec, params, _ = load_golden("tiny_big")
    m = sb.NetDisEmbd3(**dict(params, B=6))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda().train()
    bucket = FlatBucket(m)
    g = torch.Generator().manual_seed(11)
    mix = 0.1 * torch.randn(4, 6, 192 * 150 + 96, generator=g)
    tgt = 0.05 * torch.randn(4, 1, 192 * 150, generator=g)
    dis = torch.eye(3)[torch.arange(4) % 3]
    old = (ops.BPTT, ops.FWD_OVERLAP, ops.BWD_OVERLAP)
    ops.BPTT, ops.FWD_OVERLAP, ops.BWD_OVERLAP = mode, False, False          # reference: one process, plain order
    try:
        bucket.zero_grad()
        loss, _ = SnrlpLossFn.apply(m({"mixture": mix.cuda(), "dis_embed": dis.cuda()}, pad=False)["output"], tgt.cuda(), 100.0)
        loss.backward()
        g_ref = bucket.grad.cpu().numpy()
    finally:
        ops.BPTT, ops.FWD_OVERLAP, ops.BWD_OVERLAP = old
    # the shared negative-sample term of SNRLP is shard-invariant only for all-positive batches of equal size: none of these
    # targets is silent, so mean-of-local-means == global mean
    err = rel_l2(g_dp, g_ref)
    assert err < (2e-3 if mode == "compact" else 2e-5), err


def _rccl_world_of_one(port, q):
    """child process: RCCL itself (backend "nccl" on ROCm) with a world of one rank on GPU 0"""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import sound_bubble_amd as sb
    from sound_bubble_amd import ops, train
    from sound_bubble_amd.train import FlatBucket, FusedAdam, train_step
    ops.OVERLAP_MIN_FILL = 0.0                        # 19 inter-frame tiles: force the overlapped schedules on
    rec, params, _ = load_golden("tiny_big")
    g = torch.Generator().manual_seed(11)
    mix = (0.1 * torch.randn(2, 6, 192 * 150, generator=g)).cuda()    # pad=True adds the 96 look-ahead samples: T = 150
    tgt = (0.05 * torch.randn(2, 1, 192 * 150, generator=g)).cuda()
    dis = torch.eye(3)[torch.arange(2) % 3].cuda()
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import allreduce_grads
    torch.manual_seed(3)
    m = sb.NetDisEmbd3(**dict(params, B=6)).cuda().train()
    bucket = FlatBucket(m)
    optim = FusedAdam(bucket, lr=1e-3)
    inputs = {"mixture": mix, "dis_embed": dis}
    ops.sched_counts_reset()
    same, adam_err = [], []
    for it in range(3):
        # train_step's own sequence with the gradient bucket read back on either side of the collective
        bucket.zero_grad()
        loss, _ = SnrlpLossFn.apply(m(inputs)["output"], tgt, 100.0)
        loss.backward()
        g0 = bucket.grad.clone()
        w = allreduce_grads(bucket, force=True)          # RCCL's kernel on the 2 MB bucket: a sum over one rank
        assert w == 1
        g1 = bucket.grad.clone()
        p0, m0, v0 = bucket.flat.clone(), optim.m.clone(), optim.v.clone()
        optim.step(grad_clip=None, world_size=w)         # reads what the collective left, in stream order
        t = optim.step_count
        m1 = 0.9 * m0 + 0.1 * g1
        v1 = 0.999 * v0 + 0.001 * g1 * g1
        ref = p0 - 1e-3 * (m1 / (1 - 0.9 ** t)) / ((v1 / (1 - 0.999 ** t)).sqrt() + 1e-8)
        same.append(bool(torch.equal(g0, g1)) and bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0)
        adam_err.append(float((bucket.flat - ref).abs().max() / ref.abs().max()))
    train.FORCE_ALLREDUCE = True                         # ... and through train_step itself
    for it in range(2):
        train_step(m, bucket, optim, inputs, tgt, 100.0, grad_clip=1.0)
    torch.cuda.synchronize()
    ops.check_sched_status()
    counts = dict(ops.SCHED_COUNTS)
    still = ops.overlap_reprobe() if ops.overlap_available() else None
    ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    q.put((same, adam_err, counts, bool(torch.isfinite(bucket.flat).all()), still, ver, [e[2:] for e in ops.OVERLAP_LOG],
           dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world_of_one_train_step_with_forced_bucket_allreduce():
    """RCCL executes (VERDICT r3 #1b): `init_process_group("nccl", world_size=1)` on the one GPU there is, train steps of the
    six-block big model with the bucket all-reduce FORCED and the overlapped schedules on.  A sum over one rank is the
    identity, so the stream ordering RCCL's kernel gets between the backward's last gradient write and `optim.step` is checked
    to the bit: the bucket read back after the collective equals the bucket read back before it, and the parameters after the
    fused Adam equal Adam applied to that gradient.  No watchdog trip; the side stream still runs concurrently afterwards
    (the probe log is printed).  (Two RUNS of the step are not bit-comparable: the overlapped schedules deal their work by
    atomic draws, so the order of the weight-gradient partial sums differs from run to run.)"""
    import torch
    import torch.multiprocessing as mp
    assert torch.cuda.is_available()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=_rccl_world_of_one, args=(_free_port(), q))
    p.start()
    same, adam_err, counts, finite, still, ver, log, backend = q.get()
    p.join(300)
    assert p.exitcode == 0
    print(f"RCCL {ver} (backend {backend}); side stream concurrent after the run: {still}; probe log: {log}; "
          f"schedules: {counts}; Adam-after-collective error {adam_err}")
    assert backend == "nccl"
    assert same == [True, True, True]                              # gradient bucket untouched by the one-rank sum, bit for bit
    assert max(adam_err) < 2e-6 and finite
    if still is not None:
        assert counts["fwd_overlapped"] > 0 and counts["bwd_overlapped"] > 0, counts


def _bench(args, env_extra, timeout=900):
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` WITHOUT torchrun around it (the shape of the driver's N = 1 command with another N): the
    script launches two ranks itself and the line says n_gpus 2, with the process group and every rank's schedules in it."""
    import json
    out = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "small", "--batch", "4", "--no-exact"],
                 {"SB_FORCE_DEVICE": "0", "SB_DIST_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl"]["world"] == 2 and len(d["rccl"]["devices"]) == 2
    assert d["rccl"]["allreduce_in_step"] and d["rccl"]["allreduce_bytes"] > 900000
    assert len(d["schedules"]["per_rank"]) == 2 and all(s["steps"] == 2 for s in d["schedules"]["per_rank"])


def test_bench_refuses_two_gpus_on_a_one_gpu_box():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the refusal is covered by the CPU test")
    out = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    assert out.returncode != 0 and "refusing" in out.stderr
    assert not any(l.startswith("{") for l in out.stdout.splitlines())


def test_bench_one_gpu_line_through_rccl():
    """--init-dist at --gpus 1: the process group is RCCL with one rank and the step's bucket all-reduce really runs; the line
    names the backend, the device and the schedules the timed steps took"""
    import json
    out = _bench(["--gpus", "1", "--init-dist", "--steps", "2", "--warmup", "1", "--workload", "big", "--batch", "8",
                  "--no-exact", "--no-cpu-baseline"], {})
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["rccl"]["backend"] == "nccl" and d["rccl"]["allreduce_in_step"]
    assert d["rccl"]["devices"][0]["device"] == "cuda:0" and d["rccl"].get("rccl_version")
    s = d["schedules"]["per_rank"][0]
    assert s["fwd_overlapped"] + s["fwd_plain"] == 6 * 2 and s["bwd_overlapped"] + s["bwd_plain"] == 6 * 2


# ---- turns itself on when the box has more than one GPU: one rank per GPU over RCCL --------------------------
def _multi_gpu_worker(rank, world, port, q):
    """one rank of BASELINE configs[3] in miniature: its own GPU, RCCL, its own utterances (seed + rank), radii cycling over the
    global batch; one train step of the two-block big model"""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import sound_bubble_amd as sb
    from sound_bubble_amd.functional import SnrlpLossFn
    from sound_bubble_amd.train import FlatBucket, FusedAdam, allreduce_grads, broadcast_replica
    rec, params, _ = load_golden("tiny_big")
    Bl, N = 2, 192 * 40 - 37

    def shard(r):
        g = torch.Generator().manual_seed(1234 + r)
        mix = (0.1 * torch.randn(Bl, 6, N, generator=g)).clamp(-1, 1)
        tgt = 0.05 * torch.randn(Bl, 1, N, generator=g)
        if r % 2 == 1:
            tgt[0] = 0.0                                   # silent-target utterances on the odd ranks (the shared negative term)
        dis = torch.eye(3)[(torch.arange(Bl) + r * Bl) % 3]   # radii cycle over the GLOBAL batch
        return mix, tgt, dis

    torch.manual_seed(100 + rank)                          # different initial weights per rank: broadcast_replica must sync them
    m = sb.NetDisEmbd3(**params).to(dev).train()
    bucket = FlatBucket(m)
    optim = FusedAdam(bucket, lr=1e-3)
    broadcast_replica(bucket, optim)
    mix, tgt, dis = (t.to(dev) for t in shard(rank))
    bucket.zero_grad()
    loss, _ = SnrlpLossFn.apply(m({"mixture": mix, "dis_embed": dis})["output"], tgt, 100.0)
    loss.backward()
    w = allreduce_grads(bucket)
    g_dp = (bucket.grad / w).clone()
    optim.step(grad_clip=1.0, world_size=w)
    torch.cuda.synchronize()
    # replicas identical after the step: every rank's parameter bits against rank 0's
    ref = bucket.flat.clone()
    dist.broadcast(ref, 0)
    same = torch.tensor([int(torch.equal(ref, bucket.flat))], device=dev)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if rank == 0:
        # the global-batch gradient of mean_b SNRLP on ONE GPU, same weights (the pre-step parameters were rank 0's own)
        torch.manual_seed(100)
        m1 = sb.NetDisEmbd3(**params).to(dev).train()
        b1 = FlatBucket(m1)
        b1.zero_grad()
        parts = [shard(r) for r in range(world)]
        mixg, tgtg, disg = (torch.cat([p[i] for p in parts]).to(dev) for i in range(3))
        l1, _ = SnrlpLossFn.apply(m1({"mixture": mixg, "dis_embed": disg})["output"], tgtg, 100.0)
        l1.backward()
        torch.cuda.synchronize()
        q.put((w, dist.get_backend(), int(same.item()), g_dp.cpu().numpy(), b1.grad.cpu().numpy(),
               ".".join(str(v) for v in torch.cuda.nccl.version())))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_per_gpu_over_rccl_when_the_box_has_several_gpus():
    """Skipped on a one-GPU box.  With N >= 2 devices visible: N = min(count, 8) ranks, one per GPU, backend "nccl" (= RCCL over
    xGMI), different initial weights per rank synchronised by broadcast_replica, one train step per rank on its own shard
    (seed + rank, radii cycling over the global batch, silent-target utterances on the odd ranks): RCCL reports N ranks, the
    all-reduced flat bucket / N equals the single-GPU GLOBAL-batch gradient of mean_b SNRLP (the assertion of the gloo tests
    above: hl_module:34-35's nn.DataParallel gradient), and every rank holds the same parameter bits after the fused Adam."""
    import torch
    import torch.multiprocessing as mp
    n = min(torch.cuda.device_count(), 8)
    if n < 2:
        pytest.skip("one GPU visible: the N-rank RCCL step needs >= 2 (the 2-ranks-on-one-GPU gloo tests above cover the logic)")
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_multi_gpu_worker, args=(r, n, port, q)) for r in range(n)]
    for p in procs:
        p.start()
    w, backend, same, g_dp, g_ref, ver = q.get()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    err = rel_l2(g_dp, g_ref)
    print(f"RCCL {ver}: {w} ranks on {n} GPUs; all-reduced bucket vs global-batch gradient rel-L2 {err:.2e}; replicas identical: {bool(same)}")
    assert w == n and backend == "nccl" and same == 1
    assert err < 5e-5, err          # (summation order over N shards: the 2-rank gloo form above measures 1e-5)
