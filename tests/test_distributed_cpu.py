"""N>1 data-parallel path on CPU: 2 processes, gloo backend (RCCL stands in on the GPU box).
Checks that one all-reduce of the flat gradient bucket + 1/world scaling reproduces the single-process
global-batch gradient of mean_b(SNRLP) -- including a silent-target sample in one shard only (the shared
negative term is shard-invariant)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, golden_state_dict


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from oracle.tfgridnet_oracle import OracleNet, snrlp_loss
    from sound_bubble_amd.train import FlatBucket, allreduce_grads
    rec, params, flavour = load_golden("tiny_small")
    m = OracleNet(flavour, **params).train()
    m.load_state_dict(golden_state_dict(rec, torch))
    bucket = FlatBucket(m)
    mix = torch.from_numpy(rec["mixture"])
    tgt = torch.from_numpy(rec["target"])            # sample 1 has an all-zero target
    sl = slice(rank, rank + 1)                        # shard by utterance
    bucket.zero_grad()
    est = m({"mixture": mix[sl]})["output"]
    snrlp_loss(est, tgt[sl], 100.0).mean().backward()
    w = allreduce_grads(bucket)
    assert w == world
    g = bucket.grad / w
    if rank == 0:
        q.put(g.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_matches_global_batch_gradient():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    g_dp = q.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    # single-process reference: the whole batch at once
    from oracle.tfgridnet_oracle import OracleNet, snrlp_loss
    from sound_bubble_amd.train import FlatBucket
    rec, params, flavour = load_golden("tiny_small")
    m = OracleNet(flavour, **params).train()
    m.load_state_dict(golden_state_dict(rec, torch))
    bucket = FlatBucket(m)
    est = m({"mixture": torch.from_numpy(rec["mixture"])})["output"]
    snrlp_loss(est, torch.from_numpy(rec["target"]), 100.0).mean().backward()
    g_ref = bucket.grad.numpy()
    err = np.linalg.norm(g_dp - g_ref) / np.linalg.norm(g_ref)
    assert err < 1e-5, err


def test_flat_bucket_views_alias_parameters():
    from sound_bubble_amd.train import FlatBucket
    lin = torch.nn.Linear(5, 3)
    w0 = lin.weight.detach().clone()
    b = FlatBucket(lin)
    assert torch.equal(lin.weight, w0) and b.numel % 4 == 0
    assert all(o % 4 == 0 for o in b.offsets)                       # 16-byte aligned parameters
    lin(torch.ones(2, 5)).sum().backward()
    assert torch.equal(b.grad[: 15].view(3, 5), lin.weight.grad)    # autograd accumulates into the bucket
    b.flat.mul_(2.0)
    assert torch.equal(lin.weight, 2 * w0)                          # parameters are views of the bucket
    b.zero_grad()
    assert float(lin.weight.grad.abs().sum()) == 0.0
