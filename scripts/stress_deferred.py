"""Stress of the deferred small launches (round 4): full-size big-config backward passes into a FlatBucket, K inputs in turn, the
bucket read on the main stream straight after backward() with no synchronisation; every iteration's gradient is held to the
first visit of its input (1e-4: atomics in the weight-gradient sums).  A missing join, a partial buffer freed too early or a
parked launch that never ran shows as a mismatch / NaN.  usage: python scripts/stress_deferred.py [iterations] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sound_bubble_amd as sb
from sound_bubble_amd import ops
from sound_bubble_amd.functional import SnrlpLossFn
from sound_bubble_amd.train import FlatBucket

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
N = 120000
cls, params = bench.WORKLOADS["big"][0], bench.WORKLOADS["big"][1]
torch.manual_seed(7)
m = getattr(sb, cls)(**params).cuda().train()
bucket = FlatBucket(m)
K = 3
inps, tgts = [], []
for k in range(K):
    g = torch.Generator().manual_seed(11 + k)
    inps.append({"mixture": (torch.randn(B, 6, N, generator=g) * 0.1).cuda(), "dis_embed": torch.eye(3)[(torch.arange(B) + k) % 3].cuda()})
    tgts.append((torch.randn(B, 1, N, generator=g) * 0.1).cuda())
rel = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
refs, bad, joins = [None] * K, [], 0
real_join = ops.deferred_join
def counting():
    global joins
    joins += 1 if ops._DEFER["armed"] else 0
    real_join()
ops.deferred_join = counting
for it in range(n_it):
    k = it % K
    bucket.zero_grad()
    ops.absmax_hints_clear()
    est = m(inps[k])["output"]
    loss, _ = SnrlpLossFn.apply(est, tgts[k], 100.0)
    loss.backward()
    gv = bucket.grad.clone()                      # main stream, no synchronisation
    junk = torch.full((48 * 1024 * 1024,), float("nan"), device="cuda")      # churn: whatever was freed gets poisoned
    del junk
    if refs[k] is None:
        refs[k] = gv
        continue
    e = rel(gv, refs[k])
    if not (e < 1e-4) or not bool(torch.isfinite(gv).all()):
        bad.append((it, e))
torch.cuda.synchronize()
print(f"deferred={ops.DEFER_REDUCE} overlap_available={ops.overlap_available()} iterations={n_it} B={B} joins_with_work={joins} "
      f"mismatches={len(bad)} first={bad[:5]} sched_status={ops.read_sched_status()} schedules={ops.SCHED_COUNTS}")
assert not bad
