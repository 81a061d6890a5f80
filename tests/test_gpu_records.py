"""Wide BPTT gate records as 24-bit fixed point (sb_lstm_bf_common.h, include/sound_bubble_hip.h: rec_f32): the packing the
forward recurrence stores and the backward recurrence loads, held bit for bit to a NumPy restatement of the format, and the
error bound the header states (half a grid step: 2^-25 for the sigmoid gates, 2^-24 for the tanh gate)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _roundtrip(x):
    from sound_bubble_amd import _lib as L
    xin = torch.from_numpy(x).cuda()
    out = torch.empty_like(xin)
    packed = torch.empty(x.size // 16 * 12, dtype=torch.int32, device="cuda")
    L.check(L.load().sb_rec_q24_roundtrip(C.c_void_p(xin.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(packed.data_ptr()),
                                          x.size, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sb_rec_q24_roundtrip")
    torch.cuda.synchronize()
    return out.cpu().numpy(), packed.cpu().numpy().view(np.uint32)


def _ref(x):
    """groups of 16 = gates i, f, g, o x 4 units -> (decoded fp32 values, the 24-bit codes)"""
    g = x.reshape(-1, 4, 4).astype(np.float64)
    q = np.empty(g.shape, np.int64)
    for k in (0, 1, 3):                              # [0, 1]: round-to-nearest-even of x 2^24 (exact in fp32), saturated at 2^24 - 1
        q[:, k] = np.minimum(np.rint(g[:, k] * 2.0 ** 24).astype(np.int64), 2 ** 24 - 1)
    q[:, 2] = np.minimum(np.rint(g[:, 2] * 2.0 ** 23).astype(np.int64), 2 ** 23 - 1)           # [-1, 1]: two's complement
    dec = q.astype(np.float64)
    dec[:, (0, 1, 3)] *= 2.0 ** -24
    dec[:, 2] *= 2.0 ** -23
    return dec.astype(np.float32).reshape(x.shape), (q & 0xFFFFFF)


def test_q24_records_match_the_documented_format_bit_for_bit():
    if __import__("sound_bubble_amd.ops", fromlist=["x"]).wide_rec_dwords() != 192:
        pytest.skip("fp32-record build (-DSB_REC_Q24=0)")
    rng = np.random.default_rng(5)
    n = 16 * 4096
    x = rng.random(n, dtype=np.float32).reshape(-1, 4, 4)
    x[:, 2] = x[:, 2] * 2 - 1
    # saturated activations are the common case in a trained LSTM: crowd the ends
    x[:512] = np.where(rng.random((512, 4, 4)) < 0.5, 1.0 - rng.random((512, 4, 4)).astype(np.float32) ** 8 * 1e-3, x[:512])
    x[512:1024, (0, 1, 3)] = (rng.random((512, 3, 4)).astype(np.float32) ** 8 * 1e-3)
    x[1024:1536, 2] = -1.0 + rng.random((512, 4)).astype(np.float32) ** 8 * 1e-3
    edge = np.array([0.0, 1.0, 0.5, 2.0 ** -25, 2.0 ** -24, 1 - 2.0 ** -24, 1e-9, 0.25], np.float32)
    x[1536, 0], x[1536, 1], x[1536, 3] = edge[:4], edge[4:], edge[:4]
    x[1536, 2] = np.array([-1.0, 1.0, -2.0 ** -24, 1 - 2.0 ** -23], np.float32)
    x = np.ascontiguousarray(x.reshape(-1))
    out, packed = _roundtrip(x)
    want, q = _ref(x)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    # the byte layout: three dwords per gate = [q0.0 q0.1 q0.2 q1.0 | q1.1 q1.2 q2.0 q2.1 | q2.2 q3.0 q3.1 q3.2]
    p = packed.reshape(-1, 4, 3).astype(np.int64)
    e0 = q[:, :, 0] | ((q[:, :, 1] & 0xFF) << 24)
    e1 = (q[:, :, 1] >> 8) | ((q[:, :, 2] & 0xFFFF) << 16)
    e2 = (q[:, :, 2] >> 16) | (q[:, :, 3] << 8)
    assert np.array_equal(p[:, :, 0], e0) and np.array_equal(p[:, :, 1], e1) and np.array_equal(p[:, :, 2], e2)
    # the error bound: half a step, except where the top of the range saturates one step below it
    g, o = x.reshape(-1, 4, 4).astype(np.float64), out.reshape(-1, 4, 4).astype(np.float64)
    err = np.abs(o - g)
    assert err[:, (0, 1, 3)].max() <= 2.0 ** -24 and (err[:, (0, 1, 3)][g[:, (0, 1, 3)] < 1 - 2.0 ** -24] <= 2.0 ** -25).all()
    assert err[:, 2].max() <= 2.0 ** -23 and (err[:, 2][g[:, 2] < 1 - 2.0 ** -23] <= 2.0 ** -24).all()


def test_flag_arena_hands_out_zeroed_uncached_words_once():
    """The guarded schedules' flag words: uncached device memory from the library, zeroed chunk-wise by
    write-through stores, every slice handed out once; the watchdog word is write-once (a fresh slot after a trip)."""
    from sound_bubble_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    arena = ops.flag_arena(dev)
    assert arena.kind == 3, f"expected uncached flag memory (hipDeviceMallocUncached), got kind {arena.kind}"
    a, b = ops.zeroed_flags(100, dev), ops.flag_words(37, dev)
    assert a is not None and a.numel() == 128 and b.numel() == 64 and a.data_ptr() % 256 == 0
    assert b.data_ptr() >= a.data_ptr() + 4 * a.numel() or b.data_ptr() + 4 * b.numel() <= a.data_ptr()      # disjoint
    assert int(a.cpu().abs().sum()) == 0 and int(b.cpu().abs().sum()) == 0
    lo, hi = arena.base + 4 * arena.RESERVED, arena.base + 4 * (arena.RESERVED + arena.CHUNK * arena.NCHUNK)
    assert lo <= a.data_ptr() < hi and lo <= b.data_ptr() < hi
    w0 = ops.sched_status(dev)
    assert w0.item() == 0
    w1 = arena.new_status()
    assert w1.data_ptr() != w0.data_ptr() and w1.item() == 0
    ops._SCHED_STATUS[dev.index] = w1
