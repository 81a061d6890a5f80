"""Tensor-level wrappers over the C ABI (include/sound_bubble_hip.h).

PyTorch is plumbing here: it owns device memory and the stream.  Every wrapper
checks that its tensors are fp32, contiguous and on a HIP device and raises
otherwise -- there is no eager fallback.
"""
import ctypes as C
import os

import torch

from . import _lib as L

H = 64
# Precision of the BPTT state (what the forward pass keeps for the backward pass, and what travels between the backward
# kernels):
#   "wide"    (default) -- fp32-class: fp32 c_prev + 24-bit fixed-point gate records (round 5; blocked in the kernels' lane order),
#             fp32 LayerNorm-output and hs side outputs, gradients on the fp16 matrix pipe as two terms (hi + 2^-11 lo',
#             22 mantissa bits) against hi + lo splits of activations and weights -- through the SAME fused launch
#             structure as the compact mode (lstm_bwd_rec_bf_kernel<..., XP>).
#   "compact" (SB_BPTT=compact, opt-in) -- fp16 gate / c_prev records, fp16 side outputs, single-term scaled fp16 dgates:
#             half the record bytes, a third of the gradient MFMAs, overlapped inter-frame backward; gradient error against
#             the reference up to 1.3e-3 (tiny goldens) / 4e-4 (full size) instead of <= 2e-4 / 2e-5.
#   "legacy"  (SB_BPTT=legacy or SB_EXACT_BPTT=1) -- round-1 form of the wide arithmetic: position-major fp32 records,
#             unfused kernels, fp32-input MFMA in the streaming part.  Kept as the fallback of "wide" for the layer shapes
#             the fused kernels do not cover, and as an A/B yardstick.
BPTT = os.environ.get("SB_BPTT", "legacy" if os.environ.get("SB_EXACT_BPTT", "0") == "1" else "wide")
assert BPTT in ("wide", "compact", "legacy"), BPTT


class bptt_mode:
    """context manager: run a block under another BPTT-state precision (the backward of an autograd node runs under the
    mode its forward chose; bench.py / the tests switch modes)"""

    def __init__(self, mode):
        assert mode in ("wide", "compact", "legacy"), mode
        self.mode = mode

    def __enter__(self):
        global BPTT
        self.old, BPTT = BPTT, self.mode
        return self

    def __exit__(self, *exc):
        global BPTT
        BPTT = self.old
        return False


def _compact():
    return BPTT == "compact"


def _wide():
    return BPTT == "wide"


_WIDE_REC_DW = None


def wide_rec_dwords():
    """dwords per (sequence, step, direction) of the wide gate records: 192 = the four gates as 24-bit fixed point (round 5), 256 =
    fp32 (a -DSB_REC_Q24=0 build); the library says which"""
    global _WIDE_REC_DW
    if _WIDE_REC_DW is None:
        _WIDE_REC_DW = int(L.load().sb_lstm_wide_rec_dwords())
    return _WIDE_REC_DW


def wide_rec_bytes():
    return 4.0 * (wide_rec_dwords() + H)          # gates + fp32 c_prev, per position and direction


def wide_supported(kind, Cc):
    """layer shapes the wide fused backward kernels cover: kind 'inter' (single direction, fused Linear, C in {16, 32}),
    'intra-plain' (bidirectional with the fused Linear backward, C == 32), 'intra-conv' (bidirectional, gradient of hs
    given, C == 16).  Anything else runs in "legacy" form under the wide mode."""
    if LSTM_MMA != 1 or not FUSED_BPTT:
        return False
    if kind == "inter":
        return Cc in (16, 32)
    if kind == "intra-plain":
        # the wide bidirectional backward takes hs as the forward's (hi, lo) pairs (or recomputes it): it needs the Linear
        # applied inside the forward recurrence (found by the switch matrix: SB_NO_INTRA_LIN_FUSION=1 used to reach the
        # two-kernel backward with pair-form u)
        return Cc == 32 and FUSED_BPTT_BI and INTRA_LIN_FUSION
    if kind == "intra-conv":
        return Cc == 16 and FUSED_BPTT_BI
    return False


def layer_mode(kind, Cc):
    """the BPTT-state precision a layer's forward should record under"""
    if BPTT == "wide" and not wide_supported(kind, Cc):
        return "legacy"
    return BPTT
# recurrent GEMMs: bf16 matrix pipe with exact 3-way split / 6 products (fp32-class) unless SB_LSTM_FP32=1
# forward operand split: fp16 hi+lo, 3 products (default, 2^-22) or SB_LSTM_BF16X6=1: bf16 3-way, 6 products (2^-24)
LSTM_MMA = 0 if os.environ.get("SB_LSTM_FP32", "0") == "1" else (2 if os.environ.get("SB_LSTM_BF16X6", "0") == "1" else 1)
# SB_LSTM_PRODUCTS=2 (opt-in, inference forward only): the recurrent products with ONE fp16 activation term against hi + lo weights
# -- two products per MAC instead of three, 11-bit activations: NOT fp32-class.  bench.py reports its speed and its distance
# from the default arithmetic side by side (secondary.forward_*_2prod); nothing else switches it on.
LSTM_PRODUCTS = int(os.environ.get("SB_LSTM_PRODUCTS", "3"))
# bench.py: dict  kernel label -> list of (start_event, end_event, algorithmic_flops, compulsory_bytes, design_bytes)
# per launch of the recurrent kernels, HIP events on the launch stream.  compulsory = SURVEY.md 8(d): 4C in + 4C out per
# position of a fused pass; design = what this implementation must move (BPTT records, side outputs, dgates).
PROFILE = None


class _Prof:
    """HIP-event pair on the launch stream around one library call (bench.py's untimed profiling pass only).  side=True: the
    call also places a kernel on the library's side stream (the overlapped consumers: one launch beside the producer, one
    behind it on the caller's stream) -- the caller's stream never sees that launch, so a second, caller-owned event pair is
    handed to the library, which records it around the launch ON the side stream (sb_overlap_time_next_side_launch).  The
    label then collects TWO entries per call, each with half of the call's algorithmic work, the side one marked: their
    average is what `rocprofv3 --kernel-trace --stats` reports as the kernel's average duration."""

    def __init__(self, label, flops, cbytes, dbytes, side=False):
        self.rec = PROFILE.setdefault(label, []) if PROFILE is not None else None
        self.vals = (flops, cbytes, dbytes)
        self.side = side

    def __enter__(self):
        if self.rec is not None:
            if self.side:
                self.s0, self.s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.s0.record()           # (torch creates the hipEvent_t on first use; the library re-records both)
                self.s1.record()
                L.load().sb_overlap_time_next_side_launch(C.c_void_p(self.s0.cuda_event), C.c_void_p(self.s1.cuda_event))
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            fired = self.side and L.load().sb_overlap_time_next_side_launch(None, None) == 0    # 1: still armed, nothing timed
            if exc[0] is None:
                self.e1.record()
                if fired:
                    half = tuple(0.5 * v for v in self.vals)
                    self.rec.append((self.e0, self.e1) + half + ("main",))
                    self.rec.append((self.s0, self.s1) + half + ("side",))
                else:
                    self.rec.append((self.e0, self.e1) + self.vals + ("main",))
        return False


_STREAM_OVERRIDE = None


def _stream():
    if _STREAM_OVERRIDE is not None:
        return _STREAM_OVERRIDE
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class on_stream:
    """library launches inside the block go to `stream` (a handle from deferred_side(); None: the current stream as usual).
    Only for launches whose every buffer the caller keeps alive until the join (defer_small_launches(keep))."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        global _STREAM_OVERRIDE
        self.prev, _STREAM_OVERRIDE = _STREAM_OVERRIDE, self.stream
        return self

    def __exit__(self, *exc):
        global _STREAM_OVERRIDE
        _STREAM_OVERRIDE = self.prev
        return False


def _p(t, name="tensor"):
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise L.SoundBubbleHipError(f"{name}: expected a float32 tensor on the GPU, got "
                                    f"{type(t).__name__} {getattr(t, 'dtype', '')} {getattr(t, 'device', '')}")
    if not t.is_contiguous():
        raise L.SoundBubbleHipError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _ph(t, name="tensor"):
    """pointer of a contiguous GPU tensor that may be fp32 or fp16 (the fp16 side outputs of lstm_fwd)"""
    if t is None:
        return None
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in (torch.float32, torch.float16) and t.is_contiguous()):
        raise L.SoundBubbleHipError(f"{name}: expected a contiguous float32 / float16 tensor on the GPU")
    return C.c_void_p(t.data_ptr())


def _poff(t, off_floats):
    """pointer to element `off_floats` of contiguous fp32 tensor t"""
    _p(t)
    return C.c_void_p(t.data_ptr() + 4 * int(off_floats))


class Geom:
    """Sequence geometry of one recurrent pass over the dense position grid."""

    def __init__(self, nseq, nsteps, n_inner, p_outer, p_inner, p_step):
        self.nseq, self.nsteps, self.n_inner = int(nseq), int(nsteps), int(n_inner)
        self.p_outer, self.p_inner, self.p_step = int(p_outer), int(p_inner), int(p_step)
        self.P = self.nseq * self.nsteps

    @staticmethod
    def intra(n_rows, n_steps):     # sequences = rows (b,t), steps along the last grid axis
        return Geom(n_rows, n_steps, n_rows, 0, n_steps, 1)

    @staticmethod
    def inter(B, T, F):             # sequences = (b,f), steps along t
        return Geom(B * F, T, F, T * F, 1, F)


# ---- time-segmented scheduling of single-direction passes (include/sound_bubble_hip.h: seg_state / sched_status) ----
# SCHED_OVERRIDE = (workers, segments): force the schedule with that many resident workgroups / segments (API fields
# sched_workers / sched_segments; the parity tests force it on small problems, a CU-masked deployment would lower
# `workers`).  None = automatic.
SCHED_OVERRIDE = None
_SCHED_STATUS = {}


class FlagWords:
    """n int32 words of UNCACHED device memory (a slice of the per-device flag arena; include/sound_bubble_hip.h, "flag memory of
    the guarded schedules"): stands in for the int32 tensor the flags used to be -- data_ptr / numel / cpu / item / zero_."""
    __slots__ = ("ptr", "n", "dev")

    def __init__(self, ptr, n, dev):
        self.ptr, self.n, self.dev = ptr, n, dev

    def data_ptr(self):
        return self.ptr

    def numel(self):
        return self.n

    def cpu(self):
        """synchronises the current stream; -> int32 tensor on the host"""
        out = (C.c_int * self.n)()
        with torch.cuda.device(self.dev):
            L.check(L.load().sb_flags_read(C.c_void_p(self.ptr), self.n, out, _stream()), "sb_flags_read")
        return torch.tensor(list(out), dtype=torch.int32)

    def item(self):
        assert self.n == 1
        return int(self.cpu()[0])

    def zero_(self):
        with torch.cuda.device(self.dev):
            L.check(L.load().sb_flags_zero(C.c_void_p(self.ptr), self.n, _stream()), "sb_flags_zero")
        return self


class _FlagArena:
    """One uncached allocation per device, cut into CHUNKS that are handed out round-robin; a chunk is zeroed (write-through
    stores, on the stream that activates it) when it becomes current, its words are handed out once, and by the time the ring
    comes back to it -- NCHUNK chunks = hundreds of train steps later -- the device is synchronised once, so no kernel can still
    be polling the words.  The first RESERVED ints are 64 slots of 64 for the watchdog word, which is WRITE-ONCE: after a trip
    the next slot becomes the word (new_status), the old one is never cleared in place -- in round 5 a word the host had zeroed
    was seen to come back, two epochs later, with the code it had held."""
    CHUNK = 1 << 15           # ints (128 KB): ~4 train steps of the big model's flags
    NCHUNK = 64
    RESERVED = 64 * 64

    def __init__(self, dev_index):
        self.dev = dev_index
        ptr, kind = C.c_void_p(), C.c_int(-1)
        with torch.cuda.device(dev_index):
            L.check(L.load().sb_flags_alloc(4 * (self.RESERVED + self.CHUNK * self.NCHUNK), C.byref(ptr), C.byref(kind)), "sb_flags_alloc")
            self.base, self.kind = ptr.value, kind.value
            torch.cuda.synchronize(dev_index)
            L.check(L.load().sb_flags_zero(C.c_void_p(self.base), self.RESERVED, _stream()), "sb_flags_zero")
            torch.cuda.synchronize(dev_index)
        if self.kind != 3:
            import warnings
            warnings.warn(f"cuda:{dev_index}: no uncached device memory for the schedule flags (got kind {self.kind}); the "
                          "overlapped schedules run on cached flag words, which were seen to go stale once in a few thousand steps")
        self.next_chunk, self.gen = 0, [0] * self.NCHUNK
        self.status_slot = 0
        self.status = FlagWords(self.base, 1, dev_index)

    def new_status(self):
        """the watchdog word moves to the next slot (zeroed, device synchronised: rare path, after a trip)"""
        self.status_slot = (self.status_slot + 1) % 63
        self.status = FlagWords(self.base + 4 * 64 * self.status_slot, 1, self.dev)
        with torch.cuda.device(self.dev):
            torch.cuda.synchronize(self.dev)
            L.check(L.load().sb_flags_zero(C.c_void_p(self.status.ptr), 64, _stream()), "sb_flags_zero")
            torch.cuda.synchronize(self.dev)
        return self.status

    def status_debug(self):
        """the 8 ints at the watchdog word (a -DSB_TRIP_DEBUG library fills [1..3]: polls, the word as read, XCC << 16 | workgroup)"""
        return FlagWords(self.status.ptr, 56, self.dev).cpu().tolist()

    def new_chunk(self):
        """-> (chunk index, byte address, generation) of a fresh chunk, zeroed on the current stream"""
        k = self.next_chunk
        self.next_chunk = (k + 1) % self.NCHUNK
        if k == 0 and self.gen[0] > 0 and not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize(self.dev)                 # (once per NCHUNK chunks: every word of the ring is dead)
            # ... and every (device, stream) pool lets go of the chunk it was cutting slices from: another stream's pool could
            # otherwise keep a chunk k != 0 across this wrap and go on handing out words that this ring is about to re-zero
            # under kernels still polling them (two streams running guarded schedules on one device)
            for key in [key for key in _FLAGS.cur if key[0] == self.dev]:
                del _FLAGS.cur[key]
        self.gen[k] += 1
        addr = self.base + 4 * (self.RESERVED + k * self.CHUNK)
        with torch.cuda.device(self.dev):
            L.check(L.load().sb_flags_zero(C.c_void_p(addr), self.CHUNK, _stream()), "sb_flags_zero")
        return k, addr, self.gen[k]


_FLAG_ARENAS = {}


def flag_arena(dev):
    i = dev.index if isinstance(dev, torch.device) and dev.index is not None else (dev if isinstance(dev, int) else torch.cuda.current_device())
    a = _FLAG_ARENAS.get(i)
    if a is None:
        a = _FLAG_ARENAS[i] = _FlagArena(i)
    return a


def sched_status(dev):
    """the per-device watchdog word of the guarded schedules (one int32 of the flag arena, zeroed once)"""
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _SCHED_STATUS.get(i)
    if t is None:
        t = _SCHED_STATUS[i] = flag_arena(i).status
        # the word is read by kernels on the library's SIDE stream too, which is ordered after an event recorded BEFORE this
        # fill was enqueued when the first use is an overlapped consumer (SB_NO_TIME_SEGMENTS=1: no producer-side scratch
        # creates it earlier): the consumer then saw whatever the freshly allocated word held, took it for a tripped watchdog,
        # dropped its items -- and the fill cleared the evidence (found by the switch matrix).  Once per device:
        torch.cuda.synchronize(dev)
    return t


TRIP_SITES = {1: "time-segmented forward: a tile's previous segment never published its state",
              2: "overlapped forward, intra-frame consumer: the producer's time slab never completed",
              3: "time-segmented backward recurrence: a tile's previous segment never published its state",
              4: "overlapped inter-frame backward, stream kernel: the recurrence's dgates slab never completed",
              5: "cross-pass backward, consumer: a producer tile never reached the slab count waited for",
              6: "cross-pass backward, consumer: the owner of a tile's prologue rows never raised `done`"}
LAST_TRIPS = []           # decoded watchdog words read since start-up: (device, dict)


def decode_trip(word):
    """the watchdog word a bounded wait left (include/sound_bubble_hip.h, SB_TRIP_*) -> dict"""
    word = int(word) & 0xFFFFFFFF
    site = word >> 28
    return {"word": word, "site": site, "what": TRIP_SITES.get(site, "unknown site (a pre-round-5 library writes 1)"),
            "timed_out": bool((word >> 27) & 1),       # False: the waiter left because it found the word already set
            "index": (word >> 14) & 0x1FFF, "seen": (word >> 7) & 0x7F, "wanted": word & 0x7F}


def giveups_word(dev):
    """one int32 of the flag arena: overlapped-forward consumer workgroups that stopped waiting for their producer (harmless to the
    results -- the launch behind the producer does what is left -- but a sign that a producer stood still for seconds)"""
    i = dev.index if isinstance(dev, torch.device) and dev.index is not None else (dev if isinstance(dev, int) else torch.cuda.current_device())
    a = flag_arena(i)
    return FlagWords(a.base + 4 * (a.RESERVED - 64), 1, i)         # (the last watchdog slot is never used as one: new_status cycles 0 .. 62)


def read_giveups(dev=None):
    """synchronises; -> give-ups counted on the device since start-up"""
    return giveups_word(dev if dev is not None else torch.cuda.current_device()).item()


def read_sched_status():
    """synchronises; -> list of device indices whose watchdog word was set (and clears those words; what the words said is
    appended to LAST_TRIPS)"""
    bad = []
    for i, t in list(_SCHED_STATUS.items()):
        v = int(t.item())
        if v != 0:
            d = decode_trip(v)
            d["debug_words"] = flag_arena(i).status_debug()
            _SCHED_STATUS[i] = flag_arena(i).new_status()            # (write-once words: never cleared in place)
            bad.append(i)
            LAST_TRIPS.append((i, d))
    return bad


def _trip_text(devs):
    return "; ".join(f"cuda:{i}: {d['what']} (index {d['index']}, flag {d['seen']} of {d['wanted']} mod 128, {'timed out' if d['timed_out'] else 'found the word set'})"
                     for i, d in LAST_TRIPS[-len(devs):]) if devs else ""


def check_sched_status_all_ranks():
    """check_sched_status for a data-parallel job: the verdict is all-reduced (max) first, so that EVERY rank raises when any
    rank's launch aborted -- a lone raising rank would leave the others hanging in their next collective"""
    import torch.distributed as dist
    bad = read_sched_status()
    flag = 1 if bad else 0
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([flag], dtype=torch.int32,
                         device=torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        flag = int(t.item())
    if flag:
        raise L.SoundBubbleHipError(
            f"a time-segmented / overlapped LSTM launch aborted on {'this rank (cuda:%s)' % bad if bad else 'another rank'} "
            f"[{_trip_text(bad)}] "
            "(its workgroups were not co-resident -- GPU shared or CU-masked?).  Results since the last check are invalid; set "
            "SB_NO_TIME_SEGMENTS=1 / SB_NO_FWD_OVERLAP=1 / SB_NO_BWD_OVERLAP=1 or ops.SCHED_OVERRIDE.")


def check_sched_status():
    """Synchronises and raises if a segmented launch gave up waiting for a co-resident workgroup (its outputs are then
    garbage).  Called after the timed region by bench.py, once per epoch by the harness, and by the tests."""
    for i, t in list(_SCHED_STATUS.items()):
        v = int(t.item())
        if v != 0:
            LAST_TRIPS.append((i, decode_trip(v)))
            _SCHED_STATUS[i] = flag_arena(i).new_status()
            raise L.SoundBubbleHipError(
                f"cuda:{i}: a time-segmented / overlapped LSTM launch aborted [{_trip_text([i])}] (its workgroups were not co-resident -- GPU shared or "
                "CU-masked?).  Results since the last check are invalid; set SB_NO_TIME_SEGMENTS=1 or "
                "ops.SCHED_OVERRIDE = (guaranteed_resident_workgroups, 0).")


def _seg_scratch(a, geom, dev):
    """scratch + watchdog word + overrides of the segmented schedule into an LstmFwdArgs / LstmBwdArgs"""
    ntiles = (geom.nseq + 15) // 16
    scratch = (torch.empty(ntiles * 2 * 16 * H, device=dev, dtype=torch.float32), flag_words(ntiles, dev))      # (flags: zeroed by the call)
    a.seg_state, a.seg_flags = _p(scratch[0]), C.c_void_p(scratch[1].data_ptr())
    a.sched_status = C.c_void_p(sched_status(dev).data_ptr())
    if SCHED_OVERRIDE is not None:
        a.sched_workers, a.sched_segments = int(SCHED_OVERRIDE[0]), int(SCHED_OVERRIDE[1])
    return scratch


# the 3x3 front-end convolution and the output transposed convolution on the fp16 matrix pipe with hi+lo split operands
# (SB_LINEAR_FP32=1: the fp32-input MFMA as everywhere else in sb_linear.hip)
LINEAR_F16X3 = os.environ.get("SB_LINEAR_FP32", "0") != "1"

PHASE_TIMING_BUF = None   # developer hook: scratch for a -DSB_PHASE_TIMING build (scripts/phase_timing.py)


def can_fuse_linear_fwd():
    """the single-direction forward kernel can apply the following Linear + residual itself (fp16 split path)"""
    return LSTM_MMA == 1


# SB_GATE_RECOMPUTE=1 (opt-in): bidirectional C = 32 passes with the fused Linear keep no gate records in the forward and
# the fused backward recomputes the gates on the matrix pipe.  Measured (big, B = 16, same box): forward intra-frame
# recurrence 780 -> 613 us, fused backward 1393 -> 1886 us, train step 539 -> 510 utt/s -- the backward is bound by
# instruction issue and LDS traffic, not by the record bytes, so this is a MEMORY-saving mode (BPTT records of the
# intra-frame passes 640 -> 128 B per position and direction), not a speed-up.  Gradients stay inside the test bars.
GATE_RECOMPUTE = os.environ.get("SB_GATE_RECOMPUTE", "0") == "1"


# Forward with fewer inter-frame tiles than CUs: the NEXT block's intra-frame pass starts on the idle CUs while the
# inter-frame recurrence is still running (sb_lstm_fwd_produce / sb_lstm_fwd_consume).  SB_NO_FWD_OVERLAP=1: one after
# the other.
FWD_OVERLAP = os.environ.get("SB_NO_FWD_OVERLAP", "0") != "1"
# both overlapped schedules: the under-filled pass must have between OVERLAP_MIN_FILL and 3/4 of the CUs' worth of tiles
# (big config, train step: -2 % at 37 tiles, +4 % at 73, +4 % at 145, +3.6 % at 182; the tests set 0 to run tiny grids)
OVERLAP_MIN_FILL = 0.25
OVERLAP_MAX_FILL = float(os.environ.get("SB_OVERLAP_MAX_FILL", "0.75"))
FWD_OVERLAP_SLAB = int(os.environ.get("SB_FWD_OVERLAP_SLAB", "32"))
# ... also in inference (forward-only +9 %: 2160 -> 2353 utterances/s; SB_NO_FWD_OVERLAP_INFERENCE=1: training only)
FWD_OVERLAP_INFERENCE = os.environ.get("SB_NO_FWD_OVERLAP_INFERENCE", "0") != "1"
_TILE_ORDER = {}


def tile_order_np(B, T, slab):
    """intra-frame tiles (16 consecutive frames n = b T + t) sorted by the inter-frame time slab that completes them:
    -> (order [ntiles] int32: a permutation of the tiles, need [ntiles] int32: need[i] = slab that completes tile order[i])"""
    import numpy as np
    n = np.arange((B * T + 15) // 16 * 16).reshape(-1, 16)
    need = np.where(n < B * T, (n % T) // slab, 0).max(axis=1)
    order = np.argsort(need, kind="stable")
    return order.astype(np.int32), need[order].astype(np.int32)


def _tile_order(B, T, slab, dev):
    key = (B, T, slab, dev.index if dev.index is not None else torch.cuda.current_device())
    r = _TILE_ORDER.get(key)
    if r is None:
        order, need = tile_order_np(B, T, slab)
        r = _TILE_ORDER[key] = (torch.from_numpy(order).to(dev), torch.from_numpy(need).to(dev))
    return r


_OVERLAP_OK = {}
_OVERLAP_SCRATCH = {}
# SB_OVERLAP_FORCE=1 (measurement only): the overlapped schedules even when the probe finds no concurrency -- rocprofv3 --pmc
# serialises every dispatch, and the counters should see the SHIPPED kernels (scripts/gpu_pmc.sh)
OVERLAP_FORCE = os.environ.get("SB_OVERLAP_FORCE", "0") == "1"


def _overlap_scratch(dev_index):
    t = _OVERLAP_SCRATCH.get(dev_index)
    if t is None:
        t = _OVERLAP_SCRATCH[dev_index] = torch.zeros(4, device=torch.device("cuda", dev_index), dtype=torch.float32)
    return t


def overlap_available():
    """kernels on the library's side stream really run next to those of the current stream.  First use per (device, stream):
    sb_overlap_init -- a timed probe, which synchronises the stream (a side stream that shares the hardware queue of the main
    stream would serialise the two launches); afterwards the stored verdict (sb_overlap_reprobe refreshes it)."""
    st = _stream()
    dev = torch.cuda.current_device()
    key = (dev, st.value)
    ok = _OVERLAP_OK.get(key)
    if ok is None:
        if torch.cuda.is_current_stream_capturing():
            return False
        tm = (C.c_float * 2)()
        rc = L.load().sb_overlap_init(st, _p(_overlap_scratch(dev)), tm)
        if rc != 1 and OVERLAP_FORCE:        # measurement aid (counter passes under a serialising profiler): see the header
            rc = L.load().sb_overlap_force(st)
            OVERLAP_LOG.append(("force", key, rc, float(tm[0]), float(tm[1])))
        ok = _OVERLAP_OK[key] = rc == 1
        OVERLAP_LOG.append(("init", key, rc, float(tm[0]), float(tm[1])))
        # once per (device, stream): nothing of the probe (its candidate streams, their one-per-CU busy kernels) is left in
        # flight when the first real producer / consumer pair starts
        torch.cuda.synchronize(dev)
    return ok


# which schedule the inter-frame passes took since the last reset (bench.py: `schedules`, all-gathered over the ranks -- a side
# stream lost next to RCCL's kernels shows here as plain-order counts)
SCHED_COUNTS = {"fwd_overlapped": 0, "fwd_plain": 0, "bwd_overlapped": 0, "bwd_plain": 0, "deferred_joins": 0}


def sched_counts_reset():
    for k in SCHED_COUNTS:
        SCHED_COUNTS[k] = 0


OVERLAP_LOG = []          # (event, (device, stream), verdict, back-to-back ms, pair ms) -- what SB_OVERLAP_DEBUG used to print
_OVERLAP_LOST = set()     # (device, stream) whose side stream a re-probe found serialised: re-timed every epoch, may come back


def overlap_reprobe():
    """Re-time the side stream of the current stream (two 0.2 ms launches + a synchronisation): concurrency that was there at
    start-up can be lost later (another process on the GPU, more streams alive), and the overlapped schedules then cost
    15-25 % instead of gaining 4 % -- silently.  Called once per epoch by the harness; -> True while still concurrent.  On
    loss (the best of four measurements: one alone reads a 0.1 ms host hiccup between its two launches as a loss -- that
    happened about once per 3 000 epochs in round 5's stress runs and left the rest of those runs in the plain order) the
    overlapped paths are switched off for this stream and a warning is issued; later calls keep re-timing the pair and switch
    them on again when it is concurrent again."""
    st = _stream()
    dev = torch.cuda.current_device()
    key = (dev, st.value)
    was = _OVERLAP_OK.get(key)
    if not was and key not in _OVERLAP_LOST:        # never probed, or sb_overlap_init found no side stream: nothing to re-time
        return False
    import warnings
    tm = (C.c_float * 2)()
    rc = L.load().sb_overlap_reprobe(st, _p(_overlap_scratch(dev)), tm)
    OVERLAP_LOG.append(("reprobe", key, rc, float(tm[0]), float(tm[1])))
    if rc != 1 and was:
        _OVERLAP_OK[key] = False
        _OVERLAP_LOST.add(key)
        warnings.warn(f"cuda:{dev}: the side stream of the overlapped LSTM schedules no longer runs concurrently with the main "
                      f"stream (back-to-back {tm[0]:.3f} ms, best pair of 4 {tm[1]:.3f} ms): falling back to the plain launch order")
    elif rc == 1 and not was:                        # lost at an earlier call, concurrent again now: take the overlapped paths again
        _OVERLAP_OK[key] = True
        _OVERLAP_LOST.discard(key)
        warnings.warn(f"cuda:{dev}: the side stream runs concurrently with the main stream again (back-to-back {tm[0]:.3f} ms, "
                      f"pair {tm[1]:.3f} ms): back to the overlapped LSTM schedules")
    return rc == 1


def overlap_lost():
    """a data-path call found no side stream (-1009): stop choosing the overlapped paths on this stream"""
    _OVERLAP_OK[(torch.cuda.current_device(), _stream().value)] = False


# Small launches that sit between two blocks' backward kernels -- the partial-row reductions into gradient targets, which nothing
# but the optimiser reads (kernel trace, big step: ~120 us of them and their launch gaps per block boundary, 2.4 % of the step)
# -- ride on the library's side stream behind the kernels that produced their input, and the main stream joins them ONCE, at the
# end of the backward pass (autograd's final callbacks run after the last node, before backward() returns to the caller: every
# reader of a .grad / the flat bucket is behind the join).  SB_NO_DEFERRED_REDUCE=1: everything on the main stream as before.
DEFER_REDUCE = os.environ.get("SB_NO_DEFERRED_REDUCE", "0") != "1"
_DEFER = {"armed": False, "stream": None, "dev": 0, "keep": [], "pending": []}


def deferred_flush():
    """the parked small launches (defer_launch) go to the side stream, behind everything enqueued on the main stream so far"""
    if _DEFER["pending"]:
        fns, _DEFER["pending"] = _DEFER["pending"], []
        side = deferred_side(_DEFER["stream"]) if _DEFER["armed"] else None
        for fn in fns:
            fn(side if side is not None else _DEFER["stream"])


def deferred_join():
    """the main stream waits for the deferred launches (idempotent; also the engine's end-of-backward callback)"""
    deferred_flush()
    if _DEFER["armed"]:
        _DEFER["armed"] = False
        SCHED_COUNTS["deferred_joins"] += 1          # (bench.py `schedules`: backward passes whose small launches rode on the side stream)
        # the library finds the side stream through the CALLING thread's current device: the engine's final callback may run on a
        # thread whose current device is not the model's -- join under the device the launches were deferred on
        import contextlib
        with (torch.cuda.device(_DEFER["dev"]) if torch.cuda.is_available() else contextlib.nullcontext()):
            L.check(L.load().sb_overlap_join(_DEFER["stream"]), "sb_overlap_join")
    _DEFER["keep"].clear()


def defer_small_launches(keep=()):
    """-> True when the caller may put its next small launches on the side stream (deferred_side() / reduce_on_side): only inside
    a backward pass of the autograd engine (the join is queued as its final callback) with a concurrent side stream.  `keep`:
    tensors those launches read, held until the join."""
    if not DEFER_REDUCE or torch.cuda.is_current_stream_capturing():
        return False
    st = _stream()
    if not _OVERLAP_OK.get((torch.cuda.current_device(), st.value)):
        return False
    if _DEFER["armed"] and _DEFER["stream"].value != st.value:
        return False                                    # (one main stream at a time)
    try:
        torch.autograd.Variable._execution_engine.queue_callback(deferred_join)
    except RuntimeError:                                # not inside a backward pass: nobody would run the join
        return False
    _DEFER["armed"], _DEFER["stream"], _DEFER["dev"] = True, st, torch.cuda.current_device()
    _DEFER["keep"].extend(keep)
    return True


def defer_launch(fn, keep=()):
    """park fn(stream) -- a small launch whose input exists on the main stream now -- until the next deferred_flush(): the
    cross-pass consumer flushes behind its own side-stream launch (a launch put on the side stream straight away would sit in
    front of that consumer and delay it), the end-of-backward join flushes what is left.  -> False: not deferrable, launch now"""
    if not defer_small_launches(keep):
        return False
    _DEFER["pending"].append(fn)
    return True


def deferred_side(stream=None):
    """the side stream, made to wait for everything enqueued on the (current) main stream so far (None: not available)"""
    side = C.c_void_p()
    if L.load().sb_overlap_side_fork(stream if stream is not None else _stream(), C.byref(side)) != 0 or not side.value:
        return None
    return side


class FwdOverlap:
    """One inter-frame (block k) -> intra-frame (block k + 1) boundary of the overlapped forward: created by the model,
    handed to lstm_fwd(produce=...) of the former and lstm_fwd(consume=...) of the latter."""

    def __init__(self, B, T, F_, dev):
        self.slab = FWD_OVERLAP_SLAB
        nfl = int(L.load().sb_lstm_fwd_flag_ints(T, self.slab))                  # control words + the consumer's hand-back block + one flag per slab
        self.nslabs = (T + self.slab - 1) // self.slab
        self.flags = zeroed_flags(nfl, dev)       # from the once-per-step zeroed pool (None: the library zeroes them itself)
        self.prezeroed = self.flags is not None
        if self.flags is None:
            self.flags = flag_words(nfl, dev)
        self.producer_tiles = (B * F_ + 15) // 16
        self.order, self.need = _tile_order(B, T, self.slab, dev)
        self.keep = []            # everything the producer touches stays allocated until the consumer has been launched
        self.produced = False


def can_overlap_fwd(B, T, F_, Cc, train, dev):
    if not (FWD_OVERLAP and LSTM_MMA == 1 and Cc == 32 and can_fuse_linear_fwd() and INTER_SUM3
            and intra_lin_fusion_ok(train, Cc) and SCHED_OVERRIDE is None):
        return False
    if torch.cuda.is_current_stream_capturing() or not (train or FWD_OVERLAP_INFERENCE):
        return False
    tiles, cus = (B * F_ + 15) // 16, _cu_count(dev)
    return (OVERLAP_MIN_FILL * cus <= tiles <= OVERLAP_MAX_FILL * cus and T >= 4 * FWD_OVERLAP_SLAB and B * T >= 64
            and overlap_available())


def lstm_fwd(x, ln_g, ln_b, dirs, geom, h0=None, c0=None, save=False, want_state=False, lin=None, want_hs=True,
             no_gates=False, x_part=None, x_sum=None, film=None, produce=None, consume=None):
    """x [P, C] pre-LayerNorm.  dirs: list of (w_ih, w_hh, b_ih, b_hh) per direction.
    lin = (lin_w [C, 64], lin_b [C], y [P, C]): fused  y = x + lin_w . hs + lin_b  (single direction,
    can_fuse_linear_fwd()); with want_hs=False hs is then not materialised.
    -> hs [P, ndir*64] (or None), (hN, cN) or None, save_gates or None, save_u or None"""
    lib = L.load()
    ndir, Cc = len(dirs), x.shape[-1]
    assert x.numel() == geom.P * Cc
    dev = x.device
    assert want_hs or lin is not None
    # training on the default path: the tensors only the backward kernels read (LayerNorm output u; hs when the Linear
    # is fused here) are written as fp16 -- the streaming backward takes them as single fp16 terms anyway
    aux16 = bool(save and AUX_FP16 and _compact() and DGATES_FP16 and LSTM_MMA in (1, 2))
    hs16 = aux16 and lin is not None
    wide = bool(save and _wide())
    # wide form: the tensors that only the backward kernels' matrix products read travel as the fp16 (hi, lo) term pairs
    # the forward kernel itself multiplies with -- same bytes as fp32, no split in the backward: u always ([P, 2C] halves),
    # hs when the Linear is applied inside this kernel ([P, ndir * 128] halves); kernel-private layouts (see the header)
    hsp = wide and lin is not None
    hs = (torch.empty(geom.P, ndir * H * (2 if hsp else 1), device=dev,
                      dtype=torch.float16 if (hs16 or hsp) else torch.float32) if want_hs else None)
    gates = cprev = None
    if wide:                  # fp32 records, blocked like the compact ones (see the header: rec_f32)
        assert LSTM_MMA == 1 and (not no_gates or (ndir == 1 and hsp))
        Pr = (geom.nseq + 15) // 16 * 16 * geom.nsteps
        if not no_gates:      # no_gates: c_prev + the u / hs pairs only -- the backward recomputes the gates
            gates = torch.empty(Pr, ndir, wide_rec_dwords(), device=dev, dtype=torch.float32)   # (24-bit fixed-point gates: opaque dwords)
        cprev = torch.empty(Pr, ndir, H, device=dev, dtype=torch.float32)
    elif save and _compact():
        # opaque to the host: on the 16-bit matrix path the records are blocked per (16-sequence tile, step, direction)
        # in the kernels' lane order (include/sound_bubble_hip.h), hence the rows padded to whole tiles
        Pr = (geom.nseq + 15) // 16 * 16 * geom.nsteps if LSTM_MMA else geom.P
        if no_gates:          # records without gates (the backward recomputes them): fp16 side outputs required
            assert aux16 and lin is not None and ndir == 2
        else:
            gates = torch.empty(Pr, ndir, 4 * H, device=dev, dtype=torch.float16)
        cprev = torch.empty(Pr, ndir, H, device=dev, dtype=torch.float16 if LSTM_MMA else torch.float32)
    elif save:
        gates = torch.empty(geom.P, ndir, 5, H, device=dev, dtype=torch.float32)
    u = (torch.empty(geom.P, Cc * (2 if wide else 1), device=dev, dtype=torch.float16 if (aux16 or wide) else torch.float32)
         if save else None)
    hN = torch.empty(geom.nseq, H, device=dev, dtype=torch.float32) if want_state else None
    cN = torch.empty(geom.nseq, H, device=dev, dtype=torch.float32) if want_state else None
    a = L.LstmFwdArgs()
    a.nseq, a.nsteps, a.n_inner, a.ndir, a.C = geom.nseq, geom.nsteps, geom.n_inner, ndir, Cc
    a.p_outer, a.p_inner, a.p_step = geom.p_outer, geom.p_inner, geom.p_step
    a.x, a.ln_g, a.ln_b = _p(x, "x"), _p(ln_g, "ln_g"), _p(ln_b, "ln_b")
    for d, (wi, wh, bi, bh) in enumerate(dirs):
        assert wi.shape == (4 * H, Cc) and wh.shape == (4 * H, H)
        a.w_ih[d], a.w_hh[d], a.b_ih[d], a.b_hh[d] = _p(wi), _p(wh), _p(bi), _p(bh)
    a.h0, a.c0, a.hN, a.cN = _p(h0), _p(c0), _p(hN), _p(cN)
    a.hs, a.save_u = _ph(hs), _ph(u if u is not None else PHASE_TIMING_BUF)
    a.aux_f16 = 1 if aux16 else 0
    a.rec_f32 = 1 if wide else 0
    a.no_vec = 0 if VEC_LSTM else 1
    # opt-in reduced-product forward (inference calls only; never the default, never a training forward: see the header)
    a.products = 2 if (LSTM_PRODUCTS == 2 and not save and LSTM_MMA == 1) else 0
    a.save_c = C.c_void_p(cprev.data_ptr()) if cprev is not None else None
    a.mma = LSTM_MMA
    if lin is not None:       # ndir == 2: partial mode, y is [P, 2, C] (see the header)
        assert can_fuse_linear_fwd() and lin[0].shape == (Cc, ndir * H) and lin[2].numel() == geom.P * ndir * Cc
        a.lin_w, a.lin_b, a.y = _p(lin[0]), _p(lin[1]), _p(lin[2])
    a.save_gates = C.c_void_p(gates.data_ptr()) if gates is not None else None
    if x_part is not None:    # summed-input mode: x + x_part[:, 0] + x_part[:, 1] formed by the loader (see the header)
        assert ndir == 1 and lin is not None and Cc == 32 and x_part.numel() == geom.P * 2 * Cc
        a.x_part, a.x_sum = _p(x_part), _p(x_sum)
    if film is not None:      # (film_w [nseq, C], film_b [nseq, C], y_pre [P, C] or None): FiLM applied to y in the kernel
        assert ndir == 1 and lin is not None and film[0].numel() == geom.nseq * Cc
        a.film_w, a.film_b, a.y_pre = _p(film[0]), _p(film[1]), _p(film[2])
    seg_scratch = None
    if ndir == 1 and LSTM_MMA == 1 and TIME_SEGMENTS:       # scratch for time-segmented scheduling (used when it pays)
        seg_scratch = _seg_scratch(a, geom, dev)
    # design bytes of this launch
    by = 4.0 * Cc * geom.P                                               # x rows (both directions share them)
    by += hs.numel() * hs.element_size() if hs is not None else 0.0             # hidden sequence out
    if gates is not None or cprev is not None:                           # BPTT records + saved LayerNorm output
        by += geom.P * ndir * ((gates.element_size() * gates[0, 0].numel() if gates is not None else 0)
                               + (cprev.element_size() * H if cprev is not None else 0)) + u.numel() * u.element_size()
    if lin is not None:
        by += 2 * 4.0 * Cc * geom.P                                      # residual rows in, y out
    if x_part is not None:
        by += (2 + (1 if x_sum is not None else 0)) * 4.0 * Cc * geom.P - 4.0 * Cc * geom.P   # two halves in, sum out; no second read of x
    label = f"lstm_fwd_bf_kernel C={Cc} " + ("intra-frame (bidirectional)" if ndir == 2 else "inter-frame (Linear fused)"
                                             if lin is not None else "inter-frame")
    if consume is not None and not consume.produced:
        consume = None
    if consume is not None:
        a.sched_status = C.c_void_p(sched_status(dev).data_ptr())
        a.ord_giveups = C.c_void_p(giveups_word(dev).data_ptr())
    with _Prof(label + (" [producer]" if produce is not None else " [consumer, overlapped]" if consume is not None else ""),
               2.0 * 4 * H * (Cc + H) * geom.P * ndir + (2.0 * H * Cc * geom.P if lin is not None else 0.0),
               8.0 * Cc * geom.P, by, side=consume is not None):
        if produce is not None:
            assert ndir == 1 and lin is not None
            rc = lib.sb_lstm_fwd_produce_ex(C.byref(a), C.c_void_p(produce.flags.data_ptr()), produce.slab,
                                            1 if produce.prezeroed else 0, _stream())
            if rc == -1009:                # no side stream (any more): the plain call; the consumer then runs in plain order too
                overlap_lost()
                L.check(lib.sb_lstm_fwd(C.byref(a), _stream()), "sb_lstm_fwd")
                SCHED_COUNTS["fwd_plain"] += 1
            else:
                L.check(rc, "sb_lstm_fwd_produce")
                SCHED_COUNTS["fwd_overlapped"] += 1
                produce.keep += [x, ln_g, ln_b, h0, c0, hs, gates, cprev, u, hN, cN, x_part, x_sum, seg_scratch, lin, film, dirs]
                produce.produced = True
        elif consume is not None:
            assert ndir == 2 and lin is not None
            if getattr(consume, "staged_test", False):        # tests: the hand-back path staged on one stream (see the header)
                L.check(lib.sb_lstm_fwd_consume_staged_test(C.byref(a), C.c_void_p(consume.flags.data_ptr()), consume.slab,
                                                            consume.producer_tiles, consume.nslabs,
                                                            C.c_void_p(consume.order.data_ptr()),
                                                            C.c_void_p(consume.need.data_ptr()), _stream()),
                        "sb_lstm_fwd_consume_staged_test")
            else:
                L.check(lib.sb_lstm_fwd_consume(C.byref(a), C.c_void_p(consume.flags.data_ptr()), consume.slab,
                                                consume.producer_tiles, C.c_void_p(consume.order.data_ptr()),
                                                C.c_void_p(consume.need.data_ptr()), _stream()),
                        "sb_lstm_fwd_consume")
            consume.keep.clear()
            consume.produced = False
        else:
            L.check(lib.sb_lstm_fwd(C.byref(a), _stream()), "sb_lstm_fwd")
            if ndir == 1 and lin is not None:
                SCHED_COUNTS["fwd_plain"] += 1
    return hs, ((hN, cN) if want_state else None), (gates, cprev), u


def can_fuse_linear_bwd():
    """the recurrent backward can form d(hs) = dy . W_lin on the fly (bf16 split path, compact-BPTT mode)"""
    return (LSTM_MMA in (1, 2) and _compact()) or (LSTM_MMA == 1 and _wide())


# compact-BPTT mode on the bf16 path: dgates travel between the two backward kernels as fp16, scaled by a power of
# two derived from max |incoming gradient| (SB_DGATES_FP32=1 keeps them fp32)
DGATES_FP16 = os.environ.get("SB_DGATES_FP32", "0") != "1"
# ... and the forward pass writes u (and hs of the inter-frame pass) as fp16 for them (SB_AUX_FP32=1: fp32)
AUX_FP16 = os.environ.get("SB_AUX_FP32", "0") != "1"
# single-direction (inter-frame) passes with more tiles than CUs: (tile, time-segment) work items over one resident
# workgroup per CU (SB_NO_TIME_SEGMENTS=1: one workgroup per tile as everywhere else)
TIME_SEGMENTS = os.environ.get("SB_NO_TIME_SEGMENTS", "0") != "1"


class DGates:
    """dgates [P, ndir, 4, 64] (+ the device scalar max|incoming gradient| when they are the scaled fp16 form)"""

    def __init__(self, data, gmax=None):
        self.data, self.gmax = data, gmax


def absmax(x):
    out = torch.empty(1, device=x.device, dtype=torch.float32)
    n = x.numel()
    if n % 4:                       # the kernel reads 16-byte groups: the (rare) ragged tail goes through torch
        flat = x.reshape(-1)
        if n >= 4:
            L.check(L.load().sb_absmax(_p(flat), n - n % 4, _p(out), _stream()), "sb_absmax")
            return torch.maximum(out, flat[n - n % 4:].abs().max().reshape(1))
        return flat.abs().max().reshape(1)
    L.check(L.load().sb_absmax(_p(x), n, _p(out), _stream()), "sb_absmax")
    return out


# max |x| of a gradient tensor that the kernel which PRODUCED it already measured (sb_linear_args.absmax_out): saves the
# separate pass over the tensor in the next backward recurrence.  Keyed by data pointer; the entry keeps the tensor
# alive, so the address cannot be reused by another tensor while the hint exists.  Entries are consumed on use and the
# table is cleared at every train step.
_ABSMAX_HINTS = {}
ABSMAX_HINTS = os.environ.get("SB_NO_ABSMAX_HINTS", "0") != "1"


class _ScalarSlots:
    """Zeroed one-float device scalars (absmax_out targets) cut from one buffer that is re-zeroed with ONE fill per train
    step, instead of a torch.zeros(1) launch per use (40-80 fills per step)."""
    N = 256

    def __init__(self):
        self.buf, self.i = {}, {}

    def get(self, dev):
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        b = self.buf.get(key)
        if b is None or self.i[key] >= self.N or torch.cuda.is_current_stream_capturing():
            b = self.buf[key] = torch.zeros(self.N, device=dev, dtype=torch.float32)
            self.i[key] = 0
        k = self.i[key]
        self.i[key] = k + 1
        return b[k:k + 1]

    def new_step(self):
        """every slot handed out so far is dead (the hints table is cleared with it): start over on fresh zeros"""
        for key, b in self.buf.items():
            if self.i[key]:
                self.buf[key] = torch.zeros_like(b)       # a new buffer: slots still referenced by saved tensors stay valid
                self.i[key] = 0


_SLOTS = _ScalarSlots()


class _FlagPool:
    """Flag arrays of the overlapped schedules (slab counters, item counters, per-tile words), cut from the current chunk of the
    device's UNCACHED flag arena (_FlagArena): a chunk is zeroed with ONE launch when it becomes current -- not a memset in front
    of every block's producer, a launch on the critical path between two blocks -- and a slice is handed out once."""

    def __init__(self):
        self.cur = {}             # (device, stream) -> [chunk address, ints used]

    def get(self, n, dev):
        # per (device, stream): the chunk's zeroing is ordered in front of the slice's users only on the stream that ran it
        i = dev.index if dev.index is not None else torch.cuda.current_device()
        key = (i, torch.cuda.current_stream().cuda_stream)
        n = (n + 63) & ~63
        arena = flag_arena(i)
        assert n <= arena.CHUNK, n
        c = self.cur.get(key)          # [chunk index, address, generation, ints used]
        if c is None or c[3] + n > arena.CHUNK or arena.gen[c[0]] != c[2]:      # (last: another stream's pool has gone round the ring)
            c = self.cur[key] = list(arena.new_chunk()) + [0]
        w = FlagWords(c[1] + 4 * c[3], n, i)
        c[3] += n
        return w

    def new_step(self):
        pass                      # (chunks are consumed linearly: nothing to do per step)


_FLAGS = _FlagPool()


def zeroed_flags(n, dev):
    """n zeroed int32 words of uncached flag memory (zeroed in stream order before any later launch on the current stream), or None
    when the caller should take flag_words() and let the library zero them (SB_NO_DEFERRED_REDUCE=1: the round-3 launch order)"""
    if not DEFER_REDUCE or torch.cuda.is_current_stream_capturing():
        return None
    return _FLAGS.get(n, dev)


def flag_words(n, dev):
    """n int32 words of uncached flag memory for a call that zeroes its flags itself (same chunks: they happen to be zero too).
    Under stream capture: a plain int32 tensor from the graph's own pool, as before round 5 -- activating an arena chunk inside a
    capture would record its zeroing into the graph, and every replay would wipe slices handed out since (the guarded schedules
    are not used under capture; only the never-engaged segment flags of the streaming chunk step come through here)."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(int(n), 1), device=dev, dtype=torch.int32)
    return _FLAGS.get(n, dev)


def zero_scalar(dev):
    return _SLOTS.get(dev)


def absmax_hint_put(t, gmax):
    if ABSMAX_HINTS:
        if len(_ABSMAX_HINTS) > 64:
            _ABSMAX_HINTS.clear()
        _ABSMAX_HINTS[t.data_ptr()] = (t, gmax, t._version)


def absmax_hints_clear():
    _ABSMAX_HINTS.clear()
    _SLOTS.new_step()
    _FLAGS.new_step()


def absmax_or_hint(x):
    ent = _ABSMAX_HINTS.pop(x.data_ptr(), None)
    if ent is not None and ent[0].numel() == x.numel() and ent[0].dtype == x.dtype and ent[0]._version == ent[2]:
        return ent[1]
    return absmax(x)


def lstm_bwd_rec(w_hh_list, gates, dhs, geom, dy=None, w_lin=None, gmax=None):
    """dhs [P, ndir*64] -- or, fused: dhs=None, dy [P, C] and w_lin [C, ndir*64] (see can_fuse_linear_bwd).
    -> DGates"""
    lib = L.load()
    ndir = len(w_hh_list)
    dev = dhs.device if dhs is not None else dy.device
    rec, cprev = gates
    dg16 = DGATES_FP16 and LSTM_MMA in (1, 2) and cprev is not None and cprev.dtype == torch.float16 and _compact()
    gmax = (gmax if gmax is not None else absmax_or_hint(dy if dy is not None else dhs)) if dg16 else None
    dg = torch.empty(geom.P, ndir, 4, H, device=dev, dtype=torch.float16 if dg16 else torch.float32)
    a = L.LstmBwdArgs()
    a.nseq, a.nsteps, a.n_inner, a.ndir = geom.nseq, geom.nsteps, geom.n_inner, ndir
    a.p_outer, a.p_inner, a.p_step = geom.p_outer, geom.p_inner, geom.p_step
    for d, wh in enumerate(w_hh_list):
        a.w_hh[d] = _p(wh)
    a.save_gates = C.c_void_p(rec.data_ptr())
    a.save_c = C.c_void_p(cprev.data_ptr()) if cprev is not None else None
    a.dhs, a.dgates = _p(dhs), C.c_void_p(dg.data_ptr())
    a.gmax = _p(gmax)
    a.mma = LSTM_MMA
    seg_scratch = None
    if ndir == 1 and dg16 and TIME_SEGMENTS:                # scratch for time-segmented scheduling (see lstm_fwd)
        seg_scratch = _seg_scratch(a, geom, dev)
    if dy is not None:
        assert can_fuse_linear_bwd() and w_lin.shape == (dy.shape[-1], ndir * H)
        a.dy, a.w_lin, a.C_lin = _p(dy), _p(w_lin), dy.shape[-1]
    Cl = dy.shape[-1] if dy is not None else 0
    rec_b = (rec.element_size() * rec[0, 0].numel() + (cprev.element_size() * H if cprev is not None else 0))
    by = geom.P * ndir * (rec_b + dg.element_size() * 4 * H) + (4.0 * Cl * geom.P if dy is not None else 4.0 * H * ndir * geom.P)
    with _Prof(f"lstm_bwd_rec_bf_kernel ndir={ndir} (recurrence only, dgates to HBM)",
               (2.0 * 4 * H * H + 2.0 * H * Cl) * geom.P * ndir, 8.0 * max(Cl, 16) * geom.P, by):
        L.check(lib.sb_lstm_bwd_rec(C.byref(a), _stream()), "sb_lstm_bwd_rec")
    if ndir == 1:
        SCHED_COUNTS["bwd_plain"] += 1
    return DGates(dg, gmax)


# Inter-frame backward with fewer tiles than CUs: the streaming part starts on the idle CUs while the recurrence is still
# running (sb_lstm_bwd_inter_overlapped).  SB_NO_BWD_OVERLAP=1: the two launches one after the other.
BWD_OVERLAP = os.environ.get("SB_NO_BWD_OVERLAP", "0") != "1"
BWD_OVERLAP_SLAB = int(os.environ.get("SB_BWD_OVERLAP_SLAB", "32"))


# measurement aid (scripts/gpu_profiles_r3.sh): the two kernels of the overlapped inter-frame backward in plain order, so that a
# serialising profiler (rocprofv3 --pmc) can count the pair's HBM traffic
BWD_PAIR_SERIAL = os.environ.get("SB_BWD_PAIR_SERIAL", "0") == "1"


# Wide gate recomputation for the inter-frame pass (C = 32, the overlapped backward pair): the forward stores c_prev and
# the u / hs pairs but NO gate records (1 KB of the 2 KB it writes per position), and the backward recurrence recomputes the
# gates with the forward kernel's own arithmetic, bit for bit.  OPT-IN (SB_INTER_GATE_RECOMPUTE=1), a bytes / memory saver:
# measured on the big train step (same box) the forward gains 1.0 ms (10.95 -> 9.95 ms: its producer
# 1.10 -> 0.85 ms) and the backward pair loses 1.8 ms (6 x 1.46 -> 1.76 ms: +36 MFMAs and +170 VALU / copy instructions per
# step on the serial chain of the recurrence, which IS the pair's critical path) -- 518 -> 500 utterances/s.
INTER_GATE_RECOMPUTE = os.environ.get("SB_INTER_GATE_RECOMPUTE", "0") == "1"
INTER_GATE_RECOMPUTE_FORCE = False       # tests: wherever the pair can run at all (tiny geometries: in plain order)


def inter_gate_recompute_ok(geom, Cc, dev):
    """forward-time decision: store no gate records for this single-direction pass?  Only where the backward will run as the
    recurrence + stream-kernel pair (overlapped when a side stream is there, in plain order otherwise): wide mode, C = 32,
    fused Linear, and a geometry the pair is chosen for (or merely CAN run on, when forced)."""
    if not ((INTER_GATE_RECOMPUTE or INTER_GATE_RECOMPUTE_FORCE) and _wide() and LSTM_MMA == 1 and Cc == 32 and can_fuse_linear_fwd() and FUSED_BPTT
            and BWD_OVERLAP and STREAM_LIN_WGRAD and FUSED_LN_BWD and SCHED_OVERRIDE is None):
        return False
    ntiles, cus = (geom.nseq + 15) // 16, _cu_count(dev)
    if cus - ntiles < 16 or geom.n_inner * geom.nsteps < 32:
        return False
    if INTER_GATE_RECOMPUTE_FORCE:
        return True
    return OVERLAP_MIN_FILL * cus <= ntiles <= OVERLAP_MAX_FILL * cus and geom.nsteps >= 4 * BWD_OVERLAP_SLAB


def can_overlap_inter_bwd(geom, u, hs):
    """the overlapped form pays when the recurrence leaves a good part of the chip idle and has enough slabs to pipeline"""
    if _wide():               # wide form: u / hs are the (hi, lo) pair tensors
        if not (BWD_OVERLAP and STREAM_LIN_WGRAD and FUSED_LN_BWD and LSTM_MMA == 1 and u is not None and hs is not None
                and u.dtype == torch.float16 and hs.dtype == torch.float16 and u.shape[-1] in (32, 64)
                and hs.shape[-1] == 2 * H):
            return False
    elif not (BWD_OVERLAP and STREAM_LIN_WGRAD and FUSED_LN_BWD and DGATES_FP16 and _compact() and LSTM_MMA in (1, 2)
              and can_fuse_linear_bwd() and u is not None and hs is not None and u.dtype == torch.float16
              and hs.dtype == torch.float16 and u.shape[-1] in (16, 32)):
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    ntiles = (geom.nseq + 15) // 16
    cus = _cu_count(u.device)
    return (OVERLAP_MIN_FILL * cus <= ntiles <= OVERLAP_MAX_FILL * cus and geom.nsteps >= 4 * BWD_OVERLAP_SLAB
            and geom.n_inner * geom.nsteps >= 32 and (BWD_PAIR_SERIAL or overlap_available()))


def lstm_bwd_inter_overlapped(w_hh, gates, geom, dy, w_lin, u, hs, w_ih, targets, lin_targets, ln, recompute=None, serial=False):
    """Inter-frame backward of one block, recurrence and streaming part overlapped (see can_overlap_inter_bwd):
    dy [P, C]; u / hs the fp16 side outputs; targets = (dW_ih, dW_hh, db_ih, db_hh), lin_targets = (dW_lin, db_lin),
    ln = (x [P, C], ln_g, d_ln_g, d_ln_b).  -> dx [P, C] = LN-backward(du) + dy (max |dx| left as a hint)."""
    lib = L.load()
    rec, cprev = gates
    dev = dy.device
    Cc = dy.shape[-1]
    P = geom.P
    F_ = geom.n_inner
    gmax = absmax_or_hint(dy)
    wide = cprev.dtype == torch.float32                # wide form: dgates rows [hi x 256 | scaled lo x 256] halves
    dg = torch.empty(P, 1, 8 if wide else 4, H, device=dev, dtype=torch.float16)
    a = L.LstmBwdArgs()
    a.nseq, a.nsteps, a.n_inner, a.ndir = geom.nseq, geom.nsteps, geom.n_inner, 1
    a.p_outer, a.p_inner, a.p_step = geom.p_outer, geom.p_inner, geom.p_step
    a.w_hh[0] = _p(w_hh)
    a.save_gates, a.save_c = (C.c_void_p(rec.data_ptr()) if rec is not None else None), C.c_void_p(cprev.data_ptr())
    a.dgates, a.gmax, a.mma = C.c_void_p(dg.data_ptr()), _p(gmax), LSTM_MMA
    a.wide = 1 if wide else 0
    a.dy, a.w_lin, a.C_lin = _p(dy), _p(w_lin), Cc
    if recompute is not None:          # (b_ih, b_hh, h0 or None): no gate records -- recomputed from the u / hs pairs
        assert wide and Cc == 32
        a.recompute, a.u, a.hs, a.w_ih, a.C = 1, _ph(u), _ph(hs), _p(w_ih), Cc
        a.b_ih[0], a.b_hh[0] = _p(recompute[0]), _p(recompute[1])
        a.h0 = _p(recompute[2]) if recompute[2] is not None else None
    else:
        assert rec is not None
    s = L.LstmStreamArgs()
    s.P, s.ndir, s.C = P, 1, Cc
    s.shift_pos, s.seg_len, s.skip = F_, geom.nsteps * F_, F_
    s.dgates, s.u, s.hs = C.c_void_p(dg.data_ptr()), _ph(u), _ph(hs)
    s.gmax, s.u_f16, s.hs_f16, s.split_bf16 = _p(gmax), 1, 1, 1
    s.wide = 1 if wide else 0
    s.w_ih[0] = _p(w_ih)
    s.dW_ih[0], s.dW_hh[0], s.db_ih[0], s.db_hh[0] = (_p(t) for t in targets)
    dx = torch.empty(P, Cc, device=dev, dtype=torch.float32)
    s.ln_x, s.ln_g, s.ln_res, s.dx, s.d_ln_g, s.d_ln_b = _p(ln[0]), _p(ln[1]), _p(dy), _p(dx), _p(ln[2]), _p(ln[3])
    s.d_lin_w, s.d_lin_b = _p(lin_targets[0]), _p(lin_targets[1])
    rows = lib.sb_lstm_overlap_rows(P, geom.nseq)
    scratch = torch.empty(rows * (4 * H * (Cc + H) + 4 * H + 2 * Cc + Cc * H + Cc), device=dev, dtype=torch.float32)
    s.scratch = _p(scratch)
    s.sched_status = C.c_void_p(sched_status(dev).data_ptr())
    gm = None
    if ABSMAX_HINTS:
        gm = zero_scalar(dev)
        s.absmax_out = _p(gm)
    slab = BWD_OVERLAP_SLAB
    flags = flag_words((geom.nsteps + slab - 1) // slab + 4, dev)      # + 4 control words (uncached flag memory; zeroed by the call)
    serial = serial or BWD_PAIR_SERIAL
    rec_b = (256.0 + 256.0 + 4.0 * Cc if recompute is not None else wide_rec_bytes()) if wide else 640.0    # recurrence reads per position
    by = P * ((rec_b + 3 * 1024.0 + 256 * 2 + 4.0 * Cc if wide else 640.0 + 2 * 512.0 + 128 * 2 + 2.0 * Cc) + 4 * 4.0 * Cc)
    with _Prof(f"lstm_bwd inter overlapped C={Cc} (recurrence || stream kernel)" + (" [wide]" if wide else "")
               + (" [gates recomputed]" if recompute is not None else "")
               + (" [in plain order]" if serial else ""),
               (2.0 * 4 * H * H + 2.0 * H * Cc + 2.0 * 4 * H * (Cc + H) + 2.0 * 4 * H * Cc
                + (2.0 * 4 * H * (Cc + H) if recompute is not None else 0.0)) * P, 8.0 * Cc * P, by):
        fn = lib.sb_lstm_bwd_inter_pair_serial if serial else lib.sb_lstm_bwd_inter_overlapped
        rc = fn(C.byref(a), C.byref(s), C.c_void_p(flags.data_ptr()), slab, _stream())
        if rc == -1009:                    # no side stream (any more): the caller takes the two plain launches
            overlap_lost()
            return None
        L.check(rc, "sb_lstm_bwd_inter_overlapped")
        SCHED_COUNTS["bwd_plain" if serial else "bwd_overlapped"] += 1
    if gm is not None:
        absmax_hint_put(dx, gm)
    return dx


# single-direction passes on the default path: the streaming part runs inside the backward recurrence (the dgates stay in
# LDS); SB_NO_FUSED_BPTT=1 keeps the two-kernel form
FUSED_BPTT = os.environ.get("SB_NO_FUSED_BPTT", "0") != "1"
FUSED_LN_BWD = os.environ.get("SB_NO_FUSED_LN", "0") != "1"      # ... and the block's LayerNorm backward with it (C = 16)


_CU_COUNT = {}


def _cu_count(dev):
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    if i not in _CU_COUNT:
        _CU_COUNT[i] = torch.cuda.get_device_properties(i).multi_processor_count
    return _CU_COUNT[i]


def can_fuse_stream(u, hs, geom=None):
    """fp16 side outputs present (default training path).  With a geometry: also whether fusing pays -- the extra chunk
    arithmetic lengthens the serial chain of every tile, which only wins where the pass is memory-bound, i.e. the tiles
    fill the chip (measured: 290 tiles on 256 CUs +7.5 % train step, 145 tiles -1.2 %)."""
    if _wide():               # wide form: the fused kernel is the only one that reads the blocked fp32 records
        return (FUSED_BPTT and LSTM_MMA == 1 and u is not None and hs is not None and u.dtype == torch.float16
                and hs.dtype == torch.float16 and u.shape[-1] in (32, 64) and hs.shape[-1] == 2 * H)
    ok = (FUSED_BPTT and DGATES_FP16 and _compact() and LSTM_MMA in (1, 2) and u is not None and hs is not None
          and u.dtype == torch.float16 and hs.dtype == torch.float16 and u.shape[-1] in (16, 32))
    if ok and geom is not None and os.environ.get("SB_FORCE_FUSED_BPTT", "0") != "1":
        ok = 4 * ((geom.nseq + 15) // 16) >= 3 * _cu_count(u.device)
    return ok


# Backward overlapped ACROSS the two passes of a block (round 4): the inter-frame backward as ONE fused role-split launch on its
# 145 CUs (producer) and the intra-frame bidirectional backward of the same block on the CUs it leaves idle, tile by tile as the
# producer's time slabs complete (consumer, with the block's inter-frame LayerNorm backward as its per-tile prologue).
# SB_NO_BWD_CROSS_OVERLAP=1: the recurrence + stream-kernel pair / plain order as before.
BWD_CROSS_OVERLAP = os.environ.get("SB_NO_BWD_CROSS_OVERLAP", "0") != "1"
BWD_CROSS_SLAB = int(os.environ.get("SB_BWD_CROSS_SLAB", "32"))
CROSS_PENDING = {}        # data_ptr of the (not yet computed) dy1 buffer -> CrossBwd, from InterFn.backward to IntraPlainFn.backward
_HANDOVER = {"armed": False}


def arm_handover_check():
    """The cross-pass backward hands an UNFILLED gradient buffer through autograd (CROSS_PENDING: the next node's kernel fills it),
    and the fused LayerNorm + FiLM backward marks a gradient that already carries the FiLM factor (FILM_DONE); both hand-overs are
    keyed by data_ptr and assume the very tensor reaches the next node.  A tensor hook, retain_grad or gradient accumulation in
    between hands the node a COPY: the entry is then never consumed and the gradients are silently wrong.  Queued once
    per backward pass as an engine final callback: anything left over raises."""
    if _HANDOVER["armed"]:
        return
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_handover_check)
    except RuntimeError:                                # not inside a backward pass (direct calls in tests): nothing to guard
        return
    _HANDOVER["armed"] = True


def handover_reset():
    """Start of a training forward (no backward pass is in flight then): drop whatever a FAILED backward left behind.  PyTorch
    does not run the engine's final callbacks when a backward raises (out of memory, a user hook, an error from a node): without
    this the `armed` flag would stay set for the rest of the process -- the leftover check never queued again -- and stale
    data_ptr keys in CROSS_PENDING / FILM_DONE could match a recycled allocator address in a later pass."""
    _HANDOVER["armed"] = False
    if CROSS_PENDING:
        for pend in CROSS_PENDING.values():
            pend.keep.clear()
        CROSS_PENDING.clear()
    if FILM_DONE:
        FILM_DONE.clear()


def _handover_check():
    _HANDOVER["armed"] = False
    left_c, left_f = len(CROSS_PENDING), len(FILM_DONE)
    for pend in CROSS_PENDING.values():
        pend.keep.clear()
    CROSS_PENDING.clear()
    FILM_DONE.clear()
    if left_c or left_f:
        raise L.SoundBubbleHipError(
            f"backward pass ended with {left_c} cross-pass and {left_f} FiLM hand-over(s) unconsumed: a gradient tensor between "
            "InterFn and IntraPlainFn was copied on the way (tensor hook / retain_grad / accumulation?), the gradients of this pass "
            "are invalid.  SB_NO_BWD_CROSS_OVERLAP=1 / SB_NO_LN_FILM_FUSION=1 run the plain order.")


class CrossBwd:
    """hand-over between the two autograd nodes of a block: what the consumer's prologue needs, and everything the producer
    touches (kept allocated until the consumer has been enqueued: the side stream is not ordered after later main-stream work)"""

    def __init__(self, flags, slab, producer_tiles, order, need, du, x, res, ln_g, d_ln_g, d_ln_b, dy1, keep):
        self.flags, self.slab, self.producer_tiles, self.order, self.need = flags, slab, producer_tiles, order, need
        self.du, self.x, self.res, self.ln_g, self.d_ln_g, self.d_ln_b, self.dy1, self.keep = du, x, res, ln_g, d_ln_g, d_ln_b, dy1, keep

    def materialize(self):
        """plain order after all: dy1 by the LayerNorm-backward kernel on the current stream (behind the producer)"""
        ln_bwd(self.du.view(-1, 1, self.du.shape[-1]), self.x, self.ln_g, res=self.res, d_g=self.d_ln_g, d_b=self.d_ln_b,
               out=self.dy1)
        self.keep.clear()


def cross_tile_order_np(B, T, F_, slab):
    """intra-frame tiles (16 consecutive frames n = b T + t) sorted by the slab of the inter-frame BACKWARD (which walks t from
    T - 1 down: slab k covers steps T - 1 - k slab .. ) that completes them -> (order [ntiles] int32, packed [ntiles] int32):
    packed[i] = need | lo << 12 | hi << 22 for tile order[i], lo .. hi the inter-frame (producer) tiles -- 16 consecutive
    sequences b F + f -- that hold the sequences of the tile's batch entries"""
    import numpy as np
    n = np.arange((B * T + 15) // 16 * 16).reshape(-1, 16)
    ok = n < B * T
    need = np.where(ok, (T - 1 - n % T) // slab, 0).max(axis=1)
    b = np.minimum(n // T, B - 1)
    lo = (b.min(axis=1) * F_) // 16
    hi = (b.max(axis=1) * F_ + F_ - 1) // 16
    assert need.max() < 4096 and hi.max() < 1024
    order = np.argsort(need, kind="stable")
    packed = need | (lo << 12) | (hi << 22)
    return order.astype(np.int32), packed[order].astype(np.int32)


_CROSS_ORDER = {}


def _cross_order(B, T, F_, slab, dev):
    key = (B, T, F_, slab, dev.index if dev.index is not None else torch.cuda.current_device())
    r = _CROSS_ORDER.get(key)
    if r is None:
        order, need = cross_tile_order_np(B, T, F_, slab)
        r = _CROSS_ORDER[key] = (torch.from_numpy(order).to(dev), torch.from_numpy(need).to(dev))
    return r


def _cross_static_ok(geom, Cc, dev):
    """switches and geometry of the cross-pass backward (everything but the tensors at hand and the side stream)"""
    if not (BWD_CROSS_OVERLAP and BWD_OVERLAP and _wide() and LSTM_MMA == 1 and Cc == 32 and ROLE_SPLIT and FUSED_BPTT and FUSED_BPTT_BI
            and HS_FROM_RECORDS and INTRA_LIN_FUSION and can_fuse_linear_bwd() and SCHED_OVERRIDE is None):
        return False
    ntiles, cus = (geom.nseq + 15) // 16, _cu_count(dev)
    return (OVERLAP_MIN_FILL * cus <= ntiles <= OVERLAP_MAX_FILL * cus and cus - ntiles >= 16
            and geom.nsteps >= 4 * BWD_CROSS_SLAB)


def can_cross_overlap_bwd(geom, Cc, u, hs):
    """inter-frame geometry `geom` (B F sequences x T steps): run its backward as the cross-pass producer?  Needs the wide
    role-split kernels of both passes (C = 32, h recomputed from the records in the bidirectional one), an under-filled
    producer and a side stream that really runs beside the main one.  hs None: not stored (inter_hs_from_records_ok)."""
    if not (_cross_static_ok(geom, Cc, u.device if u is not None else None)
            and u is not None and u.dtype == torch.float16 and (hs is None or hs.dtype == torch.float16)):
        return False
    if torch.cuda.is_current_stream_capturing():
        return False
    return overlap_available()


def inter_hs_from_records_ok(geom, Cc, dev):
    """forward of a single-direction pass with the fused Linear, training: leave hs unstored?  Yes where the backward will be the
    wide role-split fused kernel WITHOUT time segments -- the cross-pass producer, or (side stream lost in between) the same
    kernel in plain order: its recurrence role recomputes h from the records (HREC), as the bidirectional kernel's does."""
    return bool(_cross_static_ok(geom, Cc, dev) and not torch.cuda.is_current_stream_capturing() and overlap_available())


def lstm_bwd_fused(w_hh, gates, geom, dy, w_lin, u, hs, w_ih, targets, lin_targets=None, ln=None, produce=None):
    """Backward of a single-direction LSTM whose Linear is fused (see lstm_fwd(lin=...)): recurrence + streaming part in
    one launch.  dy [P, C]; u [P, C], hs [P, 64] fp16 side outputs of the forward; targets = (dW_ih, dW_hh, db_ih,
    db_hh), lin_targets = (dW_lin [C, 64], db_lin [C]) (optional): accumulated into.
    ln = (x [P, C] pre-LayerNorm input, ln_g, d_ln_g, d_ln_b) (C == 16 only): the LayerNorm backward runs in the
    kernel as well and the result is dx = LN-backward(du) + dy instead of du.
    -> du [P, C] (gradient w.r.t. the LayerNorm output), or dx [P, C] with ln"""
    lib = L.load()
    rec, cprev = gates
    dev = dy.device
    Cc = dy.shape[-1]
    # hs None (wide role-split C = 32 form, no time segments): not stored, the kernel recomputes h from the records
    assert (can_fuse_stream(u, hs) or (hs is None and _wide() and ROLE_SPLIT and Cc == 32 and FUSED_BPTT and LSTM_MMA == 1))
    assert cprev is not None and w_lin.shape == (Cc, H) and u.numel() * u.element_size() in (2 * geom.P * Cc, 4 * geom.P * Cc)
    gmax = absmax_or_hint(dy)
    a = L.LstmBwdArgs()
    a.nseq, a.nsteps, a.n_inner, a.ndir = geom.nseq, geom.nsteps, geom.n_inner, 1
    a.p_outer, a.p_inner, a.p_step = geom.p_outer, geom.p_inner, geom.p_step
    a.w_hh[0] = _p(w_hh)
    a.save_gates = C.c_void_p(rec.data_ptr())
    a.save_c = C.c_void_p(cprev.data_ptr())
    a.gmax, a.mma = _p(gmax), LSTM_MMA
    a.wide = 1 if rec.dtype == torch.float32 else 0
    a.split = 1 if ROLE_SPLIT else 0
    a.dy, a.w_lin, a.C_lin = _p(dy), _p(w_lin), Cc
    ntiles = (geom.nseq + 15) // 16
    seg_scratch = None
    if TIME_SEGMENTS and produce is None and hs is not None:
        seg_scratch = _seg_scratch(a, geom, dev)
    du = torch.empty(geom.P, Cc, device=dev, dtype=torch.float32)
    wpart = torch.empty(ntiles, 4 * H * (Cc + H) + 4 * H + Cc * H + Cc + (2 * Cc if ln is not None else 0), device=dev,
                        dtype=torch.float32)
    a.u, a.hs, a.w_ih, a.C = _ph(u), _ph(hs), _p(w_ih), Cc
    a.wpart = _p(wpart)
    if ln is not None:
        assert Cc == 16
        a.ln_x, a.ln_g, a.dx, a.d_ln_g, a.d_ln_b = _p(ln[0]), _p(ln[1]), _p(du), _p(ln[2]), _p(ln[3])
    else:
        a.du = _p(du)
    a.dW_ih, a.dW_hh, a.db_ih, a.db_hh = (_p(t) for t in targets)
    if lin_targets is not None:
        assert lin_targets[0].shape == (Cc, H)
        a.dW_lin, a.db_lin = _p(lin_targets[0]), _p(lin_targets[1])
    by = (geom.P * ((wide_rec_bytes() if a.wide else 640.0) + 4.0 * Cc + 4.0 * Cc + (8.0 * Cc if ln is not None else 0.0))
          + (hs.numel() * hs.element_size() if hs is not None else 0) + u.numel() * u.element_size())
    fl = (2.0 * 4 * H * H + 2.0 * H * Cc + 2.0 * 4 * H * (Cc + H) + 2.0 * 4 * H * Cc + 2.0 * H * Cc) * geom.P
    with _Prof(f"lstm_bwd_rec_bf_kernel C={Cc} inter-frame fused BPTT" + (" + LayerNorm backward" if ln is not None else "")
               + (" [wide]" if a.wide else "") + (" [role-split]" if a.split else "")
               + (" [cross-pass producer]" if produce is not None else ""),
               fl, 8.0 * Cc * geom.P, by):
        if produce is not None:            # (flags, slab): publish du slab by slab for the intra-frame backward of the same block
            rc = lib.sb_lstm_bwd_cross_produce_ex(C.byref(a), C.c_void_p(produce[0].data_ptr()), produce[0].numel(), produce[1],
                                                  1 if (len(produce) > 2 and produce[2]) else 0, _stream())
            if rc == -1009:                # no side stream (any more): the plain launch; the caller materialises dy1 itself
                overlap_lost()
                L.check(lib.sb_lstm_bwd_rec(C.byref(a), _stream()), "sb_lstm_bwd_rec (fused)")
                SCHED_COUNTS["bwd_plain"] += 1
                return du, False, [wpart, gmax, seg_scratch]
            L.check(rc, "sb_lstm_bwd_cross_produce")
            SCHED_COUNTS["bwd_overlapped"] += 1
            return du, True, [wpart, gmax, dy, u, hs, rec, cprev, w_hh, w_lin, w_ih]
        L.check(lib.sb_lstm_bwd_rec(C.byref(a), _stream()), "sb_lstm_bwd_rec (fused)")
    SCHED_COUNTS["bwd_plain"] += 1
    return du


FUSED_BPTT_BI = os.environ.get("SB_NO_FUSED_BPTT_BI", "0") != "1"
# ... with role-split workgroups: four recurrence waves + four chunk-arithmetic waves per workgroup, two waves per SIMD
# (SB_NO_ROLE_SPLIT=1: the one-role kernel, 4 waves with 500+ registers each)
ROLE_SPLIT = os.environ.get("SB_NO_ROLE_SPLIT", "0") != "1"


# wide bidirectional C = 32 layers (Linear applied inside the forward kernel) under the role-split backward: hs is neither
# stored by the forward pass nor read by the backward kernel -- its recurrence waves recompute h from the records
# (SB_NO_HS_RECOMPUTE=1: hs pairs travel through HBM as before)
HS_FROM_RECORDS = os.environ.get("SB_NO_HS_RECOMPUTE", "0") != "1"


def bi_hs_from_records(Cc):
    return bool(HS_FROM_RECORDS and ROLE_SPLIT and _wide() and Cc == 32 and FUSED_BPTT and FUSED_BPTT_BI and LSTM_MMA == 1
                and can_fuse_linear_bwd())


def can_fuse_stream_bi(u, hs):
    """bidirectional passes: fused form with fp32 hs, or (C == 32, partial-Linear forward) fp16 hs (see lstm_bwd_fused_bi)"""
    if _wide():               # u as (hi, lo) pairs; hs as pairs (C = 32, Linear applied in the forward kernel) or fp32 (C = 16)
        if hs is None:        # not stored: the role-split kernel recomputes h (bi_hs_from_records)
            return bool(u is not None and u.dtype == torch.float16 and u.shape[-1] == 64 and bi_hs_from_records(32))
        return (FUSED_BPTT and FUSED_BPTT_BI and LSTM_MMA == 1 and u is not None and u.dtype == torch.float16
                and ((u.shape[-1] == 64 and hs.dtype == torch.float16 and hs.shape[-1] == 4 * H)
                     or (u.shape[-1] == 32 and hs.dtype == torch.float32)))
    return (FUSED_BPTT and FUSED_BPTT_BI and DGATES_FP16 and _compact() and LSTM_MMA in (1, 2) and u is not None
            and u.dtype == torch.float16 and u.shape[-1] in (16, 32)
            and (hs.dtype == torch.float32 or (hs.dtype == torch.float16 and u.shape[-1] == 32)))


# bidirectional (intra-frame) passes: the Linear(128 -> C) is applied inside the forward recurrence as two per-direction
# partial products + one elementwise pass, and hs travels as fp16 (SB_NO_INTRA_LIN_FUSION=1: hs fp32 + a Linear kernel)
INTRA_LIN_FUSION = os.environ.get("SB_NO_INTRA_LIN_FUSION", "0") != "1"


# inference with at most this many (sequence, direction) chains in a recurrent pass: sb_lstm_fwd runs them one workgroup per
# chain on the vector ALU (sb_lstm_vec.hip: ~4x shorter step than a 16-sequence tile holding one sequence) when the call
# asks for hs alone -- so the intra-frame Linear is then left to its own kernel (SB_NO_VEC_LSTM=1: tile kernels always)
VEC_LSTM = os.environ.get("SB_NO_VEC_LSTM", "0") != "1"
VEC_LSTM_MAX_CHAINS = 256


def vec_lstm_ok(train, nseq, ndir):
    return VEC_LSTM and not train and nseq is not None and nseq * ndir <= VEC_LSTM_MAX_CHAINS


def intra_lin_fusion_ok(train, Cc, nseq=None):
    """inference: any time the fp16x3 forward is on; training: only with the fused bidirectional backward (the one kernel
    that takes hs as fp16 [P, 128]).  nseq (sequences of the pass, when the caller knows them): a handful of sequences in
    inference go to the vector-ALU kernel, which leaves the Linear to its own launch."""
    if not (INTRA_LIN_FUSION and can_fuse_linear_fwd() and Cc == 32):
        return False
    if vec_lstm_ok(train, nseq, 2):
        return False
    if train and _wide():
        return can_fuse_linear_bwd() and FUSED_BPTT and FUSED_BPTT_BI
    return (not train) or (can_fuse_linear_bwd() and FUSED_BPTT and FUSED_BPTT_BI and DGATES_FP16 and AUX_FP16)


# ... and the elementwise pass that finishes it (x + part0 + part1) is done by the loader of the inter-frame forward
# recurrence that follows (SB_NO_INTER_SUM3=1: sb_add3 as a separate pass)
INTER_SUM3 = os.environ.get("SB_NO_INTER_SUM3", "0") != "1"
# FiLM of the next block applied in the inter-frame forward kernel's y epilogue (SB_NO_INTER_FILM=1: sb_film_fwd pass)
INTER_FILM = os.environ.get("SB_NO_INTER_FILM", "0") != "1"
# the inter-frame Linear's weight gradient rides in the stream kernel next to the LayerNorm backward
# (SB_NO_STREAM_LIN_WGRAD=1: its own sb_wgrad launch)
STREAM_LIN_WGRAD = os.environ.get("SB_NO_STREAM_LIN_WGRAD", "0") != "1"


def tail_rows(inp, bias, out, rows, F_, Fm, Cc):
    """out[r, f, :] = inp[r, f, :] (+ bias) for Fm <= f < F_ of every row (inp, out [rows, F_, C]): the conv-LSTM intra path's
    residual at the frequencies its k = s = down convolutions do not reach"""
    L.check(L.load().sb_tail_rows(_p(inp), _p(bias), _p(out), rows, F_, Fm, Cc, _stream()), "sb_tail_rows")


def add3(x, part):
    """x [P, C] + part[:, 0] + part[:, 1]  (part [P, 2, C])"""
    y = torch.empty_like(x)
    Cc = x.shape[-1]
    L.check(L.load().sb_add3(_p(x), _p(part), _p(y), x.numel() // Cc, Cc, _stream()), "sb_add3")
    return y


def lstm_bwd_fused_bi(w_hh_list, gates, geom, u, hs, w_ih_list, targets, dhs=None, dy=None, w_lin=None, gmax=None,
                      lin_targets=None, biases=None, consume=None, defer_ok=False):
    """Backward of a bidirectional LSTM pass, recurrence + streaming part in one launch (persistent workgroups, dgates in
    LDS).  Incoming gradient: dhs [P, 128], or dy [P, C] with w_lin [C, 128] (fused Linear backward, C == 32).
    u [P, C] fp16, hs [P, 128] fp32; targets[d] = (dW_ih, dW_hh, db_ih, db_hh) accumulated into.  -> du [P, 2, C]"""
    lib = L.load()
    rec, cprev = gates
    dev = u.device
    Cc = w_ih_list[0].shape[1]
    assert can_fuse_stream_bi(u, hs) and cprev is not None and len(w_hh_list) == 2
    if consume is not None:       # (CrossBwd) dy does not exist yet: the kernel's per-tile prologue forms it and its own scale
        gmax = zero_scalar(dev)
    elif gmax is None:
        gmax = absmax_or_hint(dy if dy is not None else dhs)
    a = L.LstmBwdArgs()
    a.nseq, a.nsteps, a.n_inner, a.ndir = geom.nseq, geom.nsteps, geom.n_inner, 2
    a.p_outer, a.p_inner, a.p_step = geom.p_outer, geom.p_inner, geom.p_step
    a.w_hh[0], a.w_hh[1] = _p(w_hh_list[0]), _p(w_hh_list[1])
    if rec is None:           # the forward kept no gate records: recompute them (needs the forward biases)
        assert biases is not None and hs.dtype == torch.float16 and dy is not None
        a.recompute = 1
        for d in range(2):
            a.b_ih[d], a.b_hh[d] = _p(biases[d][0]), _p(biases[d][1])
    else:
        a.save_gates = C.c_void_p(rec.data_ptr())
    a.save_c = C.c_void_p(cprev.data_ptr())
    a.gmax, a.mma = _p(gmax), LSTM_MMA
    a.wide = 1 if cprev.dtype == torch.float32 else 0
    a.split = 1 if (ROLE_SPLIT and rec is not None) else 0
    if dy is not None:
        assert w_lin.shape == (Cc, 2 * H) and dy.shape[-1] == Cc
        a.dy, a.w_lin, a.C_lin = _p(dy), _p(w_lin), Cc
    else:
        a.dhs = _p(dhs)
    ntiles = (geom.nseq + 15) // 16
    rows = 2 * min(ntiles, max(1, _cu_count(dev) // 2))
    if consume is not None:
        rows = lib.sb_lstm_bwd_cross_rows(geom.nseq, consume.producer_tiles)
    du = torch.empty(geom.P, 2, Cc, device=dev, dtype=torch.float32)
    wpart = torch.empty(rows, 4 * H * (Cc + H) + 4 * H + (Cc * 2 * H + Cc if dy is not None else 0)
                        + (2 * Cc if consume is not None else 0), device=dev, dtype=torch.float32)
    if lin_targets is not None:          # (dW_lin [C, 128], db_lin [C]) of the fused Linear (dy form only)
        assert dy is not None and lin_targets[0].shape == (Cc, 2 * H)
        a.dW_lin, a.db_lin = _p(lin_targets[0]), _p(lin_targets[1])
    a.u, a.hs, a.C = _ph(u), (_ph(hs) if hs is not None else None), Cc
    assert hs is not None or (a.wide and a.split and dy is not None and Cc == 32)
    a.hs_f16 = int(hs is not None and hs.dtype == torch.float16 and not a.wide)
    assert hs is None or not (a.hs_f16 or (a.wide and hs.dtype == torch.float16)) or (dy is not None and Cc == 32)
    a.w_ih, a.w_ih1 = _p(w_ih_list[0]), _p(w_ih_list[1])
    a.du, a.wpart = _p(du), _p(wpart)
    a.dW_ih, a.dW_hh, a.db_ih, a.db_hh = (_p(t) for t in targets[0])
    a.dW_ih1, a.dW_hh1, a.db_ih1, a.db_hh1 = (_p(t) for t in targets[1])
    by = (geom.P * (2 * ((wide_rec_bytes() if a.wide else 640.0) if rec is not None else 128.0) + (4.0 * Cc if dy is not None else 8.0 * H)
                    + 8.0 * Cc) + (hs.numel() * hs.element_size() if hs is not None else 0) + u.numel() * u.element_size())
    fl = 2 * (2.0 * 4 * H * H + (2.0 * H * Cc if dy is not None else 0.0) + 2.0 * 4 * H * (Cc + H) + 2.0 * 4 * H * Cc
              + (2.0 * H * Cc if lin_targets is not None else 0.0)) * geom.P
    if consume is not None:
        assert a.wide and a.split and dy is not None and dy.data_ptr() == consume.dy1.data_ptr() and hs is None
        a.pro_du, a.pro_x, a.pro_res, a.pro_ln_g, a.pro_dy = (_p(consume.du), _p(consume.x), _p(consume.res), _p(consume.ln_g),
                                                              _p(consume.dy1))
        a.d_ln_g, a.d_ln_b = _p(consume.d_ln_g), _p(consume.d_ln_b)
        a.sched_status = C.c_void_p(sched_status(dev).data_ptr())
        by += geom.P * 4 * 4.0 * Cc                                   # the prologue: du, x, res in, dy1 out (per direction: twice)
    with _Prof(f"lstm_bwd_rec_bf_kernel C={Cc} intra-frame fused BPTT (bidirectional, persistent)" + (" [wide]" if a.wide else "")
               + (" [role-split]" if a.split else "") + (" [cross-pass consumer, overlapped]" if consume is not None else ""), fl,
               8.0 * Cc * geom.P, by, side=consume is not None):
        if consume is not None:
            # defer_ok: every target is a buffer nobody reads before the end of the backward pass (the flat bucket) -- NOT a fresh
            # tensor handed back to autograd, whose AccumulateGrad would read it on the main stream straight after this node
            on_side = 1 if defer_ok and defer_small_launches((wpart,) + tuple(t for d in targets for t in d) + tuple(lin_targets or ())
                                                + (consume.d_ln_g, consume.d_ln_b)) else 0
            rc = lib.sb_lstm_bwd_cross_consume_ex(C.byref(a), C.c_void_p(consume.flags.data_ptr()), consume.slab,
                                                  consume.producer_tiles, C.c_void_p(consume.order.data_ptr()),
                                                  C.c_void_p(consume.need.data_ptr()), on_side, _stream())
            if rc == -1009:                # the side stream went away between the two calls: plain order
                overlap_lost()
                return None
            L.check(rc, "sb_lstm_bwd_cross_consume")
            consume.keep.clear()
            deferred_flush()               # (the reductions parked by the block behind us: behind this consumer on the side stream)
            return du
        L.check(lib.sb_lstm_bwd_rec(C.byref(a), _stream()), "sb_lstm_bwd_rec (fused, bidirectional)")
    return du


def can_fuse_stream_ln(dg, u, hs):
    """the streaming kernel can carry the LayerNorm backward (single direction, scaled fp16 dgates, fp16 u / hs)"""
    return (FUSED_LN_BWD and isinstance(dg, DGates) and dg.gmax is not None and dg.data.shape[1] == 1
            and u.dtype == torch.float16 and hs.dtype == torch.float16)


def lstm_bwd_stream(dg, u, hs, w_ih_list, shift_pos, seg_len, skip, targets=None, ln=None, lin_targets=None):
    """One pass over dgates [P, ndir, 4, 64]: -> [(dW_ih, dW_hh, db_ih, db_hh)] per direction, du_part [P, ndir, C].
    targets (optional): per direction 4 buffers the gradients are ACCUMULATED into (instead of fresh zero tensors).
    ln = (x [P, C] pre-LayerNorm input, ln_g, res [P, C], d_ln_g, d_ln_b) (see can_fuse_stream_ln): the LayerNorm backward
    runs in the kernel and dx [P, C] = LN-backward(du) + res is returned instead of du_part (max |dx| left as a hint).
    lin_targets = (d_lin_w [C, 64], d_lin_b [C]) (with ln): += the gradients of the Linear whose output gradient res is."""
    lib = L.load()
    if not isinstance(dg, DGates):
        dg = DGates(dg)
    gmax, dg = dg.gmax, dg.data
    P, ndir = dg.shape[0], dg.shape[1]
    Cc = u.shape[-1]
    dev = dg.device
    a = L.LstmStreamArgs()
    a.P, a.ndir, a.C = P, ndir, Cc
    a.shift_pos, a.seg_len, a.skip = shift_pos, seg_len, skip
    a.dgates, a.u, a.hs = C.c_void_p(dg.data_ptr()), _ph(u), _ph(hs)
    a.gmax = _p(gmax)
    a.u_f16, a.hs_f16 = int(u.dtype == torch.float16), int(hs.dtype == torch.float16)
    grads = []
    for d in range(ndir):
        g = targets[d] if targets is not None else (
            torch.zeros(4 * H, Cc, device=dev), torch.zeros(4 * H, H, device=dev),
            torch.zeros(4 * H, device=dev), torch.zeros(4 * H, device=dev))
        grads.append(g)
        a.w_ih[d] = _p(w_ih_list[d])
        a.dW_ih[d], a.dW_hh[d], a.db_ih[d], a.db_hh[d] = _p(g[0]), _p(g[1]), _p(g[2]), _p(g[3])
    du = torch.empty(P, ndir, Cc, device=dev, dtype=torch.float32) if ln is None else torch.empty(P, Cc, device=dev,
                                                                                                  dtype=torch.float32)
    ng = lib.sb_lstm_stream_grid(P)
    extras = (2 * Cc if ln is not None else 0) + (Cc * H + Cc if lin_targets is not None else 0)
    scratch = torch.empty(ndir * ng * (4 * H * (Cc + H) + 4 * H + extras), device=dev, dtype=torch.float32)
    a.scratch = _p(scratch)
    gm = None
    if ln is not None:
        assert ndir == 1 and gmax is not None
        a.ln_x, a.ln_g, a.ln_res, a.dx, a.d_ln_g, a.d_ln_b = _p(ln[0]), _p(ln[1]), _p(ln[2]), _p(du), _p(ln[3]), _p(ln[4])
        if lin_targets is not None:
            a.d_lin_w, a.d_lin_b = _p(lin_targets[0]), _p(lin_targets[1])
        if ABSMAX_HINTS:
            gm = zero_scalar(dev)
            a.absmax_out = _p(gm)
    else:
        assert lin_targets is None
        a.du_part = _p(du)
    a.split_bf16 = 1 if _compact() else 0         # legacy mode keeps the fp32 matrix path
    by = P * ndir * (dg.element_size() * 4.0 * H + hs.element_size() * H + 4.0 * Cc) + u.element_size() * Cc * P
    with _Prof(f"lstm_bwd_stream kernel C={Cc} ndir={ndir}", (2.0 * 4 * H * (Cc + H) + 2.0 * 4 * H * Cc) * P * ndir,
               8.0 * Cc * P, by):
        L.check(lib.sb_lstm_bwd_stream(C.byref(a), _stream()), "sb_lstm_bwd_stream")
    if gm is not None:
        absmax_hint_put(du, gm)
    return grads, du


def ln_bwd(du_part, xin, ln_g, prelu_a=None, res=None, d_g=None, d_b=None, d_a=None, hint=False, out=None):
    """LayerNorm(+PReLU) backward.  du_part [P, ndir, C] (summed over ndir), xin [P, C] pre-LN input.
    -> out [P, C], d_ln_g [C], d_ln_b [C], d_prelu [1] or None  (d_g / d_b / d_a: optional accumulation targets)"""
    lib = L.load()
    Cc = xin.shape[-1]
    P = xin.numel() // Cc
    ndir = du_part.numel() // (P * Cc)
    dev = xin.device
    out = torch.empty(P, Cc, device=dev, dtype=torch.float32) if out is None else out
    ng = lib.sb_ln_bwd_grid(P)
    partials = torch.empty(ng, 2 * Cc + 1, device=dev, dtype=torch.float32)
    a = L.LnBwdArgs()
    a.P, a.ndir, a.C = P, ndir, Cc
    a.du_part, a.xin, a.ln_g, a.prelu_a, a.res = _p(du_part), _p(xin), _p(ln_g), _p(prelu_a), _p(res)
    a.out, a.partials = _p(out), _p(partials)
    gm = None
    if hint and ABSMAX_HINTS:          # out is the dy of the next backward recurrence: measure max |out| on the way
        gm = zero_scalar(dev)
        a.absmax_out = _p(gm)
    L.check(lib.sb_ln_bwd(C.byref(a), _stream()), "sb_ln_bwd")
    if gm is not None:
        absmax_hint_put(out, gm)
    d_g = torch.zeros(Cc, device=dev, dtype=torch.float32) if d_g is None else d_g
    d_b = torch.zeros(Cc, device=dev, dtype=torch.float32) if d_b is None else d_b
    reduce_partials(partials, Cc, d_g, 0)
    reduce_partials(partials, Cc, d_b, Cc)
    if prelu_a is not None:
        d_a = torch.zeros(1, device=dev, dtype=torch.float32) if d_a is None else d_a
        reduce_partials(partials, 1, d_a, 2 * Cc)
    else:
        d_a = None
    return out, d_g, d_b, d_a


def linear(inp, w, bias, out, grid, in_strides, out_strides, K, N, *, kseg=None, is_seg=0, n_valid=None,
           epi=L.EPI_NONE, res=None, res_strides=None, prelu_a=None, ln_g=None, ln_b=None, aux_in=None,
           aux_out=None, want_partials=False, accumulate=False, in_off=0, out_off=0, res_off=0, absmax_out=None,
           f16x3=False):
    """out[p, :N] = epi(W[N,K] . in(p, :K) + bias).  grid = (B, T, F); strides in floats.
    N is chunked into <=128-wide launches when needed (not for LN epilogues).
    absmax_out (EPI_NONE / EPI_RES): zeroed [1] tensor that receives max |out| (the gmax of a following lstm_bwd_rec)."""
    lib = L.load()
    B_, T_, F_ = grid
    n_valid = N if n_valid is None else n_valid
    assert w.shape == (N, K) and w.is_contiguous()
    partials = None
    n0 = 0
    # the kernel stages its [nc, K] weight slice in LDS (160 KB): long-K layers (the loss STFTs) get narrower slices
    cap = max(16, min(128, (160 * 1024 // (4 * (K + 4))) // 16 * 16))
    cap = max(c for c in (16, 32, 48, 64, 80, 96, 128) if c <= cap)
    # a handful of positions (the streaming chunk step): ONE launch, column-sliced over grid.y inside the library -- the
    # 304 x 288 STFT basis is then read by 19 workgroups instead of three launches of one workgroup each
    # the fp16x3 form is built for N <= 32 and K = 9 x 32 / 9 x 16 (C in {16, 32}); wider layers (the reference constructor's
    # D = 64) take the fp32-input kernel
    f16x3 = bool(f16x3 and LINEAR_F16X3 and N <= 32 and (K + 31) // 32 in (9, 5))
    one_launch = (B_ * T_ * F_ <= 256 and epi in (L.EPI_NONE, L.EPI_RES) and not want_partials and N >= 64 and N % 16 == 0
                  and not f16x3)
    while n0 < N:
        nc = N if one_launch else min(cap, N - n0)
        if not one_launch and nc not in (16, 32, 48, 64, 80, 96, 128):       # NT in {1,2,3,4,5,6,8}
            nc = 96 if nc > 96 else 64
        a = L.LinearArgs()
        a.B, a.T, a.F, a.N, a.K = B_, T_, F_, nc, K
        a.n_valid = max(0, min(nc, n_valid - n0))
        a.kseg = K if kseg is None else kseg
        a.epi = epi
        a.inp = _poff(inp, in_off)
        a.is_b, a.is_t, a.is_f = in_strides
        a.is_seg = is_seg
        a.w = _poff(w, n0 * K)
        a.bias = _poff(bias, n0) if bias is not None else None
        a.out = _poff(out, out_off + n0)
        a.os_b, a.os_t, a.os_f = out_strides
        if res is not None:
            a.res = _poff(res, res_off + n0)
            a.rs_b, a.rs_t, a.rs_f = res_strides if res_strides is not None else out_strides
        a.prelu_a, a.ln_g, a.ln_b = _p(prelu_a), _p(ln_g), _p(ln_b)
        if absmax_out is not None:
            assert epi in (L.EPI_NONE, L.EPI_RES)
            a.absmax_out = _p(absmax_out)
        a.aux_in, a.aux_out = _p(aux_in), _p(aux_out)
        if want_partials:
            assert n0 == 0 and nc == N
            g = lib.sb_linear_grid(B_ * T_ * F_)
            partials = torch.empty(g, 2 * N + 1, device=out.device, dtype=torch.float32)
            a.partials = _p(partials)
        a.accumulate = 1 if accumulate else 0
        a.mma = 1 if f16x3 else 0      # long-K narrow convolutions on the fp16 pipe (fp32-class split)
        if a.n_valid > 0:
            L.check(lib.sb_linear_fwd(C.byref(a), _stream()), "sb_linear_fwd")
        n0 += nc
    return partials


def dense(P, ld):
    """(grid, strides) describing P dense rows of length ld"""
    return (1, 1, int(P)), (0, 0, int(ld))


def wgrad(g, ldg, N, inp, in_strides, grid, K, dW, *, g_off=0, in_off=0, kseg=None, is_seg=0, in2=None, ld2=0,
          in2_off=0, shift2=0, K2=0, dW2=None, seg_len=None, skip_first=0, skip_last=0, dbias=None, dbias2=None,
          transpose_out=False, perm_k=0, perm_n=0, bias_mod=0, wview=None, f16=False, gmax=None, gen_f16=False):
    """dW[N,K] += sum_p g[p,:N]^T in(p,:K);  dW2[N,K2] += sum_p g^T in2[p*ld2+shift2 : +K2] (segment-masked);
    dbias (+dbias2) += column sums of g.  One pass over g."""
    lib = L.load()
    B_, T_, F_ = grid
    P = B_ * T_ * F_
    a = L.WgradArgs()
    a.B, a.T, a.F, a.N, a.K = B_, T_, F_, N, K
    a.kseg = ((K + 15) // 16) * 16 if kseg is None else kseg
    a.g, a.ldg = _poff(g, g_off), ldg
    if inp.dtype == torch.float16:
        assert in_off == 0
        a.inp, a.in_f16 = _ph(inp), 1
    else:
        a.inp = _poff(inp, in_off)
    a.is_b, a.is_t, a.is_f = in_strides
    a.is_seg = is_seg
    if in2 is not None:
        a.in2, a.ld2, a.shift2, a.K2 = _poff(in2, in2_off), ld2, shift2, K2
    a.seg_len = P if seg_len is None else seg_len
    a.skip_first, a.skip_last = skip_first, skip_last
    a.transpose_out = 1 if transpose_out else 0
    a.perm_k, a.perm_n, a.bias_mod = perm_k, perm_n, bias_mod     # native-layout destinations (see the header)
    if wview is not None:                                         # ... general form: a weight view over dW's tensor
        a.wv = wview
    # 3x3 convolutions' dW on the fp16 pipe, hi+lo operands: built for N <= 32, K = 9 x 32 (or 9 x 16 with N <= 16)
    a.mma = 1 if (f16 and LINEAR_F16X3 and ((N <= 32 and K == 288 and a.kseg == 96) or (N <= 16 and K == 144 and a.kseg == 48))) else 0
    if a.mma:
        a.gmax = _p(gmax if gmax is not None else absmax_or_hint(g))   # power-of-two scale against fp16 underflow
    elif gen_f16 and gmax is not None:
        a.mma, a.gmax = 2, _p(gmax)       # the generic tiled form (if this shape takes it) on fp16 hi + lo operands, g scaled by gmax
    # partial rows: four per workgroup for the shapes with a register-accumulator kernel; the library's generic tiled form (any
    # other shape: the D = 64 / H = 128 layers) says how many position ranges it will use
    rows = lib.sb_wgrad_scratch_rows(C.byref(a))
    L.check(min(rows, 0), "sb_wgrad_scratch_rows")
    scratch = torch.empty(rows, N * (K + K2) + N, device=dW.device, dtype=torch.float32)   # one row per wave / position range (fully written)
    if _STREAM_OVERRIDE is not None:
        _DEFER["keep"].append(scratch)                # (a side-stream launch: held until the join, see on_stream)
    a.dW, a.dW2, a.dbias, a.dbias2, a.scratch = _p(dW, "dW"), _p(dW2), _p(dbias), _p(dbias2), _p(scratch)
    L.check(lib.sb_wgrad(C.byref(a), _stream()), "sb_wgrad")


def colsum(g, P, ldg, N, out, g_off=0):
    """out[:N] += sum_p g[p*ldg + :N]"""
    lib = L.load()
    scratch = torch.empty(256, N, device=out.device, dtype=torch.float32)
    L.check(lib.sb_colsum(_poff(g, g_off), P, ldg, N, _p(out), _p(scratch), _stream()), "sb_colsum")


def reduce_partials(partials, n, out, col_off=0, stream=None):
    """out[:n] += sum_rows partials[:, col_off:col_off+n]"""
    lib = L.load()
    rows, ld = partials.shape
    L.check(lib.sb_reduce_rows(_poff(partials, col_off), rows, ld, n, _p(out), stream if stream is not None else _stream()),
            "sb_reduce_rows")


def features(spec, ld_spec, zp, B, M, T, F):
    L.check(L.load().sb_features(_p(spec), ld_spec, _p(zp), B, M, T, F, _stream()), "sb_features")


def _film_bank_args(dis, W_e, ln_w, ln_b, conv):
    """conv: [w.weight [C, d_in, 1], w.bias [C], b.weight, b.bias] per FiLM layer (scale plane first, then shift)"""
    n = len(conv) // 4
    d_in = ln_w.shape[0]
    if n > L.FILM_BANK_MAX_LAYERS:
        raise L.SoundBubbleHipError(f"FiLM bank: {n} layers, the kernel's pointer table holds {L.FILM_BANK_MAX_LAYERS}")
    a = L.FilmBankArgs()
    a.B, a.F, a.C, a.n, a.K, a.d_in = dis.shape[0], W_e.shape[0] // d_in, conv[0].shape[0], n, dis.shape[1], d_in
    assert W_e.shape == (a.F * d_in, a.K) and ln_b.shape == (d_in,)
    a.dis, a.W_e, a.ln_w, a.ln_b = _p(dis, "dis_embed"), _p(W_e), _p(ln_w), _p(ln_b)
    for k in range(n):
        for wh in range(2):
            w, b = conv[4 * k + 2 * wh], conv[4 * k + 2 * wh + 1]
            assert w.numel() == a.C * d_in and b.numel() == a.C
            a.conv_w[2 * k + wh], a.conv_b[2 * k + wh] = _p(w), _p(b)
    return a


def film_bank_fwd(dis, W_e, ln_w, ln_b, conv):
    """-> planes [2n, B, F, C] (Dis_Embed_Conv + the 1x1 convolutions of every FilmLayer: one launch; sb_film_bank_fwd)"""
    a = _film_bank_args(dis, W_e, ln_w, ln_b, conv)
    planes = torch.empty(2 * a.n, a.B, a.F, a.C, device=dis.device, dtype=torch.float32)
    a.planes = _p(planes)
    L.check(L.load().sb_film_bank_fwd(C.byref(a), _stream()), "sb_film_bank_fwd")
    return planes


def film_bank_bwd(G, dis, W_e, ln_w, ln_b, conv, dW_e, d_ln_w, d_ln_b, d_conv):
    """G [2n, B, F, C]; every target is ACCUMULATED into (d_conv in the order of conv).  Two launches, deterministic."""
    lib = L.load()
    a = _film_bank_args(dis, W_e, ln_w, ln_b, conv)
    assert G.numel() == 2 * a.n * a.B * a.F * a.C
    a.G, a.dW_e, a.d_ln_w, a.d_ln_b = _p(G), _p(dW_e), _p(d_ln_w), _p(d_ln_b)
    for k in range(a.n):
        for wh in range(2):
            a.d_conv_w[2 * k + wh], a.d_conv_b[2 * k + wh] = _p(d_conv[4 * k + 2 * wh]), _p(d_conv[4 * k + 2 * wh + 1])
    partials = torch.empty(lib.sb_film_bank_bwd_scratch(a.F, a.C, a.n, a.d_in), device=G.device, dtype=torch.float32)
    a.partials = _p(partials)
    L.check(lib.sb_film_bank_bwd(C.byref(a), _stream()), "sb_film_bank_bwd")


def film_fwd(x, w, b):
    B_, T_, F_, Cc = x.shape
    y = torch.empty_like(x)
    L.check(L.load().sb_film_fwd(_p(x), _p(w), _p(b), _p(y), B_, T_, F_, Cc, _stream()), "sb_film_fwd")
    return y


def film_bwd(x, w, dy, out=None):
    """out = (dw, db): pre-zeroed [B, F, C] buffers the T-reductions are accumulated into (default: fresh zeros)"""
    B_, T_, F_, Cc = x.shape
    dx = torch.empty_like(x)
    if out is not None:
        dw, db = out
    else:
        dw = torch.zeros(B_, F_, Cc, device=x.device, dtype=torch.float32)
        db = torch.zeros_like(dw)
    gm = zero_scalar(x.device) if ABSMAX_HINTS else None   # dx is a next dy
    L.check(L.load().sb_film_bwd(_p(x), _p(w), _p(dy), _p(dx), _p(dw), _p(db), B_, T_, F_, Cc, _p(gm), _stream()),
            "sb_film_bwd")
    if gm is not None:
        absmax_hint_put(dx, gm)
    return dx, dw, db


# LayerNorm backward of an intra-frame pass + FiLM backward of the block in front of it in one kernel (round 4): the inter-frame
# forward whose epilogue applied that FiLM leaves (f_w, y_pre, (dw, db) bank slices) here under the data pointer of its output;
# the IntraPlainFn that takes that tensor as its input picks the entry up, and its backward runs the fused kernel and marks the
# gradient it returns in FILM_DONE so that the inter-frame backward skips its own FiLM pass.  SB_NO_LN_FILM_FUSION=1: two kernels.
LN_FILM_FUSION = os.environ.get("SB_NO_LN_FILM_FUSION", "0") != "1"
FILM_OF = {}
FILM_DONE = {}


def ln_film_bwd(du_part, xin, ln_g, res, film_x, film_w, dw, db, d_g, d_b, dims, defer_ok=False):
    """-> out [P, 32] = (LN-backward(du_part[:, 0] + du_part[:, 1]; xin) + res) * film_w; dw / db / d_g / d_b accumulated into"""
    lib = L.load()
    B_, T_, F_, Cc = dims
    dev = xin.device
    out = torch.empty(B_ * T_ * F_, Cc, device=dev, dtype=torch.float32)
    rows = lib.sb_ln_film_bwd_rows(B_, T_, F_)
    partials = torch.empty(rows, 2 * Cc, device=dev, dtype=torch.float32)
    gm = zero_scalar(dev) if ABSMAX_HINTS else None
    with _Prof("ln_film_bwd (intra-frame LayerNorm backward + FiLM backward)", 0.0, 8.0 * Cc * B_ * T_ * F_, 24.0 * Cc * B_ * T_ * F_):
        L.check(lib.sb_ln_film_bwd(_p(du_part), _p(xin), _p(ln_g), _p(res), _p(film_x), _p(film_w), _p(out), _p(dw), _p(db),
                                   _p(partials), B_, T_, F_, Cc, _p(gm), _stream()), "sb_ln_film_bwd")
    if gm is not None:
        absmax_hint_put(out, gm)
    # the LayerNorm parameter sums: off the critical path when the side stream is there (the next block's producer is waiting)
    def red(st):
        reduce_partials(partials, Cc, d_g, 0, st)
        reduce_partials(partials, Cc, d_b, Cc, st)

    if not (defer_ok and defer_launch(red, (partials, d_g, d_b))):      # (defer_ok: as in lstm_bwd_fused_bi)
        red(None)
    return out


def overlap_add(frames, B, T, win, hop):
    wave = torch.empty(B, hop * T, device=frames.device, dtype=torch.float32)
    L.check(L.load().sb_overlap_add(_p(frames), _p(wave), B, T, win, hop, _stream()), "sb_overlap_add")
    return wave


def multi_copy(pairs):
    """dst.copy_(src) for every (src, dst) of `pairs` (dense fp32 CUDA tensors of equal numel) in ceil(len / 16) launches
    (sb_multi_copy): the streaming chunk step's state write-back as one graph node"""
    lib = L.load()
    for i in range(0, len(pairs), L.MULTI_COPY_MAX):
        a = L.MultiCopyArgs()
        chunk = pairs[i:i + L.MULTI_COPY_MAX]
        for j, (src, dst) in enumerate(chunk):
            if not (src.is_cuda and dst.is_cuda and src.dtype == dst.dtype == torch.float32 and src.is_contiguous()
                    and dst.is_contiguous() and src.numel() == dst.numel()):
                raise L.SoundBubbleHipError("multi_copy: dense float32 GPU tensors of equal size expected")
            a.src[j], a.dst[j], a.n[j] = src.data_ptr(), dst.data_ptr(), src.numel()
        a.njobs = len(chunk)
        L.check(lib.sb_multi_copy(C.byref(a), _stream()), "sb_multi_copy")


def overlap_add_bwd(dwave, B, T, win, hop):
    df = torch.empty(B, T + 1, win, device=dwave.device, dtype=torch.float32)
    L.check(L.load().sb_overlap_add_bwd(_p(dwave), _p(df), B, T, win, hop, _stream()), "sb_overlap_add_bwd")
    return df


def deconv_bwd_data(dspec, w, B, T, F, Cc):
    dy = torch.empty(B, T, F, Cc, device=dspec.device, dtype=torch.float32)
    gm = zero_scalar(dspec.device) if ABSMAX_HINTS else None      # max |dy| measured on the way (the last block's backward reads dy next)
    L.check(L.load().sb_deconv_bwd_data(_p(dspec), _p(w), _p(dy), B, T, F, Cc, _p(gm), _stream()), "sb_deconv_bwd_data")
    if gm is not None:
        absmax_hint_put(dy, gm)
    return dy


SNR_LOSS_MODES = {"snr": 0, "sisdr": 1, "fused": 2, "max_fused": 3, "sdsdr": 4, "full": 5}     # src/losses/SNRLosses.py:10-29


def snrlp_loss_fwd(est, gt, neg_weight, mode=0):
    """est, gt [B, N] -> loss_vec [B], mean_b loss_vec [1], stats [B, 12] (what snrlp_loss_bwd needs)"""
    B_, N = est.shape
    stats = torch.empty(B_, 12, device=est.device, dtype=torch.float32)
    lv = torch.empty(B_, device=est.device, dtype=torch.float32)
    mean = torch.empty(1, device=est.device, dtype=torch.float32)
    L.check(L.load().sb_snrlp_loss_fwd(_p(est), _p(gt), B_, N, float(neg_weight), int(mode), _p(stats), _p(lv), _p(mean),
                                       _stream()), "sb_snrlp_loss_fwd")
    return lv, mean, stats


def snrlp_loss_bwd(est, gt, neg_weight, stats, gout, mode=0):
    """-> gout * d(mean_b loss)/d est [B, N]; gout: device scalar tensor [1] (or None = 1)"""
    B_, N = est.shape
    dest = torch.empty_like(est)
    L.check(L.load().sb_snrlp_loss_bwd(_p(est), _p(gt), B_, N, float(neg_weight), int(mode), _p(stats), _p(gout), _p(dest),
                                       _stream()), "sb_snrlp_loss_bwd")
    return dest


def stage_frames(state, src, dst, B, Tp, F_, Cs, Cd):
    """carried 2-frame context (+ src rows) -> the zero-bordered channels-last staging tensor of a 3x3 convolution"""
    L.check(L.load().sb_stage_frames(_p(state), _p(src), _p(dst), B, Tp, F_, Cs, Cd, _stream()), "sb_stage_frames")


def frames_to_state(rows, B, Tp, F_, Cs, Cd, r0):
    st = torch.empty(B, Cs, 2, F_, device=rows.device, dtype=torch.float32)
    L.check(L.load().sb_frames_to_state(_p(rows), _p(st), B, Tp, F_, Cs, Cd, r0, _stream()), "sb_frames_to_state")
    return st


def spec_rows(rows, buf, B, T, F_, ld, mode):
    L.check(L.load().sb_spec_rows(_p(rows), _p(buf), B, T, F_, ld, mode, _stream()), "sb_spec_rows")


def snrlp_loss(est, gt, neg_weight, want_grad, mode=0):
    """est, gt [B, N] -> loss_vec [B], d(mean loss)/d est or None; mode: SNR_LOSS_MODES[snr_loss_name]"""
    B_, N = est.shape
    stats = torch.empty(B_, 12, device=est.device, dtype=torch.float32)
    lv = torch.empty(B_, device=est.device, dtype=torch.float32)
    dest = torch.empty_like(est) if want_grad else None
    L.check(L.load().sb_snrlp_loss_ex(_p(est), _p(gt), B_, N, float(neg_weight), int(mode), _p(stats), _p(lv), _p(dest),
                                      _stream()), "sb_snrlp_loss")
    return lv, dest


def head_ln(x, gamma, beta, out, B, T, F, Hh, D, rows, t_off, ldo, res=None, ldi=None, prelu_a=None):
    ldi = Hh * D if ldi is None else ldi
    L.check(L.load().sb_head_ln(_p(x), _p(gamma), _p(beta), _p(out), _p(res), B, T, F, Hh, D, rows, t_off, ldo, ldi,
                                _p(prelu_a), _stream()), "sb_head_ln")


def head_ln_bwd(x, gamma, dout, B, T, F, Hh, D, rows, t_off, ldo, ldi, prelu_a=None):
    """-> (din [B*T*F, ldi] (columns beyond Hh*D zero), dgamma [F*D], dbeta [F*D], dalpha [1])"""
    lib = L.load()
    n = F * Hh * D
    din = torch.empty(B * T * F, ldi, device=x.device, dtype=torch.float32)      # (the kernel writes columns < Hh*D of every row)
    if ldi > Hh * D:
        din[:, Hh * D:].zero_()
    part = torch.empty(lib.sb_head_ln_bwd_grid(B, T), 2 * n + 1, device=x.device, dtype=torch.float32)
    L.check(lib.sb_head_ln_bwd(_p(x), _p(gamma), _p(dout), _p(din), _p(part), B, T, F, Hh, D, rows, t_off, ldo, ldi,
                               _p(prelu_a), _stream()), "sb_head_ln_bwd")
    red = torch.zeros(2 * n + 1, device=x.device, dtype=torch.float32)
    reduce_partials(part, 2 * n + 1, red)
    dg = red[:n].view(F, Hh, D).sum(1).reshape(F * D)
    db = red[n:2 * n].view(F, Hh, D).sum(1).reshape(F * D)
    return din, dg, db, red[2 * n:2 * n + 1]


def attn_core(Q, K, V, out, BH, Hh, T, F, Cv, Lw, ldk, ldv, scale, lse=None):
    a = L.AttnArgs()
    a.BH, a.Hh, a.T, a.F, a.Cv, a.L = BH, Hh, T, F, Cv, Lw
    a.NRp = (Lw + 15 + 15) // 16 * 16
    a.ldk, a.ldv, a.scale = ldk, ldv, scale
    a.Q, a.K, a.V, a.out, a.lse = _p(Q), _p(K), _p(V), _p(out), _p(lse)
    L.check(L.load().sb_attn_core(C.byref(a), _stream()), "sb_attn_core")


def attn_core_bwd(Q, K, V, dO, lse, BH, Hh, T, F, Cv, Lw, ldk, ldv, scale):
    """dO [BH, T, ldv] head-major -> dQ [BH,T,ldk], dK [BH,T,ldk], dV [BH,T,ldv] (current-frame rows)"""
    dev = Q.device
    a = L.AttnBwdArgs()
    a.BH, a.Hh, a.T, a.F, a.Cv, a.L = BH, Hh, T, F, Cv, Lw
    a.NRp = (Lw + 15 + 15) // 16 * 16
    a.ldk, a.ldv, a.scale = ldk, ldv, scale
    delta = torch.empty(BH, T, device=dev, dtype=torch.float32)
    dQ = torch.empty(BH, T, ldk, device=dev, dtype=torch.float32)
    dK = torch.empty(BH, T, ldk, device=dev, dtype=torch.float32)
    dV = torch.empty(BH, T, ldv, device=dev, dtype=torch.float32)
    a.Q, a.K, a.V, a.dO, a.lse = _p(Q), _p(K), _p(V), _p(dO), _p(lse)
    a.delta, a.dQ, a.dK, a.dV = _p(delta), _p(dQ), _p(dK), _p(dV)
    L.check(L.load().sb_attn_core_bwd(C.byref(a), _stream()), "sb_attn_core_bwd")
    return dQ, dK, dV


def signal_stats(est, gt, mix_ref):
    """est, gt [B, N]; mix_ref [B, N] view (row stride given by mix_ref.stride(0)) -> moments [B, 8]"""
    B_, N = est.shape
    out = torch.empty(B_, 8, device=est.device, dtype=torch.float32)
    assert mix_ref.stride(1) == 1
    lib = L.load()
    mp = C.c_void_p(mix_ref.data_ptr())
    L.check(lib.sb_signal_stats(_p(est), _p(gt), mp, B_, N, mix_ref.stride(0), _p(out), _stream()), "sb_signal_stats")
    return out


def sumsq(g, out, accumulate=True):
    """out[0] (+)= sum g^2 (accumulate=False: plain store -- no zero-fill in front of the call)"""
    L.check(L.load().sb_sumsq_ex(_p(g), g.numel(), _p(out), 1 if accumulate else 0, _stream()), "sb_sumsq_ex")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, gscale=1.0, clip=0.0, sumsq_buf=None, skipped=None):
    """skipped (int32 [1], optional): the update is GUARDED by the device's watchdog word when a guarded schedule has ever run
    on it -- a step whose launches gave up a bounded wait leaves parameters and moments untouched and counts itself here"""
    dev = p.device.index if p.device.index is not None else torch.cuda.current_device()
    guard = _SCHED_STATUS.get(dev)
    L.check(L.load().sb_adam_step_guarded(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, int(step),
                                          float(gscale), float(clip), _p(sumsq_buf),
                                          C.c_void_p(guard.data_ptr()) if guard is not None else None,
                                          C.c_void_p(skipped.data_ptr()) if (skipped is not None and guard is not None) else None,
                                          _stream()), "sb_adam_step_guarded")


# ---- fine-tune loss (include/sound_bubble_hip.h: multi-resolution STFT magnitude L1 + waveform L1) ----
def fir(x, taps):
    """y[b, n] = sum_k taps[k] x[b, n + k - ntaps // 2] (zero padded); x [B, N]"""
    y = torch.empty_like(x)
    L.check(L.load().sb_fir(_p(x), _p(taps), _p(y), x.shape[0], x.shape[1], taps.numel(), _stream()), "sb_fir")
    return y


def reflect_pad(x, pad, ldp):
    xp = torch.empty(x.shape[0], ldp, device=x.device, dtype=torch.float32)
    L.check(L.load().sb_reflect_pad(_p(x), _p(xp), x.shape[0], x.shape[1], pad, ldp, _stream()), "sb_reflect_pad")
    return xp


def stft_mag_l1(spec_x, spec_y, rows, nbins, ld, eps, gscale, loss, loss_scale, want_grad):
    """loss[0] += loss_scale * sum | |X| - |Y| |; -> d(loss)/d spec_x * (gscale / loss_scale) or None"""
    lib = L.load()
    dsx = torch.empty(rows, ld, device=spec_x.device, dtype=torch.float32) if want_grad else None
    part = torch.empty(lib.sb_stft_mag_l1_grid(rows, ld), device=spec_x.device, dtype=torch.float32)
    L.check(lib.sb_stft_mag_l1(_p(spec_x), _p(spec_y), rows, nbins, ld, eps, gscale, _p(dsx), _p(part), loss_scale, _p(loss),
                               1, _stream()), "sb_stft_mag_l1")
    return dsx


def stft_mag_terms(spec_x, spec_y, rows, nbins, ld, eps, w_lin, w_log, w_sc, scale, loss, want_grad):
    """all three auraloss STFT terms of one resolution: loss[0] += scale * (w_lin L1 + w_log log-L1 + w_sc SC);
    -> d(that) / d spec_x or None"""
    lib = L.load()
    dev = spec_x.device
    dsx = torch.empty(rows, ld, device=dev, dtype=torch.float32) if want_grad else None
    part = torch.empty(4 * lib.sb_stft_mag_l1_grid(rows, ld), device=dev, dtype=torch.float32)
    sums = torch.empty(4, device=dev, dtype=torch.float32)
    L.check(lib.sb_stft_mag_terms(_p(spec_x), _p(spec_y), rows, nbins, ld, eps, w_lin, w_log, w_sc, scale, _p(dsx), _p(part),
                                  _p(sums), _p(loss), _stream()), "sb_stft_mag_terms")
    return dsx


def stft_f64acc(xp, w, spec, B_, nframes, ldp, hop, off, K, N, lo_off=0):
    """spec[(b, t), :N] = frames of xp (row stride ldp, hop, first sample off) x w[N, K]^T, accumulated in double;
    lo_off: element distance to the low plane of a (hi, lo) pair signal (fir_pair), 0 = none"""
    L.check(L.load().sb_stft_f64acc(_p(xp), _p(w), _p(spec), B_, nframes, ldp, hop, off, K, N, lo_off, _stream()), "sb_stft_f64acc")


def fir_pair(x, taps):
    """fir() with the sum formed in double -> [2, B, N]: planes hi, lo with hi + lo the double result"""
    y = torch.empty(2, *x.shape, device=x.device, dtype=torch.float32)
    L.check(L.load().sb_fir_pair(_p(x), _p(taps), _p(y[0]), _p(y[1]), x.shape[0], x.shape[1], taps.numel(), _stream()), "sb_fir_pair")
    return y


def frames_fold(dframes, dx, nframes, K, ldk, hop, off, pad, accumulate):
    B_, N = dx.shape
    L.check(L.load().sb_frames_fold(_p(dframes), _p(dx), B_, N, nframes, K, ldk, hop, off, pad, 1 if accumulate else 0,
                                    _stream()), "sb_frames_fold")


def l1_grad(x, y, gscale, dx, accumulate, loss, loss_scale):
    n = x.numel()
    part = torch.empty((n + 255) // 256, device=x.device, dtype=torch.float32)
    L.check(L.load().sb_l1_grad(_p(x), _p(y), n, gscale, _p(dx), 1 if accumulate else 0, _p(part), loss_scale, _p(loss), 1,
                                _stream()), "sb_l1_grad")


# ---- generic-shape recurrence (sb_lstm_gen.hip): the reference constructor's own widths, D = 64 / H = 128 ----
def lstm_gen_supported(Cc, Hh):
    return bool(L.load().sb_lstm_gen_supported(int(Cc), int(Hh)))


def lstm_gen_fwd(x, ln_g, ln_b, dirs, geom, h0=None, c0=None, save=False, want_state=False):
    """LayerNorm(C) + LSTM forward for any (C, H) sb_lstm_gen_supported names.  x [P, C] pre-LayerNorm; dirs: list of
    (w_ih [4H, C], w_hh [4H, H], b_ih, b_hh) per direction.
    -> hs [P, ndir*H], (hN, cN) or None, records [P, ndir, 5, H] or None, u [P, C] or None"""
    lib = L.load()
    ndir, Cc, Hh = len(dirs), x.shape[-1], dirs[0][1].shape[1]
    assert x.numel() == geom.P * Cc
    dev = x.device
    hs = torch.empty(geom.P, ndir * Hh, device=dev, dtype=torch.float32)
    rec = torch.empty(geom.P, ndir, 5, Hh, device=dev, dtype=torch.float32) if save else None
    u = torch.empty(geom.P, Cc, device=dev, dtype=torch.float32) if save else None
    hN = torch.empty(geom.nseq, Hh, device=dev, dtype=torch.float32) if want_state else None
    cN = torch.empty(geom.nseq, Hh, device=dev, dtype=torch.float32) if want_state else None
    a = L.LstmGenFwdArgs()
    a.nseq, a.nsteps, a.n_inner, a.ndir, a.C, a.H = geom.nseq, geom.nsteps, geom.n_inner, ndir, Cc, Hh
    a.p_outer, a.p_inner, a.p_step = geom.p_outer, geom.p_inner, geom.p_step
    a.x, a.ln_g, a.ln_b = _p(x, "x"), _p(ln_g, "ln_g"), _p(ln_b, "ln_b")
    for d, (wi, wh, bi, bh) in enumerate(dirs):
        assert wi.shape == (4 * Hh, Cc) and wh.shape == (4 * Hh, Hh)
        a.w_ih[d], a.w_hh[d], a.b_ih[d], a.b_hh[d] = _p(wi), _p(wh), _p(bi), _p(bh)
    a.h0, a.c0, a.hN, a.cN = _p(h0), _p(c0), _p(hN), _p(cN)
    a.hs, a.save_gates, a.save_u = _p(hs), _p(rec), _p(u)
    by = 4.0 * Cc * geom.P + 4.0 * hs.numel() + (4.0 * (rec.numel() + u.numel()) if save else 0.0)
    with _Prof(f"lstm_gen_fwd_kernel C={Cc} H={Hh} " + ("bidirectional" if ndir == 2 else "single direction"),
               2.0 * 4 * Hh * (Cc + Hh) * geom.P * ndir, 8.0 * Cc * geom.P, by):
        L.check(lib.sb_lstm_gen_fwd(C.byref(a), _stream()), "sb_lstm_gen_fwd")
    return hs, ((hN, cN) if want_state else None), rec, u


def lstm_gen_bwd(dirs, rec, dhs, u, hs, geom, targets):
    """BPTT of lstm_gen_fwd: the recurrence (dgates of every step), then the position-wise GEMMs over the dgates.
    dhs [P, ndir*H]: gradient w.r.t. hs; targets: per direction (dW_ih, dW_hh, db_ih, db_hh), accumulated into.
    -> du_part [P, ndir, C]: per-direction gradient w.r.t. the LayerNorm output (ln_bwd sums the directions)."""
    lib = L.load()
    ndir, Cc, Hh = len(dirs), u.shape[-1], dirs[0][1].shape[1]
    dev = u.device
    P = geom.P
    dg = torch.empty(P, ndir, 4 * Hh, device=dev, dtype=torch.float32)
    a = L.LstmGenBwdArgs()
    a.nseq, a.nsteps, a.n_inner, a.ndir, a.H = geom.nseq, geom.nsteps, geom.n_inner, ndir, Hh
    a.p_outer, a.p_inner, a.p_step = geom.p_outer, geom.p_inner, geom.p_step
    for d, (wi, wh, bi, bh) in enumerate(dirs):
        a.w_hh[d] = _p(wh)
    a.save_gates, a.dhs, a.dgates = _p(rec), _p(dhs), _p(dg)
    gmax = absmax(dhs)                  # the fp16 scale of the recurrence's dgates operand (as the tuned kernels' gmax)
    a.gmax = _p(gmax)
    with _Prof(f"lstm_gen_bwd_rec_kernel H={Hh} " + ("bidirectional" if ndir == 2 else "single direction"),
               2.0 * 4 * Hh * Hh * P * ndir, 8.0 * Cc * P, 4.0 * (rec.numel() + dhs.numel() + dg.numel())):
        L.check(lib.sb_lstm_gen_bwd_rec(C.byref(a), _stream()), "sb_lstm_gen_bwd_rec")
    du = torch.empty(P, ndir, Cc, device=dev, dtype=torch.float32)
    gP, sC = dense(P, Cc)
    ldg, ldh = ndir * 4 * Hh, ndir * Hh
    for d, (wi, wh, bi, bh) in enumerate(dirs):
        # du[:, d] = dgates[:, d] . W_ih   (position-wise GEMM, K = 4H)
        linear(dg, wi.t().contiguous(), None, du, gP, (0, 0, ldg), (0, 0, ndir * Cc), 4 * Hh, Cc, in_off=d * 4 * Hh, out_off=d * Cc)
        # dW_ih += dgates^T u ; dW_hh += dgates^T h_prev ; db_ih, db_hh += column sums -- one pass over the dgates.  h_prev of a
        # position is hs one step earlier in the direction's own walk, zero at the first step of every sequence
        back = -geom.p_step if d == 0 else geom.p_step
        t_wi, t_wh, t_bi, t_bh = targets[d]
        wgrad(dg, ldg, 4 * Hh, u, sC, gP, Cc, t_wi, g_off=d * 4 * Hh, in2=hs, ld2=ldh, in2_off=d * Hh, shift2=back * ldh, K2=Hh,
              dW2=t_wh, seg_len=geom.nsteps * geom.p_step, skip_first=geom.p_step if d == 0 else 0,
              skip_last=geom.p_step if d == 1 else 0, dbias=t_bi, dbias2=t_bh, gen_f16=True, gmax=gmax)
    return du
