// Recurrent LSTM kernels on the 16-bit matrix pipes with split fp32 operands.
//
// Why: on gfx950 the fp32-input MFMA runs at the fp32 vector rate and does not overlap the cell-update VALU work
// of a co-resident wave (measured: two workgroups per CU give no throughput over one), so the fp32 kernels in
// sb_lstm.hip top out at ~45-65 % of the 157 TFLOP/s fp32 peak.  The bf16 / fp16 matrix pipes are 16x faster and
// truly concurrent with the VALU.  Every fp32 operand is split into 16-bit terms and the product evaluated term by
// term, each 16-bit x 16-bit product being exact in the fp32 accumulator:
//   fp16x3 (default):      x = hi + lo  (11 + 11 mantissa bits),  a*b = lo*hi + hi*lo + hi*hi   (dropped <= 2^-22)
//   bf16x6 (SB_LSTM_BF16X6): x = h + m + l (8 + 8 + 8 bits),  a*b = l*h + h*l + m*m + m*h + h*m + h*h (<= 2^-24)
// Both are fp32-class for this network (identical measured error against the reference goldens, 9e-7 .. 3e-6 rel-L2
// on the output): required because the recurrence amplifies rounding noise over 625 steps and the parity bar is 1e-3.
// In compact-BPTT mode the backward recurrence additionally carries its dgates as (scaled) fp16 -- see below.
//
// v_mfma_f32_16x16x32_{bf16,f16}: A lane l holds A[i = l&15][k = 8*(l>>4)..+7], B lane l holds B[k = 8*(l>>4)..+7][j = l&15],
// C/D as in sb_common.h.  K is walked in chunks of 32: chunk 0 = the (LayerNormed) input u (zero-padded to 32),
// chunks 1,2 = the hidden state.  Step pipeline: A: hidden part (MFMA) with the LayerNorm of row s+2 in its issue
// gaps; B: input part of step s+1 (MFMA) || cell update; C: h -> LDS, stores, barrier.
//
// This header: what the forward TU (sb_lstm_bf_fwd.hip) and the backward TU (sb_lstm_bf_bwd.hip) share -- operand formats and
// splitters, the bounded hand-off wait of the guarded schedules, LDS row paddings, launch helpers.  (One 2 400-line file with
// ~170 kernel instantiations took two minutes to compile; the two translation units build in parallel.)
#ifndef SB_LSTM_BF_COMMON_H
#define SB_LSTM_BF_COMMON_H
#include <type_traits>
#include "sb_common.h"
#ifndef SB_EXP_SKIP
#define SB_EXP_SKIP 0
#endif
// Developer experiment (scripts/exp_gate_recompute.py; 0 in the shipped library): cost model of RECOMPUTING the four
// gates in the backward recurrence from (u, h_prev) instead of loading the forward's gate records -- bit 0: issue the
// recompute's instruction mix per step (12 fp16 MFMAs 16x16x32 = W[4 gates][3 K-chunks] . [u | h_prev], 16 gate
// activations = 16 v_exp + 16 v_rcp, results folded into the gates at 1e-30 so nothing is eliminated); bit 1: do not
// load the gate records (their bytes are what a recompute would save; the c_prev record stays).
#ifndef SB_EXP_RECOMPUTE
#define SB_EXP_RECOMPUTE 0
#endif
#include "../../include/sound_bubble_hip.h"


#ifdef SB_PHASE_TIMING
#define SB_TICK(name) const unsigned long long name = __builtin_readcyclecounter()
#else
#define SB_TICK(name) do {} while (0)
#endif

namespace {

constexpr int H = SB_H;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));      // what the transposing LDS read builtin returns
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// Wide BPTT gate records (sb_lstm_fwd_args.rec_f32) as 24-BIT FIXED POINT (round 5): the four post-activation gates of a unit lie in
// [0, 1] (i, f, o) / [-1, 1] (g), where a uniform 24-bit grid -- step 2^-24 resp. 2^-23, error <= half a step -- is as fine as fp32
// is at the top of the range and finer nowhere it matters for a gradient (an ABSOLUTE gate error e changes every product the
// backward forms with it by e times a factor of order one, whatever the gate's own size).  A lane's 16 gate values (4 gates x
// 4 units) travel as 12 dwords = three 16-byte pieces instead of four: 768 instead of 1024 B per position and direction, records
// 1280 -> 1024 B.  Layout per (tile, step, direction) block: [wave][piece 0..2][lane][4 dwords] (one contiguous KB per store /
// load instruction, as before); dwords 3g .. 3g + 2 of the lane hold gate g's four values q0 .. q3 as bytes
// [q0.0 q0.1 q0.2 q1.0 | q1.1 q1.2 q2.0 q2.1 | q2.2 q3.0 q3.1 q3.2].  c_prev stays fp32 (unbounded).
// -DSB_REC_Q24=0: fp32 gate records as in rounds 3 / 4 (A/B builds; sb_lstm_wide_rec_dwords() tells the host which).
#ifndef SB_REC_Q24
#define SB_REC_Q24 1
#endif
constexpr int kWideGateDwords = SB_REC_Q24 ? 3 * SB_H : 4 * SB_H;      // per sequence, step and direction
// (bits of a float BY VALUE: __builtin_bit_cast applied straight to an ext-vector element expression, bit_cast(unsigned, v[k]),
//  read element 0 for every k with this compiler -- found by the record round-trip test)
SB_DEVINL unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
SB_DEVINL unsigned q24_sig(float x) {            // [0, 1] -> round(x 2^24), saturated
  // (round to nearest: x 2^24 is already an integer for x >= 0.5 -- adding 0.5 and truncating rounded every odd one of those UP,
  //  a whole step of error where the code is exact; found by the record test's error bound)
  const unsigned u = (unsigned)__builtin_rintf(x * 16777216.0f);
  return u < 0xFFFFFFu ? u : 0xFFFFFFu;
}
SB_DEVINL unsigned q24_tanh(float x) {           // [-1, 1] -> round(x 2^23) as 24-bit two's complement, saturated at 1 - 2^-23
  const int i = (int)__builtin_rintf(x * 8388608.0f);
  return (unsigned)(i < 8388607 ? i : 8388607);  // (bits above 23 are dropped by the byte permutes below)
}
// Forward side: the gates of a step are parked as their 24-bit CODES (same sixteen registers as the values: q24_codes), and each
// 16-byte piece (0 .. 2) of the lane's record is formed by four byte permutes when it is stored -- few temporaries at a time
// (packing all twelve dwords at the end of the cell update, where the register pressure of the forward step peaks, or quantising
// at store time, cost the two-workgroup bidirectional kernels 7-8 spilled registers).
SB_DEVINL f32x4 q24_codes(const f32x4 g, bool tanh_gate) {
  f32x4 r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = __builtin_bit_cast(float, tanh_gate ? q24_tanh(g[k]) : q24_sig(g[k]));
  return r;
}
SB_DEVINL f32x4 q24_piece(int p, const f32x4 ci, const f32x4 cf, const f32x4 cg, const f32x4 co) {
  auto u = [](float x) { return f2u(x); };
  unsigned e0, e1, e2, e3;
  if (p == 0) {
    e0 = __builtin_amdgcn_perm(u(ci[1]), u(ci[0]), 0x04020100u); e1 = __builtin_amdgcn_perm(u(ci[2]), u(ci[1]), 0x05040201u);
    e2 = __builtin_amdgcn_perm(u(ci[3]), u(ci[2]), 0x06050402u); e3 = __builtin_amdgcn_perm(u(cf[1]), u(cf[0]), 0x04020100u);
  } else if (p == 1) {
    e0 = __builtin_amdgcn_perm(u(cf[2]), u(cf[1]), 0x05040201u); e1 = __builtin_amdgcn_perm(u(cf[3]), u(cf[2]), 0x06050402u);
    e2 = __builtin_amdgcn_perm(u(cg[1]), u(cg[0]), 0x04020100u); e3 = __builtin_amdgcn_perm(u(cg[2]), u(cg[1]), 0x05040201u);
  } else {
    e0 = __builtin_amdgcn_perm(u(cg[3]), u(cg[2]), 0x06050402u); e1 = __builtin_amdgcn_perm(u(co[1]), u(co[0]), 0x04020100u);
    e2 = __builtin_amdgcn_perm(u(co[2]), u(co[1]), 0x05040201u); e3 = __builtin_amdgcn_perm(u(co[3]), u(co[2]), 0x06050402u);
  }
  return f32x4{__builtin_bit_cast(float, e0), __builtin_bit_cast(float, e1), __builtin_bit_cast(float, e2), __builtin_bit_cast(float, e3)};
}
// three dwords -> the four values, each TOP-ALIGNED in its 32 bits (q << 8): no sign extension, and the conversion below is exact
SB_DEVINL void q24_unpack3(unsigned e0, unsigned e1, unsigned e2, unsigned (&q)[4]) {
  q[0] = e0 << 8;
  q[1] = __builtin_amdgcn_perm(e1, e0, 0x0504030Cu);
  q[2] = __builtin_amdgcn_perm(e2, e1, 0x0403020Cu);
  q[3] = e2 & 0xFFFFFF00u;
}
SB_DEVINL void q24_unpack(const f32x4 p0, const f32x4 p1, const f32x4 p2, f32x4& gi, f32x4& gf, f32x4& gg, f32x4& go) {
  unsigned e[12];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    e[k] = f2u(p0[k]); e[4 + k] = f2u(p1[k]); e[8 + k] = f2u(p2[k]);
  }
  unsigned q[4];
  q24_unpack3(e[0], e[1], e[2], q);
#pragma unroll
  for (int k = 0; k < 4; ++k) gi[k] = (float)q[k] * 0x1p-32f;
  q24_unpack3(e[3], e[4], e[5], q);
#pragma unroll
  for (int k = 0; k < 4; ++k) gf[k] = (float)q[k] * 0x1p-32f;
  q24_unpack3(e[6], e[7], e[8], q);
#pragma unroll
  for (int k = 0; k < 4; ++k) gg[k] = (float)(int)q[k] * 0x1p-31f;
  q24_unpack3(e[9], e[10], e[11], q);
#pragma unroll
  for (int k = 0; k < 4; ++k) go[k] = (float)q[k] * 0x1p-32f;
}

struct Split3 { bf16x8 h, m, l; };

SB_DEVINL void split1(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  float r = x - (float)h;
  m = (__bf16)r;
  r -= (float)m;
  l = (__bf16)r;
}
SB_DEVINL Split3 split8(const float (&x)[8]) {
  Split3 s;
#pragma unroll
  for (int k = 0; k < 8; ++k) { __bf16 h, m, l; split1(x[k], h, m, l); s.h[k] = h; s.m[k] = m; s.l[k] = l; }
  return s;
}
SB_DEVINL f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
struct SplitH { h16x8 hi, lo; };            // fp32 = fp16 hi + fp16 lo (22 mantissa bits)
SB_DEVINL SplitH splith8(const float (&x)[8]) {
  SplitH s;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 h = (_Float16)x[k];
    s.hi[k] = h;
    s.lo[k] = (_Float16)(x[k] - (float)h);
  }
  return s;
}

// Operand format of the forward split products.
//   F16 = false: bf16, x = t0 + t1 + t2 (8+8+8 bits), six products, dropped terms <= 2^-24  ("bf16x6", fp32-exact class)
//   F16 = true : fp16, x = t0 + t1      (11+11 bits), three products, dropped term   <= 2^-22  ("fp16x3"): half the
//                MFMAs and a cheaper split; fp16 range is ample for LayerNorm outputs, hidden states in (-1, 1) and
//                the weights, and an underflowing low term costs < 6e-8 absolute.
template <bool F16> struct Prec;
template <> struct Prec<false> {
  typedef __bf16 elem;
  typedef bf16x8 vec8;
  typedef bf16x4 vec4;
  static constexpr int NT = 3;
  static SB_DEVINL f32x4 mma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Prec<true> {
  typedef _Float16 elem;
  typedef h16x8 vec8;
  typedef h16x4 vec4;
  static constexpr int NT = 2;
  static SB_DEVINL f32x4 mma(vec8 a, vec8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <bool F16> struct SplitN { typename Prec<F16>::vec8 t[Prec<F16>::NT]; };      // t[0] = leading term
template <bool F16>
SB_DEVINL void splitn1(float x, typename Prec<F16>::elem (&out)[Prec<F16>::NT]) {
  float r = x;
#pragma unroll
  for (int k = 0; k < Prec<F16>::NT; ++k) {
    out[k] = (typename Prec<F16>::elem)r;
    r -= (float)out[k];
  }
}
template <bool F16>
SB_DEVINL SplitN<F16> splitn8(const float (&x)[8]) {
  SplitN<F16> s;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    typename Prec<F16>::elem e[Prec<F16>::NT];
    splitn1<F16>(x[k], e);
#pragma unroll
    for (int n = 0; n < Prec<F16>::NT; ++n) s.t[n][k] = e[n];
  }
  return s;
}

// Hand-off wait of the time-segmented schedule: thread 0 polls the tile's flag until the predecessor segment has published
// its state.  Bounded: progress depends on all workgroups of the launch being co-resident, which the launcher checks
// against the kernel's occupancy but cannot guarantee on a shared / CU-masked device -- after kSegSpinLimit polls (seconds)
// the watchdog word is set and the whole workgroup leaves; every other waiter sees the word and leaves too, so the
// launch ends with garbage outputs and *status != 0 instead of hanging the process.  Returns false on abort (uniform
// over the workgroup).
constexpr unsigned kSegSpinLimit = kSpinLimit;
SB_DEVINL bool seg_wait(const int* flags, int tile, int seg, int* status, int site) {
  __shared__ int seg_abort;
  if (threadIdx.x == 0) {
    int bad = 0;
    unsigned spins = 0;
    while (sb_poll(flags + tile) < seg) {
      ++spins;
      if (sb_wait_over(status, spins, site, tile, flags + tile, seg)) { bad = 1; break; }
      sb_poll_pause();
    }
    seg_abort = bad;
  }
  __syncthreads();
  return seg_abort == 0;
}

constexpr int UP = 32 + 8;    // padded 16-bit row of the input-term tiles  (80 B)
constexpr int HP16 = 64 + 8;  // padded 16-bit row of the hidden-term tiles (144 B)

template <int C>
struct XVec { float v[C / 16]; };

// LIN (single-direction passes): the Linear(64 -> C) + residual that follows the LSTM is applied in the kernel,
// y[p] = x[p] + W_lin h[p] + b_lin, one step behind the recurrence from the hidden-state tiles that are in LDS anyway
// (waves w < C/16 own channel tile w); hs is then only written when the caller wants it (training).
// SEG (single-direction passes with more tiles than CUs): the time axis of every tile is cut into a.seg_count
// segments and the (tile, segment) items are dealt round-robin to one resident workgroup per CU; a segment starts from
// the (h, c) state its predecessor left in a.seg_state, published through a.seg_flags (release / acquire at agent
// scope).  A tile is an indivisible serial chain, and a second co-resident tile costs ~1.8x, so 290 tiles on 256 CUs
// run 1.8 T with most CUs idle half the time; cut into k segments the makespan is ceil(290 k / 256) / k ~ 1.14 T.
// Item i = segment * ntiles + tile goes to workgroup i mod W in increasing order; its predecessor i - ntiles lies in an
// earlier round (ntiles >= W), so every wait is on an item some resident workgroup is already past or working on.
// SUM3 (single-direction passes that follow a bidirectional pass in partial-Linear mode): the input row is
// x[p] + x_part[p, 0, :] + x_part[p, 1, :] -- the residual and the two directions' halves of the intra-frame Linear --
// summed by the loader as it fetches the row (the separate elementwise pass, 16 C bytes per position, is gone); the sum
// is written once to x_sum for the backward kernels (training) and handed to the fused Linear's residual through a
// four-row LDS ring (the residual is needed two barriers after the row was normalised).
// Overlapped forward (sb_lstm_fwd_produce / sb_lstm_fwd_consume): an inter-frame pass with fewer tiles than CUs publishes
// its y rows (write-through, sc1) and counts itself into slab_flags[k] after every slab_len time steps (runtime flag:
// a.slab_flags on a single-direction LIN launch); the NEXT block's intra-frame pass (ORD: 1-D grid, direction = workgroup
// parity, tiles tile_order[0 .. ntiles) sorted by the latest time slab their 16 frames need, drawn by the workgroups of BOTH
// launches of the pass from one atomic counter per direction) starts on the idle CUs and each item waits for
// slab_flags[tile_need[i]] to reach slab_need.  An input row is one 128-byte line that only its frame's items ever read,
// so no line of an unfinished slab enters the reader's L2.  Guarded launch (ord_guard, the one that runs NEXT to the
// producer): a workgroup that does not find every producer workgroup started within ~50 us leaves at once -- should the
// dispatcher have placed this launch first, it must not sit on the CUs the producer needs.
SB_DEVINL void st4_sc1(float* p, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

}  // namespace

// defined in sb_lstm_stream.hip: dW_ih / dW_hh / db += sum over `rows` partial rows of [256 * (C + 64) + 256] floats
int sb_launch_stream_reduce(const float* partials, int rows, int64_t ld, int C, float* dW_ih, float* dW_hh, float* db_ih,
                            float* db_hh, hipStream_t st, int n_extra = 0, const int* ex_off = nullptr,
                            const int* ex_n = nullptr, float* const* ex_out = nullptr);

// launch helpers used by sb_lstm.hip's C entry points (same argument structs)
// Number of (tile, time-segment) work items per workgroup slot: pick the segment count k that minimises the makespan
// ceil(ntiles * k / W) / k (in units of one tile's serial time) plus a small per-hand-off cost.
static int choose_segments(int ntiles, int W, int S, double* cost_out) {
  int best = 1;
  double best_cost = (double)((ntiles + W - 1) / W);
  for (int k = 2; k <= 16 && S / k >= 24; ++k) {
    const double cost = (double)(((long)ntiles * k + W - 1) / W) / k + 0.006 * k;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = k; }
  }
  *cost_out = best_cost;
  return best;
}
// one resident workgroup per CU is what the segmented schedule relies on: refuse it when the kernel does not fit a CU
template <auto Kern, int BS = 256>
static bool fits_one_per_cu() {
  static const bool ok = [] {
    int n = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, Kern, BS, 0) == hipSuccess && n >= 1;
  }();
  return ok;
}
static int device_cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus;
  }();
  return n;
}
#endif
