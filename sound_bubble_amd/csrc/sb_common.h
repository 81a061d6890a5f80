// Shared device helpers for the Sound-Bubble gfx950 kernels.
//
// MFMA convention used everywhere ("transposed form"):
//   D[16 x 16] += A[16 x 4] * B[4 x 16]   via v_mfma_f32_16x16x4_f32 (exact fp32 fma chain)
//   A operand : lane l holds A[i = l & 15][k = l >> 4]
//   B operand : lane l holds B[k = l >> 4][j = l & 15]
//   C/D       : lane l holds D[row = 4*(l >> 4) + r][col = l & 15], r = 0..3
// Rows of D are output features (gate units / channels), columns are the 16
// positions (sequences) a wave owns, so the weight matrix is the A operand and
// can stay resident in registers or LDS, and a lane ends up with 4 consecutive
// features of ONE position -> 16-byte coalesced loads/stores of activations.
// The K dimension is consumed in chunks of 16 with the permutation
//   k(chunk m, lane-quad q, sub-step r) = 16 m + 4 q + r
// so that both operands are fetched as one float4 per lane per chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SB_DEVINL __device__ __forceinline__

// One poll of a flag another workgroup raises (bounded waits of the guarded schedules).  -DSB_EXP_POLL_RMW (developer A/B): a
// returning read-modify-write, which executes at the point of coherence, instead of an sc1 load, which this XCD's L2 serves.
SB_DEVINL int sb_poll(const int* p) {
#ifdef SB_EXP_POLL_RMW
  return __hip_atomic_fetch_or(const_cast<int*>(p), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Watchdog word of the guarded schedules (sb_*_args.sched_status): 0 = fine.  The FIRST bounded wait that gives up leaves a
// code saying which one it was -- site << 28 | timed_out << 27 | index << 14 | value seen << 7 | value wanted (include/
// sound_bubble_hip.h: SB_TRIP_*; timed_out = 0: the waiter left because it found the word already set) -- every later waiter
// finds the word set and leaves without touching it.  -DSB_TRIP_DEBUG (developer builds; the word then needs 8 ints, as the
// library's own flag arena gives it): [1] polls done, [2] the word's value as the waiter read it, [3] XCC id << 16 | workgroup.
// The pause between two polls of a bounded wait.  Round 5, measured (scripts/stress_train_loop.py, -DSB_TRIP_DEBUG builds): about
// once in 5 000-10 000 train steps an overlapped FORWARD producer stops making progress while ~170 single-lane consumer pollers
// wait on its slab counter -- the counters, read back by returning atomics from the timing-out waiter, stand still in memory
// itself (no stale cache line: the words are uncached), tiles frozen at different steps -- and it resumes the moment the pollers
// leave: with the watchdog at 2^22 polls the freeze lasts 0.5 s, at 2^24 polls 16.3 s, every time.  One hot word is incremented by
// all 82 producer tiles and polled by every waiter; the cross-pass backward, whose producer tiles each own a word, never showed it
// in 300 000 launches -- yet a forward build with one progress word per producer tile showed the same events.  A slower poll
// (s_sleep 32) and the poll flavour (sc1 load / returning atomic) did not lower the rate of the event either: cause not established.
// What the library does about it: a forward consumer whose wait runs out (~2 ms) hands its item back to the launch behind the
// producer (sb_lstm_bf_fwd.hip: ord_next), so the event costs a boundary ~2 ms and nothing else; polls are ~3 us apart (a slab
// takes ~40 us).
// SB_POLL_SLEEP: the s_sleep operand (64 clocks each); the watchdog's poll budget scales with it (~2 s).
#ifndef SB_POLL_SLEEP
#define SB_POLL_SLEEP 100
#endif
#ifdef SB_SPIN_LIMIT_LOG2
constexpr unsigned kSpinLimit = 1u << SB_SPIN_LIMIT_LOG2;
#else
constexpr unsigned kSpinLimit = (3u << 24) / (SB_POLL_SLEEP > 0 ? SB_POLL_SLEEP : 1);     // polls before a bounded wait gives up (~2 s)
#endif
// ... and of a wait whose item can be handed back (the overlapped forward's consumer next to its producer: giving up costs its help
// only): ~2 ms, two producer passes -- an item legitimately waits for at most one.  (A first value of ~6 ms made one event cost a
// forward pass up to 5 boundaries x 6 ms: seen once inside a 10-step forward-only bench region, 6.7 ms median, 9.1 ms mean.)
#ifndef SB_HELP_DIV
#define SB_HELP_DIV 1200
#endif
constexpr unsigned kHelpSpinLimit = kSpinLimit / SB_HELP_DIV > 64 ? kSpinLimit / SB_HELP_DIV : 64;
// overlapped forward: the control block between the four control words and the slab flags (see lstm_fwd_bf_kernel: ord_next)
constexpr int kOrdRet = 256;                         // returned items per direction (at most one per workgroup of the side launch)
constexpr int kOrdCtl = 8 + 2 * kOrdRet;             // ints
SB_DEVINL void sb_poll_pause() {
#if SB_POLL_SLEEP > 0
  __builtin_amdgcn_s_sleep(SB_POLL_SLEEP);
#endif
}
SB_DEVINL void sb_trip(int* status, int site, int index, int seen, int want, unsigned spins, int status_seen) {
  int expected = 0;
  const int code = (site << 28) | ((spins > kSpinLimit ? 1 : 0) << 27) | ((index & 0x1FFF) << 14) | ((seen & 0x7F) << 7) | (want & 0x7F);
  const bool first = __hip_atomic_compare_exchange_strong(status, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SB_TRIP_DEBUG
  if (first) {
    __hip_atomic_store(status + 1, (int)spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(status + 2, status_seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(status + 3, (int)((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16) | (blockIdx.x & 0xFFFF)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#else
  (void)first; (void)status_seen;
#endif
}
// one look at the watchdog word from inside a bounded wait (every 64th poll): -> true when the wait must end (and the word is set)
SB_DEVINL bool sb_wait_over(int* status, unsigned spins, int site, int index, const int* flag, int want) {
  if ((spins & 63u) != 0) return false;
#ifdef SB_TRIP_DEBUG
  if ((spins & 0xFFFFu) == 0)        // the longest wait seen so far, in polls (granularity 65 536): [52]
    __hip_atomic_fetch_max(status + 52, (int)spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  const int sv = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (spins <= kSpinLimit && sv == 0) return false;
#ifdef SB_TRIP_DEBUG
  if (sv == 0) {      // the timing-out waiter itself: the flag and its neighbours read two ways -- sc1 load vs returning read-modify-write
    const int ld = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int rmw = __hip_atomic_fetch_or(const_cast<int*>(flag), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ld2 = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(status + 4, ld, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(status + 5, rmw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(status + 6, ld2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int* base = flag - index;                  // slab / tile flag 0 of the array (index = offset of `flag` in it)
    for (int i = -4; i < 36; ++i) {                  // control words (-4 .. -1) and the first 36 flags, by RMW
      const int v = __hip_atomic_fetch_or(const_cast<int*>(base + i), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(status + 12 + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(status + 7, (int)(__builtin_readcyclecounter() >> 10), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
  sb_trip(status, site, index, __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want, spins, sv);
  return true;
}

SB_DEVINL f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc += A4 (4 k-substeps of the A operand) x B4 (matching 4 k-substeps of B)
SB_DEVINL f32x4 mfma16x4(const f32x4 a, const f32x4 b, f32x4 c) {
  c = mfma16(a[0], b[0], c);
  c = mfma16(a[1], b[1], c);
  c = mfma16(a[2], b[2], c);
  c = mfma16(a[3], b[3], c);
  return c;
}

SB_DEVINL f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
SB_DEVINL void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
// write-once / read-once streams (the blocked BPTT records: one contiguous KB per wave instruction): non-temporal hint --
// +1.2 % on the big train step (same box, round 4: the overlapped intra-frame consumer 0.59 -> 0.52 ms; round 2 had measured
// -20 % with the position-major record layout, whose 64-byte pieces no longer combined in L2)
SB_DEVINL f32x4 ld4_rec(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
// read-once input streams of the streaming passes (ln_film_bwd, ln_bwd): non-temporal; -DSB_EXP_NO_NT_STREAMS: plain loads (A/B builds)
#ifdef SB_EXP_NO_NT_STREAMS
SB_DEVINL f32x4 ld4_once(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
SB_DEVINL float ld1_once(const float* p) { return *p; }
#else
SB_DEVINL f32x4 ld4_once(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
SB_DEVINL float ld1_once(const float* p) { return __builtin_nontemporal_load(p); }
#endif
SB_DEVINL void st4_rec(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }
// side outputs only the backward reads (u / hs pairs, x_sum, y_pre): experiment switch
#ifdef SB_EXP_NT_SIDE
template <class T> SB_DEVINL void st_side(T* p, T v) { __builtin_nontemporal_store(v, p); }
#else
template <class T> SB_DEVINL void st_side(T* p, T v) { *p = v; }
#endif
SB_DEVINL f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

// v_exp_f32 / v_rcp_f32 (1 ulp each): absolute error ~1e-7, far inside the 1e-3 parity bar
// (v_exp_f32 is 2^x: the log2(e) factor and the sign are one multiply -- or none at all when the caller has folded
//  them into the weights: *_pre take z = -log2(e) * x  resp.  z = -2 log2(e) * x)
constexpr float SB_NLOG2E = -1.4426950408889634f;
SB_DEVINL float sigmoid_pre(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)); }
SB_DEVINL float tanh_pre(float z) { return __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z)), -1.0f); }
SB_DEVINL float sigmoidf_fast(float x) { return sigmoid_pre(SB_NLOG2E * x); }
SB_DEVINL float tanhf_fast(float x) { return tanh_pre(2.0f * SB_NLOG2E * x); }

// sum over the 4 lanes {l, l^16, l^32, l^48} (same l & 15): the feature quads of one position
SB_DEVINL float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
// sum over the 16 lanes sharing l >> 4 (all positions of a tile): DPP butterflies (VALU latency) instead of
// ds_bpermute shuffles (LDS crossbar round trips): quad xor-1, quad xor-2, half-row mirror, row mirror.
template <int CTRL>
SB_DEVINL float dpp_add(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
  return v + __int_as_float(t);
}
SB_DEVINL float row16_sum(float v) {
  v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);   // row_half_mirror
  v = dpp_add<0x140>(v);   // row_mirror
  return v;
}
SB_DEVINL float row8_sum(float v) {   // sum over the 8 lanes sharing l >> 3
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  return v;
}
SB_DEVINL float wave_sum(float v) {
  v = row16_sum(v);
  return quad_sum(v);
}

#define SB_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return -(int)e__;                \
  } while (0)
