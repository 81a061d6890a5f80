// Forward recurrence of the LSTM passes on the 16-bit matrix pipes with split fp32 operands (see sb_lstm_bf_common.h for the
// arithmetic): lstm_fwd_bf_kernel and its launcher.  Backward: sb_lstm_bf_bwd.hip.
#ifdef SB_FWD_2P
#undef SB_PHASE_TIMING                                  // (the phase tables belong to the default build of this file)
#endif
#include "sb_lstm_bf_common.h"

// Phase timing (developer tool): build with -DSB_PHASE_TIMING and pass a scratch buffer (save_u with save_gates == NULL) --
// lane 0 of every wave of the first 4 tiles writes the average s_memtime cycles per step of each phase.
// scripts/phase_timing.py prints them.
#ifdef SB_PHASE_TIMING
// the last launch of every forward kernel kind also leaves its table here (any SAVE mode: training steps of the whole model);
// kind 0 plain, 1 fused Linear, 2 summed input + fused Linear (the inter-frame producer), 3 ordered consumer, 4 bidirectional
// partial-Linear; read with sb_debug_phase_fwd()
__device__ float g_phase_fwd[5][16][8];
extern "C" int sb_debug_phase_fwd(float* host_out) {
  const int rc = -(int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_fwd), sizeof(g_phase_fwd));
  static float zeros[5 * 16 * 8];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_phase_fwd), zeros, sizeof(zeros));      // the next read shows only what ran since
  return rc;
}
#endif

#ifndef SB_Q24_LATE
#define SB_Q24_LATE 1
#endif
#ifndef SB_EPI_EARLY
#define SB_EPI_EARLY 1
#endif
namespace {

// The bidirectional C = 32 passes with the fused partial Linear (first block's intra-frame pass, ordered consumer) are meant to run
// two workgroups per CU: at most 256 registers.  The inference forms stay below on their own (247 / 245); the training form of the
// plain pass sat at exactly 256 and any edit of this file pushed it over: capped (second __launch_bounds__ argument = waves per
// SIMD; the four registers it spills are outside the time loop).
template <int C, int SAVE, bool FULL, bool F16, bool LIN, bool SEG, bool SUM3 = false, bool ORD = false>
__global__ __launch_bounds__(256, (LIN && C == 32 && !SUM3 && F16 && !SEG && SAVE == 4 && !ORD) ? 2 : 1)
void lstm_fwd_bf_kernel(sb_lstm_fwd_args a) {
  typedef Prec<F16> PR;
  typedef typename PR::elem elem;
  typedef typename PR::vec8 vec8;
  typedef typename PR::vec4 vec4;
  constexpr int NT = PR::NT;
  constexpr int VPT = C / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = ORD ? (int)blockIdx.x & 1 : (int)blockIdx.y;
  const int S = a.nsteps;
  const bool rev = dir == 1;
  const bool prod = LIN && !ORD && !SEG && a.slab_flags != nullptr;      // producer side of the overlapped forward
  if (prod && tid == 0) __hip_atomic_fetch_add(a.ord_started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __shared__ int ord_item;
  // next (tile, this direction) item; uniform over the workgroup.  The draw is one atomic add on the direction's counter (items are
  // sorted by the producer slab they need).  NEXT to the producer (ord_guard != 0) the workgroup then waits for that slab -- and a
  // wait that runs out HANDS THE ITEM BACK (per-direction return stack in the control block a.ord_ret) instead of dropping it;
  // the launch BEHIND the producer (ord_guard == 0, every slab complete) drains the counter, then the return stacks, and leaves
  // only when no workgroup of the other launch still holds an unprocessed item ([0] of the block).  So whatever happens to a waiting
  // workgroup -- about once in 10 000 train steps a producer stands still for as long as its pollers wait (sb_common.h,
  // SB_POLL_SLEEP) -- every item is processed exactly once and the outputs are those of the plain order; the event costs its step
  // the ~2 ms of the help timeout and counts itself into *ord_giveups (nullable).  (Round 4 dropped the item: garbage activations
  // behind a watchdog word read once per epoch.  A first round-5 form claimed an item only once its slab was complete -- nothing
  // to hand back -- and serialised the claims of a slab on one compare-and-swap word: forward-only 2 295 -> 2 020 utt/s, slower
  // than no overlap at all.)
  // control block (ints, zeroed with the flags): [0] items held by waiting workgroups, [1 + dir] pushed, [3 + dir] popped,
  // [8 + dir * kOrdRet + k] returned item k of the direction, stored as item + 1 (0 = not written yet)
  auto ord_next = [&]() -> int {
    if (tid == 0) {
      const int nt = (a.nseq + 15) / 16;
      int it = __hip_atomic_fetch_add(a.ord_counter + dir, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if constexpr (ORD) {
        int* const ctl = a.ord_ret;
        if (a.ord_guard && it < nt) {
          __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int* fl = a.slab_flags + a.tile_need[it];
          unsigned spins = 0;
          bool ok = true;
          int hold_dec = -1;
          while (sb_poll(fl) < a.slab_need) {
            if (++spins > kHelpSpinLimit) { ok = false; break; }
            sb_poll_pause();
          }
          if (!ok) {                                   // hand the item back, stop helping
            const int k = __hip_atomic_fetch_add(ctl + 1 + dir, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int old = -1;
            if (k < kOrdRet) old = __hip_atomic_exchange(ctl + 8 + dir * kOrdRet + k, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else sb_trip(a.sched_status, SB_TRIP_FWD_CONSUMER, it, k, kOrdRet & 0x7F, kSpinLimit + 1, 0);   // (cannot happen: <= one per side workgroup)
            if (a.ord_giveups) __hip_atomic_fetch_add(a.ord_giveups, 1 + (old > 0 ? 1 << 20 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            it = nt;
            // ORDER (round 6): the drain side may leave once it reads hold == 0 and an unchanged `pushed`, so the push
            // must have been PERFORMED before the release of the hold below is issued.  Every step is a returning atomic on an
            // uncached word whose result the next step needs -- pushed++ returns k, the slot address; the exchange returns `old`
            // -- and the empty asm makes the decrement's operand depend on `old` for the compiler as well (with ord_giveups ==
            // NULL nothing else used it): no fence, no L2 write-back, just the data dependence spelled out.
            asm volatile("" : "+v"(hold_dec), "+v"(old));
          }
          __hip_atomic_fetch_add(ctl, hold_dec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (!a.ord_guard && it >= nt) {         // counter exhausted: what did the launch next to the producer hand back?
          unsigned spins = 0;
          for (;;) {
            const int t = __hip_atomic_load(ctl + 3 + dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int n = __hip_atomic_load(ctl + 1 + dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < n && t < kOrdRet) {
              int expected = t;
              if (!__hip_atomic_compare_exchange_strong(ctl + 3 + dir, &expected, t + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT)) continue;
              int v = 0;
              while ((v = sb_poll(ctl + 8 + dir * kOrdRet + t)) == 0 && ++spins <= kSpinLimit) sb_poll_pause();
              if (v > 0) { it = v - 1; break; }
              sb_trip(a.sched_status, SB_TRIP_FWD_CONSUMER, t, 0, 1, spins, 0);      // a push that never landed: fatal
              break;
            }
            // (the re-read of `pushed` is ISSUED only after `hold` has come back as 0: the asm pins the first load's result in
            //  front of the second load for the compiler; the hardware returns a wave's loads in issue order)
            int held = __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("" : "+v"(held) : : "memory");
            if (held == 0 && __hip_atomic_load(ctl + 1 + dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n) break;   // nobody holds, nothing new
            if (++spins > kSpinLimit) { sb_trip(a.sched_status, SB_TRIP_FWD_CONSUMER, n, t, 0, spins, 0); break; }
            sb_poll_pause();
          }
        }
      }
      ord_item = it;
    }
    __syncthreads();
    return ord_item;
  };
  int ord_first = 0;
  if constexpr (ORD) {
    if (a.ord_guard) {
      if (tid == 0) {
        int ok = 0;
        for (int i = 0; i < 200 && !ok; ++i) {
          ok = __hip_atomic_load(a.ord_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.slab_need;
          if (!ok) __builtin_amdgcn_s_sleep(8);
        }
        ord_item = ok;
      }
      __syncthreads();
      if (!ord_item) return;
      __syncthreads();
    }
    ord_first = ord_next();                           // before the weights are fetched: most late workgroups find nothing
    if (ord_first >= (a.nseq + 15) / 16) return;
  }

  __shared__ __attribute__((aligned(16))) elem U16[2][NT][16][UP];      // [buf][term][seq][channel]
  __shared__ __attribute__((aligned(16))) elem H16[2][NT][16][HP16];    // [buf][term][seq][unit]
  __shared__ __attribute__((aligned(16))) float Bias[4][H];
  // TWO (kernels with the fused Linear): two-stage loader.  The y epilogue runs on the C / 16 waves that own an output-channel
  // tile, and with one barrier per step the others wait for them (phase table r03: 320-400 of 2 070 ticks per step).  So the
  // LayerNorm + operand split of the input rows moves to waves WITHOUT a tile, and into fewer, wider passes:
  // stage 1 (every wave, 4 sequences each, as before): the fetched (summed) row of step s + 3 goes raw into a four-row LDS ring;
  // stage 2 (one barrier later): the loader waves -- 2 and 3 for C = 32 (8 sequences each), 3 for C = 16 (all 16) -- read the rows
  // back FOUR channels per lane (C / 4 lanes per row), normalise, split and store the operand tiles of step s + 2.
  // Measured (r04 phase table): a pass of the 16-lanes-per-row LayerNorm costs 200-270 ticks however few rows it holds -- its
  // price is instructions, not lanes -- so four such passes (one per wave) cost every wave as much as the y epilogue costs the
  // tile owners; two passes of twice the width on the waves that have no epilogue take the LayerNorm off the tile owners'
  // step altogether.  The ring doubles as the residual's source for the fused Linear (it always did in the summed-input form).
  // Not for the bidirectional C = 32 passes (LIN without SUM3: the first block's intra-frame pass and the ordered consumer): they
  // run TWO workgroups per CU -- 253 / 238 registers -- and the two-stage loader's state would cost them that (measured: ordered
  // consumer 0.31 -> 0.41 ms at 264 registers); with a second wave on every SIMD the imbalance is the other workgroup's gain.
  // The two loaders form bit-identical operand tiles (same summation trees: see ln2_piece).
  constexpr bool HEAVY = !(LIN && C == 32 && !SUM3);
  constexpr bool TWO = LIN && HEAVY;
  __shared__ __attribute__((aligned(16))) float XS[TWO ? 4 : 1][TWO ? 16 : 1][TWO ? C + 4 : 1];      // raw input rows (ring)

  // ---- weights -> registers, split once: Wt[gate][chunk]: rows g*64+16w+j, k = 8q..8q+7 of the chunk ----
  const float* __restrict__ wih = a.w_ih[dir];
  const float* __restrict__ whh = a.w_hh[dir];
  // The activations are evaluated as rcp(1 + 2^z): the factors z = -log2(e) x (sigmoid gates i, f, o) and
  // z = -2 log2(e) x (tanh gate g) are folded into the weight and bias rows here, once, instead of a multiply
  // per gate value and step.
  SplitN<F16> Wt[4][3];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = g * H + 16 * w + j;
    const float gsc = (g == 2 ? 2.0f : 1.0f) * SB_NLOG2E;
    float t[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) t[kk] = (8 * q + kk < C) ? gsc * wih[(size_t)row * C + 8 * q + kk] : 0.f;
    Wt[g][0] = splitn8<F16>(t);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = gsc * whh[(size_t)row * H + 32 * c + 8 * q + kk];
      Wt[g][1 + c] = splitn8<F16>(t);
    }
  }
  if (tid < 4 * H) Bias[tid >> 6][tid & 63] = ((tid >> 6) == 2 ? 2.0f : 1.0f) * SB_NLOG2E * (a.b_ih[dir][tid] + a.b_hh[dir][tid]);
  const bool linw = LIN && w < C / 16;                  // this wave owns output channels 16w .. 16w+15 of y
  // bidirectional passes (a.ndir == 2): PARTIAL mode -- each direction writes its half of the Linear(128 -> C),
  // y[p, dir, :] = W_lin[:, dir*64 .. +63] . h_dir[p] (+ b_lin in direction 0), no residual: the two directions visit a
  // position at different times in different workgroups, so the sum x + y[p,0] + y[p,1] is a cheap elementwise pass
  // (sb_add3) instead of a pass over hs [P, 128] fp32 -- and hs itself only travels as the fp16 side output.
  const bool lin_part = LIN && a.ndir == 2;
  SplitN<F16> Wl[2];
  f32x4 lbias = zero4(), yacc = zero4(), xres = zero4();
  if constexpr (LIN) {
    const int ldl = a.ndir * H;
#pragma unroll
    for (int ck = 0; ck < 2; ++ck) {
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = linw ? a.lin_w[(size_t)(16 * w + j) * ldl + dir * H + 32 * ck + 8 * q + kk] : 0.f;
      Wl[ck] = splitn8<F16>(t);
    }
    if (linw && dir == 0) lbias = ld4(a.lin_b + 16 * w + 4 * q);
  }
  // zero the padded channels of the input tiles once (C = 16: channels 16..31 stay zero)
  for (int i = tid; i < 2 * NT * 16 * UP; i += 256) (&U16[0][0][0][0])[i] = (elem)0.f;
  __syncthreads();

  // ---- loader role ----
  const int ls = tid >> 4, cpart = tid & 15;
  bool lvalid = false, cvalid = false;          // per work item (tile): set by set_tile()
  int64_t lbase = 0, cbase = 0;
  // Per-lane element offsets of the tile, fixed at set_tile(): a per-step address is  pointer + lane offset + (uniform step
  // term), i.e. one scalar multiply and a 64-bit add.  Formed as (lane base + st * p_step) * row width they cost a
  // quarter-rate 64-bit multiply chain per address and step -- VALU time, which on this chip adds to the matrix time of
  // the wave (microbenchmarks in scripts/micro/: a SIMD runs MFMA and VALU instructions back to back, never side by side).
  int64_t lo_x = 0, lo_xp = 0, co_y = 0, co_h = 0;
  const int ndir = a.ndir;
  // stage-2 rows of this wave (TWO): lane -> sequence l2s, channels 4 l2c .. + 3
  constexpr int LPR = C / 4;                                          // lanes per row
  const bool lw2 = TWO && (C == 32 ? w >= 2 : w == 3);                // loader wave
  const int l2c = lane & (LPR - 1);
  const int l2s = C == 32 ? 8 * (w & 1) + (lane >> 3) : lane >> 2;
  bool lvalid2 = false;
  int64_t lo_u2 = 0;
  // ... and the uniform step terms are RUNNING sums: one 64-bit scalar add per step and quantity instead of a 64-bit scalar
  // multiply chain per address (with one wave per SIMD every instruction, scalar ones included, costs its ~4 issue cycles:
  // the ~70 scalar instructions of the old epilogue were 290 ticks of a 2 070-tick step).  Row s of the walk:
  //   run_x = st(s) p_step C (x, x_sum, u, y, y_pre, residual), run_h = st(s) p_step ndir 64 (hs), run_blk = record block;
  // set by walk_begin(), advanced at the end of every step.
  int64_t run_x = 0, run_h = 0, run_blk = 0, run_x_last = 0, pend_h = 0, pend_blk = 0;
  const int64_t d_x = (rev ? -1 : 1) * a.p_step * C, d_h = (rev ? -1 : 1) * a.p_step * (ndir * H), d_blk = rev ? -ndir : ndir;
  int nc = 0;
  int64_t rec_tile = 0;                         // tile * S: compact records are blocked per (tile, step, direction)
  // FiLM of the NEXT block (dis_embd3 :509-513) in the y epilogue: y <- y * film_w[n] + film_b[n] with the planes of
  // sequence n = (b, f) -- constant along this kernel's time walk, so eight registers per lane, loaded once per tile.
  // The pre-FiLM value goes to y_pre when the backward needs it (training).  Replaces a pass over [B, T, F, C].
  const bool film = LIN && a.film_w != nullptr;
  f32x4 fw = {1.f, 1.f, 1.f, 1.f}, fb = zero4();
  auto set_tile = [&](int tile) {
    rec_tile = (int64_t)tile * S;
    const int nl = tile * 16 + ls;
    lvalid = FULL || nl < a.nseq;
    lbase = lvalid ? ((int64_t)(nl / a.n_inner) * a.p_outer + (int64_t)(nl % a.n_inner) * a.p_inner) : 0;
    nc = tile * 16 + j;
    cvalid = FULL || nc < a.nseq;
    cbase = cvalid ? ((int64_t)(nc / a.n_inner) * a.p_outer + (int64_t)(nc % a.n_inner) * a.p_inner) : 0;
    lo_x = lbase * C + cpart * VPT;
    lo_xp = lbase * (2 * C) + cpart * VPT;
    if constexpr (TWO) {
      const int n2 = tile * 16 + l2s;
      lvalid2 = FULL || n2 < a.nseq;
      lo_u2 = (lvalid2 ? ((int64_t)(n2 / a.n_inner) * a.p_outer + (int64_t)(n2 % a.n_inner) * a.p_inner) : 0) * C + 4 * l2c;
    }
    co_y = ((LIN && a.ndir == 2) ? cbase * 2 + dir : cbase) * C + 16 * w + 4 * q;
    co_h = (cbase * ndir + dir) * H + 16 * w + 4 * q;
    if (film && linw && cvalid) {
      fw = ld4(a.film_w + (size_t)nc * C + 16 * w + 4 * q);
      fb = ld4(a.film_b + (size_t)nc * C + 16 * w + 4 * q);
    }
  };
  float gam[VPT], bet[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) { gam[v] = a.ln_g[cpart * VPT + v]; bet[v] = a.ln_b[cpart * VPT + v]; }

  auto load_x = [&](int s) {
    XVec<C> r;
    const int st = rev ? S - 1 - s : s;
    const int64_t sp = (int64_t)st * a.p_step;        // uniform
    const float* p = a.x + sp * C + lo_x;
#pragma unroll
    for (int v = 0; v < VPT; ++v) r.v[v] = (SB_EXP_SKIP & 64) ? (float)((s + v + lane) & 7) * 0.25f : (lvalid ? p[v] : 0.f);
    if constexpr (SUM3) {
      const float* p0 = a.x_part + sp * (2 * C) + lo_xp;
#pragma unroll
      for (int v = 0; v < VPT; ++v) r.v[v] = lvalid ? (r.v[v] + p0[v]) + p0[C + v] : 0.f;      // (x + part0) + part1, as sb_add3
    }
    return r;
  };
  // LayerNorm + operand split of one input row piece: values xr of sequence `seq` (16 lanes x VPT channels), position offset
  // `lo`; operand terms to U16[buf], the saved copy of u (training) to row offset ex
  auto ln_row = [&](const float (&xr)[VPT], int seq, bool lv, int64_t lo, int buf, int64_t ex) {
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) sum += xr[v];
    const float mean = row16_sum(sum) * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) { const float d = xr[v] - mean; sq = __builtin_fmaf(d, d, sq); }
    const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(row16_sum(sq), 1.0f / C, 1e-5f));   // v_rsq_f32, 1 ulp
    float u[VPT];
    elem uterm[2][VPT];                            // the two leading terms (SAVE == 4 stores them)
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      u[v] = __builtin_fmaf((xr[v] - mean) * rstd, gam[v], bet[v]);
      // (u is rounded to fp32 HERE: left alone, hipcc folds the fma into the fp16 conversion of the leading term in some
      //  instantiations -- v_fma_mixlo_f16, one rounding instead of two -- and the two loaders' operand tiles differ in rare bits)
      asm volatile("" : "+v"(u[v]));
      elem e[NT];
      splitn1<F16>(u[v], e);
#pragma unroll
      for (int n = 0; n < NT; ++n) U16[buf][n][seq][cpart * VPT + v] = e[n];
      uterm[0][v] = e[0]; uterm[1][v] = e[1];
    }
    if (SAVE && lv && dir == 0 && !(SB_EXP_SKIP & 8)) {      // both directions normalise the same rows: one copy is enough
      const int64_t uo = ex + lo;
      if constexpr (SAVE == 3) {            // only the streaming backward reads u, as a single fp16 term
        _Float16* p = reinterpret_cast<_Float16*>(a.save_u) + uo;
        if constexpr (VPT == 2) *reinterpret_cast<h16x2*>(p) = h16x2{(_Float16)u[0], (_Float16)u[1]};
        else p[0] = (_Float16)u[0];
      } else if constexpr (SAVE == 4 && F16) {
        // wide form: u travels as the fp16 hi + lo terms this kernel's own products use (same bytes as fp32, and the
        // backward kernels take them as matrix operands without a split): [P][C/2][hi0, hi1, lo0, lo1] resp. [P][C][hi, lo]
        _Float16* p = reinterpret_cast<_Float16*>(a.save_u) + 2 * uo;
        if constexpr (VPT == 2) st_side(reinterpret_cast<h16x4*>(p), h16x4{(_Float16)uterm[0][0], (_Float16)uterm[0][1], (_Float16)uterm[1][0], (_Float16)uterm[1][1]});
        else st_side(reinterpret_cast<h16x2*>(p), h16x2{(_Float16)uterm[0][0], (_Float16)uterm[1][0]});
      } else {
        float* p = a.save_u + uo;
#pragma unroll
        for (int v = 0; v < VPT; ++v) p[v] = u[v];
      }
    }
  };
  // one-stage loader (!TWO): the fetched row of step s straight through the LayerNorm
  auto ln_store = [&](const XVec<C>& xv, int buf, int s, int64_t ex) {      // ex = st(s) p_step C (uniform)
    ln_row(xv.v, ls, lvalid, lo_x, buf, ex);
  };
  // two-stage loader, stage 1: the fetched (summed) row of step s -> ring slot s & 3 (and x_sum for the backward kernels)
  auto raw_store = [&](const XVec<C>& xv, int s, int64_t ex) {
    if constexpr (TWO) {
#pragma unroll
      for (int v = 0; v < VPT; ++v) XS[s & 3][ls][cpart * VPT + v] = xv.v[v];
      if constexpr (SUM3) {
        if (a.x_sum && lvalid) {
          float* p = a.x_sum + ex + lo_x;
#pragma unroll
          for (int v = 0; v < VPT; ++v) st_side(p + v, xv.v[v]);
        }
      }
    }
  };
  // ... stage 2 (loader waves; the row has been in the ring since before the last barrier): LPR lanes x 4 channels per row
  f32x4 gam4 = zero4(), bet4 = zero4();
  if constexpr (TWO) { gam4 = ld4(a.ln_g + 4 * l2c); bet4 = ld4(a.ln_b + 4 * l2c); }
  auto rowsum2 = [&](float v) -> float {
    v = dpp_add<0xB1>(v);                          // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);                          // quad_perm [2,3,0,1]
    if constexpr (LPR == 8) v = dpp_add<0x141>(v); // row_half_mirror
    return v;
  };
  // The pass is cut into SIX pieces that the hidden part issues one behind each of its product groups (h_part / mma6, like the
  // deferred record stores): the pass is one dependent chain -- LDS read, sum, three DPP steps, deviations, three DPP steps,
  // v_rsq, scale, split, LDS write -- and run as a block of its own it cost ~420 ticks for ~45 instructions (a SIMD with one
  // wave has nothing else to issue while a link of the chain is in flight); a product group between two links covers that.
  f32x4 l2x = zero4(), l2d = zero4();
  float l2a = 0.f;
  int l2buf = 0, l2slot = 0;
  int64_t l2ex = 0;
  auto ln2_begin = [&](int buf, int s, int64_t ex) { l2buf = buf; l2slot = s & 3; l2ex = ex; };
  auto ln2_piece = [&](int k) __attribute__((always_inline)) {
    if constexpr (TWO) {
      // (the empty asm statements keep the optimiser from sinking a piece's arithmetic down to its first use in a later piece;
      //  the scheduling barriers behind the pieces only bind the machine scheduler)
      if (k == 0) l2x = ld4(&XS[l2slot][l2s][4 * l2c]);
      if (k == 1) {
        l2a = dpp_add<0xB1>((l2x[0] + l2x[1]) + (l2x[2] + l2x[3]));                     // quad_perm [1,0,3,2]
        asm volatile("" : "+v"(l2a));
      }
      if (k == 2) {
        l2a = dpp_add<0x4E>(l2a);                                                       // quad_perm [2,3,0,1]
        if constexpr (LPR == 8) l2a = dpp_add<0x141>(l2a);                              // row_half_mirror
        asm volatile("" : "+v"(l2a));
      }
      if (k == 3) {
        // (sums in the one-stage loader's tree -- 16 lanes x C / 16 channels, then the DPP butterfly -- so that both loaders
        //  round alike: C = 32: a lane pair's fma chains, added; C = 16: four squares, added pairwise)
        const float mean = l2a * (1.0f / C);
        float sq, d0, d1, d2, d3;
#pragma unroll
        for (int v = 0; v < 4; ++v) l2d[v] = l2x[v] - mean;
        if constexpr (C == 32) sq = __builtin_fmaf(l2d[1], l2d[1], __builtin_fmaf(l2d[0], l2d[0], 0.f)) +
                                    __builtin_fmaf(l2d[3], l2d[3], __builtin_fmaf(l2d[2], l2d[2], 0.f));
        else sq = (__builtin_fmaf(l2d[0], l2d[0], 0.f) + __builtin_fmaf(l2d[1], l2d[1], 0.f)) +
                  (__builtin_fmaf(l2d[2], l2d[2], 0.f) + __builtin_fmaf(l2d[3], l2d[3], 0.f));
        l2a = dpp_add<0xB1>(sq);
        d0 = l2d[0]; d1 = l2d[1]; d2 = l2d[2]; d3 = l2d[3];
        asm volatile("" : "+v"(l2a), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
        l2d[0] = d0; l2d[1] = d1; l2d[2] = d2; l2d[3] = d3;
      }
      if (k == 4) {
        l2a = dpp_add<0x4E>(l2a);
        if constexpr (LPR == 8) l2a = dpp_add<0x141>(l2a);
        l2a = __builtin_amdgcn_rsqf(__builtin_fmaf(l2a, 1.0f / C, 1e-5f));               // v_rsq_f32, 1 ulp
        asm volatile("" : "+v"(l2a));
      }
      if (k == 5) {
        f32x4 u;
        vec4 ut[NT];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          u[v] = __builtin_fmaf(l2d[v] * l2a, gam4[v], bet4[v]);
          { float uv = u[v]; asm volatile("" : "+v"(uv)); u[v] = uv; }      // (rounded to fp32 here: see ln_row)
          elem e[NT];
          splitn1<F16>(u[v], e);
#pragma unroll
          for (int n = 0; n < NT; ++n) ut[n][v] = e[n];
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) *reinterpret_cast<vec4*>(&U16[l2buf][n][l2s][4 * l2c]) = ut[n];
        if (SAVE && lvalid2 && dir == 0 && !(SB_EXP_SKIP & 8)) {      // both directions normalise the same rows: one copy is enough
          const int64_t uo = l2ex + lo_u2;
          if constexpr (SAVE == 3) {            // only the streaming backward reads u, as a single fp16 term
            h16x4 t;
#pragma unroll
            for (int v = 0; v < 4; ++v) t[v] = (_Float16)u[v];
            *reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(a.save_u) + uo) = t;
          } else if constexpr (SAVE == 4 && F16) {
            // wide form: the fp16 hi + lo terms of this kernel's own products (see ln_row): [P][C/2][hi0, hi1, lo0, lo1] resp.
            // [P][C][hi, lo] -- this lane's four channels are 16 contiguous bytes either way
            h16x8 t;
            if constexpr (C == 32) {
              t = h16x8{(_Float16)ut[0][0], (_Float16)ut[0][1], (_Float16)ut[1][0], (_Float16)ut[1][1],
                        (_Float16)ut[0][2], (_Float16)ut[0][3], (_Float16)ut[1][2], (_Float16)ut[1][3]};
            } else {
              t = h16x8{(_Float16)ut[0][0], (_Float16)ut[1][0], (_Float16)ut[0][1], (_Float16)ut[1][1],
                        (_Float16)ut[0][2], (_Float16)ut[1][2], (_Float16)ut[0][3], (_Float16)ut[1][3]};
            }
            st_side(reinterpret_cast<h16x8*>(reinterpret_cast<_Float16*>(a.save_u) + 2 * uo), t);
          } else {
            st4(a.save_u + uo, u);
          }
        }
      }
    }
  };
  auto ln_stage2 = [&](int buf, int s, int64_t ex) {       // the whole pass at once (prologue of a work item)
    ln2_begin(buf, s, ex);
#pragma unroll
    for (int k = 0; k < 6; ++k) ln2_piece(k);
  };

  // ---- compute role ----
  const int uoff = 16 * w + 4 * q;
  f32x4 c = zero4(), h = zero4();
  vec4 htv[NT];                                   // terms of the hidden state this lane stored last (SAVE == 4 writes them out)
  auto store_h = [&](int buf, const f32x4& hv) {   // split the 4 hidden values of this lane, 3 x 8-byte LDS stores
    vec4 (&tv)[NT] = htv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      elem e[NT];
      splitn1<F16>(hv[r], e);
#pragma unroll
      for (int n = 0; n < NT; ++n) tv[n][r] = e[n];
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) *reinterpret_cast<vec4*>(&H16[buf][n][j][uoff]) = tv[n];
  };
  // acc[g] += W[g][chunk] * B (6-term split product).  The four gate accumulators are walked round-robin and the
  // groups are fenced against MFMA reordering (mask: everything but MFMA may cross): hipcc otherwise chains all 12
  // products of a gate back to back on one accumulator, and any VALU instruction that lands between two MFMAs on
  // the SAME accumulator costs ~40 cycles.
  constexpr int kNoMfmaCross = 0x7F6;
  constexpr int kNoMfmaVmemCross = 0x786;
  // Deferred record stores (compact / wide records): the ~6 sixteen-byte-per-lane stores of a step, issued back to back by all
  // four waves right before the barrier, block the waves while the CU's store path drains (phase table: 700-800 ticks per
  // step and wave in training against 90 in inference).  They are kept in registers instead and issued one at a time between
  // the product groups of the NEXT step's W_hh h phase -- ~70 ticks of matrix work apart, no wave ever finds the path busy.
  // (not for the C = 16 inter-frame walk: one wave carries its whole y epilogue there and is the pole of every step;
  // measured 1.11 -> 1.17 ms with the stores moved into its phase A)
#ifdef SB_NO_DEFER                                   // A/B switch (developer builds)
  constexpr bool DEFER = false;
#else
  constexpr bool DEFER = SAVE >= 2 && !(LIN && C == 16);
#endif
  f32x4 rgi = zero4(), rgf = zero4(), rgg = zero4(), rgo = zero4(), rcp = zero4();     // records of step s_pend
  int s_pend = -1;
  // QLATE (late round 6): in the one-workgroup-per-CU kernels the parked gates are quantised to their 24-bit codes WHERE THEY ARE
  // STORED -- in the issue gaps of the next step's hidden-part products -- instead of at the end of the cell update, where the
  // 24 instructions ran exposed on a SIMD with nothing else to issue.  Same function on the same values: same bits.  (The
  // two-workgroup bidirectional kernels keep the early form: quantising at store time cost them 7-8 spilled registers, round 5.)
  // (... and the ordered consumer: same-box A/B, producer 1.055 -> 1.015 ms per launch, consumer 1.098 -> 1.109: SB_Q24_LATE = 2
  //  includes it)
  constexpr bool QLATE = SB_Q24_LATE != 0 && SAVE == 4 && SB_REC_Q24 != 0 && DEFER &&
                         !(LIN && C == 32 && !SUM3 && F16 && !SEG && !(ORD && SB_Q24_LATE == 2));
  auto rec_piece = [&](int k) __attribute__((always_inline)) {
    if constexpr (SAVE >= 2) {
      if (s_pend >= 0 && cvalid) {
        const int64_t blk = pend_blk;
        if (k == (SAVE == 4 ? 5 : 3)) {                  // hs of the step (h / htv still hold it: they change in phases B / C)
          const int64_t ho = pend_h + co_h;
          if constexpr (LIN && SAVE == 3) {   // the Linear is applied here: hs only feeds the backward kernels (fp16 terms)
            h16x4 h16;
#pragma unroll
            for (int r = 0; r < 4; ++r) h16[r] = (_Float16)h[r];
            if (a.hs && !(SB_EXP_SKIP & 4)) *reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(a.hs) + ho) = h16;
          } else if constexpr (LIN && SAVE == 4 && F16) {
            // wide form with the Linear applied here: hs only feeds the backward kernels' matrix products -- it travels as the
            // fp16 hi + lo terms just stored to LDS: [P][ndir][16 unit quads][hi x 4, lo x 4], same bytes as fp32
            h16x8 hp;
#pragma unroll
            for (int r = 0; r < 4; ++r) { hp[r] = (_Float16)htv[0][r]; hp[4 + r] = (_Float16)htv[1][r]; }
            if (a.hs) st_side(reinterpret_cast<h16x8*>(reinterpret_cast<_Float16*>(a.hs) + ho * 2), hp);
          } else {
            if ((!LIN || a.hs) && !(SB_EXP_SKIP & 4)) st4(a.hs + ho, h);
          }
        }
        if constexpr (SAVE == 4) {
          // wide records (sb_lstm_fwd_args.rec_f32): fp32, blocked per (tile, step, direction) in lane order like the
          // compact ones -- [wave][gate][lane][4 floats] and [wave][lane][4 floats], one contiguous KB per store
          // save_gates == NULL: records WITHOUT the gates (c_prev only) -- for a backward that recomputes them from u and
          // h_prev with the forward weights (sb_lstm_bwd_args.recompute with `wide`)
          if (k < 4 && a.save_gates) {
#if SB_REC_Q24
            // 24-bit fixed-point gates: three 16-byte pieces per lane, each packed here from the parked gates
            float* rec = a.save_gates + blk * (16 * kWideGateDwords) + (w * 192 + lane) * 4;
            if constexpr (QLATE) {                       // (pieces are issued in order: 0 needs i and f, 1 adds g, 2 adds o)
              if (k == 0) { rgi = q24_codes(rgi, false); rgf = q24_codes(rgf, false); }
              if (k == 1) rgg = q24_codes(rgg, true);
              if (k == 2) rgo = q24_codes(rgo, false);
            }
            if (k < 3) st4_rec(rec + 256 * k, q24_piece(k, rgi, rgf, rgg, rgo));
#else
            float* rec = a.save_gates + blk * (16 * 4 * H) + (w * 256 + lane) * 4;
            if (k == 0) st4_rec(rec, rgi);
            if (k == 1) st4_rec(rec + 256, rgf);
            if (k == 2) st4_rec(rec + 512, rgg);
            if (k == 3) st4_rec(rec + 768, rgo);
#endif
          }
          if (k == 4) st4_rec(a.save_c + blk * (16 * H) + (w * 64 + lane) * 4, rcp);
        } else {
          // Compact records are private to this kernel and the backward recurrence, which walks the same (tile, step)
          // grid with the same lane ownership, so they are laid out per (tile, step, direction) block in LANE order:
          // [wave][(i,f) | (g,o)][lane][8 halves] and [wave][lane][4 halves].  Every store instruction then writes one
          // contiguous KB (512 B for c_prev).  In the position-major layout adjacent lanes (sequences j, j+1) hit
          // different rows and the L1 splits each instruction into 64 sixteen-byte writes: measured 19 % of the
          // inter-frame forward, 27 % of the intra-frame one.
          if ((k == 0 || k == 1) && (SAVE != 3 || a.save_gates) && !(SB_EXP_SKIP & 1)) {   // SAVE == 3, no save_gates: recompute mode
            _Float16* rec = reinterpret_cast<_Float16*>(a.save_gates) + blk * (16 * 4 * H) + (w * 128 + lane) * 8;
            h16x8 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = (_Float16)(k == 0 ? rgi[r] : rgg[r]);
              v[4 + r] = (_Float16)(k == 0 ? rgf[r] : rgo[r]);
            }
            *reinterpret_cast<h16x8*>(rec + 512 * k) = v;
          }
          if (k == 2 && !(SB_EXP_SKIP & 2)) {
            h16x4 c16;
#pragma unroll
            for (int r = 0; r < 4; ++r) c16[r] = (_Float16)rcp[r];
            *reinterpret_cast<h16x4*>(reinterpret_cast<_Float16*>(a.save_c) + blk * (16 * H) + (w * 64 + lane) * 4) = c16;
          }
        }
      }
    }
  };
  auto rec_flush = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) rec_piece(k);
    s_pend = -1;
  };
  // hook >= 0: after product group pi the pending record store number hook + pi is issued (see rec_piece) and pinned there
  // (second fence: VMEM may not cross either)
  auto mma6 = [&](f32x4 (&acc)[4], int chunk, const vec8 (&b)[NT], int hook = -1, int lnhook = -1) {
    // (weight term, operand term) pairs, smallest products first
#ifdef SB_FWD_2P
    // opt-in reduced-product build of this translation unit (sb_lstm_fwd_args.products == 2): activations as ONE fp16 term --
    // W_lo x_hi + W_hi x_hi, two products per MAC; the low activation terms are still formed and parked in LDS (same code,
    // same layout) but no product reads them
    static_assert(F16, "two-product mode: fp16 operands");
    constexpr int NP = 2;
    constexpr int WT[6] = {1, 0, 0, 0, 0, 0};
    constexpr int XT[6] = {0, 0, 0, 0, 0, 0};
#else
    constexpr int NP = F16 ? 3 : 6;
    constexpr int WT[6] = {F16 ? 1 : 2, F16 ? 0 : 0, F16 ? 0 : 1, 1, 0, 0};
    constexpr int XT[6] = {F16 ? 0 : 0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};
#endif
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = PR::mma(Wt[g][chunk].t[WT[pi]], b[XT[pi]], acc[g]);
      __builtin_amdgcn_sched_barrier(kNoMfmaCross);
      if (hook >= 0) {
        rec_piece(hook + pi);
        __builtin_amdgcn_sched_barrier(kNoMfmaVmemCross);
      }
      if (lnhook >= 0) {                               // piece lnhook + pi of the stage-2 LayerNorm, pinned behind this group
        ln2_piece(lnhook + pi);
#ifdef SB_FWD_2P
        if (pi == NP - 1) ln2_piece(lnhook + pi + 1);  // (three pieces per chunk over two product groups)
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto x_part = [&](f32x4 (&acc)[4], int buf) {
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = ld4(&Bias[g][uoff]);
    vec8 b[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const vec8*>(&U16[buf][n][j][8 * q]);
    mma6(acc, 0, b);
  };
  // Phase B with the cell update PINNED between the products of the input part (fp16x3 form): a SIMD issues in order, its matrix
  // pipe takes a v_mfma every ~18 ticks and covers ~2 transcendentals issued behind it (r03 microbenchmarks: 1 MFMA + 2 trans =
  // 27.6 ticks against 18.6 + 20), but left to itself hipcc issues ~26 of the step's 40 v_exp / v_rcp in front of the first
  // product (its operands come from LDS) and 14 among them.  Here every product is followed by one piece of the update of ONE
  // of the lane's four units -- 4 v_exp, then 4 v_rcp, then the cell state and its v_exp -- so 36 of the 40 sit behind a
  // product; the operands are fetched before the first piece.  Same expressions as sigmoid_pre / tanh_pre / tanhf_fast: same bits.
#ifdef SB_EXP_NO_CELLPIN
  constexpr bool CELLPIN = false;
#else
  // (same-box A/B, r04: C = 32 inference producer 0.607 -> 0.595 ms; C = 16 0.62 -> 0.655 ms and the store-bound training forms
  //  unchanged or slower: inference, C = 32 only)
  constexpr bool CELLPIN = F16 && HEAVY && C == 32 && SAVE == 0;
#endif
  auto x_part_cell = [&](f32x4 (&accn)[4], int buf, const f32x4 (&acc)[4], f32x4& gi, f32x4& gf, f32x4& gg, f32x4& go, f32x4& tc) {
    constexpr int WT[3] = {1, 0, 0}, XT[3] = {0, 1, 0};          // (weight term, operand term), smallest products first (as mma6)
#pragma unroll
    for (int g = 0; g < 4; ++g) accn[g] = ld4(&Bias[g][uoff]);
    vec8 b[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const vec8*>(&U16[buf][n][j][8 * q]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pi = 0; pi < 3; ++pi) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#ifdef SB_FWD_2P
        if (pi != 1)                                             // (no W_hi x_lo product; the cell-update piece stays where it is)
#endif
        accn[g] = PR::mma(Wt[g][0].t[WT[pi]], b[XT[pi]], accn[g]);
        const int r = g;                                         // this piece's unit
        if (pi == 0) {
          float e0 = __builtin_amdgcn_exp2f(acc[0][r]), e1 = __builtin_amdgcn_exp2f(acc[1][r]);
          float e2 = __builtin_amdgcn_exp2f(acc[2][r]), e3 = __builtin_amdgcn_exp2f(acc[3][r]);
          asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
          gi[r] = e0; gf[r] = e1; gg[r] = e2; go[r] = e3;
        } else if (pi == 1) {
          float e0 = __builtin_amdgcn_rcpf(1.0f + gi[r]), e1 = __builtin_amdgcn_rcpf(1.0f + gf[r]);
          float e2 = __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + gg[r]), -1.0f), e3 = __builtin_amdgcn_rcpf(1.0f + go[r]);
          asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
          gi[r] = e0; gf[r] = e1; gg[r] = e2; go[r] = e3;
        } else {
          float cn = __builtin_fmaf(gf[r], c[r], gi[r] * gg[r]);
          float e = __builtin_amdgcn_exp2f(2.0f * SB_NLOG2E * cn);
          asm volatile("" : "+v"(cn), "+v"(e));
          c[r] = cn; tc[r] = e;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // y_tag: std::true_type = this wave owns an output-channel tile of the fused Linear (its W_lin . h products ride along);
  // the caller branches on the wave's role ONCE and calls the matching copy, so that each role's phase A is one basic block
  // (hipcc interleaves the stage-2 LayerNorm's dependent chain -- LDS round trip, two DPP sums, v_rsq -- with the products only
  // inside a block: behind a branch of its own the chain ran exposed, ~420 ticks for ~45 instructions)
  auto h_part = [&](f32x4 (&acc)[4], int buf, auto y_tag, auto ln_tag) {
    // y_tag: 0 = no output-channel tile, 1 = this wave owns one, 2 = decided here (linw: the one-stage loader's kernels)
    constexpr int Y = decltype(y_tag)::value;
#ifdef SB_EXP_NO_LNHOOK
    constexpr bool LNH = false;
    if constexpr (decltype(ln_tag)::value) ln_stage2(l2buf, l2slot, l2ex);
#else
    constexpr bool LNH = decltype(ln_tag)::value && F16;      // (three product groups per chunk)
#endif
#pragma unroll
    for (int ck = 0; ck < 2; ++ck) {
      vec8 b[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const vec8*>(&H16[buf][n][j][32 * ck + 8 * q]);
      mma6(acc, 1 + ck, b, DEFER ? 3 * ck : -1, LNH ? 3 * ck : -1);
      if constexpr (LIN && Y != 0) {                   // W_lin . h of the step that produced this buffer
        if (Y == 1 || linw) {
          if (ck == 0) yacc = zero4();
          if constexpr (F16) {
            yacc = PR::mma(Wl[ck].t[1], b[0], yacc);
#ifndef SB_FWD_2P
            yacc = PR::mma(Wl[ck].t[0], b[1], yacc);
#endif
            yacc = PR::mma(Wl[ck].t[0], b[0], yacc);
          }
        }
      }
    }
  };
  // y of step sy (its W_lin h is in yacc, its residual row in xres)
  auto store_y = [&](int64_t ey) {                  // ey = st(sy) p_step C of the step sy whose y this is (uniform)
    if (linw && cvalid && !(SB_EXP_SKIP & 16)) {
      f32x4 v = yacc + lbias + xres;
      if (film) {
        if (a.y_pre) st_side(reinterpret_cast<f32x4*>(a.y_pre + ey + co_y), v);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], fw[r], fb[r]);
      }
      float* yp = a.y + (lin_part ? 2 * ey : ey) + co_y;
      if (prod) st4_sc1(yp, v); else st4(yp, v);
    }
  };
  // the tile owners' epilogue of a step as ONE block (one-workgroup-per-CU kernels): y of the previous step, then this step's
  // residual.  FiLM is always applied (planes 1 / 0 without it: v 1 + 0 = v), and the item's first step, which has no previous y,
  // writes what it has to its OWN row -- the next step replaces it (same lane, same address, program order) -- so the block has
  // two uniform tests left (y_pre, write-through) where store_y + load_res had seven: ~16 ticks each on a one-wave SIMD.
  auto epilogue = [&](int64_t ey, int sy, int64_t er) {
    if (cvalid) {
      f32x4 v = yacc + lbias + xres;
      if (a.y_pre) st_side(reinterpret_cast<f32x4*>(a.y_pre + ey + co_y), v);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(v[r], fw[r], fb[r]);
      float* yp = a.y + (lin_part ? 2 * ey : ey) + co_y;
      if (prod) st4_sc1(yp, v); else st4(yp, v);
      if (!lin_part) {
        if constexpr (TWO) xres = ld4(&XS[sy & 3][j][16 * w + 4 * q]);
        else xres = ld4(a.x + er + co_y);
      }
    }
  };
  auto load_res = [&](int sy, int64_t ey) {
    if (linw && cvalid && !lin_part && !(SB_EXP_SKIP & 32)) {
      if constexpr (TWO) xres = ld4(&XS[sy & 3][j][16 * w + 4 * q]);         // the raw row is in the ring
      else xres = ld4(a.x + ey + co_y);
    }
  };

  // Input rows are fetched FOUR steps before their LayerNorm: loads and stores share one in-order counter (vmcnt),
  // so waiting for a row also waits for every older store, and in training the record stores of the inter-frame walk
  // take longer than two steps to be acknowledged (measured: SQ_WAIT_ANY 1050 cycles per step against 310 without
  // the stores).  The loop body covers four steps and issues the rows of the next four at its top.
  XVec<C> xa, xb, xc, xd;
  f32x4 accx[4];
  int s_begin = 0;                              // first step of the current work item

#ifdef SB_PHASE_TIMING
  unsigned long long tph[5] = {0, 0, 0, 0, 0};
#endif
  auto step = [&](int s, const XVec<C>& xrow) {        // xrow: input row s + 2 (TWO: s + 3)
    const int cur = s & 1;
    SB_TICK(c0);
    // ---- A: hidden part on the matrix pipe || LayerNorm of row s+2 in the issue gaps (it does not depend on
    //         this step; its U16[cur] buffer was last read in phase B of step s-1) ----
    f32x4 acc[4];
    {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = accx[g];
      if constexpr (TWO) {
        raw_store(xrow, s + 3, s + 3 <= S - 1 ? run_x + 3 * d_x : run_x_last);
        typedef std::integral_constant<int, 0> Y0;
        typedef std::integral_constant<int, 1> Y1;
        if (linw) {
          h_part(acc, cur, Y1{}, std::false_type{});
        } else if (C == 32 || lw2) {                   // (C = 32: every wave without a tile is a loader wave)
          ln2_begin(cur, s + 2, s + 2 <= S - 1 ? run_x + 2 * d_x : run_x_last);
          if constexpr (F16) h_part(acc, cur, Y0{}, std::true_type{});
          else { ln_stage2(cur, s + 2, l2ex); h_part(acc, cur, Y0{}, std::false_type{}); }
        } else {
          h_part(acc, cur, Y0{}, std::false_type{});
        }
      } else {
        ln_store(xrow, cur, min(s + 2, S - 1), s + 2 <= S - 1 ? run_x + 2 * d_x : run_x_last);
        h_part(acc, cur, std::integral_constant<int, 2>{}, std::false_type{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_TICK(c1);
    // ---- B: input part of step s+1 (matrix pipe) || cell update of step s (VALU) ----
    f32x4 gi, gf, gg, go, cprev;
    if constexpr (CELLPIN) {
      cprev = c;
      f32x4 tc;
      x_part_cell(accx, cur ^ 1, acc, gi, gf, gg, go, tc);
#pragma unroll
      for (int r = 0; r < 4; ++r) h[r] = go[r] * __builtin_fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + tc[r]), -1.0f);
    } else {
      x_part(accx, cur ^ 1);
#if SB_EPI_EARLY
      // (late round 6: the tile owners' y epilogue of the PREVIOUS step -- its W_lin h products finished in phase A -- issued here,
      //  behind the input part's products, instead of at the end of the step, where it ran exposed: the owners' phase D was 340
      //  ticks longer than the loader waves', who waited for them at the barrier)
      if constexpr (LIN && (HEAVY || (ORD && SB_EPI_EARLY == 2)) && !SB_EXP_SKIP) { if (linw) epilogue(s > s_begin ? run_x - d_x : run_x, s, run_x); }
      else if constexpr (LIN && SB_EPI_EARLY == 3) { if (s > s_begin) store_y(run_x - d_x); load_res(s, run_x); }
      __builtin_amdgcn_sched_barrier(0);
#endif
      cprev = c;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = sigmoid_pre(acc[0][r]);
        gf[r] = sigmoid_pre(acc[1][r]);
        gg[r] = tanh_pre(acc[2][r]);
        go[r] = sigmoid_pre(acc[3][r]);
        c[r] = __builtin_fmaf(gf[r], c[r], gi[r] * gg[r]);
        h[r] = go[r] * tanhf_fast(c[r]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_TICK(c2);
    store_h(cur ^ 1, h);
    __builtin_amdgcn_sched_barrier(0);
    SB_TICK(c3);
    // ---- C ----
    if constexpr (SAVE >= 2) {        // records of this step: issued in phase A of the next one (or by the flush after the walk)
      if constexpr (SAVE == 4 && SB_REC_Q24 != 0 && !QLATE) {   // parked as the record's 24-bit codes (sb_lstm_bf_common.h)
        rgi = q24_codes(gi, false); rgf = q24_codes(gf, false); rgg = q24_codes(gg, true); rgo = q24_codes(go, false); rcp = cprev;
      } else { rgi = gi; rgf = gf; rgg = gg; rgo = go; rcp = cprev; }
      s_pend = s;
      pend_blk = run_blk; pend_h = run_h;
      if constexpr (!DEFER) rec_flush();              // ... or right here
    } else {
      if (cvalid) {
        const int st = rev ? S - 1 - s : s;
        const int64_t pos = cbase + (int64_t)st * a.p_step;
        if ((!LIN || a.hs) && !(SB_EXP_SKIP & 4)) st4(a.hs + run_h + co_h, h);
        if (SAVE == 1) {
          float* rec = a.save_gates + (pos * ndir + dir) * (5 * H) + uoff;
          st4(rec, gi); st4(rec + H, gf); st4(rec + 2 * H, gg); st4(rec + 3 * H, go); st4(rec + 4 * H, cprev);
        }
      }
    }
#ifdef SB_EXP_NO_EPI
    constexpr bool EPI1 = false;
#else
    constexpr bool EPI1 = LIN && (HEAVY || (ORD && SB_EPI_EARLY == 2)) && !SB_EXP_SKIP;      // (SB_EPI_EARLY = 2: the ordered consumer too)
#endif
    if constexpr (EPI1) {
#if SB_EPI_EARLY
      if constexpr (CELLPIN) { if (linw) epilogue(s > s_begin ? run_x - d_x : run_x, s, run_x); }      // (the pinned inference form keeps it here)
#else
      if (linw) epilogue(s > s_begin ? run_x - d_x : run_x, s, run_x);
#endif
    } else if constexpr (LIN && SB_EPI_EARLY != 3) {
      if (s > s_begin) store_y(run_x - d_x);
      load_res(s, run_x);
    }
    run_x += d_x; run_h += d_h; run_blk += d_blk;       // row s + 1
    __builtin_amdgcn_sched_barrier(0);
    SB_TICK(c4);
    __syncthreads();
#ifdef SB_PHASE_TIMING
    SB_TICK(c5);
    tph[0] += c1 - c0; tph[1] += c2 - c1; tph[2] += c3 - c2; tph[3] += c4 - c3; tph[4] += c5 - c4;
#endif
  };
  const int ntiles = (a.nseq + 15) / 16;
  const int nitems = SEG ? ntiles * a.seg_count : ntiles;          // !SEG: gridDim.x == ntiles, one item each
  float* const seg_hc = SEG ? a.seg_state : nullptr;               // [ntiles][2][16][64]: c, h
  int next_slab = 0;                                               // producer: first slab not yet counted in
  auto slab_signal = [&](int k) {                                  // every y row of slab k of this tile is on its way
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.slab_flags + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  const int item1 = ORD ? ntiles : nitems;
  for (int item = ORD ? ord_first : (int)blockIdx.x; item < item1; item = ORD ? ord_next() : item + (int)gridDim.x) {
    const int seg = SEG ? item / ntiles : 0;
    const int tile = ORD ? a.tile_order[item] : SEG ? item - seg * ntiles : item;
    s_begin = SEG ? seg * a.seg_len : 0;
    const int s_end = SEG ? min(S, s_begin + a.seg_len) : S;
    if constexpr (ORD) { if (!seg_wait(a.slab_flags, a.tile_need[item], a.slab_need, a.sched_status, SB_TRIP_FWD_CONSUMER)) return; }
    set_tile(tile);
    // ---- initial state of this item ----
    c = zero4();
    h = zero4();
    if (seg == 0) {
      if (dir == 0 && cvalid) {
        if (a.c0) c = ld4(a.c0 + (size_t)nc * H + uoff);
        if (a.h0) h = ld4(a.h0 + (size_t)nc * H + uoff);
      }
    } else if constexpr (SEG) {
      // wait until the previous segment of this tile has published its state.  Flag and state travel as agent-scope
      // (sc1, write-through / cache-bypassing) accesses ordered by s_waitcnt + barrier -- no release / acquire fences:
      // those write back / invalidate the whole L2, which is full of this kernel's own record stores.
      if (!seg_wait(a.seg_flags, tile, seg, a.sched_status, SB_TRIP_FWD_SEGMENT)) return;
      const float* st = seg_hc + ((size_t)tile * 2 * 16 + j) * H + uoff;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c[r] = __hip_atomic_load(st + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        h[r] = __hip_atomic_load(st + 16 * H + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // ---- prologue: state and the first two normalised rows into LDS, four rows in flight ----
    {
      store_h(s_begin & 1, h);
      XVec<C> x0 = load_x(s_begin);
      XVec<C> x1 = load_x(min(s_begin + 1, S - 1));
      {                                                // running step terms of row s_begin (see their declaration)
        const int st0 = rev ? S - 1 - s_begin : s_begin;
        run_x = (int64_t)st0 * a.p_step * C;
        run_h = (int64_t)st0 * a.p_step * (ndir * H);
        run_blk = (rec_tile + st0) * ndir + dir;
        run_x_last = (int64_t)(rev ? 0 : S - 1) * a.p_step * C;
      }
      if constexpr (TWO) {
        const XVec<C> x2 = load_x(min(s_begin + 2, S - 1));
        raw_store(x0, s_begin, run_x);
        raw_store(x1, s_begin + 1, s_begin + 1 <= S - 1 ? run_x + d_x : run_x_last);
        raw_store(x2, s_begin + 2, s_begin + 2 <= S - 1 ? run_x + 2 * d_x : run_x_last);
        xa = load_x(min(s_begin + 3, S - 1));
        xb = load_x(min(s_begin + 4, S - 1));
        xc = load_x(min(s_begin + 5, S - 1));
        xd = load_x(min(s_begin + 6, S - 1));
        __syncthreads();
        if (lw2) {
          ln_stage2(s_begin & 1, s_begin, run_x);
          ln_stage2((s_begin + 1) & 1, s_begin + 1, s_begin + 1 <= S - 1 ? run_x + d_x : run_x_last);
        }
      } else {
        ln_store(x0, s_begin & 1, s_begin, run_x);
        ln_store(x1, (s_begin + 1) & 1, min(s_begin + 1, S - 1), s_begin + 1 <= S - 1 ? run_x + d_x : run_x_last);
        xa = load_x(min(s_begin + 2, S - 1));
        xb = load_x(min(s_begin + 3, S - 1));
        xc = load_x(min(s_begin + 4, S - 1));
        xd = load_x(min(s_begin + 5, S - 1));
      }
    }
    __syncthreads();
    x_part(accx, s_begin & 1);

    int s = s_begin;
    for (; s + 3 < s_end; s += 4) {
      const XVec<C> ca = xa, cb = xb, cc = xc, cd = xd;
      xa = load_x(min(s + 6 + (TWO ? 1 : 0), S - 1));
      xb = load_x(min(s + 7 + (TWO ? 1 : 0), S - 1));
      xc = load_x(min(s + 8 + (TWO ? 1 : 0), S - 1));
      xd = load_x(min(s + 9 + (TWO ? 1 : 0), S - 1));
      step(s, ca);
      step(s + 1, cb);
      step(s + 2, cc);
      step(s + 3, cd);
      if (prod) {                                      // y of steps .. s + 2 has been issued
        while ((next_slab + 1) * a.slab_len <= s + 3) slab_signal(next_slab++);
      }
    }
    if (s < s_end) step(s, xa);
    if (s + 1 < s_end) step(s + 1, xb);
    if (s + 2 < s_end) step(s + 2, xc);
    rec_flush();                                       // records of the last step
    if constexpr (LIN) {                               // y of the item's last step from the final hidden-state tiles
      if (linw) {
#pragma unroll
        for (int ck = 0; ck < 2; ++ck) {
          vec8 b[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const vec8*>(&H16[s_end & 1][n][j][32 * ck + 8 * q]);
          if (ck == 0) yacc = zero4();
          if constexpr (F16) {
            yacc = PR::mma(Wl[ck].t[1], b[0], yacc);
#ifndef SB_FWD_2P
            yacc = PR::mma(Wl[ck].t[0], b[1], yacc);
#endif
            yacc = PR::mma(Wl[ck].t[0], b[0], yacc);
          }
        }
      }
      store_y(run_x - d_x);                            // (run_x stands at row s_end)
      if (prod) { while (next_slab * a.slab_len < S) slab_signal(next_slab++); }
    }
    // ---- final state: to the caller after the last step, to the next segment otherwise ----
    if (s_end == S) {
      if (dir == 0 && cvalid) {
        if (a.hN) st4(a.hN + (size_t)nc * H + uoff, h);
        if (a.cN) st4(a.cN + (size_t)nc * H + uoff, c);
      }
    } else if constexpr (SEG) {
      float* st = seg_hc + ((size_t)tile * 2 * 16 + j) * H + uoff;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        __hip_atomic_store(st + r, c[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(st + 16 * H + r, h[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_s_waitcnt(0);                   // ... acknowledged at the device-coherent level ...
      __syncthreads();                                 // ... by every wave, before the flag goes up
      if (tid == 0) __hip_atomic_store(a.seg_flags + tile, seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if constexpr (SEG || ORD) __syncthreads();         // LDS tiles are reused by the next item
  }
#ifdef SB_PHASE_TIMING
  if (SAVE == 0 && a.save_u && lane == 0 && blockIdx.x < 4) {
    float* d = a.save_u + (blockIdx.x * 4 + w) * 8;
    for (int i = 0; i < 5; ++i) d[i] = (float)tph[i] / S;
  }
  if (lane == 0 && blockIdx.x < 4 && (ORD || blockIdx.y == 0)) {
    const int kind = ORD ? 3 : (LIN && a.ndir == 2) ? 4 : SUM3 ? 2 : LIN ? 1 : 0;
    for (int i = 0; i < 5; ++i) g_phase_fwd[kind][blockIdx.x * 4 + w][i] = (float)tph[i] / S;
  }
#endif
}

#ifndef SB_FWD_2P
// test hook (sb_rec_q24_roundtrip): 16 floats per thread through the record packing and back -- gates i, f, g, o x 4 units
__global__ void rec_q24_roundtrip_kernel(const float* in, float* out, unsigned* packed, int n16) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n16) return;
  f32x4 g[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k] = ld4(in + (size_t)t * 16 + 4 * k);
  const f32x4 ci = q24_codes(g[0], false), cf = q24_codes(g[1], false), cg = q24_codes(g[2], true), co = q24_codes(g[3], false);
  const f32x4 p0 = q24_piece(0, ci, cf, cg, co), p1 = q24_piece(1, ci, cf, cg, co), p2 = q24_piece(2, ci, cf, cg, co);
  if (packed) {
    unsigned* pk = packed + (size_t)t * 12;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pk[k] = f2u(p0[k]); pk[4 + k] = f2u(p1[k]); pk[8 + k] = f2u(p2[k]);
    }
  }
  f32x4 o[4];
  q24_unpack(p0, p1, p2, o[0], o[1], o[2], o[3]);
#pragma unroll
  for (int k = 0; k < 4; ++k) st4(out + (size_t)t * 16 + 4 * k, o[k]);
}
#endif

}  // namespace

#ifndef SB_FWD_2P
extern "C" int sb_lstm_wide_rec_dwords(void) { return kWideGateDwords; }
extern "C" int sb_rec_q24_roundtrip(const float* in, float* out, uint32_t* packed, int64_t n, void* stream) {
  if (!in || !out || n <= 0 || n % 16) return -1001;
  const int n16 = (int)(n / 16);
  hipLaunchKernelGGL(rec_q24_roundtrip_kernel, dim3((n16 + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, packed, n16);
  SB_CHECK_LAUNCH();
  return 0;
}
#endif

#ifdef SB_FWD_2P
#define sb_launch_lstm_fwd_bf sb_launch_lstm_fwd_bf_2p        // second build of this file: the two-product inference kernels
#endif
int sb_launch_lstm_fwd_bf(const sb_lstm_fwd_args& a_in, hipStream_t st) {
  sb_lstm_fwd_args a = a_in;
  const int ntiles = (a.nseq + 15) / 16;
  const bool full = a.nseq % 16 == 0;
  const bool f16 = a.mma != 2;                      // mma == 2: bf16x6 (fp32-exact class); default fp16x3
  if (a.aux_f16 && !a.save_c) return -1003;
  // aux_f16 with save_gates == NULL: records without the gates (the backward recomputes them: sb_lstm_bwd_args.recompute)
  // rec_f32 with save_gates == NULL: wide records without the gates (c_prev, u and hs pairs only; single direction with the
  // Linear applied here, so that hs is stored): the backward recomputes the gates
  if (a.rec_f32 && (!a.save_c || a.aux_f16 || !f16 || (!a.save_gates && (a.ndir != 1 || !a.lin_w || !a.hs)))) return -1003;
  const int save = a.rec_f32 ? 4 : a.save_gates == nullptr ? (a.save_c && a.aux_f16 ? 3 : 0)
                                                           : (a.save_c ? (a.aux_f16 ? 3 : 2) : 1);
#ifdef SB_FWD_2P
  if (save != 0 || !f16) return -1003;                         // inference only: records / side outputs keep the default arithmetic
#endif
  dim3 grid(ntiles, a.ndir);
  const bool lin = a.lin_w != nullptr;
  if (lin && (!f16 || !a.lin_b || !a.y)) return -1003;         // ndir == 2: per-direction partial products (see the kernel)
  if (a.film_w && (!lin || a.ndir != 1 || !a.film_b)) return -1003;
  if (!lin && !a.hs) return -1003;
  // time-segmented scheduling (see the kernel): single direction, scratch provided, more tiles than CUs
  const int cus = device_cu_count();
  if (a.sched_workers < 0 || a.sched_segments < 0 || a.sched_workers > cus) return -1003;
  const int W = a.sched_workers > 0 ? a.sched_workers : cus, kforce = a.sched_segments;
  bool seg = f16 && a.ndir == 1 && a.seg_state && a.seg_flags && a.sched_status && ntiles >= W &&
             ((ntiles > W && ntiles <= 2 * W) || kforce > 0);
  if (seg) {
    double cost = 0.0;
    const int k = kforce > 0 ? kforce : choose_segments(ntiles, W, a.nsteps, &cost);
    // two co-resident tiles per CU cost ~1.4-1.6 T on the CUs that get them; only segment when clearly below that
    if (k < 2 || (kforce == 0 && cost > 1.30)) seg = false;
    else {
      // segment starts on multiples of 4 steps: every step then runs through the same copy of the 4-step unrolled
      // loop body (or the same tail code) as in the plain schedule, which keeps the two schedules bit-identical
      a.seg_len = ((a.nsteps + k - 1) / k + 3) & ~3;
      a.seg_count = (a.nsteps + a.seg_len - 1) / a.seg_len;      // drop empty trailing segments
      grid.x = W;
      (void)sb_flags_zero(a.seg_flags, ntiles, st);
    }
  }
  if (a.slab_flags && (seg || !lin || !f16 || a.slab_len < 4 || (a.slab_len & 3) || !a.ord_started)) return -1003;
  if (a.tile_order) {    // consumer side of the overlapped forward: bidirectional partial-Linear pass, ordered 1-D grid
    if (!a.slab_flags || !a.tile_need || !a.sched_status || !a.ord_counter || !a.ord_started || a.ndir != 2 || a.C != 32 ||
        (save != 0 && save != 3 && save != 4) || a.ord_grid < 2 || (a.ord_grid & 1))
      return -1003;
    dim3 g1(a.ord_grid);
#define SB_LO(SV, FL) hipLaunchKernelGGL((lstm_fwd_bf_kernel<32, SV, FL, true, true, false, false, true>), g1, dim3(256), 0, st, a)
    if (save == 0) { if (full) SB_LO(0, true); else SB_LO(0, false); }
#ifndef SB_FWD_2P
    else if (save == 4) { if (full) SB_LO(4, true); else SB_LO(4, false); }
    else { if (full) SB_LO(3, true); else SB_LO(3, false); }
#endif
#undef SB_LO
    return 0;
  }
#define SB_L(CC, SV, FL, HF, LN, SG) do { \
    if (SG && !fits_one_per_cu<lstm_fwd_bf_kernel<CC, SV, FL, HF, LN, SG>>()) return -1008; \
    hipLaunchKernelGGL((lstm_fwd_bf_kernel<CC, SV, FL, HF, LN, SG>), grid, dim3(256), 0, st, a); } while (0)
  if (a.x_part) {        // summed-input mode: single direction, fused Linear, C = 32, inference or fp16 side outputs
    if (!lin || a.ndir != 1 || a.C != 32 || (save != 0 && save != 3 && save != 4)) return -1003;
#define SB_L3(SV, FL, SG) do { \
    if (SG && !fits_one_per_cu<lstm_fwd_bf_kernel<32, SV, FL, true, true, SG, true>>()) return -1008; \
    hipLaunchKernelGGL((lstm_fwd_bf_kernel<32, SV, FL, true, true, SG, true>), grid, dim3(256), 0, st, a); } while (0)
#define SB_L3F(SV) do { if (full) { if (seg) SB_L3(SV, true, true); else SB_L3(SV, true, false); } \
                        else { if (seg) SB_L3(SV, false, true); else SB_L3(SV, false, false); } } while (0)
#ifdef SB_FWD_2P
    SB_L3F(0);
#else
    if (save == 0) SB_L3F(0); else if (save == 4) SB_L3F(4); else SB_L3F(3);
#endif
#undef SB_L3F
#undef SB_L3
    return 0;
  }
#ifdef SB_FWD_2P
#define SB_LT(CC, SV, FL) do { \
    if (seg) { if (lin) SB_L(CC, SV, FL, true, true, true); else SB_L(CC, SV, FL, true, false, true); } \
    else if (lin) SB_L(CC, SV, FL, true, true, false); else SB_L(CC, SV, FL, true, false, false); } while (0)
#define SB_LC(CC) do { if (full) SB_LT(CC, 0, true); else SB_LT(CC, 0, false); } while (0)
#else
#define SB_LT(CC, SV, FL) do { \
    if (seg) { if (lin) SB_L(CC, SV, FL, true, true, true); else SB_L(CC, SV, FL, true, false, true); } \
    else if (lin) SB_L(CC, SV, FL, true, true, false); else if (f16) SB_L(CC, SV, FL, true, false, false); \
    else SB_L(CC, SV, FL, false, false, false); } while (0)
#define SB_LC(CC) do { \
    if (save == 0) { if (full) SB_LT(CC, 0, true); else SB_LT(CC, 0, false); } \
    else if (save == 1) { if (full) SB_LT(CC, 1, true); else SB_LT(CC, 1, false); } \
    else if (save == 2) { if (full) SB_LT(CC, 2, true); else SB_LT(CC, 2, false); } \
    else if (save == 4) { if (full) SB_LT(CC, 4, true); else SB_LT(CC, 4, false); } \
    else { if (full) SB_LT(CC, 3, true); else SB_LT(CC, 3, false); } } while (0)
#endif
  if (a.C == 32) SB_LC(32); else SB_LC(16);
#undef SB_LC
#undef SB_LT
#undef SB_L
  return 0;
}
