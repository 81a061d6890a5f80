// Recurrent LSTM kernels for gfx950 (MI355X): the intra-frame (over F) and
// inter-frame (over T) LSTMs of the GridNet blocks -- 90 % of the model FLOPs.
//
// Design (not a translation of cuDNN/MIOpen's "hoist the input GEMM" scheme):
//   * sequences are independent, so a workgroup owns a tile of 16 sequences for
//     the whole time loop: no grid-wide per-step synchronisation;
//   * transposed MFMA form: gates^T[256 x 16 seq] = W[256 x (C+64)] * [u;h]^T.
//     W_ih and W_hh are the A operand and stay in VGPRs for the whole loop
//     (96 VGPRs for C=32); the [C+H] input/hidden vectors are the B operand,
//     exchanged through 2 x 5 KB of double-buffered LDS (one barrier per step);
//   * wave w owns hidden units 16w..16w+15 for all four gates, so the cell
//     update is lane-local (4 units x 1 sequence per lane), h is written back
//     with one ds_write_b128 and one 16-byte global store;
//   * the LayerNorm in front of every LSTM is fused into the loader (the input
//     row is normalised while it is staged, one step ahead of its use);
//   * v_mfma_f32_16x16x4_f32: exact fp32 (bitwise an fma chain), so the 1e-3
//     parity bar of the north star holds without any mixed-precision risk.
#include <stdlib.h>
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

// bf16 split-product variants (sb_lstm_bf.hip)
int sb_launch_lstm_fwd_bf(const sb_lstm_fwd_args& a, hipStream_t st);
int sb_launch_lstm_fwd_bf_2p(const sb_lstm_fwd_args& a, hipStream_t st);         // the same TU built with -DSB_FWD_2P (two products per MAC)
bool sb_lstm_fwd_vec_ok(const sb_lstm_fwd_args& a);                         // sb_lstm_vec.hip: a handful of sequences, inference
int sb_launch_lstm_fwd_vec(const sb_lstm_fwd_args& a, hipStream_t st);
int sb_launch_lstm_bwd_bf(const sb_lstm_bwd_args& a, hipStream_t st);

namespace {

constexpr int H = SB_H;
constexpr int HP = H + 4;   // padded LDS row (conflict-free ds_read_b128 across 16 rows)

template <int C>
struct XVec { float v[C / 16]; };
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// FULL: nseq is a multiple of the 16-sequence tile -> no bounds checks, hence no exec-masked branches
// around the global stores, hence counted (not zero) vmcnt waits in the time loop.
template <int C, int SAVE, bool FULL>
__global__ __launch_bounds__(256) void lstm_fwd_kernel(sb_lstm_fwd_args a) {
  constexpr int KX = C / 16;     // 16-wide K chunks of the input part
  constexpr int VPT = C / 16;    // floats per loader thread
  constexpr int CP = C + 4;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = blockIdx.y;
  const int n0 = blockIdx.x * 16;
  const int S = a.nsteps;
  const bool rev = dir == 1;

  __shared__ __attribute__((aligned(16))) float U[2][16][CP];
  __shared__ __attribute__((aligned(16))) float Hb[2][16][HP];

  // ---- weights -> registers (A operand fragments) ----
  const float* __restrict__ wih = a.w_ih[dir];
  const float* __restrict__ whh = a.w_hh[dir];
  f32x4 Aih[4][KX], Ahh[4][4], bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = g * H + 16 * w + j;
#pragma unroll
    for (int m = 0; m < KX; ++m) Aih[g][m] = ld4(wih + (size_t)row * C + 16 * m + 4 * q);
#pragma unroll
    for (int m = 0; m < 4; ++m) Ahh[g][m] = ld4(whh + (size_t)row * H + 16 * m + 4 * q);
    const int u0 = g * H + 16 * w + 4 * q;
    bias[g] = ld4(a.b_ih[dir] + u0) + ld4(a.b_hh[dir] + u0);
  }

  // ---- loader role: thread -> (sequence ls, channel slice) ----
  const int ls = tid >> 4, cpart = tid & 15;
  const int nl = n0 + ls;
  const bool lvalid = FULL || nl < a.nseq;
  const int64_t lbase = lvalid ? ((int64_t)(nl / a.n_inner) * a.p_outer + (int64_t)(nl % a.n_inner) * a.p_inner) : 0;
  float gam[VPT], bet[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) { gam[v] = a.ln_g[cpart * VPT + v]; bet[v] = a.ln_b[cpart * VPT + v]; }

  auto load_x = [&](int s) {
    XVec<C> r;
    const int st = rev ? S - 1 - s : s;
    const float* p = a.x + (lbase + (int64_t)st * a.p_step) * C + cpart * VPT;
#pragma unroll
    for (int v = 0; v < VPT; ++v) r.v[v] = lvalid ? p[v] : 0.f;
    return r;
  };
  auto ln_store = [&](const XVec<C>& xv, int buf, int s) {
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) sum += xv.v[v];
    const float mean = row16_sum(sum) * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) { const float d = xv.v[v] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(row16_sum(sq) * (1.0f / C) + 1e-5f);
    float u[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      u[v] = (xv.v[v] - mean) * rstd * gam[v] + bet[v];
      U[buf][ls][cpart * VPT + v] = u[v];
    }
    // both directions write the (identical) normalised row: branch-free beats saving 1/10 of the store traffic
    if (SAVE && lvalid && dir == 0) {
      const int st = rev ? S - 1 - s : s;
      float* p = a.save_u + (lbase + (int64_t)st * a.p_step) * C + cpart * VPT;
#pragma unroll
      for (int v = 0; v < VPT; ++v) p[v] = u[v];
    }
  };

  // ---- compute role: lane -> (sequence j, units 16w+4q..+3) ----
  const int nc = n0 + j;
  const bool cvalid = FULL || nc < a.nseq;
  const int64_t cbase = cvalid ? ((int64_t)(nc / a.n_inner) * a.p_outer + (int64_t)(nc % a.n_inner) * a.p_inner) : 0;
  const int uoff = 16 * w + 4 * q;
  f32x4 c = zero4(), h = zero4();
  if (dir == 0 && cvalid) {
    if (a.c0) c = ld4(a.c0 + (size_t)nc * H + uoff);
    if (a.h0) h = ld4(a.h0 + (size_t)nc * H + uoff);
  }
  st4(&Hb[0][j][uoff], h);

  // ---- software pipeline (per step s) ----
  //   A: acc = accx + W_hh * h_{s-1}                (64 MFMAs, needs the barrier of step s-1)
  //   B: accx' = bias + W_ih * u_{s+1}  (MFMAs, independent)  ||  cell update of step s (VALU on A's result)
  //      || LayerNorm of the prefetched row x_{s+2} -> U (VALU)   => VALU work hides in the MFMA shadow
  //   C: ds_write h_s, global stores of step s, prefetch x_{s+3}, barrier
  // Indices clamp at the end of the sequence (redundant re-normalisation of the last row) so that the
  // steady-state body is branch-free.
  {
    XVec<C> x0 = load_x(0);
    XVec<C> x1 = load_x(min(1, S - 1));
    ln_store(x0, 0, 0);
    ln_store(x1, 1, min(1, S - 1));
  }
  XVec<C> xnext = load_x(min(2, S - 1));
  __syncthreads();
  f32x4 accx[4] = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
  for (int m = 0; m < KX; ++m) {
    const f32x4 b4 = ld4(&U[0][j][16 * m + 4 * q]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int g = 0; g < 4; ++g) accx[g] = mfma16(Aih[g][m][r], b4[r], accx[g]);
  }

  const int ndir = a.ndir;
  for (int s = 0; s < S; ++s) {
    const int cur = s & 1;
    // ---- A ----
    f32x4 acc[4] = {accx[0], accx[1], accx[2], accx[3]};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 b4 = ld4(&Hb[cur][j][16 * m + 4 * q]);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = mfma16(Ahh[g][m][r], b4[r], acc[g]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- B ----
#pragma unroll
    for (int g = 0; g < 4; ++g) accx[g] = bias[g];
#pragma unroll
    for (int m = 0; m < KX; ++m) {
      const f32x4 b4 = ld4(&U[cur ^ 1][j][16 * m + 4 * q]);      // u_{s+1}
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int g = 0; g < 4; ++g) accx[g] = mfma16(Aih[g][m][r], b4[r], accx[g]);
    }
    f32x4 gi, gf, gg, go, cprev = c;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      gi[r] = sigmoidf_fast(acc[0][r]);
      gf[r] = sigmoidf_fast(acc[1][r]);
      gg[r] = tanhf_fast(acc[2][r]);
      go[r] = sigmoidf_fast(acc[3][r]);
      c[r] = gf[r] * c[r] + gi[r] * gg[r];
      h[r] = go[r] * tanhf_fast(c[r]);
    }
    ln_store(xnext, cur, min(s + 2, S - 1));                    // u_{s+2} -> U[s & 1]
    // interleave request for region B: the U reads first, then 1 MFMA : VPM VALU
    __builtin_amdgcn_sched_group_barrier(0x100, KX, 0);
#pragma unroll
    for (int i = 0; i < 16 * KX; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, C == 32 ? 6 : 11, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- C ----
    st4(&Hb[cur ^ 1][j][uoff], h);
    if (cvalid) {
      const int st = rev ? S - 1 - s : s;
      const int64_t pos = cbase + (int64_t)st * a.p_step;
      st4(a.hs + (pos * ndir + dir) * H + uoff, h);
      if (SAVE == 1) {
        float* rec = a.save_gates + (pos * ndir + dir) * (5 * H) + uoff;
        st4(rec, gi); st4(rec + H, gf); st4(rec + 2 * H, gg); st4(rec + 3 * H, go); st4(rec + 4 * H, cprev);
      } else if (SAVE == 2) {
        // compact BPTT record: the four gates as fp16 (values in [-1,1]; 2^-12 relative rounding), c_prev in fp32;
        // lane-contiguous 32 B: [w][q][gate][r]
        h16x8 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[r] = (_Float16)gi[r]; lo[4 + r] = (_Float16)gf[r];
          hi[r] = (_Float16)gg[r]; hi[4 + r] = (_Float16)go[r];
        }
        _Float16* rec = reinterpret_cast<_Float16*>(a.save_gates) + (pos * ndir + dir) * (4 * H) + (w * 4 + q) * 16;
        *reinterpret_cast<h16x8*>(rec) = lo;
        *reinterpret_cast<h16x8*>(rec + 8) = hi;
        st4(a.save_c + (pos * ndir + dir) * H + uoff, cprev);
      }
    }
    xnext = load_x(min(s + 3, S - 1));
    __syncthreads();
  }
  if (dir == 0 && cvalid) {
    if (a.hN) st4(a.hN + (size_t)nc * H + uoff, h);
    if (a.cN) st4(a.cN + (size_t)nc * H + uoff, c);
  }
}

// Backward through time, recurrent part: dgates for every step and the
// dh/dc recurrences.  Wave w owns gate rows {g*64+16w..+15}: its dgates are the
// B operand straight from registers, partial dh^T = W_hh^T[:, slice] * dgates
// is reduced across the 4 waves through LDS (one barrier per step).
template <bool FULL, bool REC16>
__global__ __launch_bounds__(256) void lstm_bwd_rec_kernel(sb_lstm_bwd_args a) {
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = blockIdx.y;
  const int n0 = blockIdx.x * 16;
  const int S = a.nsteps, ndir = a.ndir;
  const bool rev = dir == 1;
  __shared__ __attribute__((aligned(16))) float P[2][4][4][64][4];

  const float* __restrict__ whh = a.w_hh[dir];
  f32x4 At[4][4];   // [out tile ot][gate g] over r : W_hh[g*64+16w+4q+r][16*ot + j]
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) At[ot][g][r] = whh[(size_t)(g * H + 16 * w + 4 * q + r) * H + 16 * ot + j];

  const int nc = n0 + j;
  const bool valid = FULL || nc < a.nseq;
  const int64_t base = valid ? ((int64_t)(nc / a.n_inner) * a.p_outer + (int64_t)(nc % a.n_inner) * a.p_inner) : 0;
  const int uoff = 16 * w + 4 * q;

  // Raw (as stored) record of one step; converted to fp32 only at its point of use so that the prefetch of
  // step s-1, issued before the MFMAs of step s, is not waited for inside step s.
  struct Raw { f32x4 r0, r1, r2, r3, cp, dh; };
  auto load_raw = [&](int s) {   // s = forward processing index
    Raw r;
    const int st = rev ? S - 1 - s : s;
    const int64_t pos = base + (int64_t)st * a.p_step;
    if (valid) {
      if constexpr (REC16) {
        const float* rec = a.save_gates + ((pos * ndir + dir) * (4 * H) + (w * 4 + q) * 16) / 2;   // fp16 units -> floats
        r.r0 = ld4(rec); r.r1 = ld4(rec + 4);
        r.r2 = r.r3 = zero4();
        r.cp = ld4(a.save_c + (pos * ndir + dir) * H + uoff);
      } else {
        const float* rec = a.save_gates + (pos * ndir + dir) * (5 * H) + uoff;
        r.r0 = ld4(rec); r.r1 = ld4(rec + H); r.r2 = ld4(rec + 2 * H); r.r3 = ld4(rec + 3 * H); r.cp = ld4(rec + 4 * H);
      }
      r.dh = ld4(a.dhs + (pos * ndir + dir) * H + uoff);
    } else {
      r.r0 = r.r1 = r.r2 = r.r3 = r.cp = r.dh = zero4();
    }
    return r;
  };
  struct Rec { f32x4 i, f, g, o; };
  auto unpack = [&](const Raw& r) {
    Rec c;
    if constexpr (REC16) {
      const h16x8 lo = __builtin_bit_cast(h16x8, r.r0), hi = __builtin_bit_cast(h16x8, r.r1);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c.i[k] = (float)lo[k]; c.f[k] = (float)lo[4 + k]; c.g[k] = (float)hi[k]; c.o[k] = (float)hi[4 + k];
      }
    } else {
      c.i = r.r0; c.f = r.r1; c.g = r.r2; c.o = r.r3;
    }
    return c;
  };

  f32x4 dc = zero4(), dhrec = zero4();
  Raw nxt = load_raw(S - 1);
  for (int s = S - 1; s >= 0; --s) {
    const int cur = s & 1;
    const Raw raw = nxt;
    const Rec rc = unpack(raw);
    // ---- cell backward (lane-local: 4 units of one sequence) ----
    f32x4 dG[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = raw.dh[r] + dhrec[r];
      const float cc = rc.f[r] * raw.cp[r] + rc.i[r] * rc.g[r];
      const float tc = tanhf_fast(cc);
      const float dO = dh * tc;
      const float dct = dc[r] + dh * rc.o[r] * (1.0f - tc * tc);
      dG[0][r] = dct * rc.g[r] * rc.i[r] * (1.0f - rc.i[r]);
      dG[1][r] = dct * raw.cp[r] * rc.f[r] * (1.0f - rc.f[r]);
      dG[2][r] = dct * rc.i[r] * (1.0f - rc.g[r] * rc.g[r]);
      dG[3][r] = dO * rc.o[r] * (1.0f - rc.o[r]);
      dc[r] = dct * rc.f[r];
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- dgates out, then prefetch the previous step's record (youngest ops = the loads) ----
    if (valid) {
      const int st = rev ? S - 1 - s : s;
      const int64_t pos = base + (int64_t)st * a.p_step;
      float* dg = a.dgates + (pos * ndir + dir) * (4 * H) + uoff;
      st4(dg, dG[0]); st4(dg + H, dG[1]); st4(dg + 2 * H, dG[2]); st4(dg + 3 * H, dG[3]);
    }
    nxt = load_raw(max(s - 1, 0));
    __builtin_amdgcn_sched_barrier(0);
    // ---- partial dh^T = W_hh^T[:, this wave's gate rows] * dgates ; reduce over waves through LDS ----
    f32x4 part[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) part[ot] = mfma16(At[ot][g][r], dG[g][r], part[ot]);
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) st4(&P[cur][w][ot][lane][0], part[ot]);
    __syncthreads();
    dhrec = ld4(&P[cur][0][w][lane][0]) + ld4(&P[cur][1][w][lane][0]) + ld4(&P[cur][2][w][lane][0]) +
            ld4(&P[cur][3][w][lane][0]);
  }
}

}  // namespace

template <int C>
static void launch_fwd(const sb_lstm_fwd_args& a, dim3 grid, hipStream_t st) {
  const bool full = a.nseq % 16 == 0;
  const int save = a.save_gates == nullptr ? 0 : (a.save_c ? 2 : 1);
#define SB_L(SV, FL) hipLaunchKernelGGL((lstm_fwd_kernel<C, SV, FL>), grid, dim3(256), 0, st, a)
  if (save == 0) { if (full) SB_L(0, true); else SB_L(0, false); }
  else if (save == 1) { if (full) SB_L(1, true); else SB_L(1, false); }
  else { if (full) SB_L(2, true); else SB_L(2, false); }
#undef SB_L
}

extern "C" int sb_lstm_fwd(const sb_lstm_fwd_args* a, void* stream) {
  if (!a || a->nseq <= 0 || a->nsteps <= 0 || (a->ndir != 1 && a->ndir != 2)) return -1001;
  if (a->C != 16 && a->C != 32) return -1002;
  if (a->save_gates && !a->save_u) return -1003;
  dim3 grid((a->nseq + 15) / 16, a->ndir);
  if (a->lin_w && a->mma != 1) return -1003;                         // fused Linear: fp16 path only
  if (a->products != 0 && a->products != 3 && !(a->products == 2 && a->mma == 1)) return -1003;
  if (a->products == 2 && (a->save_gates || a->save_c || a->save_u)) return -1003;      // two products: inference calls only
  if (sb_lstm_fwd_vec_ok(*a)) {                                      // <= 256 (sequence, direction) chains, hs only: one
    const int rc = sb_launch_lstm_fwd_vec(*a, (hipStream_t)stream);  // workgroup per chain, fp32 matrix-vector products
    if (rc) return rc;
    SB_CHECK_LAUNCH();
    return 0;
  }
  if (a->mma == 1 && a->products == 2) { const int rc = sb_launch_lstm_fwd_bf_2p(*a, (hipStream_t)stream); if (rc) return rc; }
  else if (a->mma == 1 || a->mma == 2) { const int rc = sb_launch_lstm_fwd_bf(*a, (hipStream_t)stream); if (rc) return rc; }
  else if (a->C == 32) launch_fwd<32>(*a, grid, (hipStream_t)stream);
  else launch_fwd<16>(*a, grid, (hipStream_t)stream);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_lstm_bwd_rec(const sb_lstm_bwd_args* a, void* stream) {
  if (!a || a->nseq <= 0 || a->nsteps <= 0 || (a->ndir != 1 && a->ndir != 2)) return -1001;
  dim3 grid((a->nseq + 15) / 16, a->ndir), block(256);
  hipStream_t st = (hipStream_t)stream;
  const bool full = a->nseq % 16 == 0, r16 = a->save_c != nullptr;
  if (a->mma == 1 || a->mma == 2) {
    const int rc = sb_launch_lstm_bwd_bf(*a, st);
    if (rc) return rc;
    SB_CHECK_LAUNCH();
    return 0;
  }
  if (a->gmax) return -1003;                       // compact fp16 dgates: bf16 path only
#define SB_B(FL, R16) hipLaunchKernelGGL((lstm_bwd_rec_kernel<FL, R16>), grid, block, 0, st, *a)
  if (full) { if (r16) SB_B(true, true); else SB_B(true, false); }
  else { if (r16) SB_B(false, true); else SB_B(false, false); }
#undef SB_B
  SB_CHECK_LAUNCH();
  return 0;
}
