// Full-band local self-attention of the GridNet block (forward), gfx950.
// Reference: dis_embd3/tfgridnet_causal.py:639-684 (modules), :856-898 (forward), :722-744 (causal unfold).
//
//   Q,K : Linear(C -> E*heads) + PReLU -> per head [F*E]   -> LayerNorm(F*E)        (sb_linear_fwd + sb_head_ln)
//   V   : Linear(C -> C)       + PReLU -> per head [F*C/h] -> LayerNorm(F*C/h)
//   frame t attends to the L = local_atten_len most recent frames (carried K/V buffers supply the history; the
//   zero-filled buffer rows of a fresh state are NOT masked, as in the reference):
//        p = softmax_l( q_t . k_{t-L+1+l} / sqrt(F*E) ),   o_t = sum_l p_l v_{t-L+1+l}                 (sb_attn_core)
//   heads merged -> Linear(C -> C) + PReLU -> LayerNorm(F*C) -> + residual                  (sb_linear_fwd + sb_head_ln)
//
// sb_attn_core: one workgroup per (batch*head, 16 query frames).  Both contractions run on the fp32-input MFMA
// (exact): scores^T[window rows x 16 queries] = K_window * Q^T with both operands fetched as 16-byte rows, softmax
// over the window in LDS, then out[16 queries x features] = P^T * V_window streaming V rows from L2/HBM.
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

// LayerNorm over the (f, d) elements of each head for one (b, t) per workgroup.
//   in  [B, T, F, ldi] (head h, component d at column h*D + d; ldi >= Hh*D); prelu_a (nullable): PReLU applied on load
//   out row (b*Hh + h, t_off + t) of a [B*Hh, rows, ldo] matrix, element f*D + d; columns [F*D, ldo) zeroed
//   res (nullable, Hh == 1 only): out = res[b,t,:] + LN(...)
template <int MAXV, bool FIX>
__global__ __launch_bounds__(256) void head_ln_kernel(const float* __restrict__ in, const float* __restrict__ gam,
                                                      const float* __restrict__ bet, float* __restrict__ out,
                                                      const float* __restrict__ res, int B, int T, int F, int Hh, int D,
                                                      int rows, int t_off, int ldo, int ldi,
                                                      const float* __restrict__ prelu_a) {
  // F*Hh*D <= 256*MAXV (145*32 = 4640 needs 20; the Q / K heads, 145*4*2 = 1160, need 5: round 6 instantiates 5 / 10 / 20 --
  // the fully unrolled, masked slot loops cost their instructions whether a slot holds an element or not)
  // FIX (256 % (Hh*D) == 0 -- every shipped configuration): slot k of thread tid is element tid + 256 k either way, but then its
  // column tid % (Hh*D) -- head and component -- is the same for every k and its frequency is k (256 / (Hh*D)) + tid / (Hh*D): one
  // integer division per thread instead of two per element and loop (the kernel was instruction-bound on them: 2.1 TB/s), and
  // one running sum per thread instead of eight masked ones.  Same elements per thread, same order of every sum: same bits.
  const int bt = blockIdx.x, b = bt / T, t = bt % T;
  const int n = F * Hh * D, HD = Hh * D, FD = F * D;
  const float* x = in + (size_t)bt * F * ldi;
  const float pa = prelu_a ? prelu_a[0] : 1.0f;
  const int fpi = 256 / HD, fl = threadIdx.x / HD, hdt = threadIdx.x % HD, ht = hdt / D, dt = hdt % D;      // (FIX)
  float v[MAXV];
  __shared__ float red[4][8];                    // [wave][head] (Hh <= 8)
  __shared__ float stat[8][2];
  float part[FIX ? 1 : 8];
#pragma unroll
  for (int h = 0; h < (FIX ? 1 : 8); ++h) part[h] = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int i = threadIdx.x + 256 * k;
    v[k] = 0.f;
    if constexpr (FIX) {
      const int f = k * fpi + fl;
      if (f < F) {
        const float xv = x[f * ldi + hdt];
        v[k] = xv > 0.f ? xv : pa * xv;
        part[0] += v[k];
      }
    } else if (i < n) {
      const float xv = x[(size_t)(i / HD) * ldi + i % HD];
      v[k] = xv > 0.f ? xv : pa * xv;
      const int h = (i % HD) / D;
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) part[hh] += hh == h ? v[k] : 0.f;
    }
  }
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = threadIdx.x & 63;
  for (int h = 0; h < Hh; ++h) { const float s = wave_sum(FIX ? (h == ht ? part[0] : 0.f) : part[h]); if (ln == 0) red[wv][h] = s; }
  __syncthreads();
  if (threadIdx.x < Hh) stat[threadIdx.x][0] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / FD;
  __syncthreads();
#pragma unroll
  for (int h = 0; h < (FIX ? 1 : 8); ++h) part[h] = 0.f;
  const float mean_t = FIX ? stat[ht][0] : 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int i = threadIdx.x + 256 * k;
    if constexpr (FIX) {
      if (k * fpi + fl < F) { const float d = v[k] - mean_t; part[0] += d * d; }
    } else if (i < n) {
      const int h = (i % HD) / D;
      const float d = v[k] - stat[h][0];
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) part[hh] += hh == h ? d * d : 0.f;
    }
  }
  for (int h = 0; h < Hh; ++h) { const float s = wave_sum(FIX ? (h == ht ? part[0] : 0.f) : part[h]); if (ln == 0) red[wv][h] = s; }
  __syncthreads();
  if (threadIdx.x < Hh) stat[threadIdx.x][1] = 1.0f / sqrtf((red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / FD + 1e-5f);
  __syncthreads();
  if constexpr (FIX) {
    const float rstd_t = stat[ht][1];
    float* orow = out + ((size_t)(b * Hh + ht) * rows + t_off + t) * ldo + dt;
    const float* rrow = res ? res + (size_t)bt * n + hdt : nullptr;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int f = k * fpi + fl;
      if (f < F) {
        const int e = f * D + dt;
        float y = (v[k] - mean_t) * rstd_t * gam[e] + bet[e];
        if (res) y += rrow[f * HD];
        orow[f * D] = y;
      }
    }
  } else {
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < n) {
      const int f = i / HD, hd = i % HD, h = hd / D, d = hd % D;
      const int e = f * D + d;
      float y = (v[k] - stat[h][0]) * stat[h][1] * gam[e] + bet[e];
      if (res) y += res[(size_t)bt * n + i];
      out[((size_t)(b * Hh + h) * rows + t_off + t) * ldo + e] = y;
    }
  }
  }
  for (int h = 0; h < Hh; ++h)
    for (int e = FD + threadIdx.x; e < ldo; e += 256) out[((size_t)(b * Hh + h) * rows + t_off + t) * ldo + e] = 0.f;
}

// Workgroup -> (batch*head, 16-frame tile), XCD-aware (round 6).  Neighbouring tiles of one (batch, head) share 99 of their 115
// K / V window rows, but workgroups are dealt to the 8 XCDs round-robin by their linear id and every XCD has its own L2: with
// tiles along blockIdx.x the 40 tiles of a head landed on all 8 XCDs and each fetched its whole window from HBM -- 2.8 / 5.0 /
// 3.2 GB per launch against 0.5 GB of K, V, Q, O (profiles/r05_pmc_traffic_big_attn_wide.json: 5.6-7x, at 0.75-0.84 of the HBM
// peak).  Here the linear id g goes to XCD g % 8, so head bh = g % 8 + 8 (g / 8 / ntiles) keeps ALL tiles of a head on ONE XCD,
// in time order: a tile finds its predecessor's window rows in that L2.  (The heads beyond a multiple of 8 take the plain order.)
SB_DEVINL void attn_tile(int ntiles, int BH, int& bh, int& tile) {
  const int g = blockIdx.x, full = BH & ~7, nfull = full * ntiles;
  if (g < nfull) { const int i = g >> 3; bh = (g & 7) + 8 * (i / ntiles); tile = i % ntiles; }
  else { const int i = g - nfull; bh = full + i / ntiles; tile = i % ntiles; }
}

// Both contractions below are LATENCY-bound as first written (one or two loads, then the MFMAs that need them, per loop trip:
// ~2 loads in flight per wave, 0.17 of the fp32 matrix peak -- round 5): the loads of a whole GROUP of K chunks are now issued
// before the first MFMA of the group (8 x 16-byte row pieces, resp. 32 scalar column elements in flight per lane); indices past
// the end are clamped to valid memory and meet a zero operand, so the loops have no tail branch.
// D[i][j] = sum_k X[xrow(lane&15)][k] * Y[yrow(lane&15)][k] over ld features (both fetched as 16-byte row pieces);
// lane (j, q) ends up with D[4q + r][j].
__device__ __forceinline__ f32x4 rowdot_tile(const float* __restrict__ X, size_t xrow, const float* __restrict__ Y,
                                             size_t yrow, int ld, int q) {
  constexpr int G = 4;
  f32x4 acc = zero4();
  const int nk = ld / 16;
  const float* __restrict__ xp = X + xrow * ld + 4 * q;
  const float* __restrict__ yp = Y + yrow * ld + 4 * q;
  for (int m0 = 0; m0 < nk; m0 += G) {
    f32x4 xa[G], ya[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int m = min(m0 + g, nk - 1);
      xa[g] = ld4(xp + 16 * m);
      ya[g] = ld4(yp + 16 * m);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) acc = mfma16x4(xa[g], (m0 + g < nk) ? ya[g] : zero4(), acc);
  }
  return acc;
}
// D[i][j] = sum_k Pl[i][k] * Z[clamp(zrow0 + k)][col]  (Pl: LDS, leading dim ldp; k < 16*nk16)
__device__ __forceinline__ f32x4 lds_times_rows(const float* Pl, int ldp, int nk16, const float* __restrict__ Z,
                                                int zrow0, int zmax, int ld, int col, int j, int q) {
  constexpr int G = 8;
  f32x4 acc = zero4();
  for (int m0 = 0; m0 < nk16; m0 += G) {
    f32x4 b[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int r = 0; r < 4; ++r) b[g][r] = Z[(size_t)min(zrow0 + 16 * (m0 + g) + 4 * q + r, zmax) * ld + col];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const f32x4 a4 = ld4(&Pl[j * ldp + 16 * min(m0 + g, nk16 - 1) + 4 * q]);
      acc = mfma16x4((m0 + g < nk16) ? a4 : zero4(), b[g], acc);
    }
  }
  return acc;
}

// attention core: 1-D grid of ceil(T/16) * B*Hh workgroups (attn_tile)
__global__ __launch_bounds__(256) void attn_core_kernel(sb_attn_args a) {
  extern __shared__ __attribute__((aligned(16))) float PT[];     // [16 queries][NRp + 4]  scores -> probabilities
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  int bh, tile0;
  attn_tile((a.T + 15) / 16, a.BH, bh, tile0);
  const int t0 = tile0 * 16;
  const int L = a.L, NRp = a.NRp, ldp = NRp + 4;
  const int rows = L - 1 + a.T;
  const float* __restrict__ Kb = a.K + (size_t)bh * rows * a.ldk;
  const float* __restrict__ Qb = a.Q + (size_t)bh * a.T * a.ldk;
  const float* __restrict__ Vb = a.V + (size_t)bh * rows * a.ldv;
  const float scale = a.scale;

  // ---- scores^T[window row][query] = K_window . Q^T ; row r <-> concatenated frame t0 + r ----
  const int nrt = NRp / 16;
  const int tq = min(t0 + j, a.T - 1);                         // clamped query row for the B operand
  for (int rt = w; rt < nrt; rt += 4) {
    const int krow = min(t0 + 16 * rt + j, rows - 1);
    const f32x4 acc = rowdot_tile(Kb, krow, Qb, tq, a.ldk, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + 4 * q + r;                       // window row, query j
      const bool ok = row >= j && row < j + L && (t0 + row) < rows;
      PT[j * ldp + row] = ok ? acc[r] * scale : -INFINITY;
    }
  }
  __syncthreads();
  // ---- softmax over the window of each query: 16 threads per query ----
  {
    const int qi = tid >> 4, sub = tid & 15;
    float mx = -INFINITY;
    for (int r = sub; r < NRp; r += 16) mx = fmaxf(mx, PT[qi * ldp + r]);
    for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float s = 0.f;
    for (int r = sub; r < NRp; r += 16) { const float e = __expf(PT[qi * ldp + r] - mx); PT[qi * ldp + r] = e; s += e; }
    s = row16_sum(s);
    if (a.lse && sub == 0 && t0 + qi < a.T) a.lse[(size_t)bh * a.T + t0 + qi] = mx + __logf(s);
    const float inv = 1.0f / s;
    for (int r = sub; r < NRp; r += 16) PT[qi * ldp + r] *= inv;
  }
  __syncthreads();
  // ---- out[query][feature] = P^T . V_window ----
  const int nft = a.ldv / 16;
  const int b = bh / a.Hh, h = bh % a.Hh;
  for (int nt = w; nt < nft; nt += 4) {
    // A[i = query j][k = window rows], B[k = row][j = feature 16nt + j]
    const f32x4 acc = lds_times_rows(PT, ldp, nrt, Vb, t0, rows - 1, a.ldv, 16 * nt + j, j, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + 4 * q + r;                               // query frame, feature n = 16nt + j
      const int nfe = 16 * nt + j;
      if (t < a.T && nfe < a.F * a.Cv) {
        const int f = nfe / a.Cv, cv = nfe % a.Cv;
        a.out[(((size_t)b * a.T + t) * a.F + f) * (a.Hh * a.Cv) + h * a.Cv + cv] = acc[r];
      }
    }
  }
}

// Backward of head_ln_kernel (+ the PReLU applied on load).  One workgroup walks `rpb` consecutive (b, t) rows so the
// gamma/beta/alpha gradients accumulate in registers; partials[blockIdx.x] = [dgamma_i (n)][dbeta_i (n)][dalpha], with
// i the flat (f, h, d) element index (the host folds the heads: gamma/beta are shared by them).
//   dout: head-major rows as written by the forward (row (b*Hh+h, t_off+t) of [B*Hh, rows, ldo]);  din [B,T,F,ldi]
// Thread mapping: thread = (frequency lane tid / HD, column hd = tid % HD), so a thread's head is fixed and the
// per-head statistics are scalars (Hh masked wave reductions + one LDS exchange per statistic pair).
template <int MAXV>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void head_ln_bwd_kernel(const float* __restrict__ in, const float* __restrict__ gam,
                                                          const float* __restrict__ dout, float* __restrict__ din,
                                                          float* __restrict__ partials, int B, int T, int F, int Hh,
                                                          int D, int rows, int t_off, int ldo, int ldi,
                                                          const float* __restrict__ prelu_a, int rpb) {
  // ceil(F / (256 / (Hh*D))) <= MAXV (instantiated for 5 / 10 / 20: see head_ln_kernel)
  const int HD = Hh * D, n = F * HD;
  const float invFD = 1.0f / (F * D);
  const float pa = prelu_a ? prelu_a[0] : 1.0f;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), ln = threadIdx.x & 63;
  const int fpi = 256 / HD;
  const bool active = (int)threadIdx.x < fpi * HD;
  const int fl = threadIdx.x / HD, hd = threadIdx.x % HD, h = hd / D, d = hd % D;
  __shared__ float red[2][4][8][2];
  int phase = 0;
  auto head_sums = [&](float& a0, float& a1) {    // per-head block sums of (a0, a1); every thread gets its head's
    for (int hh = 0; hh < Hh; ++hh) {
      const float s0 = wave_sum(h == hh ? a0 : 0.f), s1 = wave_sum(h == hh ? a1 : 0.f);
      if (ln == 0) { red[phase][wv][hh][0] = s0; red[phase][wv][hh][1] = s1; }
    }
    __syncthreads();
    a0 = red[phase][0][h][0] + red[phase][1][h][0] + red[phase][2][h][0] + red[phase][3][h][0];
    a1 = red[phase][0][h][1] + red[phase][1][h][1] + red[phase][2][h][1] + red[phase][3][h][1];
    phase ^= 1;
  };
  float dgam[MAXV], dbet[MAXV], dalpha = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) dgam[k] = dbet[k] = 0.f;

  for (int bt = blockIdx.x * rpb; bt < min((blockIdx.x + 1) * rpb, B * T); ++bt) {
    const int b = bt / T, t = bt % T;
    // uniform base pointers + 32-bit per-thread offsets (keeps the 40 in-flight loads off 64-bit address registers)
    const float* x = in + (size_t)bt * F * ldi;
    const float* dy = dout + ((size_t)b * Hh * rows + t_off + t) * ldo;
    float* dx = din + (size_t)bt * F * ldi;
    const int xo = fl * ldi + hd, xs = fpi * ldi;
    const int yo = h * rows * ldo + fl * D + d, ys = fpi * D;
    float xh[MAXV], gy[MAXV];
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int f = k * fpi + fl;
      const bool ok = active & (f < F);
      const float pr = ok ? x[xo + k * xs] : 0.f;
      gy[k] = ok ? dy[yo + k * ys] : 0.f;                         // raw dout for now
      xh[k] = pr > 0.f ? pr : pa * pr;
      s0 += xh[k];
    }
    head_sums(s0, s1);
    const float mean = s0 * invFD;
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const bool ok = active & (k * fpi + fl < F);
      xh[k] = ok ? xh[k] - mean : 0.f;
      s1 += xh[k] * xh[k];
    }
    head_sums(s0, s1);
    const float rstd = 1.0f / sqrtf(s1 * invFD + 1e-5f);
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int f = min(k * fpi + fl, F - 1);
      xh[k] *= rstd;                                              // normalised value (0 for inactive slots)
      dgam[k] += gy[k] * xh[k];
      dbet[k] += gy[k];
      gy[k] *= gam[f * D + d];
      s0 += gy[k];
      s1 += gy[k] * xh[k];
    }
    head_sums(s0, s1);
    const float m1 = s0 * invFD, m2 = s1 * invFD;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
      const int f = k * fpi + fl;
      if (active & (f < F)) {
        const float dv = rstd * (gy[k] - m1 - xh[k] * m2);
        const float pr = x[xo + k * xs];                            // pre-activation again (cache hit)
        const bool pos = pr > 0.f;
        dalpha += pos ? 0.f : dv * pr;
        dx[xo + k * xs] = pos ? dv : pa * dv;
      }
    }
  }
  float* prow = partials + (size_t)blockIdx.x * (2 * n + 1);
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int f = k * fpi + fl;
    if (active & (f < F)) { prow[f * HD + hd] = dgam[k]; prow[n + f * HD + hd] = dbet[k]; }
  }
  const float da = wave_sum(dalpha);
  __syncthreads();
  if (ln == 0) red[0][wv][0][0] = da;
  __syncthreads();
  if (threadIdx.x == 0) prow[2 * n] = red[0][0][0][0] + red[0][1][0][0] + red[0][2][0][0] + red[0][3][0][0];
}

// dQ (+ delta_t = sum_l p_l dp_l): grid (ceil(T/16), B*Hh), same tiling as the forward.
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(sb_attn_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) float SM[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  int bh, tile0;
  attn_tile((a.T + 15) / 16, a.BH, bh, tile0);
  const int t0 = tile0 * 16;
  const int L = a.L, NRp = a.NRp, ldp = NRp + 4, rows = L - 1 + a.T;
  float* PT = SM;                       // [16 queries][ldp] probabilities
  float* DP = SM + 16 * ldp;            // [16 queries][ldp] dP -> dS
  const float* __restrict__ Kb = a.K + (size_t)bh * rows * a.ldk;
  const float* __restrict__ Qb = a.Q + (size_t)bh * a.T * a.ldk;
  const float* __restrict__ Vb = a.V + (size_t)bh * rows * a.ldv;
  const float* __restrict__ Gb = a.dO + (size_t)bh * a.T * a.ldv;
  const int nrt = NRp / 16;
  const int tq = min(t0 + j, a.T - 1);
  const float lse = a.lse[(size_t)bh * a.T + tq];
  for (int rt = w; rt < nrt; rt += 4) {
    const int krow = min(t0 + 16 * rt + j, rows - 1);
    const f32x4 s = rowdot_tile(Kb, krow, Qb, tq, a.ldk, q);
    const f32x4 dp = rowdot_tile(Vb, krow, Gb, tq, a.ldv, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * rt + 4 * q + r;
      const bool ok = row >= j && row < j + L && (t0 + row) < rows;
      PT[j * ldp + row] = ok ? __expf(s[r] * a.scale - lse) : 0.f;
      DP[j * ldp + row] = dp[r];
    }
  }
  __syncthreads();
  {
    const int qi = tid >> 4, sub = tid & 15;
    float d = 0.f;
    for (int r = sub; r < NRp; r += 16) d += PT[qi * ldp + r] * DP[qi * ldp + r];
    d = row16_sum(d);
    if (sub == 0 && t0 + qi < a.T) a.delta[(size_t)bh * a.T + t0 + qi] = d;
    for (int r = sub; r < NRp; r += 16) DP[qi * ldp + r] = PT[qi * ldp + r] * (DP[qi * ldp + r] - d) * a.scale;
  }
  __syncthreads();
  for (int nt = w; nt < a.ldk / 16; nt += 4) {
    const f32x4 acc = lds_times_rows(DP, ldp, nrt, Kb, t0, rows - 1, a.ldk, 16 * nt + j, j, q);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = t0 + 4 * q + r;
      if (t < a.T) a.dQ[((size_t)bh * a.T + t) * a.ldk + 16 * nt + j] = acc[r];
    }
  }
}

// dK, dV of the current-frame rows (the carried buffer rows get no gradient): grid (ceil(T/16), B*Hh); the tile of
// 16 key rows r = L-1 + 16*bx + j gathers from the queries 16*bx .. 16*bx + L + 14 that attend to them.
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(sb_attn_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) float SM[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  int bh, tile0;
  attn_tile((a.T + 15) / 16, a.BH, bh, tile0);
  const int tb = tile0 * 16;
  const int L = a.L, NRp = a.NRp, ldp = NRp + 4, rows = L - 1 + a.T;
  float* PT = SM;                       // [16 key rows][ldp queries]
  float* DS = SM + 16 * ldp;
  const float* __restrict__ Kb = a.K + (size_t)bh * rows * a.ldk;
  const float* __restrict__ Qb = a.Q + (size_t)bh * a.T * a.ldk;
  const float* __restrict__ Vb = a.V + (size_t)bh * rows * a.ldv;
  const float* __restrict__ Gb = a.dO + (size_t)bh * a.T * a.ldv;
  const int nqt = NRp / 16;
  const int r0 = L - 1 + tb;
  const int krow = min(r0 + j, rows - 1);
  for (int qt = w; qt < nqt; qt += 4) {
    const int tqa = min(tb + 16 * qt + j, a.T - 1);             // A-operand row: query 16qt + j
    const f32x4 s = rowdot_tile(Qb, tqa, Kb, krow, a.ldk, q);    // D[query 4q+r][row j]
    const f32x4 dp = rowdot_tile(Gb, tqa, Vb, krow, a.ldv, q);
    f32x4 p4, ds4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int wq = 16 * qt + 4 * q + r, tq = tb + wq;
      const int off = L - 1 + j - wq;                            // position of row r0+j inside query tq's window
      const bool ok = (tq < a.T) & (off >= 0) & (off < L) & (r0 + j < rows);
      const int tqc = min(tq, a.T - 1);
      const float p = ok ? __expf(s[r] * a.scale - a.lse[(size_t)bh * a.T + tqc]) : 0.f;
      p4[r] = p;
      ds4[r] = p * (dp[r] - a.delta[(size_t)bh * a.T + tqc]) * a.scale;
    }
    st4(&PT[j * ldp + 16 * qt + 4 * q], p4);
    st4(&DS[j * ldp + 16 * qt + 4 * q], ds4);
  }
  __syncthreads();
  const int nk = a.ldk / 16, nv = a.ldv / 16;
  for (int nt = w; nt < nk + nv; nt += 4) {
    const bool isk = nt < nk;
    const int c = isk ? nt : nt - nk;
    const f32x4 acc = isk ? lds_times_rows(DS, ldp, nqt, Qb, tb, a.T - 1, a.ldk, 16 * c + j, j, q)
                          : lds_times_rows(PT, ldp, nqt, Gb, tb, a.T - 1, a.ldv, 16 * c + j, j, q);
    float* o = isk ? a.dK : a.dV;
    const int ld = isk ? a.ldk : a.ldv;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = tb + 4 * q + r;
      if (t < a.T) o[((size_t)bh * a.T + t) * ld + 16 * c + j] = acc[r];
    }
  }
}

}  // namespace

extern "C" int sb_head_ln(const float* in, const float* gamma, const float* beta, float* out, const float* res, int B,
                          int T, int F, int Hh, int D, int rows, int t_off, int ldo, int ldi, const float* prelu_a,
                          void* stream) {
  if (Hh > 8 || F * Hh * D > 256 * 20 || (res && Hh != 1) || ldi < Hh * D) return -1002;
  const int need = (F * Hh * D + 255) / 256;
#define SB_HL(M_, X_) hipLaunchKernelGGL((head_ln_kernel<M_, X_>), dim3(B * T), dim3(256), 0, (hipStream_t)stream, in, gamma, beta, out, res, B, T, F, \
                                         Hh, D, rows, t_off, ldo, ldi, prelu_a)
  const bool fix = 256 % (Hh * D) == 0;             // a thread's column is the same in every slot: see the kernel
  if (fix) { if (need <= 5) SB_HL(5, true); else if (need <= 10) SB_HL(10, true); else SB_HL(20, true); }
  else { if (need <= 5) SB_HL(5, false); else if (need <= 10) SB_HL(10, false); else SB_HL(20, false); }
#undef SB_HL
  SB_CHECK_LAUNCH();
  return 0;
}

static constexpr int kHeadLnBwdRows = 8;          // (b, t) rows walked by one workgroup
extern "C" int sb_head_ln_bwd_grid(int B, int T) { return (B * T + kHeadLnBwdRows - 1) / kHeadLnBwdRows; }

extern "C" int sb_head_ln_bwd(const float* in, const float* gamma, const float* dout, float* din, float* partials, int B,
                              int T, int F, int Hh, int D, int rows, int t_off, int ldo, int ldi, const float* prelu_a,
                              void* stream) {
  if (Hh > 8 || Hh * D > 256 || ldi < Hh * D) return -1002;
  const int need = (F + 256 / (Hh * D) - 1) / (256 / (Hh * D));
  if (need > 20) return -1002;
#define SB_HB(M_) hipLaunchKernelGGL(head_ln_bwd_kernel<M_>, dim3(sb_head_ln_bwd_grid(B, T)), dim3(256), 0, (hipStream_t)stream, in, gamma, \
                                     dout, din, partials, B, T, F, Hh, D, rows, t_off, ldo, ldi, prelu_a, kHeadLnBwdRows)
  if (need <= 5) SB_HB(5); else if (need <= 10) SB_HB(10); else SB_HB(20);
#undef SB_HB
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_attn_core_bwd(const sb_attn_bwd_args* ap, void* stream) {
  if (!ap || ap->ldk % 16 || ap->ldv % 16 || ap->NRp % 16 || ap->NRp < ap->L + 15) return -1002;
  const size_t lds = (size_t)2 * 16 * (ap->NRp + 4) * sizeof(float);
  if (lds > 64 * 1024) return -1005;
  dim3 grid(((ap->T + 15) / 16) * ap->BH);
  hipLaunchKernelGGL(attn_bwd_dq_kernel, grid, dim3(256), lds, (hipStream_t)stream, *ap);
  SB_CHECK_LAUNCH();
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, grid, dim3(256), lds, (hipStream_t)stream, *ap);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_attn_core(const sb_attn_args* ap, void* stream) {
  if (!ap || ap->ldk % 16 || ap->ldv % 16 || ap->NRp % 16 || ap->NRp < ap->L + 15) return -1002;
  const size_t lds = (size_t)16 * (ap->NRp + 4) * sizeof(float);
  dim3 grid(((ap->T + 15) / 16) * ap->BH);
  hipLaunchKernelGGL(attn_core_kernel, grid, dim3(256), lds, (hipStream_t)stream, *ap);
  SB_CHECK_LAUNCH();
  return 0;
}
