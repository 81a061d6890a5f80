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
//   in  [B, T, F, Hh*D]  (head h, component d at column h*D + d)
//   out row (b*Hh + h, t_off + t) of a [B*Hh, rows, ldo] matrix, element f*D + d; columns [F*D, ldo) zeroed
//   res (nullable, Hh == 1 only): out = res[b,t,:] + LN(...)
__global__ __launch_bounds__(256) void head_ln_kernel(const float* __restrict__ in, const float* __restrict__ gam,
                                                      const float* __restrict__ bet, float* __restrict__ out,
                                                      const float* __restrict__ res, int B, int T, int F, int Hh, int D,
                                                      int rows, int t_off, int ldo) {
  constexpr int MAXV = 20;                       // F*Hh*D <= 256*MAXV  (145*32 = 4640 fits)
  const int bt = blockIdx.x, b = bt / T, t = bt % T;
  const int n = F * Hh * D, HD = Hh * D, FD = F * D;
  const float* x = in + (size_t)bt * n;
  float v[MAXV];
  __shared__ float red[4][8];                    // [wave][head] (Hh <= 8)
  __shared__ float stat[8][2];
  float part[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) part[h] = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int i = threadIdx.x + 256 * k;
    v[k] = i < n ? x[i] : 0.f;
    if (i < n) {
      const int h = (i % HD) / D;
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) part[hh] += hh == h ? v[k] : 0.f;
    }
  }
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  for (int h = 0; h < Hh; ++h) { const float s = wave_sum(part[h]); if (ln == 0) red[wv][h] = s; }
  __syncthreads();
  if (threadIdx.x < Hh) stat[threadIdx.x][0] = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / FD;
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 8; ++h) part[h] = 0.f;
#pragma unroll
  for (int k = 0; k < MAXV; ++k) {
    const int i = threadIdx.x + 256 * k;
    if (i < n) {
      const int h = (i % HD) / D;
      const float d = v[k] - stat[h][0];
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) part[hh] += hh == h ? d * d : 0.f;
    }
  }
  for (int h = 0; h < Hh; ++h) { const float s = wave_sum(part[h]); if (ln == 0) red[wv][h] = s; }
  __syncthreads();
  if (threadIdx.x < Hh) stat[threadIdx.x][1] = 1.0f / sqrtf((red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / FD + 1e-5f);
  __syncthreads();
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
  for (int h = 0; h < Hh; ++h)
    for (int e = FD + threadIdx.x; e < ldo; e += 256) out[((size_t)(b * Hh + h) * rows + t_off + t) * ldo + e] = 0.f;
}

// attention core: grid (ceil(T/16), B*Hh)
__global__ __launch_bounds__(256) void attn_core_kernel(sb_attn_args a) {
  extern __shared__ __attribute__((aligned(16))) float PT[];     // [16 queries][NRp + 4]  scores -> probabilities
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, j = lane & 15;
  const int bh = blockIdx.y, t0 = blockIdx.x * 16;
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
    f32x4 acc = zero4();
    for (int m = 0; m < a.ldk / 16; ++m) {
      const f32x4 a4 = ld4(Kb + (size_t)krow * a.ldk + 16 * m + 4 * q);
      const f32x4 b4 = ld4(Qb + (size_t)tq * a.ldk + 16 * m + 4 * q);
      acc = mfma16x4(a4, b4, acc);
    }
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
    const float inv = 1.0f / s;
    for (int r = sub; r < NRp; r += 16) PT[qi * ldp + r] *= inv;
  }
  __syncthreads();
  // ---- out[query][feature] = P^T . V_window ----
  const int nft = a.ldv / 16;
  const int b = bh / a.Hh, h = bh % a.Hh;
  for (int nt = w; nt < nft; nt += 4) {
    f32x4 acc = zero4();
    for (int m = 0; m < nrt; ++m) {
      const f32x4 a4 = ld4(&PT[j * ldp + 16 * m + 4 * q]);         // A[i = query j][k = rows 16m+4q..+3]
      f32x4 b4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int vrow = min(t0 + 16 * m + 4 * q + r, rows - 1);
        b4[r] = Vb[(size_t)vrow * a.ldv + 16 * nt + j];           // B[k = row][j = feature 16nt + j]
      }
      acc = mfma16x4(a4, b4, acc);
    }
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

}  // namespace

extern "C" int sb_head_ln(const float* in, const float* gamma, const float* beta, float* out, const float* res, int B,
                          int T, int F, int Hh, int D, int rows, int t_off, int ldo, void* stream) {
  if (Hh > 8 || F * Hh * D > 256 * 20 || (res && Hh != 1)) return -1002;
  hipLaunchKernelGGL(head_ln_kernel, dim3(B * T), dim3(256), 0, (hipStream_t)stream, in, gamma, beta, out, res, B, T, F,
                     Hh, D, rows, t_off, ldo);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_attn_core(const sb_attn_args* ap, void* stream) {
  if (!ap || ap->ldk % 16 || ap->ldv % 16 || ap->NRp % 16 || ap->NRp < ap->L + 15) return -1002;
  const size_t lds = (size_t)16 * (ap->NRp + 4) * sizeof(float);
  dim3 grid((ap->T + 15) / 16, ap->BH);
  hipLaunchKernelGGL(attn_core_kernel, grid, dim3(256), lds, (hipStream_t)stream, *ap);
  SB_CHECK_LAUNCH();
  return 0;
}
