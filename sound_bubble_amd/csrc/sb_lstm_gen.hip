// Generic-shape recurrent LSTM kernels for gfx950: the layer widths the tuned fp16x3 kernels of sb_lstm_bf_*.hip are not
// built for -- first of all the reference constructor's own defaults, D = 64 / H = 128
// (src/models/tfgridnet_realtime_clean_dis_embd3/net.py:21-26, ..._optim/net.py:21-26; no shipped experiment JSON uses them).
//
// Same tiling as everywhere in this library: a workgroup owns 16 sequences for the whole walk, gates^T[4H x 16] =
// W[4H x (C + H)] . [u; h]^T on the matrix pipe with the weights as the A operand, wave w owns hidden units 16w .. 16w + 15
// of all four gates (lane-local cell update), LayerNorm fused into the loader.  What differs from the tuned kernels:
//   * H / 16 waves per workgroup (8 for H = 128: two per SIMD, 256 registers each), so W_hh stays register-resident
//     (H registers per lane) and W_ih is parked in LDS in MFMA-fragment order -- one conflict-free ds_read_b128 per lane and
//     K chunk (131 KB for C = 64, H = 128: the kernel asks for 157 KB of the CU's 160 KB);
//   * forward: fp16 hi + lo operands, three products per multiply-accumulate on v_mfma_f32_16x16x32_f16 -- the tuned forward's
//     arithmetic (late round 6: the exact fp32 form, v_mfma_f32_16x16x4_f32, was matrix-pipe-bound at 7.1 us per step at H = 128 /
//     C = 64; this one measures 2.7x faster end to end at the reference constructor's widths and holds the same 5e-6 bars);
//     backward recurrence: v_mfma_f32_16x16x4_f32, exact fp32 products (its operands are gradients: an fp16 form needs the
//     tuned kernels' scaling, not repeated here); position-major fp32 records (i, f, g, o, c_prev: the "legacy" record form of
//     sb_lstm_fwd), fp32 dgates; the byte diets of the tuned kernels (blocked Q24 records, fused streaming part) are not repeated;
//   * the backward recurrence passes its dgates through LDS as the B operand of dh^T = W_hh^T . dgates (wave w owns output
//     tile w over the full K = 4H), one barrier per step, instead of reducing per-wave partial products.
// The weight / input gradients that follow are position-wise GEMMs over the dgates (sb_linear_fwd, sb_wgrad's generic form).
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------------------
typedef _Float16 g16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 g16x4 __attribute__((ext_vector_type(4)));
struct HiLo8 { g16x8 hi, lo; };
// x = hi + lo with both terms fp16 (the tuned kernels' operand form: sb_lstm_bf_common.h splitn1)
SB_DEVINL void split1(float x, _Float16& hi, _Float16& lo) { hi = (_Float16)x; lo = (_Float16)(x - (float)hi); }
SB_DEVINL HiLo8 split8(const float (&x)[8]) {
  HiLo8 r;
#pragma unroll
  for (int k = 0; k < 8; ++k) { _Float16 h, l; split1(x[k], h, l); r.hi[k] = h; r.lo[k] = l; }
  return r;
}
// acc += W x over one 32-wide K chunk with W = Wh + Wl, x = xh + xl: the three leading products, smallest first (the order of
// the tuned forward's mma6); Wl xl (2^-22 relative) is dropped
SB_DEVINL f32x4 mma3(const HiLo8& W, const g16x8 xh, const g16x8 xl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.lo, xh, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(W.hi, xl, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(W.hi, xh, acc, 0, 0, 0);
}

template <int C, int H, bool SAVE>
__global__ __launch_bounds__(H * 4) void lstm_gen_fwd_kernel(sb_lstm_gen_fwd_args a) {
  constexpr int NW = H / 16;             // waves
  constexpr int KX = (C + 31) / 32;      // 32-wide K chunks of the input part (C = 16: one chunk, upper half zero)
  constexpr int KH = H / 32;             // ... of the hidden part
  constexpr int VPT = C / 16;            // floats per loader thread
  constexpr int CP = 32 * KX + 8, HP = H + 8;      // LDS rows in halves (16-byte aligned, 4-dword bank shift per row)
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = blockIdx.y;
  const int n0 = blockIdx.x * 16;
  const int S = a.nsteps, ndir = a.ndir;
  const bool rev = dir == 1;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  g16x8* WL = reinterpret_cast<g16x8*>(smem);                                       // [4][NW][KX][2 terms][64] fragments of W_ih
  _Float16* U = reinterpret_cast<_Float16*>(WL + 4 * NW * KX * 2 * 64);             // [2 bufs][2 terms][16][CP]
  _Float16* Hb = U + 2 * 2 * 16 * CP;                                               // [2 bufs][2 terms][16][HP]

  // ---- weights as fp16 hi + lo fragments (A operand of v_mfma_f32_16x16x32_f16: lane (i = j, k = 8q .. 8q + 7)):
  //      W_hh -> registers (H per lane, as the fp32 fragments of round 6's first version), W_ih -> LDS, bias -> registers ----
  const float* __restrict__ wih = a.w_ih[dir];
  const float* __restrict__ whh = a.w_hh[dir];
  HiLo8 Ahh[4][KH];
  f32x4 bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = g * H + 16 * w + j;
#pragma unroll
    for (int m = 0; m < KX; ++m) {
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const int col = 32 * m + 8 * q + k; t[k] = col < C ? wih[(size_t)row * C + col] : 0.f; }
      const HiLo8 f = split8(t);
      WL[(((g * NW + w) * KX + m) * 2 + 0) * 64 + lane] = f.hi;
      WL[(((g * NW + w) * KX + m) * 2 + 1) * 64 + lane] = f.lo;
    }
#pragma unroll
    for (int m = 0; m < KH; ++m) {
      float t[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) t[k] = whh[(size_t)row * H + 32 * m + 8 * q + k];
      Ahh[g][m] = split8(t);
    }
    const int u0 = g * H + 16 * w + 4 * q;
    bias[g] = ld4(a.b_ih[dir] + u0) + ld4(a.b_hh[dir] + u0);
  }
  if constexpr (32 * KX > C) {               // the K padding of the input operand stays zero (the loader writes columns < C)
    for (int i = tid; i < 2 * 2 * 16 * CP; i += H * 4) U[i] = (_Float16)0.f;
    __syncthreads();
  }

  // ---- loader role (the first 256 threads): thread -> (sequence ls, channel slice) ----
  const bool loader = tid < 256;
  const int ls = (tid >> 4) & 15, cpart = tid & 15;
  const int nl = n0 + ls;
  const bool lvalid = loader && nl < a.nseq;
  const int64_t lbase = lvalid ? ((int64_t)(nl / a.n_inner) * a.p_outer + (int64_t)(nl % a.n_inner) * a.p_inner) : 0;
  float gam[VPT], bet[VPT];
#pragma unroll
  for (int v = 0; v < VPT; ++v) { gam[v] = a.ln_g[cpart * VPT + v]; bet[v] = a.ln_b[cpart * VPT + v]; }

  struct XV { float v[VPT]; };
  auto load_x = [&](int s) {
    XV r;
    const int st = rev ? S - 1 - s : s;
    const float* p = a.x + (lbase + (int64_t)st * a.p_step) * C + cpart * VPT;
#pragma unroll
    for (int v = 0; v < VPT; ++v) r.v[v] = lvalid ? p[v] : 0.f;
    return r;
  };
  auto ln_store = [&](const XV& xv, int buf, int s) {
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) sum += xv.v[v];
    const float mean = row16_sum(sum) * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) { const float d = xv.v[v] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(row16_sum(sq) * (1.0f / C) + 1e-5f);
    if (!loader) return;
    float u[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      u[v] = (xv.v[v] - mean) * rstd * gam[v] + bet[v];
      _Float16 hh, ll;
      split1(u[v], hh, ll);
      U[((buf * 2 + 0) * 16 + ls) * CP + cpart * VPT + v] = hh;
      U[((buf * 2 + 1) * 16 + ls) * CP + cpart * VPT + v] = ll;
    }
    if (SAVE && lvalid && dir == 0) {
      const int st = rev ? S - 1 - s : s;
      float* p = a.save_u + (lbase + (int64_t)st * a.p_step) * C + cpart * VPT;
#pragma unroll
      for (int v = 0; v < VPT; ++v) p[v] = u[v];
    }
  };

  // ---- compute role: lane -> (sequence j, units 16w + 4q .. + 3) ----
  const int nc = n0 + j;
  const bool cvalid = nc < a.nseq;
  const int64_t cbase = cvalid ? ((int64_t)(nc / a.n_inner) * a.p_outer + (int64_t)(nc % a.n_inner) * a.p_inner) : 0;
  const int uoff = 16 * w + 4 * q;
  f32x4 c = zero4(), h = zero4();
  if (dir == 0 && cvalid) {
    if (a.c0) c = ld4(a.c0 + (size_t)nc * H + uoff);
    if (a.h0) h = ld4(a.h0 + (size_t)nc * H + uoff);
  }
  auto store_h = [&](int buf) {              // h of this lane's four units as hi / lo halves
    g16x4 hh, ll;
#pragma unroll
    for (int r = 0; r < 4; ++r) { _Float16 x, y; split1(h[r], x, y); hh[r] = x; ll[r] = y; }
    *reinterpret_cast<g16x4*>(&Hb[((buf * 2 + 0) * 16 + j) * HP + uoff]) = hh;
    *reinterpret_cast<g16x4*>(&Hb[((buf * 2 + 1) * 16 + j) * HP + uoff]) = ll;
  };
  store_h(0);

  {
    XV x0 = load_x(0);
    XV x1 = load_x(min(1, S - 1));
    ln_store(x0, 0, 0);
    ln_store(x1, 1, min(1, S - 1));
  }
  XV xnext = load_x(min(2, S - 1));
  __syncthreads();
  // accx = bias + W_ih . u_s, one step ahead of its use
  auto input_part = [&](int buf, f32x4 (&out)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g) out[g] = bias[g];
#pragma unroll
    for (int m = 0; m < KX; ++m) {
      const g16x8 xh = *reinterpret_cast<const g16x8*>(&U[((buf * 2 + 0) * 16 + j) * CP + 32 * m + 8 * q]);
      const g16x8 xl = *reinterpret_cast<const g16x8*>(&U[((buf * 2 + 1) * 16 + j) * CP + 32 * m + 8 * q]);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        HiLo8 W;
        W.hi = WL[(((g * NW + w) * KX + m) * 2 + 0) * 64 + lane];
        W.lo = WL[(((g * NW + w) * KX + m) * 2 + 1) * 64 + lane];
        out[g] = mma3(W, xh, xl, out[g]);
      }
    }
  };
  f32x4 accx[4];
  input_part(0, accx);

  for (int s = 0; s < S; ++s) {
    const int cur = s & 1;
    // ---- gates = accx + W_hh . h_{s-1} ----
    f32x4 acc[4] = {accx[0], accx[1], accx[2], accx[3]};
#pragma unroll
    for (int m = 0; m < KH; ++m) {
      const g16x8 xh = *reinterpret_cast<const g16x8*>(&Hb[((cur * 2 + 0) * 16 + j) * HP + 32 * m + 8 * q]);
      const g16x8 xl = *reinterpret_cast<const g16x8*>(&Hb[((cur * 2 + 1) * 16 + j) * HP + 32 * m + 8 * q]);
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[g] = mma3(Ahh[g][m], xh, xl, acc[g]);
    }
    // ---- input part of step s + 1 (independent MFMAs) next to the cell update of step s ----
    input_part(cur ^ 1, accx);
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
    ln_store(xnext, cur, min(s + 2, S - 1));                    // u_{s+2} -> U[s & 1] (its readers passed the last barrier)
    store_h(cur ^ 1);
    if (cvalid) {
      const int st = rev ? S - 1 - s : s;
      const int64_t pos = cbase + (int64_t)st * a.p_step;
      st4(a.hs + (pos * ndir + dir) * H + uoff, h);
      if (SAVE) {
        float* rec = a.save_gates + (pos * ndir + dir) * (5 * H) + uoff;
        st4(rec, gi); st4(rec + H, gf); st4(rec + 2 * H, gg); st4(rec + 3 * H, go); st4(rec + 4 * H, cprev);
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

template <int C, int H>
constexpr size_t fwd_lds_bytes() {       // W_ih fragments (hi + lo, 16 bytes per lane each) + the U and H operand tiles (halves)
  return (size_t)4 * (H / 16) * ((C + 31) / 32) * 2 * 64 * 16 + (size_t)(2 * 2 * 16 * (32 * ((C + 31) / 32) + 8) + 2 * 2 * 16 * (H + 8)) * 2;
}

template <int C, int H>
int launch_gen_fwd(const sb_lstm_gen_fwd_args& a, hipStream_t st) {
  dim3 grid((a.nseq + 15) / 16, a.ndir), block(H * 4);
  constexpr size_t lds = fwd_lds_bytes<C, H>();
  static_assert(lds <= 160 * 1024, "LDS budget of a CU");
  if (a.save_gates) {
    (void)hipFuncSetAttribute((const void*)lstm_gen_fwd_kernel<C, H, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((lstm_gen_fwd_kernel<C, H, true>), grid, block, lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)lstm_gen_fwd_kernel<C, H, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((lstm_gen_fwd_kernel<C, H, false>), grid, block, lds, st, a);
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// backward through time, recurrent part: dgates of every step and the dh / dc recurrences
// ---------------------------------------------------------------------------------------------------------------------------
template <int H>
__global__ __launch_bounds__(H * 4) void lstm_gen_bwd_rec_kernel(sb_lstm_gen_bwd_args a) {
  // dh^T[H x 16] = W_hh^T[H x 4H] . dgates[4H x 16] on fp16 operands (late round 6; the exact fp32 form chained 4H / 4 dependent
  // v_mfma_f32_16x16x4_f32 on one accumulator every step): W_hh^T as hi + lo fragments in registers (H per lane), the step's
  // dgates as S dgates = hi + 2^-11 lo' through LDS -- S = 2^-ceil(log2 gmax) from the incoming gradient's maximum
  // (sb_lstm_gen_bwd_args.gmax; the recurrence is linear in it: the tuned kernels' DG16 scaling, sb_lstm_bf_bwd.hip) --, three
  // products per multiply-accumulate on six independent accumulators.  The dgates that LEAVE are the fp32 values, unscaled.
  constexpr int KG = 4 * H / 32;          // 32-wide K chunks over the 4H gate rows
  constexpr int GP = 4 * H + 8;           // padded dgates row of one sequence (halves)
  constexpr float kLoUp = 2048.0f, kLoDn = 1.0f / 2048.0f;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = blockIdx.y;
  const int n0 = blockIdx.x * 16;
  const int S = a.nsteps, ndir = a.ndir;
  const bool rev = dir == 1;
  extern __shared__ __attribute__((aligned(16))) float smem_b[];
  _Float16* DG = reinterpret_cast<_Float16*>(smem_b);      // [2 bufs][2 terms][16][GP]

  float gS = 1.0f;
  if (a.gmax) {
    const float m = a.gmax[0];
    gS = (m > 0.f && m < 3.0e38f) ? exp2f(-ceilf(log2f(m))) : 1.0f;
  }
  gS = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(gS)));
  const float invS = 1.0f / gS;

  // A operand: lane (i = j, k = 8q .. 8q + 7 of chunk m) holds W_hh[32m + 8q + k][16w + j]
  const float* __restrict__ whh = a.w_hh[dir];
  HiLo8 At[KG];
#pragma unroll
  for (int m = 0; m < KG; ++m) {
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = whh[(size_t)(32 * m + 8 * q + k) * H + 16 * w + j];
    At[m] = split8(t);
  }

  const int nc = n0 + j;
  const bool valid = nc < a.nseq;
  const int64_t base = valid ? ((int64_t)(nc / a.n_inner) * a.p_outer + (int64_t)(nc % a.n_inner) * a.p_inner) : 0;
  const int uoff = 16 * w + 4 * q;

  struct Raw { f32x4 i, f, g, o, cp, dh; };
  auto load_raw = [&](int s) {
    Raw r;
    const int st = rev ? S - 1 - s : s;
    const int64_t pos = base + (int64_t)st * a.p_step;
    if (valid) {
      const float* rec = a.save_gates + (pos * ndir + dir) * (5 * H) + uoff;
      r.i = ld4(rec); r.f = ld4(rec + H); r.g = ld4(rec + 2 * H); r.o = ld4(rec + 3 * H); r.cp = ld4(rec + 4 * H);
      r.dh = ld4(a.dhs + (pos * ndir + dir) * H + uoff);
    } else {
      r.i = r.f = r.g = r.o = r.cp = r.dh = zero4();
    }
    return r;
  };

  f32x4 dc = zero4(), dhrec = zero4();
  Raw nxt = load_raw(S - 1);
  for (int s = S - 1; s >= 0; --s) {
    const int cur = s & 1;
    const Raw rc = nxt;
    f32x4 dG[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = rc.dh[r] + dhrec[r];
      const float cc = rc.f[r] * rc.cp[r] + rc.i[r] * rc.g[r];
      const float tc = tanhf_fast(cc);
      const float dO = dh * tc;
      const float dct = dc[r] + dh * rc.o[r] * (1.0f - tc * tc);
      dG[0][r] = dct * rc.g[r] * rc.i[r] * (1.0f - rc.i[r]);
      dG[1][r] = dct * rc.cp[r] * rc.f[r] * (1.0f - rc.f[r]);
      dG[2][r] = dct * rc.i[r] * (1.0f - rc.g[r] * rc.g[r]);
      dG[3][r] = dO * rc.o[r] * (1.0f - rc.o[r]);
      dc[r] = dct * rc.f[r];
    }
    _Float16* rowh = &DG[((cur * 2 + 0) * 16 + j) * GP];
    _Float16* rowl = &DG[((cur * 2 + 1) * 16 + j) * GP];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      g16x4 hh, ll;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = dG[g][r] * gS;
        const _Float16 h1 = (_Float16)v;
        hh[r] = h1;
        ll[r] = (_Float16)__builtin_fmaf((float)h1, -kLoUp, v * kLoUp);
      }
      *reinterpret_cast<g16x4*>(rowh + g * H + uoff) = hh;
      *reinterpret_cast<g16x4*>(rowl + g * H + uoff) = ll;
    }
    if (valid) {
      const int st = rev ? S - 1 - s : s;
      const int64_t pos = base + (int64_t)st * a.p_step;
      float* dg = a.dgates + (pos * ndir + dir) * (4 * H) + uoff;
      st4(dg, dG[0]); st4(dg + H, dG[1]); st4(dg + 2 * H, dG[2]); st4(dg + 3 * H, dG[3]);
    }
    nxt = load_raw(max(s - 1, 0));
    __syncthreads();
    // dh_{s-1}^T tile w = sum over the 4H gate rows (all waves' dgates, from LDS)
    f32x4 ah[4] = {zero4(), zero4(), zero4(), zero4()}, al[2] = {zero4(), zero4()};
#pragma unroll
    for (int m = 0; m < KG; ++m) {
      const g16x8 bh = *reinterpret_cast<const g16x8*>(rowh + 32 * m + 8 * q);
      const g16x8 bl = *reinterpret_cast<const g16x8*>(rowl + 32 * m + 8 * q);
      al[m & 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(At[m].hi, bl, al[m & 1], 0, 0, 0);
      ah[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(At[m].lo, bh, ah[m & 3], 0, 0, 0);
      ah[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(At[m].hi, bh, ah[m & 3], 0, 0, 0);
    }
    const f32x4 sh = (ah[0] + ah[1]) + (ah[2] + ah[3]), sl = al[0] + al[1];
#pragma unroll
    for (int r = 0; r < 4; ++r) dhrec[r] = __builtin_fmaf(sl[r], kLoDn, sh[r]) * invS;
  }
}

template <int H>
int launch_gen_bwd(const sb_lstm_gen_bwd_args& a, hipStream_t st) {
  dim3 grid((a.nseq + 15) / 16, a.ndir), block(H * 4);
  constexpr size_t lds = (size_t)2 * 2 * 16 * (4 * H + 8) * sizeof(_Float16);
  (void)hipFuncSetAttribute((const void*)lstm_gen_bwd_rec_kernel<H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((lstm_gen_bwd_rec_kernel<H>), grid, block, lds, st, a);
  return 0;
}

}  // namespace

extern "C" int sb_lstm_gen_supported(int C, int H) {
  return (H == 64 || H == 128) && (C == 16 || C == 32 || C == 64) ? 1 : 0;
}

extern "C" int sb_lstm_gen_fwd(const sb_lstm_gen_fwd_args* ap, void* stream) {
  if (!ap || ap->nseq <= 0 || ap->nsteps <= 0 || (ap->ndir != 1 && ap->ndir != 2)) return -1001;
  const sb_lstm_gen_fwd_args& a = *ap;
  if (!sb_lstm_gen_supported(a.C, a.H)) return -1002;
  if (!a.x || !a.ln_g || !a.ln_b || !a.hs) return -1001;
  for (int d = 0; d < a.ndir; ++d) if (!a.w_ih[d] || !a.w_hh[d] || !a.b_ih[d] || !a.b_hh[d]) return -1001;
  if (a.save_gates && !a.save_u) return -1003;
  hipStream_t st = (hipStream_t)stream;
#define SB_G(C_, H_) if (a.C == C_ && a.H == H_) launch_gen_fwd<C_, H_>(a, st); else
  SB_G(16, 64) SB_G(32, 64) SB_G(64, 64) SB_G(16, 128) SB_G(32, 128) SB_G(64, 128) return -1002;
#undef SB_G
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_lstm_gen_bwd_rec(const sb_lstm_gen_bwd_args* ap, void* stream) {
  if (!ap || ap->nseq <= 0 || ap->nsteps <= 0 || (ap->ndir != 1 && ap->ndir != 2)) return -1001;
  const sb_lstm_gen_bwd_args& a = *ap;
  if (a.H != 64 && a.H != 128) return -1002;
  if (!a.save_gates || !a.dhs || !a.dgates) return -1001;
  for (int d = 0; d < a.ndir; ++d) if (!a.w_hh[d]) return -1001;
  hipStream_t st = (hipStream_t)stream;
  if (a.H == 128) launch_gen_bwd<128>(a, st); else launch_gen_bwd<64>(a, st);
  SB_CHECK_LAUNCH();
  return 0;
}
