// Position-wise MFMA GEMMs for gfx950: every Linear / k=s Conv1d / 3x3 Conv2d /
// STFT filter bank of the Sound-Bubble hot path is "out[p, :] = W * in(p, :)"
// over the dense (b, t, f) position grid, and every weight gradient is the
// matching "dW = sum_p g(p,:)^T in(p,:)".
//
// sb_linear_fwd : weights (<= 152 KB) staged ONCE per workgroup in LDS as the A
//   operand; each wave streams 16 positions at a time, fetching its B operand
//   (the activations) straight from HBM as 16-byte loads (64 B contiguous per
//   position per K-chunk), and finishes with a fused epilogue (bias, residual,
//   PReLU, LayerNorm forward, LayerNorm backward).  Persistent grid.
// sb_wgrad      : TN GEMM over positions with per-workgroup register
//   accumulators (no atomics; deterministic two-stage reduction).
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

constexpr int LIN_MAX_WG = 1024;   // persistent grid cap (4 WG/CU on 256 CUs)
constexpr int SB_LINEAR_SLICE_MAX_P = 256;   // at most this many positions: column-sliced launch (see sb_linear_fwd)
constexpr int WG_MAX_WG = 512;

// Position p = (b*T + t)*F + f  ->  element offset b*sb + t*st + f*sf.  Two 64-bit integer divisions per position and
// lane cost several hundred VALU instructions and dominated the narrow kernels of this file, so: positions are 32-bit
// (checked by the launchers), a dense operand (st == F*sf, sb == T*st) needs no division at all (offset = p*sf), one
// (b, t, f) split serves all operands of a position, and the weight-gradient kernels split once per 16-position tile
// and walk the lanes' positions from there by carry.
struct Pos3 { unsigned b, t, f; };
SB_DEVINL Pos3 split_pos(unsigned p, unsigned T, unsigned F) {
  const unsigned tf = T * F;
  Pos3 r;
  r.b = p / tf;
  const unsigned rem = p - r.b * tf;
  r.t = rem / F;
  r.f = rem - r.t * F;
  return r;
}
SB_DEVINL bool dense_strides(int T, int F, int64_t sb, int64_t st, int64_t sf) {
  return st == (int64_t)F * sf && sb == (int64_t)T * st;
}
SB_DEVINL int64_t off3(const Pos3& x, int64_t sb, int64_t st, int64_t sf) {
  return (int64_t)x.b * sb + (int64_t)x.t * st + (int64_t)x.f * sf;
}
// offset of position (base + delta), 0 <= delta < 16, from the split of `base`
SB_DEVINL int64_t off3_delta(const Pos3& base, unsigned delta, unsigned T, unsigned F, int64_t sb, int64_t st, int64_t sf) {
  unsigned f = base.f + delta, t = base.t, b = base.b;
  while (f >= F) { f -= F; ++t; }
  while (t >= T) { t -= T; ++b; }
  return (int64_t)b * sb + (int64_t)t * st + (int64_t)f * sf;
}

template <int NT, int EPI>
__global__ __launch_bounds__(256) void linear_kernel(sb_linear_args a, int64_t P) {
  extern __shared__ __attribute__((aligned(16))) float Wl[];   // [N][K+4]
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int N = NT * 16, K = a.K, KP = K + 4;
  if (gridDim.y > 1) {
    // column slices (sb_linear_fwd: few positions or N > 128): workgroup row y owns output columns [y N, (y + 1) N) -- its
    // own rows of W in LDS -- so that a call with a handful of positions spreads the weight read over many CUs
    const int y = blockIdx.y;
    a.w += (size_t)y * N * K;
    if (a.bias) a.bias += y * N;
    a.out += y * N;
    if (a.res) a.res += y * N;
    a.n_valid = min(max(a.n_valid - y * N, 0), N);
  }
  // stage weights
  for (int idx = tid * 4; idx < N * K; idx += 256 * 4) {
    const int n = idx / K, k = idx - n * K;
    st4(&Wl[n * KP + k], ld4(a.w + idx));
  }
  __syncthreads();

  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = a.bias ? ld4(a.bias + 16 * nt + 4 * q) : zero4();

  // LN-backward parameter-gradient accumulators (this lane's 4*NT features)
  constexpr int NB = EPI == SB_EPI_LNBWD ? NT : 1;
  f32x4 dgam[NB], dbet[NB];
  float dalpha = 0.f;
#pragma unroll
  for (int nt = 0; nt < NB; ++nt) { dgam[nt] = zero4(); dbet[nt] = zero4(); }
  const float alpha = a.prelu_a ? a.prelu_a[0] : 0.f;

  float amax = 0.f;                                  // max |stored value| (a.absmax_out)
  const int64_t ntiles = (P + 63) / 64;
  const int nchunk = K / 16;
  const bool in_dense = dense_strides(a.T, a.F, a.is_b, a.is_t, a.is_f);
  const bool out_dense = dense_strides(a.T, a.F, a.os_b, a.os_t, a.os_f);
  const bool res_dense = !a.res || dense_strides(a.T, a.F, a.rs_b, a.rs_t, a.rs_f);
  const bool all_dense = in_dense && out_dense && res_dense;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t p_raw = tile * 64 + 16 * w + j;
    const bool valid = p_raw < P;
    // out-of-range lanes recompute the last position (columns of the MFMA are independent) and skip the store:
    // no predicated loads, so the K loop stays one basic block
    const int64_t p = valid ? p_raw : P - 1;
    Pos3 ps = {0, 0, 0};
    if (!all_dense) ps = split_pos((unsigned)p, a.T, a.F);
    const int64_t ioff = in_dense ? p * a.is_f : off3(ps, a.is_b, a.is_t, a.is_f);
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = bias[nt];

    int kin_n = 0;                                   // K chunks are walked in order: segment / offset by carry
    int64_t seg_off = 0;
    auto load_next = [&]() -> f32x4 {
      const f32x4 v = ld4(a.in + ioff + seg_off + kin_n + 4 * q);
      kin_n += 16;
      if (kin_n >= a.kseg) { kin_n = 0; seg_off += a.is_seg; }
      return v;
    };
    f32x4 bcur = load_next();
    for (int m = 0; m < nchunk; ++m) {
      const f32x4 b4 = bcur;
      if (m + 1 < nchunk) bcur = load_next();
      const float* wrow = &Wl[j * KP + 16 * m + 4 * q];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 a4 = ld4(wrow + nt * 16 * KP);
        acc[nt] = mfma16x4(a4, b4, acc[nt]);
      }
    }

    // ---------------- epilogue: lane holds features 16nt+4q..+3 of position p ----------------
    const int64_t ooff = out_dense ? p * a.os_f : off3(ps, a.os_b, a.os_t, a.os_f);
    if constexpr (EPI == SB_EPI_RES) {
      {
        const int64_t roff = res_dense ? p * a.rs_f : off3(ps, a.rs_b, a.rs_t, a.rs_f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] += ld4(a.res + roff + 16 * nt + 4 * q);
      }
    } else if constexpr (EPI == SB_EPI_PRELU) {
      if (valid && a.aux_out)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) st4(a.aux_out + p * N + 16 * nt + 4 * q, acc[nt]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nt][r] = acc[nt][r] > 0.f ? acc[nt][r] : alpha * acc[nt][r];
    } else if constexpr (EPI == SB_EPI_LN) {
      if (valid && a.aux_out)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) st4(a.aux_out + p * N + 16 * nt + 4 * q, acc[nt]);
      float s = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) s += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
      const float mean = quad_sum(s) * (1.0f / N);
      float sq = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = acc[nt][r] - mean; sq += d * d; }
      const float rstd = 1.0f / sqrtf(quad_sum(sq) * (1.0f / N) + 1e-5f);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 g4 = ld4(a.ln_g + 16 * nt + 4 * q), b4 = ld4(a.ln_b + 16 * nt + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nt][r] = (acc[nt][r] - mean) * rstd * g4[r] + b4[r];
      }
    } else if constexpr (EPI == SB_EPI_LNBWD) {
      // acc = dL/d(LN output).  x = pre-LN input (optionally PReLU of the stored pre-activation).
      f32x4 raw[NT], xh[NT];
      float s = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        raw[nt] = ld4(a.aux_in + p * N + 16 * nt + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = (a.prelu_a && raw[nt][r] <= 0.f) ? alpha * raw[nt][r] : raw[nt][r];
          xh[nt][r] = x;
          s += x;
        }
      }
      const float mean = quad_sum(s) * (1.0f / N);
      float sq = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = xh[nt][r] - mean; sq += d * d; }
      const float rstd = 1.0f / sqrtf(quad_sum(sq) * (1.0f / N) + 1e-5f);
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 g4 = ld4(a.ln_g + 16 * nt + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xhat = (xh[nt][r] - mean) * rstd;
          xh[nt][r] = xhat;
          const float du = valid ? acc[nt][r] : 0.f;
          dgam[nt][r] += du * xhat;
          dbet[nt][r] += du;
          const float gg = du * g4[r];
          acc[nt][r] = gg;
          m1 += gg;
          m2 += gg * xhat;
        }
      }
      m1 = quad_sum(m1) * (1.0f / N);
      m2 = quad_sum(m2) * (1.0f / N);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float dx = rstd * (acc[nt][r] - m1 - xh[nt][r] * m2);
          if (a.prelu_a) {
            if (raw[nt][r] <= 0.f) { dalpha += valid ? dx * raw[nt][r] : 0.f; dx *= alpha; }
          }
          acc[nt][r] = dx;
        }
      if (valid && a.res) {
        const int64_t roff = res_dense ? p * a.rs_f : off3(ps, a.rs_b, a.rs_t, a.rs_f);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] += ld4(a.res + roff + 16 * nt + 4 * q);
      }
    }
    if (valid) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n0 = 16 * nt + 4 * q;
        float* o = a.out + ooff + n0;
        if (n0 + 4 <= a.n_valid) {
          if (a.accumulate) { acc[nt] += ld4(o); st4(o, acc[nt]); } else st4(o, acc[nt]);
          if constexpr (EPI == SB_EPI_NONE || EPI == SB_EPI_RES)
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(acc[nt][0]), fabsf(acc[nt][1]))), fmaxf(fabsf(acc[nt][2]), fabsf(acc[nt][3])));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < a.n_valid) {
              const float v = a.accumulate ? o[r] + acc[nt][r] : acc[nt][r];
              o[r] = v;
              amax = fmaxf(amax, fabsf(v));
            }
        }
      }
    }
  }

  if constexpr (EPI == SB_EPI_NONE || EPI == SB_EPI_RES) {
    if (a.absmax_out) {                                // one atomic per workgroup (same-address atomics serialise in L2)
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      __shared__ float wm[4];
      if (lane == 0) wm[w] = amax;
      __syncthreads();
      if (tid == 0) atomicMax(reinterpret_cast<unsigned*>(a.absmax_out),
                              __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
    }
  }
  if constexpr (EPI == SB_EPI_LNBWD) if (a.partials) {
    float* part = a.partials + (size_t)blockIdx.x * (2 * N + 1);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float g = row16_sum(dgam[nt][r]), b = row16_sum(dbet[nt][r]);
        if (j == 0) {
          atomicAdd(part + 16 * nt + 4 * q + r, g);
          atomicAdd(part + N + 16 * nt + 4 * q + r, b);
        }
      }
    const float da = wave_sum(dalpha);
    if (lane == 0 && a.prelu_a) atomicAdd(part + 2 * N, da);
  }
}

// ---------------------------------------------------------------------------------------------
// fp16x3 form of the position-wise GEMM for the long-K, narrow-N layers (the 3x3 front-end convolution K = 9*32, the
// output transposed convolution K = 9*C; sb_linear_args.mma == 1).  On the fp32-input MFMA these ran at 36-46 % matrix
// pipe busy and 0.4 of the fp32 peak (profiles/r02_pmc_sq_*.json); here every fp32 operand is split into fp16 hi + lo
// (11 + 11 bits) and the product evaluated as lo*hi + hi*lo + hi*hi on v_mfma_f32_16x16x32_f16 -- the same fp32-class
// arithmetic as the recurrent kernels (dropped term <= 2^-22), one sixth of the matrix-pipe time.  Weights are
// register-resident (A operand, split once); the activations of a whole 16-position tile are fetched up front
// (KC * 32 bytes per lane in flight) and split as they arrive.
typedef _Float16 lh16x8 __attribute__((ext_vector_type(8)));
struct LSplitH { lh16x8 hi, lo; };
SB_DEVINL LSplitH lsplit8(const f32x4 lo4, const f32x4 hi4) {
  LSplitH s;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const _Float16 h0 = (_Float16)lo4[k], h1 = (_Float16)hi4[k];
    s.hi[k] = h0; s.hi[4 + k] = h1;
    s.lo[k] = (_Float16)(lo4[k] - (float)h0); s.lo[4 + k] = (_Float16)(hi4[k] - (float)h1);
  }
  return s;
}

template <int NT, int KC, int EPI>
__global__ __launch_bounds__(256) void linear16_kernel(sb_linear_args a, int64_t P) {
  static_assert(EPI == SB_EPI_NONE || EPI == SB_EPI_RES || EPI == SB_EPI_LN, "fp16x3 form: bias / residual / LayerNorm");
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  constexpr int N = NT * 16;
  const int K = a.K;
  // this lane's K offsets: chunk m covers k = 32 m + 8 q .. + 7 (kseg is a multiple of 8, so the 8 values share a segment)
  int64_t koff[KC];
  bool kval[KC];
#pragma unroll
  for (int m = 0; m < KC; ++m) {
    const int k0 = 32 * m + 8 * q;
    kval[m] = k0 < K;
    const int seg = kval[m] ? k0 / a.kseg : 0;
    koff[m] = kval[m] ? (int64_t)seg * a.is_seg + (k0 - seg * a.kseg) : 0;
  }
  // A[i = feature 16 nt + j][k = 32 m + 8 q + kk], split once.  Register-resident while it fits (NT * KC <= 9: 72 VGPRs);
  // beyond that (C = 32, K = 288: 144 VGPRs, one wave per SIMD) it lives in LDS in lane order -- every lane re-reads its
  // own 16 bytes per term and chunk, conflict-free.
  constexpr bool ALDS = NT * KC > 9;
  __shared__ __attribute__((aligned(16))) lh16x8 Al[ALDS ? 2 * NT * KC * 64 : 1];
  LSplitH A[ALDS ? 1 : NT][ALDS ? 1 : KC];
  if constexpr (ALDS) {
    for (int e = tid; e < NT * KC * 64; e += 256) {
      const int ln = e & 63, mm = (e >> 6) % KC, nt = (e >> 6) / KC;
      const int k0 = 32 * mm + 8 * (ln >> 4);
      const float* wr = a.w + (size_t)(16 * nt + (ln & 15)) * K + k0;
      const f32x4 z = zero4();
      const LSplitH t = lsplit8(k0 < K ? ld4(wr) : z, k0 < K ? ld4(wr + 4) : z);
      Al[(nt * KC + mm) * 64 + ln] = t.hi;
      Al[((NT + nt) * KC + mm) * 64 + ln] = t.lo;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < KC; ++m) {
        const float* wr = a.w + (size_t)(16 * nt + j) * K + 32 * m + 8 * q;
        const f32x4 z = zero4();
        A[nt][m] = lsplit8(kval[m] ? ld4(wr) : z, kval[m] ? ld4(wr + 4) : z);
      }
  }
  f32x4 bias[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias[nt] = a.bias ? ld4(a.bias + 16 * nt + 4 * q) : zero4();

  float amax = 0.f;
  const int64_t ntiles = (P + 63) / 64;
  const bool in_dense = dense_strides(a.T, a.F, a.is_b, a.is_t, a.is_f);
  const bool out_dense = dense_strides(a.T, a.F, a.os_b, a.os_t, a.os_f);
  const bool res_dense = !a.res || dense_strides(a.T, a.F, a.rs_b, a.rs_t, a.rs_f);
  const bool all_dense = in_dense && out_dense && res_dense;
  // XCD-aware tile order: a 3x3 window re-reads the rows of its neighbours in t, and workgroup i runs on XCD i % 8, each
  // with its own L2 -- so every XCD walks a CONTIGUOUS eighth of the position range (its workgroups on adjacent tiles)
  // instead of every eighth tile (which made all eight L2s fetch every row: 3x the compulsory reads in the PMC counters).
  const int xcd = blockIdx.x & 7, xl = blockIdx.x >> 3, nxl = (gridDim.x + 7 - xcd) >> 3;
  const int64_t tpx = (ntiles + 7) / 8;
  for (int64_t tl = xl; tl < tpx; tl += nxl) {
    const int64_t tile = (int64_t)xcd * tpx + tl;
    if (tile >= ntiles) break;
    const int64_t p_raw = tile * 64 + 16 * w + j;
    const bool valid = p_raw < P;
    const int64_t p = valid ? p_raw : P - 1;          // out-of-range lanes recompute the last position and skip the store
    Pos3 ps = {0, 0, 0};
    if (!all_dense) ps = split_pos((unsigned)p, a.T, a.F);
    const float* src = a.in + (in_dense ? p * a.is_f : off3(ps, a.is_b, a.is_t, a.is_f));
    f32x4 b0[KC], b1[KC];
#pragma unroll
    for (int m = 0; m < KC; ++m) { b0[m] = ld4(src + koff[m]); b1[m] = ld4(src + koff[m] + 4); }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = bias[nt];
    int alane = lane;
    if constexpr (ALDS) asm volatile("" : "+v"(alane));      // keeps the (tile-invariant) LDS reads inside the tile loop
#pragma unroll
    for (int m = 0; m < KC; ++m) {
      const f32x4 z = zero4();
      const LSplitH B = lsplit8(kval[m] ? b0[m] : z, kval[m] ? b1[m] : z);
      lh16x8 ah[NT], al[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (ALDS) { ah[nt] = Al[(nt * KC + m) * 64 + alane]; al[nt] = Al[((NT + nt) * KC + m) * 64 + alane]; }
        else { ah[nt] = A[nt][m].hi; al[nt] = A[nt][m].lo; }
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[nt], B.hi, acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt], B.lo, acc[nt], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt], B.hi, acc[nt], 0, 0, 0);
    }
    // ---------------- epilogue: lane holds features 16nt+4q..+3 of position p (as in linear_kernel) ----------------
    const int64_t ooff = out_dense ? p * a.os_f : off3(ps, a.os_b, a.os_t, a.os_f);
    if constexpr (EPI == SB_EPI_RES) {
      const int64_t roff = res_dense ? p * a.rs_f : off3(ps, a.rs_b, a.rs_t, a.rs_f);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] += ld4(a.res + roff + 16 * nt + 4 * q);
    } else if constexpr (EPI == SB_EPI_LN) {
      if (valid && a.aux_out)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) st4(a.aux_out + p * N + 16 * nt + 4 * q, acc[nt]);
      float sm = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) sm += acc[nt][0] + acc[nt][1] + acc[nt][2] + acc[nt][3];
      const float mean = quad_sum(sm) * (1.0f / N);
      float sq = 0.f;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = acc[nt][r] - mean; sq += d * d; }
      const float rstd = 1.0f / sqrtf(quad_sum(sq) * (1.0f / N) + 1e-5f);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 g4 = ld4(a.ln_g + 16 * nt + 4 * q), b4 = ld4(a.ln_b + 16 * nt + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[nt][r] = (acc[nt][r] - mean) * rstd * g4[r] + b4[r];
      }
    }
    if (valid) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n0 = 16 * nt + 4 * q;
        float* o = a.out + ooff + n0;
        if (n0 + 4 <= a.n_valid) {
          if (a.accumulate) { acc[nt] += ld4(o); st4(o, acc[nt]); } else st4(o, acc[nt]);
          if constexpr (EPI != SB_EPI_LN)
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(acc[nt][0]), fabsf(acc[nt][1]))), fmaxf(fabsf(acc[nt][2]), fabsf(acc[nt][3])));
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n0 + r < a.n_valid) {
              const float v = a.accumulate ? o[r] + acc[nt][r] : acc[nt][r];
              o[r] = v;
              amax = fmaxf(amax, fabsf(v));
            }
        }
      }
    }
  }
  if constexpr (EPI != SB_EPI_LN) {
    if (a.absmax_out) {
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      __shared__ float wm[4];
      if (lane == 0) wm[w] = amax;
      __syncthreads();
      if (tid == 0) atomicMax(reinterpret_cast<unsigned*>(a.absmax_out),
                              __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dW[n, k] = sum_p g(p, n) * [in1(p, :K1) | in2(p, :K2)][k]  (+ column sums of g = bias gradients).
// Used for the narrow layers (N <= 80: Linear / Conv1d / ConvTranspose1d / 3x3 convs); the 256-row LSTM
// gradients have their own kernel (sb_lstm_stream.hip).  Every wave holds ALL n-blocks and walks its own
// position tiles (tile = 4*block + wave), so all four SIMDs work even when N = 16; each wave emits one
// partial row and a second, wide kernel reduces the rows.  Source 2 is a dense row matrix addressed
// p*ld2 + shift2 with the per-segment first/last-row exclusion.
// IN16: source 1 holds fp16 elements (the fp16 hs written by the forward recurrence for the backward kernels)
template <int NTW, int KT1, int KT2, bool IN16 = false>
__global__ __launch_bounds__(256) void wgrad_kernel(sb_wgrad_args a, int64_t P) {
  constexpr int KT = KT1 + KT2;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  f32x4 acc[NTW][KT];
  float csum[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    csum[nt] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = zero4();
  }
  int64_t koff[KT1 > 0 ? KT1 : 1];
  bool kval[KT1 > 0 ? KT1 : 1];
#pragma unroll
  for (int kt = 0; kt < KT1; ++kt) {
    const int k = 16 * kt + j;
    const int seg = k / a.kseg;
    koff[kt] = (int64_t)seg * a.is_seg + (k - seg * a.kseg);
    kval[kt] = k < a.K;
  }
  int ncol[NTW];
  bool nval[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) { ncol[nt] = 16 * nt + j; nval[nt] = ncol[nt] < a.N; }

  const int64_t ntiles = (P + 15) / 16;
  const bool in_dense = dense_strides(a.T, a.F, a.is_b, a.is_t, a.is_f);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    float av[NTW][4], bv[KT][4];
    Pos3 base = {0, 0, 0};
    if (!in_dense) base = split_pos((unsigned)(tile * 16), a.T, a.F);      // wave-uniform: once per tile
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t p_raw = tile * 16 + 4 * q + r;
      const bool ok = p_raw < P;
      const int64_t p = ok ? p_raw : P - 1;                  // clamped address + select: no predicated loads
      const int64_t ioff = in_dense ? p * a.is_f
                                    : off3_delta(base, (unsigned)(p - tile * 16), a.T, a.F, a.is_b, a.is_t, a.is_f);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const float v = a.g[p * a.ldg + (nval[nt] ? ncol[nt] : 0)];
        av[nt][r] = (ok & nval[nt]) ? v : 0.f;
      }
#pragma unroll
      for (int kt = 0; kt < KT1; ++kt) {
        float v;
        if constexpr (IN16) v = (float)reinterpret_cast<const _Float16*>(a.in)[ioff + (kval[kt] ? koff[kt] : 0)];
        else v = a.in[ioff + (kval[kt] ? koff[kt] : 0)];
        bv[kt][r] = (ok & kval[kt]) ? v : 0.f;
      }
      if constexpr (KT2 > 0) {
        const int idx = (int)((unsigned)p % (unsigned)a.seg_len);
        const bool ok2 = ok & (idx >= a.skip_first) & (idx < a.seg_len - a.skip_last);
#pragma unroll
        for (int kt = 0; kt < KT2; ++kt) {
          const float v = a.in2[p * a.ld2 + (ok2 ? a.shift2 : 0) + 16 * kt + j];
          bv[KT1 + kt][r] = ok2 ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        csum[nt] += av[nt][r];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = mfma16(av[nt][r], bv[kt][r], acc[nt][kt]);
      }
  }
  const int Ktot = a.K + a.K2;
  float* part = a.scratch + ((size_t)blockIdx.x * 4 + w) * ((size_t)a.N * Ktot + a.N);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * nt + 4 * q + r;
        const int k = kt < KT1 ? 16 * kt + j : a.K + 16 * (kt - KT1) + j;
        const bool kok = kt < KT1 ? (16 * kt + j < a.K) : true;
        if (n < a.N && kok) part[(size_t)n * Ktot + k] = acc[nt][kt][r];
      }
    const float cs = quad_sum(csum[nt]);
    if (q == 0 && nval[nt]) part[(size_t)a.N * Ktot + ncol[nt]] = cs;
  }
}

// Wide-load variant (single source, whole 16-column tiles): the tiles of a K segment (and of N) are grouped 4 / 2 / 1 at
// a time and lane j loads the G consecutive columns 16*gs + G*j .. of a group with ONE 16 / 8 / 4-byte load, which
// serves tile gs + e as its column for e = 0 .. G-1.  A tile is then the column set {16*gs + G*j + e}, a permutation
// that only shows up in the index the partial sums are written to.  3-4x fewer load instructions than one element
// per load: these narrow GEMMs are bound by the latency and count of their loads, not by bytes or MFMA time.
struct TileGroup { int gs, G, e; };
SB_DEVINL constexpr TileGroup tile_group(int t, int TS) {      // greedy groups of 4, 2, 1 over the TS tiles of a segment
  int s = 0;
  while (TS - s >= 4) { if (t < s + 4) return {s, 4, t - s}; s += 4; }
  if (TS - s >= 2) { if (t < s + 2) return {s, 2, t - s}; s += 2; }
  return {s, 1, t - s};
}

template <bool IN16, int G>
SB_DEVINL void load_group(const float* base, int64_t off, float (&out)[4]) {
  if constexpr (IN16) {
    const _Float16* hp = reinterpret_cast<const _Float16*>(base) + off;
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    if constexpr (G == 4) { const h4 v = *reinterpret_cast<const h4*>(hp); for (int e = 0; e < 4; ++e) out[e] = (float)v[e]; }
    else if constexpr (G == 2) { const h2 v = *reinterpret_cast<const h2*>(hp); out[0] = (float)v[0]; out[1] = (float)v[1]; }
    else out[0] = (float)hp[0];
  } else {
    if constexpr (G == 4) { const f32x4 v = ld4(base + off); for (int e = 0; e < 4; ++e) out[e] = v[e]; }
    else if constexpr (G == 2) { const float2 v = *reinterpret_cast<const float2*>(base + off); out[0] = v.x; out[1] = v.y; }
    else out[0] = base[off];
  }
}

// U position tiles are fetched per loop trip before any of them is multiplied: one 16-position tile is a few KB, far
// less than a wave must keep in flight to cover the memory latency at two waves per SIMD.
template <int NTW, int KT1, int TS, bool IN16>
__global__ __launch_bounds__(256) void wgrad_wide_kernel(sb_wgrad_args a, int64_t P) {
  constexpr int KT = KT1, NSEG = KT1 / TS;
  constexpr int U = NTW * KT1 <= 8 ? 4 : 1;
  static_assert(KT1 % TS == 0, "whole segments");
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  f32x4 acc[NTW][KT];
  float csum[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    csum[nt] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = zero4();
  }
  const int64_t ntiles = (P + 15) / 16, tstride = (int64_t)gridDim.x * 4;
  const bool in_dense = dense_strides(a.T, a.F, a.is_b, a.is_t, a.is_f);
  for (int64_t tile0 = (int64_t)blockIdx.x * 4 + w; tile0 < ntiles; tile0 += tstride * U) {
    float av[U][NTW][4], bv[U][KT][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int64_t tile_u = tile0 + u * tstride, tile_c = tile_u < ntiles ? tile_u : ntiles - 1;
    Pos3 base = {0, 0, 0};
    if (!in_dense) base = split_pos((unsigned)(tile_c * 16), a.T, a.F);    // wave-uniform: once per tile
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t p_raw = tile_u * 16 + 4 * q + r;
      const bool ok = p_raw < P;
      const int64_t p = ok ? p_raw : P - 1;                  // clamped address + select: no predicated loads
      const int64_t ioff = in_dense ? p * a.is_f
                                    : off3_delta(base, (unsigned)(p - tile_c * 16), a.T, a.F, a.is_b, a.is_t, a.is_f);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const TileGroup tg = tile_group(nt, NTW);
        if (tg.e == 0) {
          float v[4];
          if (tg.G == 4) load_group<false, 4>(a.g, p * a.ldg + 16 * tg.gs + 4 * j, v);
          else if (tg.G == 2) load_group<false, 2>(a.g, p * a.ldg + 16 * tg.gs + 2 * j, v);
          else load_group<false, 1>(a.g, p * a.ldg + 16 * tg.gs + j, v);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < tg.G) av[u][tg.gs + e][r] = ok ? v[e] : 0.f;
        }
      }
#pragma unroll
      for (int sg = 0; sg < NSEG; ++sg)
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const TileGroup tg = tile_group(t, TS);
          if (tg.e == 0) {
            float v[4];
            const int64_t o = ioff + (int64_t)sg * a.is_seg + 16 * tg.gs;
            if (tg.G == 4) load_group<IN16, 4>(a.in, o + 4 * j, v);
            else if (tg.G == 2) load_group<IN16, 2>(a.in, o + 2 * j, v);
            else load_group<IN16, 1>(a.in, o + j, v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < tg.G) bv[u][sg * TS + tg.gs + e][r] = ok ? v[e] : 0.f;
          }
        }
    }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        csum[nt] += av[u][nt][r];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = mfma16(av[u][nt][r], bv[u][kt][r], acc[nt][kt]);
      }
  }
  const int Ktot = a.K;
  float* part = a.scratch + ((size_t)blockIdx.x * 4 + w) * ((size_t)a.N * Ktot + a.N);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const TileGroup ng = tile_group(nt, NTW);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const TileGroup kg = tile_group(kt % TS, TS);
      const int k = (kt / TS) * a.kseg + 16 * kg.gs + kg.G * j + kg.e;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * ng.gs + ng.G * (4 * q + r) + ng.e;
        part[(size_t)n * Ktot + k] = acc[nt][kt][r];
      }
    }
    const float cs = quad_sum(csum[nt]);
    if (q == 0) part[(size_t)a.N * Ktot + 16 * ng.gs + ng.G * j + ng.e] = cs;
  }
}

// fp16x3 form of the wide-load weight-gradient kernel (sb_wgrad_args.mma == 1): 32 positions per MFMA
// (v_mfma_f32_16x16x32_f16), both operands split into fp16 hi + lo, products lo*hi + hi*lo + hi*hi (fp32-class, dropped
// term <= 2^-22).  (One fp16 term for the activations was measured first: 2e-4 on random data but 2e-3 on the real
// front-end convolution gradient at full size -- a weight gradient is a heavily cancelling sum, which amplifies the
// 2^-12 per-product rounding.)  On the fp32-input MFMA the K = 288 convolutions' weight gradients ran at 42-52 %
// matrix-pipe busy (issue-stalled on the 8-pass fp32 MFMA); this form issues 3 x 4-pass MFMAs per 32 positions and tile
// instead of 8 x 8-pass ones.
template <int NTW, int NSEG, int TS>
SB_DEVINL void wgrad16_body(const sb_wgrad_args& a, int64_t P) {
  constexpr int KT = NSEG * TS;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  f32x4 acc[NTW][KT];
  float csum[NTW];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    csum[nt] = 0.f;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = zero4();
  }
  const int64_t ntiles = (P + 31) / 32, tstride = (int64_t)gridDim.x * 4;
  const bool in_dense = dense_strides(a.T, a.F, a.is_b, a.is_t, a.is_f);
  // Gradients are tiny (1e-6 .. 1e-9 here) and fp16 underflows below 6e-8: g is scaled by S = 2^-ceil(log2 max|g|) before
  // the split (exact) and the sums by 1/S at the end -- the scaled-fp16 device of the LSTM backward (sb_lstm_bwd_args.gmax).
  float gS = 1.0f;
  if (a.gmax) { const float m = a.gmax[0]; if (m > 0.f && m < 3.0e38f) gS = exp2f(-ceilf(log2f(m))); }
  const float invS = 1.0f / gS;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += tstride) {
    Pos3 base = {0, 0, 0};
    if (!in_dense) base = split_pos((unsigned)(tile * 32), a.T, a.F);          // wave-uniform: once per tile
    int64_t ioff[8];
    bool ok[8];
    float gv[NTW][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int64_t p_raw = tile * 32 + 8 * q + r;
      ok[r] = p_raw < P;
      const int64_t p = ok[r] ? p_raw : P - 1;                                 // clamped address + select
      ioff[r] = in_dense ? p * a.is_f : off3_delta(base, (unsigned)(p - tile * 32), a.T, a.F, a.is_b, a.is_t, a.is_f);
#pragma unroll
      for (int nt = 0; nt < NTW; ++nt) {
        const TileGroup tg = tile_group(nt, NTW);
        if (tg.e == 0) {
          float v[4];
          if (tg.G == 4) load_group<false, 4>(a.g, p * a.ldg + 16 * tg.gs + 4 * j, v);
          else if (tg.G == 2) load_group<false, 2>(a.g, p * a.ldg + 16 * tg.gs + 2 * j, v);
          else load_group<false, 1>(a.g, p * a.ldg + (16 * tg.gs + j < a.N ? 16 * tg.gs + j : 0), v);   // N < 16: masked
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (e < tg.G) gv[tg.gs + e][r] = (ok[r] && 16 * tg.gs + tg.G * j + e < a.N) ? v[e] * gS : 0.f;
        }
      }
    }
    lh16x8 ghi[NTW], glo[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        csum[nt] += gv[nt][r];
        const _Float16 h = (_Float16)gv[nt][r];
        ghi[nt][r] = h;
        glo[nt][r] = (_Float16)(gv[nt][r] - (float)h);
      }
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) {
      lh16x8 bh[TS], bl[TS];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const TileGroup tg = tile_group(t, TS);
          if (tg.e == 0) {
            float v[4];
            const int64_t o = ioff[r] + (int64_t)sg * a.is_seg + 16 * tg.gs;
            if (tg.G == 4) load_group<false, 4>(a.in, o + 4 * j, v);
            else if (tg.G == 2) load_group<false, 2>(a.in, o + 2 * j, v);
            else load_group<false, 1>(a.in, o + j, v);
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (e < tg.G) {
                const float x = ok[r] ? v[e] : 0.f;
                const _Float16 h = (_Float16)x;
                bh[tg.gs + e][r] = h;
                bl[tg.gs + e][r] = (_Float16)(x - (float)h);
              }
          }
        }
#pragma unroll
      for (int t = 0; t < TS; ++t)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
          acc[nt][sg * TS + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(glo[nt], bh[t], acc[nt][sg * TS + t], 0, 0, 0);
          acc[nt][sg * TS + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ghi[nt], bl[t], acc[nt][sg * TS + t], 0, 0, 0);
          acc[nt][sg * TS + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ghi[nt], bh[t], acc[nt][sg * TS + t], 0, 0, 0);
        }
    }
  }
  const int Ktot = a.K;
  float* part = a.scratch + ((size_t)blockIdx.x * 4 + w) * ((size_t)a.N * Ktot + a.N);
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const TileGroup ng = tile_group(nt, NTW);
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
      const TileGroup kg = tile_group(kt % TS, TS);
      const int k = (kt / TS) * a.kseg + 16 * kg.gs + kg.G * j + kg.e;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = 16 * ng.gs + ng.G * (4 * q + r) + ng.e;
        if (n < a.N) part[(size_t)n * Ktot + k] = acc[nt][kt][r] * invS;
      }
    }
    const float cs = quad_sum(csum[nt]);
    const int nb = 16 * ng.gs + ng.G * j + ng.e;
    if (q == 0 && nb < a.N) part[(size_t)a.N * Ktot + nb] = cs * invS;
  }
}

template <int NTW, int NSEG, int TS>
__global__ __launch_bounds__(256) void wgrad16_kernel(sb_wgrad_args a, int64_t P) { wgrad16_body<NTW, NSEG, TS>(a, P); }
// N = 32 (144 accumulator registers): capped at 256 registers so that two waves share a SIMD (uncapped it took 292 and
// ran alone: 645 us against 313 us for the fp32 kernel it replaces)
template <int NSEG, int TS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad16_n32_kernel(sb_wgrad_args a, int64_t P) {
  wgrad16_body<2, NSEG, TS>(a, P);
}

// Generic form of the weight gradient (any N, any K / K2 that are multiples of 16; fp32 sources): the fallback of sb_wgrad for
// the shapes no register-accumulator instantiation above covers -- the 512-row LSTM gradients and the 256 / 320-column
// Conv1d / ConvTranspose1d gradients of the reference constructor's default widths (D = 64, H = 128).  A workgroup owns a
// 64 x 64 tile of dW (blockIdx.y: rows n, blockIdx.z: columns k over [K | K2]) and a contiguous range of positions
// (blockIdx.x): 32 positions at a time are staged TRANSPOSED in LDS ([column][position], padded), so that both MFMA operands
// are one ds_read_b128 per lane and 16-position chunk; wave w owns rows 16w .. 16w + 15 of the tile.  One partial row per
// position range, reduced by wgrad_reduce_kernel like the rows of the kernels above (same scratch layout).
constexpr int WGEN_POS = 32, WGEN_LD = WGEN_POS + 4;
__global__ __launch_bounds__(256) void wgrad_gen_kernel(sb_wgrad_args a, int64_t P, int64_t per) {
  __shared__ __attribute__((aligned(16))) float GT[64][WGEN_LD];
  __shared__ __attribute__((aligned(16))) float IT[64][WGEN_LD];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int Ktot = a.K + a.K2;
  const int n0 = blockIdx.y * 64, k0 = blockIdx.z * 64;
  const int64_t p_begin = (int64_t)blockIdx.x * per, p_end = min(P, p_begin + per);
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()};
  float csum = 0.f;
  // staging role: thread -> column c = tid & 63, positions 4 * (tid >> 6) + 8 i' ... (8 positions per thread and operand)
  const int sc = tid & 63, sp = tid >> 6;
  const int gn = n0 + sc;
  const bool gok = gn < a.N;
  const int kk = k0 + sc;
  const bool k1 = kk < a.K, k2 = !k1 && kk < Ktot;
  int64_t koff = 0;
  if (k1) { const int seg = kk / a.kseg; koff = (int64_t)seg * a.is_seg + (kk - seg * a.kseg); }
  const bool in_dense = dense_strides(a.T, a.F, a.is_b, a.is_t, a.is_f);
  for (int64_t pc = p_begin; pc < p_end; pc += WGEN_POS) {
#pragma unroll
    for (int i = 0; i < WGEN_POS / 4; ++i) {
      const int pl = sp + 4 * i;
      const int64_t p = pc + pl;
      const bool ok = p < p_end;
      float gv = 0.f, iv = 0.f;
      if (ok && gok) gv = a.g[p * a.ldg + gn];
      if (ok && k1) {
        const int64_t ioff = in_dense ? p * a.is_f : off3(split_pos((unsigned)p, a.T, a.F), a.is_b, a.is_t, a.is_f);
        iv = a.in[ioff + koff];
      } else if (ok && k2) {
        const int idx = (int)((unsigned)p % (unsigned)a.seg_len);
        if (idx >= a.skip_first && idx < a.seg_len - a.skip_last) iv = a.in2[p * a.ld2 + a.shift2 + (kk - a.K)];
      }
      GT[sc][pl] = gv;
      IT[sc][pl] = iv;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < WGEN_POS / 16; ++m) {
      const f32x4 a4 = ld4(&GT[16 * w + j][16 * m + 4 * q]);
      csum += a4[0] + a4[1] + a4[2] + a4[3];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) acc[kt] = mfma16x4(a4, ld4(&IT[16 * kt + j][16 * m + 4 * q]), acc[kt]);
    }
    __syncthreads();
  }
  float* part = a.scratch + (size_t)blockIdx.x * ((size_t)a.N * Ktot + a.N);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * w + 4 * q + r, k = k0 + 16 * kt + j;
      if (n < a.N && k < Ktot) part[(size_t)n * Ktot + k] = acc[kt][r];
    }
  if (blockIdx.z == 0) {
    const float cs = quad_sum(csum);
    const int n = n0 + 16 * w + j;
    if (q == 0 && n < a.N) part[(size_t)a.N * Ktot + n] = cs;
  }
}
// The same tiling on the fp16 matrix pipe (sb_wgrad_args.mma == 2, late round 6): both operands as fp16 hi + lo, three
// v_mfma_f32_16x16x32_f16 per 32-position chunk and 16 x 16 tile instead of eight fp32 instructions; g is scaled by the power of
// two 2^-ceil(log2 *gmax) before its split (lo' scaled by 2^11 more, accumulated apart) and the sums scaled back: the gradients'
// range is not fp16's.  A staging thread holds 8 CONSECUTIVE positions of its column, so an operand leaves as one 16-byte LDS
// write per term; the bias sums are exact fp32 sums of what the staging threads read.
typedef _Float16 w16x8 __attribute__((ext_vector_type(8)));
constexpr int WG16_LD = WGEN_POS + 8;          // halves per LDS row (80 B: 16-byte aligned, 16 lanes x 16 B hit 64 distinct banks)
__global__ __launch_bounds__(256) void wgrad_gen16_kernel(sb_wgrad_args a, int64_t P, int64_t per) {
  __shared__ __attribute__((aligned(16))) _Float16 GT[2][64][WG16_LD];
  __shared__ __attribute__((aligned(16))) _Float16 IT[2][64][WG16_LD];
  __shared__ float BS[4][64];
  constexpr float kLoUp = 2048.0f, kLoDn = 1.0f / 2048.0f;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int Ktot = a.K + a.K2;
  const int n0 = blockIdx.y * 64, k0 = blockIdx.z * 64;
  const int64_t p_begin = (int64_t)blockIdx.x * per, p_end = min(P, p_begin + per);
  float gS = 1.0f;
  if (a.gmax) { const float m = a.gmax[0]; gS = (m > 0.f && m < 3.0e38f) ? exp2f(-ceilf(log2f(m))) : 1.0f; }
  gS = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(gS)));
  f32x4 acc[4] = {zero4(), zero4(), zero4(), zero4()}, accx[4] = {zero4(), zero4(), zero4(), zero4()};
  float bsum = 0.f;
  // staging role: thread -> column c = tid & 63, positions 8 (tid >> 6) .. + 7 of the chunk
  const int sc = tid & 63, sp = tid >> 6;
  const int gn = n0 + sc;
  const bool gok = gn < a.N;
  const int kk = k0 + sc;
  const bool k1 = kk < a.K, k2 = !k1 && kk < Ktot;
  int64_t koff = 0;
  if (k1) { const int seg = kk / a.kseg; koff = (int64_t)seg * a.is_seg + (kk - seg * a.kseg); }
  const bool in_dense = dense_strides(a.T, a.F, a.is_b, a.is_t, a.is_f);
  for (int64_t pc = p_begin; pc < p_end; pc += WGEN_POS) {
    float gv[8], iv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t p = pc + 8 * sp + i;
      const bool ok = p < p_end;
      gv[i] = 0.f; iv[i] = 0.f;
      if (ok && gok) gv[i] = a.g[p * a.ldg + gn];
      if (ok && k1) {
        const int64_t ioff = in_dense ? p * a.is_f : off3(split_pos((unsigned)p, a.T, a.F), a.is_b, a.is_t, a.is_f);
        iv[i] = a.in[ioff + koff];
      } else if (ok && k2) {
        const int idx = (int)((unsigned)p % (unsigned)a.seg_len);
        if (idx >= a.skip_first && idx < a.seg_len - a.skip_last) iv[i] = a.in2[p * a.ld2 + a.shift2 + (kk - a.K)];
      }
    }
    w16x8 gh, gl, ih, il;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bsum += gv[i];
      const float v = gv[i] * gS;
      const _Float16 h1 = (_Float16)v;
      gh[i] = h1;
      gl[i] = (_Float16)__builtin_fmaf((float)h1, -kLoUp, v * kLoUp);
      const _Float16 h2 = (_Float16)iv[i];
      ih[i] = h2;
      il[i] = (_Float16)(iv[i] - (float)h2);
    }
    *reinterpret_cast<w16x8*>(&GT[0][sc][8 * sp]) = gh;
    *reinterpret_cast<w16x8*>(&GT[1][sc][8 * sp]) = gl;
    *reinterpret_cast<w16x8*>(&IT[0][sc][8 * sp]) = ih;
    *reinterpret_cast<w16x8*>(&IT[1][sc][8 * sp]) = il;
    __syncthreads();
    {
      const w16x8 ah = *reinterpret_cast<const w16x8*>(&GT[0][16 * w + j][8 * q]);
      const w16x8 al = *reinterpret_cast<const w16x8*>(&GT[1][16 * w + j][8 * q]);
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const w16x8 bh = *reinterpret_cast<const w16x8*>(&IT[0][16 * kt + j][8 * q]);
        const w16x8 bl = *reinterpret_cast<const w16x8*>(&IT[1][16 * kt + j][8 * q]);
        accx[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accx[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[kt], 0, 0, 0);
        acc[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[kt], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  const float invS = 1.0f / gS;
  float* part = a.scratch + (size_t)blockIdx.x * ((size_t)a.N * Ktot + a.N);
#pragma unroll
  for (int kt = 0; kt < 4; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * w + 4 * q + r, k = k0 + 16 * kt + j;
      if (n < a.N && k < Ktot) part[(size_t)n * Ktot + k] = __builtin_fmaf(accx[kt][r], kLoDn, acc[kt][r]) * invS;
    }
  if (blockIdx.z == 0) {                       // (uniform over the workgroup)
    BS[sp][sc] = bsum;
    __syncthreads();
    if (tid < 64 && n0 + tid < a.N) part[(size_t)a.N * Ktot + n0 + tid] = (BS[0][tid] + BS[1][tid]) + (BS[2][tid] + BS[3][tid]);
  }
}
// position ranges (= partial rows) of the generic form: enough workgroups to fill the chip a few times over, ranges of whole
// 32-position chunks
static int wgrad_gen_rows(int64_t P, int N, int Ktot, int64_t* per_out) {
  const int tiles = ((N + 63) / 64) * ((Ktot + 63) / 64);
  int64_t rows = (1536 + tiles - 1) / tiles;
  const int64_t chunks = (P + WGEN_POS - 1) / WGEN_POS;
  if (rows > chunks) rows = chunks;
  if (rows < 1) rows = 1;
  const int64_t per = (chunks + rows - 1) / rows * WGEN_POS;
  rows = (P + per - 1) / per;
  if (per_out) *per_out = per;
  return (int)rows;
}

// out (+)= sum over partial rows; 2-D grid (columns x row-chunks) + atomics so that the reduction of
// 512 x 24 K partials is itself a wide, short kernel.  Columns route to dW1 / dW2 / the bias gradients.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partials, int rows, int N, int K1,
                                                           int K2, float* __restrict__ dW1, float* __restrict__ dW2,
                                                           float* __restrict__ db1, float* __restrict__ db2,
                                                           int transpose_out, int perm_k, int perm_n, int bias_mod,
                                                           sb_wview wv) {
  const int Ktot = K1 + K2;
  const int total = N * Ktot + N;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int rper = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rper, r1 = min(rows, r0 + rper);
  float s = 0.f, sa = 0.f, sb = 0.f, sc = 0.f;
  int r = r0;
  for (; r + 3 < r1; r += 4) {                        // four independent loads in flight
    s += partials[(size_t)r * total + i];
    sa += partials[(size_t)(r + 1) * total + i];
    sb += partials[(size_t)(r + 2) * total + i];
    sc += partials[(size_t)(r + 3) * total + i];
  }
  for (; r < r1; ++r) s += partials[(size_t)r * total + i];
  s += sa + sb + sc;
  if (i < N * Ktot) {
    const int n = i / Ktot, k = i - n * Ktot;
    if (k < K1 && wv.nmod != 0) {
      // destination through a weight view over the parameter's native layout (sb_wview)
      const int kl = k % wv.kmod, kh = k / wv.kmod;
      if (kl < wv.kvalid && n < wv.nvalid)
        atomicAdd(dW1 + wv.off + (int64_t)(n % wv.nmod) * wv.sn_lo + (int64_t)(n / wv.nmod) * wv.sn_hi +
                  (int64_t)kl * wv.sk_lo + (int64_t)kh * wv.sk_hi, s);
    } else if (k < K1) {
      // destination in the parameter's native layout (sb_wgrad_args.perm_k / perm_n)
      const int kd = perm_k > 0 ? (k % perm_k) * (K1 / perm_k) + k / perm_k : k;
      const int nd = perm_n > 0 ? (n % perm_n) * (N / perm_n) + n / perm_n : n;
      atomicAdd(dW1 + (transpose_out ? (size_t)kd * N + nd : (size_t)nd * K1 + kd), s);
    } else if (dW2) atomicAdd(dW2 + (size_t)n * K2 + (k - K1), s);
  } else {
    const int n = i - N * Ktot;
    const int nb = bias_mod > 0 ? n % bias_mod : n;
    if (wv.nmod != 0 && n >= wv.nvalid) return;
    if (db1) atomicAdd(db1 + nb, s);
    if (db2) atomicAdd(db2 + nb, s);
  }
}

// out[o(i)] += sum over rows of partials[r][i].  Block = 8 row lanes x 32 columns: eight rows are in flight per column
// (a single thread per column walking all rows is latency-bound: ~180 ns per dependent load).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partials, int rows, int64_t ld, int n,
                                                          float* __restrict__ out, int tr_N, int tr_K) {
  const int col = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + col;
  const int rper = (rows + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rper, r1 = min(rows, r0 + rper);
  float s0 = 0.f, s1 = 0.f;
  if (i < n) {
    int r = r0 + rl;
    for (; r + 8 < r1; r += 16) { s0 += partials[(size_t)r * ld + i]; s1 += partials[(size_t)(r + 8) * ld + i]; }
    if (r < r1) s0 += partials[(size_t)r * ld + i];
  }
  __shared__ float red[8][32];
  red[rl][col] = s0 + s1;
  __syncthreads();
  if (rl == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][col];
    int o = i;
    if (tr_N > 0) { const int nn = i / tr_K, kk = i - nn * tr_K; o = kk * tr_N + nn; }
    atomicAdd(out + o, s);
  }
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, int64_t P, int64_t ldg, int N,
                                                     float* __restrict__ scratch) {
  // block (x: row chunk, y: 64-column chunk): thread = (row lane rr 0..3, column c 0..63)
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), rr = threadIdx.x >> 6;
  __shared__ float red[4][64];
  float s = 0.f;
  if (c < N)
    for (int64_t p = (int64_t)blockIdx.x * 4 + rr; p < P; p += (int64_t)gridDim.x * 4) s += g[p * ldg + c];
  red[rr][threadIdx.x & 63] = s;
  __syncthreads();
  if (rr == 0 && c < N)
    scratch[(size_t)blockIdx.x * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Weight forms: every job copies the logical [N, K] matrix a view describes over a parameter's native layout into a
// dense row-major buffer (zeros where the view says so).  One launch refreshes all forms of a model (a few hundred KB).
__global__ __launch_bounds__(256) void wview_gather_kernel(const sb_wview_job* __restrict__ jobs) {
  const sb_wview_job j = jobs[blockIdx.y];
  const int total = j.N * j.K;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int n = idx / j.K, k = idx - n * j.K;
    const int kl = k % j.v.kmod, kh = k / j.v.kmod;
    const bool ok = kl < j.v.kvalid && n < j.v.nvalid;
    const int64_t ad = j.v.off + (int64_t)(n % j.v.nmod) * j.v.sn_lo + (int64_t)(n / j.v.nmod) * j.v.sn_hi +
                       (int64_t)kl * j.v.sk_lo + (int64_t)kh * j.v.sk_hi;
    j.dst[idx] = ok ? j.src[ad] : 0.f;
  }
}

template <int NT, int EPI>
int launch_linear2(const sb_linear_args& a, int64_t P, hipStream_t st, int ny) {
  const size_t lds = (size_t)NT * 16 * (a.K + 4) * sizeof(float);
  if (lds > 160 * 1024) return -1005;
  (void)hipFuncSetAttribute((const void*)linear_kernel<NT, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((linear_kernel<NT, EPI>), dim3(sb_linear_grid(P), ny), dim3(256), lds, st, a, P);
  return 0;
}
template <int NT>
int launch_linear(const sb_linear_args& a, int64_t P, hipStream_t st, int ny) {
  if (ny > 1 && a.epi != SB_EPI_NONE && a.epi != SB_EPI_RES) return -1006;      // column slices: plain / residual epilogues
  switch (a.epi) {
    case SB_EPI_NONE: return launch_linear2<NT, SB_EPI_NONE>(a, P, st, ny);
    case SB_EPI_RES: return launch_linear2<NT, SB_EPI_RES>(a, P, st, ny);
    case SB_EPI_PRELU: return launch_linear2<NT, SB_EPI_PRELU>(a, P, st, 1);
    case SB_EPI_LN: if constexpr (NT <= 2 || NT == 4) return launch_linear2<NT, SB_EPI_LN>(a, P, st, 1); else return -1006;
    case SB_EPI_LNBWD: if constexpr (NT <= 2) return launch_linear2<NT, SB_EPI_LNBWD>(a, P, st, 1); else return -1006;
    default: return -1007;
  }
}

}  // namespace

extern "C" int sb_linear_grid(int64_t positions) {
  const int64_t t = (positions + 63) / 64;
  return (int)(t < LIN_MAX_WG ? (t < 1 ? 1 : t) : LIN_MAX_WG);
}
extern "C" int sb_wgrad_grid(int64_t positions) {
  const int64_t t = (positions + 63) / 64;
  return (int)(t < WG_MAX_WG ? (t < 1 ? 1 : t) : WG_MAX_WG);
}

extern "C" int sb_reduce_rows(const float* partials, int rows, int64_t ld, int n, float* out, void* stream) {
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((n + 31) / 32, rows >= 512 ? 8 : 1), dim3(256), 0, (hipStream_t)stream,
                     partials, rows, ld, n, out, 0, 0);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_linear_fwd(const sb_linear_args* ap, void* stream) {
  if (!ap) return -1001;
  sb_linear_args a = *ap;
  if (a.N % 16 || a.K % 16 || a.kseg % 16 || a.N <= 0 || a.K <= 0) return -1002;
  if ((a.epi == SB_EPI_LN || a.epi == SB_EPI_LNBWD) && a.n_valid != a.N) return -1003;
  const int64_t P = (int64_t)a.B * a.T * a.F;
  if (P <= 0 || P >= (1ll << 31)) return -1001;          // 32-bit position arithmetic in the kernel
  hipStream_t st = (hipStream_t)stream;
  const int grid = sb_linear_grid(P);
  if (a.epi == SB_EPI_LNBWD && a.partials)
    (void)hipMemsetAsync(a.partials, 0, (size_t)grid * (2 * a.N + 1) * sizeof(float), st);
  int rc;
  if (a.mma == 1) {
    // fp16x3 form: N <= 32, K = 9 * 32 or 9 * 16 (padded to 160), kseg | 8, bias / residual / LayerNorm epilogues
    const int kc = (a.K + 31) / 32;
    if (a.N > 32 || (kc != 9 && kc != 5) || a.kseg % 8 || a.K % 16 || a.epi > SB_EPI_LN || a.epi == SB_EPI_PRELU) return -1004;
#define SB_L16(NT_, KC_, EP_) hipLaunchKernelGGL((linear16_kernel<NT_, KC_, EP_>), dim3(grid), dim3(256), 0, st, a, P)
#define SB_L16E(NT_, KC_) do { if (a.epi == SB_EPI_LN) SB_L16(NT_, KC_, SB_EPI_LN); else if (a.epi == SB_EPI_RES) SB_L16(NT_, KC_, SB_EPI_RES); \
                              else SB_L16(NT_, KC_, SB_EPI_NONE); } while (0)
    if (a.N == 16) { if (kc == 9) SB_L16E(1, 9); else SB_L16E(1, 5); }
    else { if (kc == 9) SB_L16E(2, 9); else SB_L16E(2, 5); }
#undef SB_L16E
#undef SB_L16
    SB_CHECK_LAUNCH();
    return 0;
  }
  // Column slices (grid.y): with a handful of positions (the streaming chunk step: one frame) a single workgroup would
  // stage the whole [N, K] weight block through its LDS -- 17 us for the 304 x 288 STFT basis; N / width workgroups, each
  // with `width` rows of W, read it in parallel.  Also how N > 128 is served in one launch.
  int ny = 1;
  if ((a.epi == SB_EPI_NONE || a.epi == SB_EPI_RES) && (a.N > 128 || (P <= SB_LINEAR_SLICE_MAX_P && a.N >= 64))) {
    static const int cand[] = {128, 96, 80, 64, 48, 32, 16};
    int width = 16;
    for (int c : cand)
      if (a.N % c == 0 && a.N / c >= (P <= SB_LINEAR_SLICE_MAX_P ? 8 : 1) && (size_t)c * (a.K + 4) * sizeof(float) <= 160 * 1024) {
        width = c;
        break;
      }
    ny = a.N / width;
    a.N = width;
  }
  switch (a.N / 16) {
    case 1: rc = launch_linear<1>(a, P, st, ny); break;
    case 2: rc = launch_linear<2>(a, P, st, ny); break;
    case 3: rc = launch_linear<3>(a, P, st, ny); break;
    case 4: rc = launch_linear<4>(a, P, st, ny); break;
    case 5: rc = launch_linear<5>(a, P, st, ny); break;
    case 6: rc = launch_linear<6>(a, P, st, ny); break;
    case 8: rc = launch_linear<8>(a, P, st, ny); break;
    default: return -1004;
  }
  if (rc) return rc;
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_wview_gather(const sb_wview_job* jobs, int njobs, int max_elems, void* stream) {
  if (!jobs || njobs <= 0 || max_elems <= 0) return -1001;
  int gx = (max_elems + 255) / 256;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(wview_gather_kernel, dim3(gx, njobs), dim3(256), 0, (hipStream_t)stream, jobs);
  SB_CHECK_LAUNCH();
  return 0;
}

// One body for the launch and for the question "how many partial rows will this call write" (sb_wgrad_scratch_rows): the
// dispatch below is the only place that knows which kernel serves a shape.  launch == false: nothing is launched, *rows_out is set.
static int wgrad_dispatch(const sb_wgrad_args& a, void* stream, bool launch, int* rows_out) {
  const int64_t P = (int64_t)a.B * a.T * a.F;
  if (P <= 0 || P >= (1ll << 31)) return -1001;          // 32-bit position arithmetic in the kernels
  const int nblk = (a.N + 15) / 16, kt1 = (a.K + 15) / 16, kt2 = a.K2 / 16, ntw = nblk;
  if (a.K2 % 16 || (a.K2 && a.K % 16)) return -1002;
  if ((a.perm_k > 0 && a.K % a.perm_k) || (a.perm_n > 0 && a.N % a.perm_n) || a.perm_k < 0 || a.perm_n < 0 || a.bias_mod < 0)
    return -1002;
  if (a.wv.nmod != 0 && (a.wv.nmod < 0 || a.wv.kmod <= 0 || a.perm_k || a.perm_n || a.transpose_out)) return -1002;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(sb_wgrad_grid(P)), block(256);
  // wide-load kernel: single source, whole tiles, whole segments, rows aligned for 16-byte (fp16: 8-byte) loads
  const int ts = a.kseg % 16 == 0 ? a.kseg / 16 : 0;
  const int64_t es = a.in_f16 ? 2 : 4;
  const bool wide = kt2 == 0 && a.K % 16 == 0 && a.N % 16 == 0 && ts > 0 && kt1 % ts == 0 && a.is_b % 4 == 0 &&
                    a.is_t % 4 == 0 && a.is_f % 4 == 0 && a.is_seg % 4 == 0 && a.ldg % 4 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.in) % (4 * es)) == 0 && (reinterpret_cast<uintptr_t>(a.g) % 16) == 0;
  bool launched = false, generic = false;
  // fp16 matrix-pipe form for the K = 288 / 144 convolutions (mma == 1): single source, whole segments of 6 or 3 tiles
  if (a.mma == 1) {
    const bool ok16 = kt2 == 0 && a.K % 16 == 0 && ts > 0 && kt1 % ts == 0 && a.is_b % 4 == 0 && a.is_t % 4 == 0 &&
                      a.is_f % 4 == 0 && a.is_seg % 4 == 0 && (reinterpret_cast<uintptr_t>(a.in) % 16) == 0 &&
                      ((a.N % 16 == 0 && a.ldg % 4 == 0 && (reinterpret_cast<uintptr_t>(a.g) % 16) == 0) || ntw == 1);
    if (!ok16 || a.in_f16 || ntw > 2) return -1004;
#define SB_W16(NTW_, NSEG_, TS_) \
    if (!launched && ntw == NTW_ && kt1 == NSEG_ * TS_ && ts == TS_) { \
      if (launch) hipLaunchKernelGGL((wgrad16_kernel<NTW_, NSEG_, TS_>), grid, block, 0, st, a, P); launched = true; }
    SB_W16(1, 3, 6) SB_W16(1, 3, 3)
#undef SB_W16
    if (!launched && ntw == 2 && kt1 == 18 && ts == 6) {
      if (launch) hipLaunchKernelGGL((wgrad16_n32_kernel<3, 6>), grid, block, 0, st, a, P);
      launched = true; }
    if (!launched) return -1004;
  }
#define SB_WW(NTW_, KT1_, TS_, H16_) \
  if (!launched && wide && ntw == NTW_ && kt1 == KT1_ && ts == TS_ && (a.in_f16 != 0) == H16_) { \
    if (launch) hipLaunchKernelGGL((wgrad_wide_kernel<NTW_, KT1_, TS_, H16_>), grid, block, 0, st, a, P); launched = true; }
  SB_WW(1, 4, 4, true) SB_WW(2, 4, 4, true)
  SB_WW(1, 1, 1, false) SB_WW(1, 2, 2, false) SB_WW(1, 4, 4, false) SB_WW(1, 8, 8, false) SB_WW(1, 5, 5, false)
  SB_WW(1, 9, 3, false) SB_WW(1, 18, 6, false) SB_WW(2, 1, 1, false) SB_WW(2, 2, 2, false) SB_WW(2, 4, 4, false)
  SB_WW(2, 18, 6, false) 
  SB_WW(1, 3, 3, false) SB_WW(1, 6, 6, false)
  SB_WW(2, 5, 5, false) SB_WW(2, 6, 6, false)
#undef SB_WW
  if (launched) {
  } else if (a.in_f16) {
    if (kt1 != 4 || kt2 != 0 || ntw > 2) return -1004;
    if (!launch) {}
    else if (ntw == 1) hipLaunchKernelGGL((wgrad_kernel<1, 4, 0, true>), grid, block, 0, st, a, P);
    else hipLaunchKernelGGL((wgrad_kernel<2, 4, 0, true>), grid, block, 0, st, a, P);
  } else
#define SB_WG(NTW_, KT1_, KT2_) \
  if (ntw == NTW_ && kt1 == KT1_ && kt2 == KT2_) { if (launch) hipLaunchKernelGGL((wgrad_kernel<NTW_, KT1_, KT2_>), grid, block, 0, st, a, P); } else
  SB_WG(1, 1, 0) SB_WG(1, 2, 0) SB_WG(1, 4, 0) SB_WG(1, 8, 0) SB_WG(1, 5, 0) SB_WG(1, 9, 0) SB_WG(1, 18, 0)
  SB_WG(2, 1, 0) SB_WG(2, 2, 0) SB_WG(2, 4, 0) SB_WG(2, 8, 0) SB_WG(2, 10, 0) SB_WG(2, 18, 0)
  SB_WG(3, 8, 0) SB_WG(4, 8, 0) SB_WG(5, 8, 0) SB_WG(8, 8, 0) SB_WG(10, 8, 0) SB_WG(1, 3, 0) SB_WG(1, 6, 0) SB_WG(2, 5, 0) SB_WG(2, 6, 0) SB_WG(1, 1, 4) SB_WG(2, 2, 4)
  { generic = true; }
#undef SB_WG
  int rows = (int)grid.x * 4;
  if (generic) {
    // any other shape: the generic tiled form (fp32 sources, whole 16-column K tiles)
    if (a.kseg <= 0) return -1004;
    int64_t per = 0;
    rows = wgrad_gen_rows(P, a.N, a.K + a.K2, &per);
    if (launch) {
      if (a.mma == 2) hipLaunchKernelGGL(wgrad_gen16_kernel, dim3(rows, (a.N + 63) / 64, (a.K + a.K2 + 63) / 64), block, 0, st, a, P, per);
      else hipLaunchKernelGGL(wgrad_gen_kernel, dim3(rows, (a.N + 63) / 64, (a.K + a.K2 + 63) / 64), block, 0, st, a, P, per);
    }
  }
  if (rows_out) *rows_out = rows;
  if (!launch) return 0;
  SB_CHECK_LAUNCH();
  const int total = a.N * (a.K + a.K2) + a.N;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256, rows >= 256 ? 16 : 1), dim3(256), 0, st, a.scratch,
                     rows, a.N, a.K, a.K2, a.dW, a.dW2, a.dbias, a.dbias2, a.transpose_out, a.perm_k, a.perm_n,
                     a.bias_mod, a.wv);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_wgrad(const sb_wgrad_args* ap, void* stream) {
  if (!ap) return -1001;
  return wgrad_dispatch(*ap, stream, true, nullptr);
}

extern "C" int sb_wgrad_scratch_rows(const sb_wgrad_args* ap) {
  if (!ap) return -1001;
  int rows = 0;
  const int rc = wgrad_dispatch(*ap, nullptr, false, &rows);
  return rc ? rc : rows;
}

extern "C" int sb_colsum(const float* g, int64_t P, int64_t ldg, int N, float* out, float* scratch, void* stream) {
  if (P <= 0 || N <= 0) return -1001;
  hipStream_t st = (hipStream_t)stream;
  int64_t gx = (P + 3) / 4;
  if (gx > 256) gx = 256;
  dim3 grid((unsigned)gx, (N + 63) / 64), block(256);
  hipLaunchKernelGGL(colsum_kernel, grid, block, 0, st, g, P, ldg, N, scratch);
  SB_CHECK_LAUNCH();
  hipLaunchKernelGGL(reduce_rows_kernel, dim3((N + 31) / 32, 1), dim3(256), 0, st, scratch, (int)gx, (int64_t)N, N,
                     out, 0, 0);
  SB_CHECK_LAUNCH();
  return 0;
}
