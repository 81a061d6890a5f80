// Fine-tune loss of the reference (src/losses/MultiResoLoss.py:6-31): auraloss MultiResolutionSTFTLoss with
// perceptual (A-) weighting and the linear-magnitude L1 term, plus l1_ratio * L1(est, gt).
//
// The STFTs are GEMMs over overlapping rows (sb_linear_fwd: frames of the reflect-padded signal times the windowed DFT
// basis restricted to the window's support), so what lives here is the streaming rest of the chain:
//   sb_fir          the 101-tap A-weighting FIR (conv1d, zero padded) -- also its backward (flipped taps)
//   sb_reflect_pad  torch.stft(center=True, pad_mode="reflect") framing input
//   sb_stft_mag_l1  |X|, |Y| from interleaved (re, im) spectra, sum |(|X| - |Y|)|, d(loss)/d(spectrum of X)
//   sb_stft_mag_terms  the same with auraloss's other two terms (log-magnitude L1, spectral convergence): two passes
//   sb_stft_f64acc  the STFT with double accumulation (what the log-magnitude term's gradient needs: see the kernel)
//   sb_frames_fold  backward of framing + reflect padding: frame gradients -> signal gradient
//   sb_l1_grad      sum |x - y| and its gradient
// All HBM-bound single passes.
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

constexpr int FIR_TILE = 1024;        // outputs per workgroup
constexpr int FIR_MAX_TAPS = 257;

// y[b, n] = sum_k taps[k] * x[b, n + k - ntaps / 2]   (torch conv1d = cross-correlation, zero padding ntaps / 2)
// PAIR: the sum is formed in double and leaves as TWO fp32 planes y + y_lo (the log-magnitude term's STFT adds them back up in
// double: the A-weighted signal's low bands sit 50 dB and more below its rounding noise floor otherwise -- see stft_f64acc_kernel)
template <bool PAIR>
__global__ __launch_bounds__(256) void fir_kernel(const float* __restrict__ x, const float* __restrict__ taps,
                                                  float* __restrict__ y, float* __restrict__ y_lo, int64_t N, int ntaps) {
  __shared__ float xs[FIR_TILE + FIR_MAX_TAPS + 3];
  __shared__ float ts[FIR_MAX_TAPS];
  const int b = blockIdx.y, tid = threadIdx.x, half = ntaps / 2;
  const int64_t n0 = (int64_t)blockIdx.x * FIR_TILE;
  const float* xb = x + (int64_t)b * N;
  for (int i = tid; i < FIR_TILE + ntaps - 1; i += 256) {
    const int64_t n = n0 + i - half;
    xs[i] = (n >= 0 && n < N) ? xb[n] : 0.f;
  }
  for (int i = tid; i < ntaps; i += 256) ts[i] = taps[i];
  __syncthreads();
  if constexpr (PAIR) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < ntaps; ++k) {
      const double t = (double)ts[k];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_fma(t, (double)xs[tid + 256 * r + k], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = n0 + tid + 256 * r;
      if (n < N) {
        const float hi = (float)acc[r];
        y[(int64_t)b * N + n] = hi;
        y_lo[(int64_t)b * N + n] = (float)(acc[r] - (double)hi);
      }
    }
  } else {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < ntaps; ++k) {
      const float t = ts[k];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = __builtin_fmaf(t, xs[tid + 256 * r + k], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t n = n0 + tid + 256 * r;
      if (n < N) y[(int64_t)b * N + n] = acc[r];
    }
  }
}

// xp[b, i] = x[b, refl(i - pad)] for i < N + 2 pad (refl(j) = -j below 0, 2 (N - 1) - j from N on), zero up to ldp
__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ xp, int64_t N,
                                                          int pad, int64_t ldp) {
  const int b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= ldp) return;
  float v = 0.f;
  if (i < N + 2 * pad) {
    int64_t j = i - pad;
    if (j < 0) j = -j;
    else if (j >= N) j = 2 * (N - 1) - j;
    v = x[(int64_t)b * N + j];
  }
  xp[(int64_t)b * ldp + i] = v;
}

// spectra: rows of ld floats, interleaved (re_k, im_k), k < nbins.  One thread per (row, k).
// partial[blockIdx] = sum over the block of | |X| - |Y| | ; dsx = gscale * sign(|X| - |Y|) * X / |X|  (0 where the
// squared magnitude sits on the clamp eps, as torch.clamp's gradient; columns k >= nbins are written as zeros)
__global__ __launch_bounds__(256) void stft_mag_l1_kernel(const float* __restrict__ sx, const float* __restrict__ sy,
                                                          int64_t rows, int nbins, int ld, float eps, float gscale,
                                                          float* __restrict__ dsx, float* __restrict__ partial) {
  const int half = ld / 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = rows * half;
  float d = 0.f;
  if (idx < total) {
    const int64_t row = idx / half;
    const int k = (int)(idx - row * half);
    const int64_t o = row * ld + 2 * k;
    float2 g = {0.f, 0.f};
    if (k < nbins) {
      const float2 a = *reinterpret_cast<const float2*>(sx + o), c = *reinterpret_cast<const float2*>(sy + o);
      const float px = a.x * a.x + a.y * a.y, py = c.x * c.x + c.y * c.y;
      const float mx = sqrtf(fmaxf(px, eps)), my = sqrtf(fmaxf(py, eps));
      const float e = mx - my;
      d = fabsf(e);
      if (px > eps && e != 0.f) {
        const float s = (e > 0.f ? gscale : -gscale) / mx;
        g.x = s * a.x;
        g.y = s * a.y;
      }
    }
    if (dsx) *reinterpret_cast<float2*>(dsx + o) = g;
  }
  d = wave_sum(d);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// The three auraloss STFT terms together (STFTLoss.forward: w_sc * SC + w_log_mag * L1(log|X|, log|Y|) + w_lin_mag *
// L1(|X|, |Y|)).  The spectral-convergence term ||  |Y| - |X|  ||_F / || |Y| ||_F is a ratio of GLOBAL norms over the
// whole [rows, nbins] magnitude array, so its gradient needs both sums first -- two passes:
//   pass 1 (stft_mag_sums_kernel): per block the four sums  S0 = sum |e|, S1 = sum |log|X| - log|Y||, S2 = sum e^2,
//           S3 = sum |Y|^2  (e = |X| - |Y|) -> partial[4][blocks]
//   mag_terms_finish_kernel: fixed-tree totals -> sums[4], *loss (+)= scale * (w_lin S0 / cnt + w_log S1 / cnt + w_sc sqrt(S2 / S3))
//   pass 2 (stft_mag_grad_kernel): dsx = scale * [ (w_lin / cnt) sign(e) + (w_log / cnt) sign(e) / |X| + w_sc e / (sqrt(S2) sqrt(S3)) ] * X / |X|
//           (log is monotone: sign(log|X| - log|Y|) = sign(e); zero on the clamp, as torch.clamp's gradient; zero when S2 == 0)
__global__ __launch_bounds__(256) void stft_mag_sums_kernel(const float* __restrict__ sx, const float* __restrict__ sy,
                                                            int64_t rows, int nbins, int ld, float eps,
                                                            float* __restrict__ partial, int64_t nblocks) {
  const int half = ld / 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (idx < rows * half) {
    const int64_t row = idx / half;
    const int k = (int)(idx - row * half);
    if (k < nbins) {
      const int64_t o = row * ld + 2 * k;
      const float2 a = *reinterpret_cast<const float2*>(sx + o), c = *reinterpret_cast<const float2*>(sy + o);
      const float mx = sqrtf(fmaxf(a.x * a.x + a.y * a.y, eps)), my = sqrtf(fmaxf(c.x * c.x + c.y * c.y, eps));
      const float e = mx - my;
      v[0] = fabsf(e);
      v[1] = fabsf(logf(mx) - logf(my));
      v[2] = e * e;
      v[3] = my * my;
    }
  }
  __shared__ float ws[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = wave_sum(v[i]);
    if ((threadIdx.x & 63) == 0) ws[i][threadIdx.x >> 6] = t;
  }
  __syncthreads();
  if (threadIdx.x < 4) partial[threadIdx.x * nblocks + blockIdx.x] = (ws[threadIdx.x][0] + ws[threadIdx.x][1]) + (ws[threadIdx.x][2] + ws[threadIdx.x][3]);
}

__global__ __launch_bounds__(256) void mag_terms_finish_kernel(const float* __restrict__ partial, int64_t n, float cnt,
                                                               float w_lin, float w_log, float w_sc, float scale,
                                                               float* __restrict__ sums, float* __restrict__ loss) {
  __shared__ double red[256];
  __shared__ double tot[4];
  for (int q = 0; q < 4; ++q) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)partial[q * n + i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) tot[q] = red[0];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    for (int q = 0; q < 4; ++q) sums[q] = (float)tot[q];
    double l = (double)w_lin * tot[0] / cnt + (double)w_log * tot[1] / cnt;
    if (w_sc != 0.f) l += (double)w_sc * sqrt(tot[2] / tot[3]);
    loss[0] += (float)(l * (double)scale);
  }
}

__global__ __launch_bounds__(256) void stft_mag_grad_kernel(const float* __restrict__ sx, const float* __restrict__ sy,
                                                            int64_t rows, int nbins, int ld, float eps, float cnt,
                                                            float w_lin, float w_log, float w_sc, float scale,
                                                            const float* __restrict__ sums, float* __restrict__ dsx) {
  const int half = ld / 2;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * half) return;
  const int64_t row = idx / half;
  const int k = (int)(idx - row * half);
  const int64_t o = row * ld + 2 * k;
  float2 g = {0.f, 0.f};
  if (k < nbins) {
    const float2 a = *reinterpret_cast<const float2*>(sx + o), c = *reinterpret_cast<const float2*>(sy + o);
    const float px = a.x * a.x + a.y * a.y;
    if (px > eps) {
      const float mx = sqrtf(px), my = sqrtf(fmaxf(c.x * c.x + c.y * c.y, eps));
      const float e = mx - my;
      const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
      float d = sg * (w_lin / cnt) + sg * (w_log / cnt) / mx;
      const float s2 = sums[2], s3 = sums[3];
      if (w_sc != 0.f && s2 > 0.f) d += w_sc * e / (sqrtf(s2) * sqrtf(s3));
      const float f = scale * d / mx;
      g.x = f * a.x;
      g.y = f * a.y;
    }
  }
  *reinterpret_cast<float2*>(dsx + o) = g;
}

// out[0] (+)= scale * sum partial[0 .. n)  -- one workgroup, fixed summation tree (bit-identical on every replica)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int64_t n, float scale,
                                                           float* __restrict__ out, int accumulate) {
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(red[0] * (double)scale);
}

// STFT of the log-magnitude term: spec[(b, t), n] = sum_k xp[b, t hop + off + k] * w[n, k], ACCUMULATED IN DOUBLE.
// d log|X| = X / |X|^2 weighs every bin by the inverse of its energy, so the gradient of auraloss's log-magnitude term is
// set by the few bins where the frame's K = 240 .. 1200 products cancel to ~1e-3 of their size: a sequential fp32 sum
// (sb_linear_fwd's MFMA chain) carries ~K eps / sqrt(2) of the term size there -- 3e-4 .. 6e-4 of the gradient's norm,
// 3 x what a fp32 FFT (log2 K stages) leaves.  Each fp32 x fp32 product is exact in double, the sum is rounded once: the
// spectrum is the correctly rounded one, and the gradient meets the float64 oracle at the 1e-4 the other terms meet.
// Vector fp64 FMA runs at the fp32 rate on gfx950; plain 64 x 64 x 8 LDS tiles, 4 x 4 outputs per thread (a loss term:
// ~80 GFLOP per 16 five-second clips, a few ms).
constexpr int DT_M = 64, DT_N = 64, DT_K = 8;
__global__ __launch_bounds__(256) void stft_f64acc_kernel(const float* __restrict__ xp, const float* __restrict__ w,
                                                          float* __restrict__ spec, int64_t rows, int nfr, int64_t ldp, int hop,
                                                          int off, int K, int N, int64_t lo_off) {
  // lo_off != 0: the signal is the pair xp[i] + xp[i + lo_off] (sb_fir_pair's two planes, reflect-padded alike)
  __shared__ double As[DT_K][DT_M + 2];
  __shared__ float Bs[DT_K][DT_N + 4];
  const int tid = threadIdx.x, tm = tid >> 4, tn = tid & 15;
  const int64_t m0 = (int64_t)blockIdx.x * DT_M;
  const int n0 = blockIdx.y * DT_N;
  // loader: thread -> (row lm, k pair lk) of the A tile and (column lm, k pair lk) of the B tile
  const int lm = tid >> 2, lk = (tid & 3) * 2;
  const int64_t ra = min(m0 + lm, rows - 1);
  const float* pa = xp + (ra / nfr) * ldp + (ra % nfr) * (int64_t)hop + off;
  const float* pb = w + (int64_t)min(n0 + lm, N - 1) * K;
  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
  for (int k0 = 0; k0 < K; k0 += DT_K) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k = k0 + lk + e;
      As[lk + e][lm] = k < K ? (double)pa[k] + (lo_off ? (double)pa[k + lo_off] : 0.0) : 0.0;
      Bs[lk + e][lm] = k < K ? pb[k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < DT_K; ++k) {
      double av[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[k][4 * tm + i];
      const f32x4 bv = *reinterpret_cast<const f32x4*>(&Bs[k][4 * tn]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_fma(av[i], (double)bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + 4 * tm + i;
    if (m >= rows) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 4 * tn + j;
      if (n < N) spec[m * N + n] = (float)acc[i][j];
    }
  }
}

// dx[b, m] (+)= sum over the padded indices i that map onto m (i = m + pad, the left mirror pad - m, the right mirror
// pad + 2 (N - 1) - m) of  sum_t dframes[b, t, i - off - t * hop]  (0 <= i - off - t hop < K)
__global__ __launch_bounds__(256) void frames_fold_kernel(const float* __restrict__ df, float* __restrict__ dx, int64_t N,
                                                          int nframes, int K, int ldk, int hop, int off, int pad,
                                                          int accumulate) {
  const int b = blockIdx.y;
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= N) return;
  const float* dfb = df + (int64_t)b * nframes * ldk;
  int64_t src[3] = {m + pad, (m >= 1 && m <= pad) ? pad - m : -1,
                    (m >= N - 1 - pad && m <= N - 2) ? pad + 2 * (N - 1) - m : -1};
  float acc = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int64_t i = src[c] - off;                  // position inside the window-support grid
    if (src[c] < 0 || i < 0) continue;
    int64_t t1 = i / hop;                            // last frame that can cover i
    if (t1 > nframes - 1) t1 = nframes - 1;
    for (int64_t t = t1; t >= 0; --t) {
      const int64_t k = i - t * hop;
      if (k >= K) break;
      acc += dfb[t * ldk + k];
    }
  }
  float* o = dx + (int64_t)b * N + m;
  *o = accumulate ? *o + acc : acc;
}

// partial[blockIdx] = sum |x - y| ; dx (+)= gscale * sign(x - y)
__global__ __launch_bounds__(256) void l1_grad_kernel(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                      float gscale, float* __restrict__ dx, int accumulate,
                                                      float* __restrict__ partial) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float d = 0.f;
  if (i < n) {
    const float e = x[i] - y[i];
    d = fabsf(e);
    if (dx) {
      const float g = e > 0.f ? gscale : (e < 0.f ? -gscale : 0.f);
      dx[i] = accumulate ? dx[i] + g : g;
    }
  }
  d = wave_sum(d);
  __shared__ float ws[4];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

}  // namespace

extern "C" int sb_fir(const float* x, const float* taps, float* y, int B, int64_t N, int ntaps, void* stream) {
  if (!x || !taps || !y || B <= 0 || N <= 0 || ntaps < 1 || ntaps > FIR_MAX_TAPS || !(ntaps & 1)) return -1001;
  hipLaunchKernelGGL(fir_kernel<false>, dim3(nblk(N, FIR_TILE), B), dim3(256), 0, (hipStream_t)stream, x, taps, y, nullptr, N,
                     ntaps);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_fir_pair(const float* x, const float* taps, float* y, float* y_lo, int B, int64_t N, int ntaps, void* stream) {
  if (!x || !taps || !y || !y_lo || B <= 0 || N <= 0 || ntaps < 1 || ntaps > FIR_MAX_TAPS || !(ntaps & 1)) return -1001;
  hipLaunchKernelGGL(fir_kernel<true>, dim3(nblk(N, FIR_TILE), B), dim3(256), 0, (hipStream_t)stream, x, taps, y, y_lo, N, ntaps);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_reflect_pad(const float* x, float* xp, int B, int64_t N, int pad, int64_t ldp, void* stream) {
  if (!x || !xp || B <= 0 || pad < 0 || N <= pad || ldp < N + 2 * pad) return -1001;
  hipLaunchKernelGGL(reflect_pad_kernel, dim3(nblk(ldp), B), dim3(256), 0, (hipStream_t)stream, x, xp, N, pad, ldp);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_stft_mag_l1_grid(int64_t rows, int ld) { return (int)nblk(rows * (ld / 2)); }

extern "C" int sb_stft_mag_l1(const float* spec_x, const float* spec_y, int64_t rows, int nbins, int ld, float eps,
                              float gscale, float* dspec_x, float* partial, float loss_scale, float* loss,
                              int accumulate, void* stream) {
  if (!spec_x || !spec_y || !partial || !loss || rows <= 0 || nbins <= 0 || ld < 2 * nbins || (ld & 1)) return -1001;
  const int64_t blocks = nblk(rows * (ld / 2));
  if (blocks >= (1ll << 31)) return -1002;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(stft_mag_l1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, spec_x, spec_y, rows, nbins, ld, eps,
                     gscale, dspec_x, partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, blocks, loss_scale, loss, accumulate);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_stft_mag_terms(const float* spec_x, const float* spec_y, int64_t rows, int nbins, int ld, float eps,
                                 float w_lin, float w_log, float w_sc, float scale, float* dspec_x, float* partial,
                                 float* sums, float* loss, void* stream) {
  if (!spec_x || !spec_y || !partial || !sums || !loss || rows <= 0 || nbins <= 0 || ld < 2 * nbins || (ld & 1)) return -1001;
  const int64_t blocks = nblk(rows * (ld / 2));
  if (blocks >= (1ll << 31)) return -1002;
  hipStream_t st = (hipStream_t)stream;
  const float cnt = (float)((double)rows * nbins);
  hipLaunchKernelGGL(stft_mag_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, spec_x, spec_y, rows, nbins, ld, eps,
                     partial, blocks);
  hipLaunchKernelGGL(mag_terms_finish_kernel, dim3(1), dim3(256), 0, st, partial, blocks, cnt, w_lin, w_log, w_sc, scale, sums,
                     loss);
  if (dspec_x)
    hipLaunchKernelGGL(stft_mag_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, spec_x, spec_y, rows, nbins, ld, eps,
                       cnt, w_lin, w_log, w_sc, scale, sums, dspec_x);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_stft_f64acc(const float* xp, const float* w, float* spec, int B, int nframes, int64_t ldp, int hop, int off,
                              int K, int N, int64_t lo_off, void* stream) {
  if (!xp || !w || !spec || B <= 0 || nframes <= 0 || hop <= 0 || off < 0 || K <= 0 || N <= 0 || lo_off < 0 ||
      ldp < (int64_t)(nframes - 1) * hop + off + K)
    return -1001;
  const int64_t rows = (int64_t)B * nframes;
  const int64_t gx = (rows + DT_M - 1) / DT_M;
  if (gx >= (1ll << 31)) return -1002;
  hipLaunchKernelGGL(stft_f64acc_kernel, dim3((unsigned)gx, (unsigned)((N + DT_N - 1) / DT_N)), dim3(256), 0,
                     (hipStream_t)stream, xp, w, spec, rows, nframes, ldp, hop, off, K, N, lo_off);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_frames_fold(const float* dframes, float* dx, int B, int64_t N, int nframes, int K, int ldk, int hop,
                              int off, int pad, int accumulate, void* stream) {
  if (!dframes || !dx || B <= 0 || N <= pad || nframes <= 0 || K <= 0 || ldk < K || hop <= 0 || off < 0 || pad < 0)
    return -1001;
  hipLaunchKernelGGL(frames_fold_kernel, dim3(nblk(N), B), dim3(256), 0, (hipStream_t)stream, dframes, dx, N, nframes, K,
                     ldk, hop, off, pad, accumulate);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_l1_grad(const float* x, const float* y, int64_t n, float gscale, float* dx, int accumulate,
                          float* partial, float loss_scale, float* loss, int accumulate_loss, void* stream) {
  if (!x || !y || !partial || !loss || n <= 0) return -1001;
  const int64_t blocks = nblk(n);
  if (blocks >= (1ll << 31)) return -1002;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(l1_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, y, n, gscale, dx, accumulate, partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partial, blocks, loss_scale, loss, accumulate_loss);
  SB_CHECK_LAUNCH();
  return 0;
}
