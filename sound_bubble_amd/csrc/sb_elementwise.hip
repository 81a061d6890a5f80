// HBM-bound streaming kernels of the Sound-Bubble hot path (features, FiLM,
// overlap-add, output-deconv data gradient, SNRLP loss, Adam).  All are
// coalesced 4/16-byte-per-lane passes; none carries reuse worth LDS tiling.
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

constexpr int ZC = 32;   // padded channel count of the front-end feature tensor

// one thread per (b, t, fp) of the padded frequency axis
template <int M>
__global__ __launch_bounds__(256) void features_kernel(const float* __restrict__ spec, int64_t ld, float* __restrict__ zp,
                                                       int B, int T, int F) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int FP = F + 2;
  const int64_t total = (int64_t)B * T * FP;
  if (idx >= total) return;
  const unsigned bt = (unsigned)idx / (unsigned)FP;       // 32-bit index split (launcher: total < 2^31)
  const int fp = (int)((unsigned)idx - bt * FP);
  const int b = (int)(bt / (unsigned)T);
  const int t = (int)(bt - (unsigned)b * T);
  float* o = zp + (((int64_t)b * (T + 2) + t + 2) * FP + fp) * ZC;
  float v[ZC];
#pragma unroll
  for (int i = 0; i < ZC; ++i) v[i] = 0.f;
  if (fp >= 1 && fp <= F) {
    const int f = fp - 1;
    float re[M], im[M], mag[M];
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const float* s = spec + ((int64_t)(b * M + m) * T + t) * ld;
      re[m] = s[f];
      im[m] = s[F + f];
      mag[m] = sqrtf(re[m] * re[m] + im[m] * im[m]);
      v[m] = re[m];
      v[M + m] = im[m];
    }
#pragma unroll
    for (int m = 1; m < M; ++m) {
      v[2 * M + m - 1] = log10f((mag[m] + 1e-6f) / (mag[0] + 1e-6f));
      const float den = mag[m] * mag[0] + 1e-6f;
      v[3 * M - 1 + 2 * (m - 1)] = (re[0] * im[m] - im[0] * re[m]) / den;       // sin
      v[3 * M - 1 + 2 * (m - 1) + 1] = (re[m] * re[0] + im[m] * im[0]) / den;   // cos
    }
  }
#pragma unroll
  for (int i = 0; i < ZC; i += 4) {
    f32x4 x = {v[i], v[i + 1], v[i + 2], v[i + 3]};
    st4(o + i, x);
  }
}

__global__ __launch_bounds__(256) void film_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int B,
                                                       int T, int F, int C) {
  const int64_t i4 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t fc4 = (int64_t)F * C / 4;
  const int64_t total = (int64_t)B * T * fc4;
  if (i4 >= total) return;
  const int64_t b = (unsigned)i4 / (unsigned)(T * fc4);      // 32-bit index split (launcher: total < 2^31)
  const int64_t r = (unsigned)i4 % (unsigned)fc4;
  const f32x4 xv = ld4(x + i4 * 4), wv = ld4(w + (b * fc4 + r) * 4), bv = ld4(bias + (b * fc4 + r) * 4);
  st4(y + i4 * 4, xv * wv + bv);
}

// thread = (b, f, c4); grid.y splits T; dw/dbias accumulated with atomics (pre-zeroed)
__global__ __launch_bounds__(256) void film_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ dy, float* __restrict__ dx,
                                                       float* __restrict__ dw, float* __restrict__ dbias, int B, int T,
                                                       int F, int C, int tchunk, float* __restrict__ absmax_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t fc4 = (int64_t)F * C / 4;
  float amax = 0.f;
  if (i < B * fc4) {
    const int64_t b = i / fc4, r = i % fc4;
    const f32x4 wv = ld4(w + i * 4);
    f32x4 aw = zero4(), ab = zero4();
    const int t0 = blockIdx.y * tchunk, t1 = min(T, t0 + tchunk);
    for (int t = t0; t < t1; ++t) {
      const int64_t off = ((b * T + t) * fc4 + r) * 4;
      const f32x4 g = ld4(dy + off), xv = ld4(x + off);
      const f32x4 d = g * wv;
      st4(dx + off, d);
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(d[0]), fabsf(d[1]))), fmaxf(fabsf(d[2]), fabsf(d[3])));
      aw += g * xv;
      ab += g;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(dw + i * 4 + k, aw[k]);
      atomicAdd(dbias + i * 4 + k, ab[k]);
    }
  }
  if (absmax_out) {                                  // max |dx|: one atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned*>(absmax_out),
                                    __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
  }
}

// LayerNorm backward of a block's intra-frame pass + the FiLM backward of the block in front of it, in ONE pass over [B, T, F, C]
// (C = 32; round 4).  Separately they were ln_bwd_kernel<32> (du planes, x, residual in; dx out: 640 B per position)
// and film_bwd_kernel (dx, y_pre in; dx * w out: 384 B per position) with dx written and read back in between; fused, dx never
// leaves the registers: 768 B per position.  Thread = (b, f, channel quad) as in film_bwd_kernel -- the 8 threads of a position are
// consecutive, so the LayerNorm sums are three DPP steps -- walking a chunk of time steps with its FiLM sums in registers; the
// LayerNorm parameter sums leave as one partial row per workgroup (x, y: grid), the FiLM sums by atomics like film_bwd_kernel's.
//   g = du[p, 0, :] + du[p, 1, :];  dx = LN-bwd(g; xin, gamma) + res;  out = dx * w[b, f, :];
//   dw[b, f, :] += sum_t dx * fx;  dbias[b, f, :] += sum_t dx;  partials[wg] = (sum g * xhat [32], sum g [32])
__global__ __launch_bounds__(256) void ln_film_bwd_kernel(const float* __restrict__ du, const float* __restrict__ xin,
                                                          const float* __restrict__ ln_g, const float* __restrict__ res,
                                                          const float* __restrict__ fx, const float* __restrict__ w,
                                                          float* __restrict__ out, float* __restrict__ dw,
                                                          float* __restrict__ dbias, float* __restrict__ partials, int B, int T,
                                                          int F, int tchunk, float* __restrict__ absmax_out) {
  constexpr int C = 32;
  const int tid = threadIdx.x, c4 = tid & 7;
  const int64_t fc4 = (int64_t)F * (C / 4);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + tid;
  const bool live = i < B * fc4;                     // (8 threads of a position are live or dead together: fc4 is a multiple of 8)
  const int64_t ii = live ? i : 0;
  const int64_t b = ii / fc4, r = ii % fc4;
  const f32x4 wv = ld4(w + ii * 4), gam = ld4(ln_g + 4 * c4);
  f32x4 aw = zero4(), ab = zero4(), dgam = zero4(), dbet = zero4();
  float amax = 0.f;
  const int t0 = blockIdx.y * tchunk, t1 = min(T, t0 + tchunk);
  // (the five rows of step t + 1 are requested before step t is worked on: the LayerNorm sums are dependent DPP chains)
  auto row_off = [&](int t) -> int64_t { return ((b * T + min(t, t1 - 1)) * fc4 + r) * 4; };
  int64_t off = row_off(t0);
  // the five input streams are read exactly once: non-temporal loads (no L2 / MALL allocation) -- isolated 285 -> 257 us, in the
  // train step 0.228 -> 0.209 ms (scripts/exp_ln_film.py; a non-temporal store of `out` and other time chunks gave nothing more)
  auto lds_ = [&](const float* p) -> f32x4 { return ld4_once(p); };
  f32x4 ng0 = lds_(du + ((off >> 5) * 2) * C + 4 * c4), ng1 = lds_(du + ((off >> 5) * 2 + 1) * C + 4 * c4);
  f32x4 nx = lds_(xin + off), nrs = lds_(res + off), nfv = lds_(fx + off);
  for (int t = t0; t < t1; ++t) {
    const f32x4 g0 = ng0, g1 = ng1, x = nx, rs = nrs, fv = nfv;
    const int64_t offn = row_off(t + 1);
    ng0 = lds_(du + ((offn >> 5) * 2) * C + 4 * c4); ng1 = lds_(du + ((offn >> 5) * 2 + 1) * C + 4 * c4);
    nx = lds_(xin + offn); nrs = lds_(res + offn); nfv = lds_(fx + offn);
    const f32x4 g = g0 + g1;
    const float mean = row8_sum((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / C);
    f32x4 d, xh, gg, dx;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) { d[v] = x[v] - mean; sq += d[v] * d[v]; }
    const float rstd = 1.0f / sqrtf(row8_sum(sq) * (1.0f / C) + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      xh[v] = d[v] * rstd;
      gg[v] = g[v] * gam[v];
      s1 += gg[v];
      s2 += gg[v] * xh[v];
    }
    const float m1 = row8_sum(s1) * (1.0f / C), m2 = row8_sum(s2) * (1.0f / C);
#pragma unroll
    for (int v = 0; v < 4; ++v) dx[v] = rstd * (gg[v] - m1 - xh[v] * m2) + rs[v];
    if (live) {
      const f32x4 o = dx * wv;
      st4(out + off, o);
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
      aw += dx * fv;
      ab += dx;
      dgam += g * xh;
      dbet += g;
    }
    off = offn;
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      atomicAdd(dw + i * 4 + k, aw[k]);
      atomicAdd(dbias + i * 4 + k, ab[k]);
    }
  }
  // LayerNorm parameter sums: the 32 threads of a wave... 32 positions x 8 channel quads per workgroup -> [64] per workgroup
  __shared__ float red[32][2 * C + 1];
  const int rowi = tid >> 3;
#pragma unroll
  for (int v = 0; v < 4; ++v) { red[rowi][4 * c4 + v] = dgam[v]; red[rowi][C + 4 * c4 + v] = dbet[v]; }
  __syncthreads();
  if (tid < 2 * C) {
    float s = 0.f;
    for (int q = 0; q < 32; ++q) s += red[q][tid];
    partials[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (2 * C) + tid] = s;
  }
  if (absmax_out) {                                  // max |out|: one atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    __shared__ float wm[4];
    if ((tid & 63) == 0) wm[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) atomicMax(reinterpret_cast<unsigned*>(absmax_out), __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
  }
}

__global__ __launch_bounds__(256) void overlap_add_kernel(const float* __restrict__ frames, float* __restrict__ wave,
                                                          int B, int T, int win, int hop) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t L = (int64_t)hop * T;
  if (i >= B * L) return;
  const int64_t b = i / L, n = i % L;
  const int64_t m = n + hop;
  const int t1 = (int)(m / hop), j1 = (int)(m % hop);
  const float* fb = frames + b * (int64_t)(T + 1) * win;
  float v = fb[(int64_t)t1 * win + j1];
  if (j1 < win - hop) v += fb[(int64_t)(t1 - 1) * win + j1 + hop];
  wave[i] = v;
}

__global__ __launch_bounds__(256) void overlap_add_bwd_kernel(const float* __restrict__ dwave,
                                                              float* __restrict__ dframes, int B, int T, int win,
                                                              int hop) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t per = (int64_t)(T + 1) * win;
  if (i >= B * per) return;
  const int64_t b = i / per, r = i % per;
  const int t = (int)(r / win), j = (int)(r % win);
  const int64_t m = (int64_t)hop * t + j;
  const int64_t L = (int64_t)hop * T;
  dframes[i] = (m >= hop && m < hop + L) ? dwave[b * L + m - hop] : 0.f;
}

// dy[b,t,f,c] = sum_{a,d,o} dspec[b, t+2-a, f+1-d, o] * W[c, o, 2-a, 2-d]   (valid t', f' only)
__global__ __launch_bounds__(256) void deconv_bwd_data_kernel(const float* __restrict__ dspec,
                                                              const float* __restrict__ w, float* __restrict__ dy,
                                                              int B, int T, int F, int C, float* __restrict__ absmax_out) {
  extern __shared__ float ws[];   // [C][18]
  for (int i = threadIdx.x; i < C * 18; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  // one thread per (position, 4 channels): the 9 x 2 gradient taps of a position serve all its channels, and the
  // index split is three 32-bit divisions per thread instead of four 64-bit ones per element (the launcher checks
  // that the element count fits 31 bits) -- the kernel was bound by that integer arithmetic, not by its 4 B/element
  const unsigned i4 = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned c4n = (unsigned)C / 4, total = (unsigned)B * T * F * c4n;
  const bool live = i4 < total;                      // (no early return: the wave reduces max |dy| below)
  const unsigned p = (live ? i4 : 0u) / c4n, c = ((live ? i4 : 0u) - p * c4n) * 4;
  const unsigned bt = p / (unsigned)F;
  const int f = (int)(p - bt * F);
  const unsigned b = bt / (unsigned)T;
  const int t = (int)(bt - b * T);
  f32x4 acc = zero4();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int tt = t + 2 - a;
    if (tt < 0 || tt >= T) continue;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int ff = f + 1 - d;
      if (ff < 0 || ff >= F) continue;
      const float2 g = *reinterpret_cast<const float2*>(dspec + (((int64_t)b * T + tt) * F + ff) * 2);
      // W[c][o][kt=2-a][kf=2-d]
      const int tap = (2 - a) * 3 + (2 - d);
#pragma unroll
      for (int k = 0; k < 4; ++k)
        acc[k] = __builtin_fmaf(g.x, ws[(c + k) * 18 + tap], __builtin_fmaf(g.y, ws[(c + k) * 18 + 9 + tap], acc[k]));
    }
  }
  if (live) st4(dy + (int64_t)i4 * 4, acc);
  if (absmax_out) {                                  // max |dy| (the fp16 scale of the backward recurrence that reads it): one atomic per wave
    float mx = live ? fmaxf(fmaxf(fabsf(acc[0]), fabsf(acc[1])), fmaxf(fabsf(acc[2]), fabsf(acc[3]))) : 0.f;
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    // 45 000 workgroups: an atomic per wave on ONE address serialises the whole kernel (measured 110 us -> 2 ms).  The word only
    // ever grows, so a (possibly stale) plain read filters out every wave that cannot raise it -- all but a handful
    if ((threadIdx.x & 63) == 0) {
      const unsigned bits = __float_as_uint(mx);
      if (bits > __hip_atomic_load(reinterpret_cast<unsigned*>(absmax_out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<unsigned*>(absmax_out), bits);
    }
  }
}

// ---------------- SNRLP loss ----------------
// stats[b*8 + k]: 0 sum e, 1 sum t, 2 max|t| (as uint bits), 3 sum t'^2, 4 sum (e'-t')^2, 5 sum |e-t|
// SNRLP loss, stats layout [B, 12]: 0 sum e, 1 sum t, 2 max |t| (bit pattern), 3 S_tt, 4 S_dd = sum ((e - me) - (t - mt))^2,
// 5 sum |e - t|, 6 S_et = sum (e - me)(t - mt), 7 is_negative, 8 grad coefficient of (e - me), 9 grad coefficient of (t - mt),
// 10 S_nn = sum ((e - me) - alpha (t - mt))^2 with alpha = S_et / (S_tt + EPS) (third pass, only for the modes with an 'sisdr' term:
// formed from the moments it is S_ee - 2 alpha S_et + alpha^2 S_tt, which cancels to nothing at high SI-SDR)
constexpr int kLs = 12;
__global__ __launch_bounds__(256) void loss_pass1_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                         int64_t N, float* __restrict__ stats) {
  const int b = blockIdx.y;
  const float* e = est + (int64_t)b * N;
  const float* t = gt + (int64_t)b * N;
  float se = 0.f, st = 0.f, mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    se += e[i]; st += t[i]; mx = fmaxf(mx, fabsf(t[i]));
  }
  se = wave_sum(se); st = wave_sum(st);
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(stats + b * kLs + 0, se);
    atomicAdd(stats + b * kLs + 1, st);
    atomicMax(reinterpret_cast<unsigned int*>(stats + b * kLs + 2), __float_as_uint(mx));
  }
}
__global__ __launch_bounds__(256) void loss_pass2_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                         int64_t N, float* __restrict__ stats) {
  const int b = blockIdx.y;
  const float* e = est + (int64_t)b * N;
  const float* t = gt + (int64_t)b * N;
  const float me = stats[b * kLs + 0] / (float)N, mt = stats[b * kLs + 1] / (float)N;
  float s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float tt = t[i] - mt, ee = e[i] - me, d = ee - tt;
    s3 += tt * tt; s4 += d * d; s5 += fabsf(e[i] - t[i]); s6 += ee * tt;
  }
  s3 = wave_sum(s3); s4 = wave_sum(s4); s5 = wave_sum(s5); s6 = wave_sum(s6);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(stats + b * kLs + 3, s3);
    atomicAdd(stats + b * kLs + 4, s4);
    atomicAdd(stats + b * kLs + 5, s5);
    atomicAdd(stats + b * kLs + 6, s6);
  }
}
__global__ __launch_bounds__(256) void loss_pass3_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                         int64_t N, float* __restrict__ stats) {
  const int b = blockIdx.y;
  const float* e = est + (int64_t)b * N;
  const float* t = gt + (int64_t)b * N;
  const float me = stats[b * kLs + 0] / (float)N, mt = stats[b * kLs + 1] / (float)N;
  const float al = stats[b * kLs + 6] / (stats[b * kLs + 3] + 1e-8f);
  float s = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float d = (e[i] - me) - al * (t[i] - mt);
    s += d * d;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) atomicAdd(stats + b * kLs + 10, s);
}
// asteroid SingleSrcNegSDR (third-party, restated: zero-mean signals, EPS = 1e-8 in the denominators and inside the log) as a
// function of the zero-mean moments; returns the loss and its derivatives w.r.t. S_ee and S_et (S_tt does not depend on est).
//   kind 0 'snr':   target / (est - target);  1 'sisdr': scaled target / (est - scaled target);  2 'sdsdr': scaled target / (est - target)
struct SdrTerm { float loss, d_see, d_set; };
__device__ inline SdrTerm sdr_term(int kind, float Stt, float Sdd, float Set, float Snn) {
  const float EPS = 1e-8f, k10 = 10.f / 2.302585092994046f;
  SdrTerm r;
  if (kind == 0) {                                   // (the round-1 formulas, kept to the bit)
    const float Sn = Sdd + EPS, R = Stt / Sn;
    r.loss = -10.f * log10f(R + EPS);
    const float c = k10 / (R + EPS) * (Stt / (Sn * Sn));          // dL/dS_dd;  S_dd = S_ee - 2 S_et + S_tt
    r.d_see = c; r.d_set = -2.f * c;
    return r;
  }
  const float da = 1.f / (Stt + EPS), al = Set * da;
  const float num = al * al * Stt, dnum_set = 2.f * al * Stt * da;
  float noise, dn_see = 1.f, dn_set;
  if (kind == 1) { noise = Snn; dn_set = -2.f * al + da * (-2.f * Set + 2.f * al * Stt); }   // = S_ee - 2 al S_et + al^2 S_tt
  else { noise = Sdd; dn_set = -2.f; }
  const float Sn = noise + EPS, R = num / Sn;
  r.loss = -10.f * log10f(R + EPS);
  const float g = -k10 / (R + EPS);                  // dL/dR
  r.d_see = g * (-num / (Sn * Sn) * dn_see);
  r.d_set = g * (dnum_set / Sn - num / (Sn * Sn) * dn_set);
  return r;
}
// max of two terms with torch.maximum's gradient (ties: half each)
__device__ inline SdrTerm sdr_max(SdrTerm x, SdrTerm y) {
  if (x.loss > y.loss) return x;
  if (y.loss > x.loss) return y;
  SdrTerm r = {x.loss, 0.5f * (x.d_see + y.d_see), 0.5f * (x.d_set + y.d_set)};
  return r;
}
// single block: per-sample loss and the two gradient coefficients.  mode (src/losses/SNRLosses.py:10-52): 0 'snr', 1 'sisdr',
// 2 'fused' = (sisdr + snr) / 2, 3 'max_fused' = max(sisdr, snr), 4 'sdsdr' = max(snr, sdsdr), 5 'full' = sisdr / 2 + max(snr, sdsdr) / 2
__global__ void loss_final_kernel(float* __restrict__ stats, int B, int64_t N, float neg_weight, int mode,
                                  float* __restrict__ loss_vec, float* __restrict__ loss_mean) {
  __shared__ float negsum;
  __shared__ int nneg;
  if (threadIdx.x == 0) {
    float s = 0.f; int c = 0;
    for (int b = 0; b < B; ++b)
      if (stats[b * kLs + 2] == 0.f) { s += stats[b * kLs + 5]; ++c; }
    negsum = s; nneg = c;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    const bool neg = stats[b * kLs + 2] == 0.f;
    if (neg) {
      loss_vec[b] = neg_weight * negsum / ((float)nneg * (float)N);
      stats[b * kLs + 8] = 0.f; stats[b * kLs + 9] = 0.f;
      stats[b * kLs + 7] = 1.f;
    } else {
      const float Stt = stats[b * kLs + 3], Sdd = stats[b * kLs + 4], Set = stats[b * kLs + 6], Snn = stats[b * kLs + 10];
      SdrTerm r;
      if (mode == 0) r = sdr_term(0, Stt, Sdd, Set, Snn);
      else if (mode == 1) r = sdr_term(1, Stt, Sdd, Set, Snn);
      else if (mode == 2 || mode == 3) {
        const SdrTerm x = sdr_term(1, Stt, Sdd, Set, Snn), y = sdr_term(0, Stt, Sdd, Set, Snn);
        if (mode == 3) r = sdr_max(x, y);
        else { r.loss = 0.5f * x.loss + 0.5f * y.loss; r.d_see = 0.5f * (x.d_see + y.d_see); r.d_set = 0.5f * (x.d_set + y.d_set); }
      } else {
        r = sdr_max(sdr_term(0, Stt, Sdd, Set, Snn), sdr_term(2, Stt, Sdd, Set, Snn));
        if (mode == 5) {
          const SdrTerm z = sdr_term(1, Stt, Sdd, Set, Snn);
          r.loss = 0.5f * z.loss + 0.5f * r.loss; r.d_see = 0.5f * (z.d_see + r.d_see); r.d_set = 0.5f * (z.d_set + r.d_set);
        }
      }
      loss_vec[b] = r.loss;
      // dL/de_n = 2 dL/dS_ee (e_n - me) + dL/dS_et (t_n - mt)   (both zero-mean: the mean subtraction's own gradient vanishes)
      stats[b * kLs + 8] = 2.f * r.d_see;
      stats[b * kLs + 9] = r.d_set;
      stats[b * kLs + 7] = 0.f;
    }
  }
  if (loss_mean) {                                   // hl_module:321: loss.mean() over the batch, in batch order
    __syncthreads();
    if (threadIdx.x == 0) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += loss_vec[b];
      loss_mean[0] = s / (float)B;
    }
  }
}
__global__ __launch_bounds__(256) void loss_grad_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                        int B, int64_t N, float neg_weight, int mode,
                                                        const float* __restrict__ stats, float* __restrict__ dest,
                                                        const float* __restrict__ gout) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * N) return;
  const int b = (int)(i / N);
  const float invB = (gout ? gout[0] : 1.0f) / (float)B;
  if (stats[b * kLs + 7] != 0.f) {
    const float d = est[i] - gt[i];
    dest[i] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) * neg_weight * invB / (float)N;
  } else {
    const float me = stats[b * kLs + 0] / (float)N, mt = stats[b * kLs + 1] / (float)N;
    if (mode == 0) dest[i] = stats[b * kLs + 8] * ((est[i] - me) - (gt[i] - mt)) * invB;       // (a = -b: the round-1 expression)
    else dest[i] = (stats[b * kLs + 8] * (est[i] - me) + stats[b * kLs + 9] * (gt[i] - mt)) * invB;
  }
}

// per-sample first/second moments of (est, gt, mix): out[b*8 + k] = sum e, t, m, ee, tt, mm, et, mt
__global__ __launch_bounds__(256) void signal_stats_kernel(const float* __restrict__ est, const float* __restrict__ gt,
                                                           const float* __restrict__ mix, int64_t N,
                                                           int64_t mix_stride, float* __restrict__ out) {
  const int b = blockIdx.y;
  const float* e = est + (int64_t)b * N;
  const float* t = gt + (int64_t)b * N;
  const float* m = mix + (int64_t)b * mix_stride;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
    const float ev = e[i], tv = t[i], mv = m[i];
    s[0] += ev; s[1] += tv; s[2] += mv; s[3] += ev * ev; s[4] += tv * tv; s[5] += mv * mv; s[6] += ev * tv;
    s[7] += mv * tv;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + b * 8 + k, v);
  }
}

// Gradient norm for clip_grad_norm_.  DETERMINISTIC (one workgroup, fixed summation tree, no atomics): every
// data-parallel rank must derive bit-identical clip factors from the bit-identical all-reduced bucket, or the replicas
// drift apart in the last bit (found by tests/test_gpu_distributed.py with the atomicAdd version).  The bucket is
// 1-2 MB: one CU reads it in ~10 us.
__global__ __launch_bounds__(1024) void sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ out, int accumulate) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const int64_t n4 = n / 4;
  int64_t i = threadIdx.x;
  for (; i < n4; i += 1024) {
    const f32x4 v = ld4(g + 4 * i);
    s0 = __builtin_fmaf(v[0], v[0], s0); s1 = __builtin_fmaf(v[1], v[1], s1);
    s2 = __builtin_fmaf(v[2], v[2], s2); s3 = __builtin_fmaf(v[3], v[3], s3);
  }
  for (int64_t k = 4 * n4 + threadIdx.x; k < n; k += 1024) s0 = __builtin_fmaf(g[k], g[k], s0);
  float s = wave_sum((s0 + s1) + (s2 + s3));
  __shared__ float ws[16];
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += ws[k];
    out[0] = accumulate ? out[0] + t : t;
  }
}

// max |x| (bit pattern of a non-negative float orders like an unsigned integer)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t n4, unsigned* __restrict__ out) {
  float m = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {               // four independent 16-byte loads in flight
    const f32x4 v0 = ld4(x + 4 * i), v1 = ld4(x + 4 * (i + stride)), v2 = ld4(x + 4 * (i + 2 * stride)),
                v3 = ld4(x + 4 * (i + 3 * stride));
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(fmaxf(m, fmaxf(fabsf(v0[r]), fabsf(v1[r]))), fmaxf(fabsf(v2[r]), fabsf(v3[r])));
  }
  for (; i < n4; i += stride) {
    const f32x4 v = ld4(x + 4 * i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)                      // one atomic per workgroup: same-address atomics serialise in L2
    atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr,
                                                   float b1, float b2, float eps, float bc1, float bc2, float gscale,
                                                   float clip, const float* __restrict__ sumsq, const int* __restrict__ guard,
                                                   int* __restrict__ skipped) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // guard: the watchdog word of the guarded schedules (sched_status).  Non-zero = a bounded wait of this step's launches gave
  // up, the gradients in g are garbage: the update is a no-op (parameters and moments keep their last good values; the host
  // finds out at its next look at the word, and `skipped` counts what that look will report)
  if (guard && __hip_atomic_load(guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
    if (i == 0 && skipped) atomicAdd(skipped, 1);
    return;
  }
  if (i >= n) return;
  float sc = gscale;
  if (clip > 0.f) {
    const float norm = sqrtf(sumsq[0]) * gscale;
    const float coef = clip / (norm + 1e-6f);
    sc *= coef < 1.f ? coef : 1.f;
  }
  const float gi = g[i] * sc;
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  p[i] -= (lr / bc1) * (mi / denom);
}

// y = x + part[:, 0] + part[:, 1]: thread = 4 channels of one position
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ x, const float* __restrict__ part,
                                                   float* __restrict__ y, int64_t n4, int c4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / c4, c = i - p * c4;
    const f32x4 a = ld4(x + 4 * i), b0 = ld4(part + 4 * ((2 * p) * c4 + c)), b1 = ld4(part + 4 * ((2 * p + 1) * c4 + c));
    st4(y + 4 * i, (a + b0) + b1);
  }
}

// out[r, f, :] = in[r, f, :] (+ bias) for the tail frequencies f in [Fm, F) of every row r (= (b, t)): the residual of the
// conv-LSTM intra path beyond down * floor(F / down) frequencies, and its backward (bias == NULL)
__global__ __launch_bounds__(256) void tail_rows_kernel(const float* __restrict__ in, const float* __restrict__ bias,
                                                        float* __restrict__ out, int64_t rows, int F, int Fm, int C) {
  const int tw = (F - Fm) * C;
  const int64_t total = rows * tw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / tw;
    const int e = (int)(i - r * tw);
    const int64_t off = (r * F + Fm) * C + e;
    const float v = in[off];
    out[off] = bias ? v + bias[e % C] : v;
  }
}

// ---- staging of the two 3x3 convolutions' inputs and of the iSTFT's spectrum rows (round 6: was ATen fills / strided copies) ----
// dst [B, Tp, F + 2, Cd] channels-last, zero frequency borders.  Frame rows 0, 1 <- the carried state [B, Cs, 2, F] (channels
// >= Cs zero); frame rows 2 .. <- src [B, Tp - 2, F, Cd] (src != NULL; the output convolution's input), borders zeroed.
__global__ __launch_bounds__(256) void stage_frames_kernel(const float* __restrict__ state, const float* __restrict__ src,
                                                           float* __restrict__ dst, int B, int Tp, int F, int Cs, int Cd) {
  const int64_t nrow = src ? (int64_t)B * Tp : (int64_t)B * 2;           // (b, frame row) pairs this launch writes
  const int rw = (F + 2) * Cd;
  const int64_t total = nrow * rw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / rw;
    const int e = (int)(i - row * rw);
    const int fp = e / Cd, c = e - fp * Cd;
    const int b = (int)(src ? row / Tp : row / 2), r = (int)(src ? row % Tp : row % 2);
    float v = 0.f;
    if (fp >= 1 && fp <= F) {
      if (r < 2) { if (c < Cs) v = state[(((int64_t)b * Cs + c) * 2 + r) * F + (fp - 1)]; }
      else v = src[(((int64_t)b * (Tp - 2) + (r - 2)) * F + (fp - 1)) * Cd + c];
    }
    dst[((int64_t)b * Tp + r) * rw + e] = v;
  }
}
// new_state [B, Cs, 2, F] <- frame rows r0, r0 + 1 of rows [B, Tp, F + 2, Cd] (interior, channels < Cs)
__global__ __launch_bounds__(256) void frames_to_state_kernel(const float* __restrict__ rows, float* __restrict__ state, int B,
                                                              int Tp, int F, int Cs, int Cd, int r0) {
  const int64_t total = (int64_t)B * Cs * 2 * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int f = (int)(i % F), r = (int)((i / F) % 2), c = (int)((i / (2 * F)) % Cs), b = (int)(i / ((int64_t)2 * F * Cs));
    state[i] = rows[(((int64_t)b * Tp + r0 + r) * (F + 2) + f + 1) * Cd + c];
  }
}
// spectrum rows [B, T + 1, ld] (interleaved re / im per frequency, columns 2F .. ld - 1 padding): mode 0 -- zero the padding
// columns of every row and fill row 0 from the carried istft_buf [B, 2, F] (re | im); mode 1 -- new istft_buf <- row T
__global__ __launch_bounds__(256) void spec_rows_kernel(float* __restrict__ rows, float* __restrict__ buf, int B, int T, int F,
                                                        int ld, int mode) {
  if (mode == 1) {
    const int64_t total = (int64_t)B * 2 * F;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
      const int f = (int)(i % F), o = (int)((i / F) % 2), b = (int)(i / (2 * F));
      buf[i] = rows[((int64_t)b * (T + 1) + T) * ld + 2 * f + o];
    }
    return;
  }
  const int pad = ld - 2 * F;
  const int64_t npad = (int64_t)B * (T + 1) * pad, nrow0 = (int64_t)B * 2 * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npad + nrow0; i += (int64_t)gridDim.x * blockDim.x) {
    if (i < npad) {
      const int64_t row = i / pad;
      rows[row * ld + 2 * F + (i - row * pad)] = 0.f;
    } else {
      const int64_t k = i - npad;
      const int f = (int)(k % F), o = (int)((k / F) % 2), b = (int)(k / (2 * F));
      rows[(int64_t)b * (T + 1) * ld + 2 * f + o] = buf[k];
    }
  }
}

inline unsigned nblk(int64_t n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

// several small dense copies in one launch (workgroup row y = job y): the streaming chunk step's state write-back
__global__ __launch_bounds__(256) void multi_copy_kernel(sb_multi_copy_args a) {
  const int jb = blockIdx.y;
  const float* __restrict__ s = a.src[jb];
  float* __restrict__ d = a.dst[jb];
  const int64_t n = a.n[jb];
  const bool al = ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
  const int64_t n4 = al ? n / 4 : 0;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  for (int64_t i = i0; i < n4; i += stride) st4(d + 4 * i, ld4(s + 4 * i));
  for (int64_t i = 4 * n4 + i0; i < n; i += stride) d[i] = s[i];
}

// ---- distance embedding -> FiLM plane bank (Dis_Embed_Conv, tfgridnet_causal.py:150-173, and the two Conv1d(4 -> C, k = 1) of
// every FilmLayer, :51-68, applied once per forward for all layers, :509-513) -------------------------------------------------
// e[b, f, :] = LN_DIN(W_e[DIN f .. DIN f + DIN - 1, :] . dis[b, :])  (DIN = 1 / 2 / 4 / 8 for dis_type conv1 .. conv4);  planes[j, b, f, c] = conv_b[j][c] + sum_i conv_w[j][c, i] e[b, f, i],
// j = 2 layer + (0 scale | 1 shift).  A few hundred KB of arithmetic: ONE launch forward, two backward (main + partial rows).
constexpr int kFbRows = 8;          // (b, f) rows per workgroup, forward
template <int DIN>
__global__ __launch_bounds__(256) void film_bank_fwd_kernel(sb_film_bank_args a) {
  __shared__ float e_s[kFbRows][DIN];
  const int BF = a.B * a.F, r0 = blockIdx.x * kFbRows, tid = threadIdx.x;
  if (tid < kFbRows && r0 + tid < BF) {
    const int b = (r0 + tid) / a.F, f = (r0 + tid) % a.F;
    float v[DIN], mean = 0.f;
#pragma unroll
    for (int i = 0; i < DIN; ++i) {
      float s = 0.f;
      for (int k = 0; k < a.K; ++k) s = __builtin_fmaf(a.W_e[(DIN * f + i) * a.K + k], a.dis[b * a.K + k], s);
      v[i] = s;
      mean += s;
    }
    mean *= 1.0f / DIN;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < DIN; ++i) { v[i] -= mean; sq += v[i] * v[i]; }
    const float rstd = 1.0f / sqrtf(sq * (1.0f / DIN) + 1e-5f);
#pragma unroll
    for (int i = 0; i < DIN; ++i) e_s[tid][i] = v[i] * rstd * a.ln_w[i] + a.ln_b[i];
  }
  __syncthreads();
  const int C = a.C, nj = 2 * a.n, per = kFbRows * C;
  for (int o = tid; o < nj * per; o += 256) {
    const int j = o / per, r = (o % per) / C, c = o % C;
    if (r0 + r >= BF) continue;
    const float* w = a.conv_w[j] + DIN * c;
    float y = a.conv_b[j][c];
#pragma unroll
    for (int i = 0; i < DIN; ++i) y = __builtin_fmaf(w[i], e_s[r][i], y);
    a.planes[((int64_t)j * BF + r0 + r) * C + c] = y;
  }
}

// backward: one workgroup per frequency bin f (it owns rows 4f .. 4f+3 of dW_e outright), thread = (plane j, channel c) over
// the batch; the sums over (b, f) of the conv / LayerNorm parameter gradients leave as one partial row per workgroup:
// [j][c][DIN] weights, [j][c] biases, d(ln_w)[DIN], d(ln_b)[DIN] -- reduced in fixed order by film_bank_reduce_kernel (deterministic)
constexpr int kFbMaxB = 64;
template <int DIN>
__global__ __launch_bounds__(1024) void film_bank_bwd_kernel(sb_film_bank_args a) {
  __shared__ float e_s[kFbMaxB][DIN], xh_s[kFbMaxB][DIN], rstd_s[kFbMaxB], de_s[kFbMaxB][DIN], red_s[16][DIN], dE_s[kFbMaxB][DIN];
  const int f = blockIdx.x, tid = threadIdx.x, C = a.C, nj = 2 * a.n, njc = nj * C, BF = a.B * a.F;
  const int lane = tid & 63, wave = tid >> 6, nwaves = (blockDim.x + 63) >> 6;
  const int j = tid / C, c = tid % C;
  const bool act = tid < njc;
  float w4[DIN], accw[DIN], accb = 0.f, dlw = 0.f, dlb = 0.f;
#pragma unroll
  for (int i = 0; i < DIN; ++i) { w4[i] = act ? a.conv_w[j][DIN * c + i] : 0.f; accw[i] = 0.f; }
  for (int b0 = 0; b0 < a.B; b0 += kFbMaxB) {
    const int nb = min(kFbMaxB, a.B - b0);
    __syncthreads();
    if (tid < nb) {                                  // the forward's e, xhat, rstd of row (b0 + tid, f)
      const int b = b0 + tid;
      float v[DIN], mean = 0.f;
#pragma unroll
      for (int i = 0; i < DIN; ++i) {
        float s = 0.f;
        for (int k = 0; k < a.K; ++k) s = __builtin_fmaf(a.W_e[(DIN * f + i) * a.K + k], a.dis[b * a.K + k], s);
        v[i] = s;
        mean += s;
      }
      mean *= 1.0f / DIN;
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < DIN; ++i) { v[i] -= mean; sq += v[i] * v[i]; }
      const float rstd = 1.0f / sqrtf(sq * (1.0f / DIN) + 1e-5f);
      rstd_s[tid] = rstd;
#pragma unroll
      for (int i = 0; i < DIN; ++i) { xh_s[tid][i] = v[i] * rstd; e_s[tid][i] = v[i] * rstd * a.ln_w[i] + a.ln_b[i]; }
    }
    __syncthreads();
    for (int bb = 0; bb < nb; ++bb) {
      const float g = act ? a.G[((int64_t)j * BF + (int64_t)(b0 + bb) * a.F + f) * C + c] : 0.f;
      accb += g;
      float p[DIN];
#pragma unroll
      for (int i = 0; i < DIN; ++i) { accw[i] = __builtin_fmaf(g, e_s[bb][i], accw[i]); p[i] = g * w4[i]; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int i = 0; i < DIN; ++i) p[i] += __shfl_xor(p[i], o, 64);
      }
      if (lane == 0) {
#pragma unroll
        for (int i = 0; i < DIN; ++i) red_s[wave][i] = p[i];
      }
      __syncthreads();
      if (tid < DIN) {
        float s = 0.f;
        for (int w = 0; w < nwaves; ++w) s += red_s[w][tid];
        de_s[bb][tid] = s;
      }
      __syncthreads();
    }
    if (tid < nb) {                                  // LayerNorm(DIN) backward of row (b0 + tid, f)
      float dxh[DIN], m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < DIN; ++i) {
        dxh[i] = de_s[tid][i] * a.ln_w[i];
        m1 += dxh[i];
        m2 += dxh[i] * xh_s[tid][i];
      }
      m1 *= 1.0f / DIN; m2 *= 1.0f / DIN;
#pragma unroll
      for (int i = 0; i < DIN; ++i) dE_s[tid][i] = rstd_s[tid] * (dxh[i] - m1 - xh_s[tid][i] * m2);
    }
    __syncthreads();
    if (tid < DIN) {                                 // d(ln_w), d(ln_b): this workgroup's share, batch order
      for (int bb = 0; bb < nb; ++bb) { dlw += de_s[bb][tid] * xh_s[bb][tid]; dlb += de_s[bb][tid]; }
    }
    if (tid >= 64 && tid < 64 + DIN * a.K) {         // dW_e[DIN f + i, k] += sum_b dE0[b, f, i] dis[b, k]: this workgroup owns the rows
      const int i = (tid - 64) / a.K, k = (tid - 64) % a.K;
      float s = 0.f;
      for (int bb = 0; bb < nb; ++bb) s = __builtin_fmaf(dE_s[bb][i], a.dis[(b0 + bb) * a.K + k], s);
      a.dW_e[(DIN * f + i) * a.K + k] += s;
    }
  }
  float* row = a.partials + (int64_t)f * (njc * (DIN + 1) + 2 * DIN);
  if (act) {
#pragma unroll
    for (int i = 0; i < DIN; ++i) row[(j * C + c) * DIN + i] = accw[i];
    row[njc * DIN + j * C + c] = accb;
  }
  if (tid < DIN) { row[njc * (DIN + 1) + tid] = dlw; row[njc * (DIN + 1) + DIN + tid] = dlb; }
}

// out[col] += sum over the F partial rows, fixed order; column -> (parameter tensor, element) through the pointer tables
template <int DIN>
__global__ __launch_bounds__(256) void film_bank_reduce_kernel(sb_film_bank_args a) {
  const int C = a.C, njc = 2 * a.n * C, ld = njc * (DIN + 1) + 2 * DIN;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ld) return;
  float s = 0.f;
  for (int r = 0; r < a.F; ++r) s += a.partials[(int64_t)r * ld + col];
  if (col < njc * DIN) a.d_conv_w[col / (DIN * C)][col % (DIN * C)] += s;
  else if (col < njc * (DIN + 1)) a.d_conv_b[(col - njc * DIN) / C][(col - njc * DIN) % C] += s;
  else if (col < njc * (DIN + 1) + DIN) a.d_ln_w[col - njc * (DIN + 1)] += s;
  else a.d_ln_b[col - njc * (DIN + 1) - DIN] += s;
}

}  // namespace

extern "C" int sb_features(const float* spec, int64_t ld_spec, float* zp, int B, int M, int T, int F, void* stream) {
  const int64_t total = (int64_t)B * T * (F + 2);
  if (total <= 0 || total >= (1ll << 31)) return -1002;
  // 5 M - 3 feature channels in the ZC = 32 padded stack: 2 <= M <= 7 (every shipped JSON: 6; the reference's constructor default: 2)
#define SB_FEAT(M_) case M_: hipLaunchKernelGGL(features_kernel<M_>, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, spec, ld_spec, zp, B, T, F); break
  switch (M) {
    SB_FEAT(2); SB_FEAT(3); SB_FEAT(4); SB_FEAT(5); SB_FEAT(6); SB_FEAT(7);
    default: return -1002;
  }
#undef SB_FEAT
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_film_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int F, int C,
                           void* stream) {
  const int64_t total = (int64_t)B * T * F * C / 4;
  if (total <= 0 || total >= (1ll << 31)) return -1002;
  hipLaunchKernelGGL(film_fwd_kernel, dim3(nblk(total)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, T, F, C);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_film_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dbias, int B,
                           int T, int F, int C, float* absmax_out, void* stream) {
  const int64_t n = (int64_t)B * F * C / 4;
  const int tchunk = 25;
  dim3 grid(nblk(n), (T + tchunk - 1) / tchunk);
  hipLaunchKernelGGL(film_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, dy, dx, dw, dbias, B, T, F, C, tchunk, absmax_out);
  SB_CHECK_LAUNCH();
  return 0;
}

static int film_bank_check(const sb_film_bank_args* a) {
  if (!a || !a->dis || !a->W_e || !a->ln_w || !a->ln_b) return -1001;
  if (a->B <= 0 || a->F <= 0 || a->C <= 0 || a->n <= 0 || a->K <= 0) return -1001;
  if (a->n > SB_FILM_BANK_MAX_LAYERS || a->K > 8 || 2 * a->n * a->C > 1024 ||
      (a->d_in != 1 && a->d_in != 2 && a->d_in != 4 && a->d_in != 8))
    return -1002;
  for (int j = 0; j < 2 * a->n; ++j) if (!a->conv_w[j] || !a->conv_b[j]) return -1001;
  return 0;
}
extern "C" int sb_film_bank_fwd(const sb_film_bank_args* a, void* stream) {
  if (int rc = film_bank_check(a)) return rc;
  if (!a->planes) return -1001;
  const int BF = a->B * a->F;
#define SB_FB(D) hipLaunchKernelGGL(film_bank_fwd_kernel<D>, dim3((BF + kFbRows - 1) / kFbRows), dim3(256), 0, (hipStream_t)stream, *a)
  if (a->d_in == 4) SB_FB(4); else if (a->d_in == 8) SB_FB(8); else if (a->d_in == 2) SB_FB(2); else SB_FB(1);
#undef SB_FB
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_film_bank_bwd_scratch(int F, int C, int n, int d_in) { return F * (2 * n * C * (d_in + 1) + 2 * d_in); }
extern "C" int sb_film_bank_bwd(const sb_film_bank_args* a, void* stream) {
  if (int rc = film_bank_check(a)) return rc;
  if (!a->G || !a->dW_e || !a->d_ln_w || !a->d_ln_b || !a->partials) return -1001;
  for (int j = 0; j < 2 * a->n; ++j) if (!a->d_conv_w[j] || !a->d_conv_b[j]) return -1001;
  const int njc = 2 * a->n * a->C;
  const int bs = (njc + 63) / 64 * 64 < 128 ? 128 : (njc + 63) / 64 * 64;
#define SB_FB(D) do { \
    hipLaunchKernelGGL(film_bank_bwd_kernel<D>, dim3(a->F), dim3(bs), 0, (hipStream_t)stream, *a); \
    hipLaunchKernelGGL(film_bank_reduce_kernel<D>, dim3((njc * (D + 1) + 2 * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a); } while (0)
  if (a->d_in == 4) SB_FB(4); else if (a->d_in == 8) SB_FB(8); else if (a->d_in == 2) SB_FB(2); else SB_FB(1);
#undef SB_FB
  SB_CHECK_LAUNCH();
  return 0;
}

constexpr int kLnFilmChunk = 25;
extern "C" int sb_ln_film_bwd_rows(int B, int T, int F) {
  return (int)(nblk((int64_t)B * F * 8) * ((T + kLnFilmChunk - 1) / kLnFilmChunk));
}

extern "C" int sb_ln_film_bwd(const float* du, const float* xin, const float* ln_g, const float* res, const float* film_x,
                              const float* film_w, float* out, float* dw, float* dbias, float* partials, int B, int T, int F,
                              int C, float* absmax_out, void* stream) {
  if (!du || !xin || !ln_g || !res || !film_x || !film_w || !out || !dw || !dbias || !partials || B <= 0 || T <= 0 || F <= 0)
    return -1001;
  if (C != 32 || (int64_t)B * T * F * 8 >= (1ll << 31)) return -1002;
  const int tchunk = kLnFilmChunk;
  dim3 grid(nblk((int64_t)B * F * 8), (T + tchunk - 1) / tchunk);
  hipLaunchKernelGGL(ln_film_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, du, xin, ln_g, res, film_x, film_w, out, dw,
                     dbias, partials, B, T, F, tchunk, absmax_out);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_add3(const float* x, const float* part, float* y, int64_t P, int C, void* stream) {
  if (P <= 0 || C <= 0 || C % 4) return -1002;
  const int64_t n4 = P * (C / 4);
  unsigned gx = nblk(n4);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(add3_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, x, part, y, n4, C / 4);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_tail_rows(const float* in, const float* bias, float* out, int64_t rows, int F, int Fm, int C, void* stream) {
  if (!in || !out || rows <= 0 || C <= 0 || Fm < 0 || Fm >= F) return -1001;
  unsigned gx = nblk(rows * (F - Fm) * C);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(tail_rows_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, in, bias, out, rows, F, Fm, C);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_overlap_add(const float* frames, float* wave, int B, int T, int win, int hop, void* stream) {
  if (win - hop > hop) return -1002;
  hipLaunchKernelGGL(overlap_add_kernel, dim3(nblk((int64_t)B * hop * T)), dim3(256), 0, (hipStream_t)stream, frames, wave, B, T, win, hop);
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_overlap_add_bwd(const float* dwave, float* dframes, int B, int T, int win, int hop, void* stream) {
  hipLaunchKernelGGL(overlap_add_bwd_kernel, dim3(nblk((int64_t)B * (T + 1) * win)), dim3(256), 0, (hipStream_t)stream, dwave, dframes, B, T, win, hop);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_deconv_bwd_data(const float* dspec, const float* w, float* dy, int B, int T, int F, int C,
                                  float* absmax_out, void* stream) {
  const int64_t total = (int64_t)B * T * F * C;
  if (C % 4 || total <= 0 || total >= (1ll << 31)) return -1002;
  hipLaunchKernelGGL(deconv_bwd_data_kernel, dim3(nblk(total / 4)), dim3(256), C * 18 * sizeof(float), (hipStream_t)stream, dspec, w, dy, B, T, F, C, absmax_out);
  SB_CHECK_LAUNCH();
  return 0;
}

static int snrlp_forward(const float* est, const float* gt, int B, int64_t N, float neg_weight, int mode, float* stats,
                         float* loss_vec, float* loss_mean, hipStream_t st) {
  if (!est || !gt || !stats || !loss_vec || B <= 0 || N <= 0) return -1001;
  if (mode < 0 || mode > 5) return -1002;
  (void)hipMemsetAsync(stats, 0, (size_t)B * kLs * sizeof(float), st);
  unsigned gx = nblk(N, 256 * 8);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(loss_pass1_kernel, dim3(gx, B), dim3(256), 0, st, est, gt, N, stats);
  hipLaunchKernelGGL(loss_pass2_kernel, dim3(gx, B), dim3(256), 0, st, est, gt, N, stats);
  if (mode == 1 || mode == 2 || mode == 3 || mode == 5)
    hipLaunchKernelGGL(loss_pass3_kernel, dim3(gx, B), dim3(256), 0, st, est, gt, N, stats);
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(64), 0, st, stats, B, N, neg_weight, mode, loss_vec, loss_mean);
  return 0;
}
extern "C" int sb_snrlp_loss_fwd(const float* est, const float* gt, int B, int64_t N, float neg_weight, int mode, float* stats,
                                 float* loss_vec, float* loss_mean, void* stream) {
  if (int rc = snrlp_forward(est, gt, B, N, neg_weight, mode, stats, loss_vec, loss_mean, (hipStream_t)stream)) return rc;
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_snrlp_loss_bwd(const float* est, const float* gt, int B, int64_t N, float neg_weight, int mode,
                                 const float* stats, const float* gout, float* dest, void* stream) {
  if (!est || !gt || !stats || !dest || B <= 0 || N <= 0) return -1001;
  if (mode < 0 || mode > 5) return -1002;
  hipLaunchKernelGGL(loss_grad_kernel, dim3(nblk((int64_t)B * N)), dim3(256), 0, (hipStream_t)stream, est, gt, B, N, neg_weight,
                     mode, stats, dest, gout);
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_snrlp_loss_ex(const float* est, const float* gt, int B, int64_t N, float neg_weight, int mode, float* stats,
                                float* loss_vec, float* dest, void* stream) {
  if (int rc = snrlp_forward(est, gt, B, N, neg_weight, mode, stats, loss_vec, nullptr, (hipStream_t)stream)) return rc;
  if (dest) hipLaunchKernelGGL(loss_grad_kernel, dim3(nblk((int64_t)B * N)), dim3(256), 0, (hipStream_t)stream, est, gt, B, N,
                               neg_weight, mode, stats, dest, (const float*)nullptr);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_signal_stats(const float* est, const float* gt, const float* mix, int B, int64_t N, int64_t mix_stride,
                               float* out, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(out, 0, (size_t)B * 8 * sizeof(float), st);
  unsigned gx = nblk(N, 256 * 8);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(signal_stats_kernel, dim3(gx, B), dim3(256), 0, st, est, gt, mix, N, mix_stride, out);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_sumsq_ex(const float* g, int64_t n, float* sumsq, int accumulate, void* stream) {
  if (n <= 0 || (reinterpret_cast<uintptr_t>(g) & 15)) return -1002;
  hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, g, n, sumsq, accumulate);
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_sumsq(const float* g, int64_t n, float* sumsq, void* stream) { return sb_sumsq_ex(g, n, sumsq, 1, stream); }

extern "C" int sb_stage_frames(const float* state, const float* src, float* dst, int B, int Tp, int F, int Cs, int Cd, void* stream) {
  if (!state || !dst || B <= 0 || Tp < 2 || F <= 0 || Cs <= 0 || Cd < Cs) return -1001;
  const int64_t total = (src ? (int64_t)B * Tp : (int64_t)B * 2) * (F + 2) * Cd;
  unsigned gx = nblk(total);
  if (gx > 8192) gx = 8192;
  hipLaunchKernelGGL(stage_frames_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, state, src, dst, B, Tp, F, Cs, Cd);
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_frames_to_state(const float* rows, float* state, int B, int Tp, int F, int Cs, int Cd, int r0, void* stream) {
  if (!rows || !state || B <= 0 || r0 < 0 || r0 + 2 > Tp || F <= 0 || Cs <= 0 || Cd < Cs) return -1001;
  unsigned gx = nblk((int64_t)B * Cs * 2 * F);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(frames_to_state_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, rows, state, B, Tp, F, Cs, Cd, r0);
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_spec_rows(float* rows, float* buf, int B, int T, int F, int ld, int mode, void* stream) {
  if (!rows || !buf || B <= 0 || T <= 0 || F <= 0 || ld < 2 * F || (mode != 0 && mode != 1)) return -1001;
  const int64_t total = mode == 1 ? (int64_t)B * 2 * F : (int64_t)B * (T + 1) * (ld - 2 * F) + (int64_t)B * 2 * F;
  unsigned gx = nblk(total);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(spec_rows_kernel, dim3(gx), dim3(256), 0, (hipStream_t)stream, rows, buf, B, T, F, ld, mode);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_absmax(const float* x, int64_t n, float* out, void* stream) {
  if (n % 4) return -1002;
  hipStream_t st = (hipStream_t)stream;
  (void)hipMemsetAsync(out, 0, sizeof(float), st);
  unsigned gx = nblk(n / 4, 256 * 4);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(absmax_kernel, dim3(gx), dim3(256), 0, st, x, n / 4, reinterpret_cast<unsigned*>(out));
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                    float beta2, float eps, int step, float gscale, float clip, const float* sumsq,
                                    const int* guard, int* skipped, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return -1001;
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adam_kernel, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, bc1, bc2,
                     gscale, clip, sumsq, guard, skipped);
  SB_CHECK_LAUNCH();
  return 0;
}
extern "C" int sb_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                            float beta2, float eps, int step, float gscale, float clip, const float* sumsq,
                            void* stream) {
  return sb_adam_step_guarded(p, g, m, v, n, lr, beta1, beta2, eps, step, gscale, clip, sumsq, nullptr, nullptr, stream);
}

extern "C" int sb_multi_copy(const sb_multi_copy_args* ap, void* stream) {
  if (!ap || ap->njobs <= 0 || ap->njobs > SB_MULTI_COPY_MAX) return -1001;
  int64_t nmax = 0;
  for (int i = 0; i < ap->njobs; ++i) {
    if (!ap->src[i] || !ap->dst[i] || ap->n[i] < 0) return -1002;
    if (ap->n[i] > nmax) nmax = ap->n[i];
  }
  if (nmax == 0) return 0;
  int64_t gx = (nmax / 4 + 255) / 256;
  gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
  hipLaunchKernelGGL(multi_copy_kernel, dim3((unsigned)gx, ap->njobs), dim3(256), 0, (hipStream_t)stream, *ap);
  SB_CHECK_LAUNCH();
  return 0;
}
