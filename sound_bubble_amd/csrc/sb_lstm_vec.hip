// Forward LSTM recurrence for a HANDFUL of sequences (inference): one workgroup per (sequence, direction), matrix-vector
// products on the vector ALU with the weights resident in registers.
//
// Where it runs: the intra-frame pass of the streaming chunk step (edge/causal_infer.py:15-26 feeds ONE frame: B*T = 1
// sequence of 145 -- or 29, conv-LSTM -- frequency steps per direction) and short-clip inference.  The tile kernels
// (sb_lstm_bf.hip) give 16 sequences to a workgroup and pay ~2 200 cycles per step whether the tile is full or holds one
// sequence: MFMA issue for 16 columns, the hi/lo operand split, LDS exchange of a [16, 64] state tile.  One sequence needs
// none of that: gate row r of thread r is 64 fused multiply-adds against h (LDS broadcast reads) -- exact fp32, the
// reference's own arithmetic (tfgridnet_causal.py:818-823 / optim :690-703: nn.LSTM after LayerNorm) -- and one barrier per
// step.  With <= 256 workgroups every sequence has a CU to itself, so the launch takes nsteps x (one step's latency).
//
// Step s of a workgroup (256 threads; wave w owns gate type w of i, f, g, o; lane j = hidden unit j):
//   z_r   = zin[s][r] + sum_k W_hh[r][k] h[k]        zin = W_ih . LN(x_s) + b_ih + b_hh, precomputed 32 steps at a time
//   act_r = sigmoid / tanh (z_r)  -> LDS (double-buffered); barrier
//   every wave redundantly: c_j = f c_j + i g, h_j = o tanh(c_j) -> its own LDS copy of h (no second barrier)
// No BPTT records (training keeps the tile kernels), no fused Linear / FiLM epilogues: sb_lstm_fwd picks this kernel only for
// calls that ask for hs (+ final state) alone.
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

constexpr int H = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int CH = 32;         // steps whose input projections are staged in LDS at a time

template <int CC>
__global__ __launch_bounds__(256) void lstm_fwd_vec_kernel(sb_lstm_fwd_args a) {
  __shared__ __attribute__((aligned(16))) float zin[CH][4 * H];      // 32 KB
  __shared__ __attribute__((aligned(16))) float u_l[CH][CC];         // LayerNorm output of the chunk's steps
  __shared__ __attribute__((aligned(16))) float act[2][4 * H];       // double-buffered: one barrier per step
  __shared__ __attribute__((aligned(16))) float h_l[4][H];           // one copy of h per wave
  __shared__ __attribute__((aligned(16))) float hs_l[CH][H];         // the chunk's hidden states: stored in bulk (no global
                                                                     // store -- and no wait for one -- inside the step)

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.x, d = blockIdx.y;
  const int64_t pos0 = (int64_t)(n / a.n_inner) * a.p_outer + (int64_t)(n % a.n_inner) * a.p_inner;
  const int r = tid;                                                 // gate row: type w, unit lane

  // weights of this row: registers for the whole launch
  f32x2 whh2[H / 2];
  float wih[CC];
  {
    const float* p = a.w_hh[d] + (size_t)r * H;
#pragma unroll
    for (int k = 0; k < H; k += 4) { const f32x4 v = ld4(p + k); whh2[k / 2] = (f32x2){v[0], v[1]}; whh2[k / 2 + 1] = (f32x2){v[2], v[3]}; }
    const float* q = a.w_ih[d] + (size_t)r * CC;
#pragma unroll
    for (int k = 0; k < CC; k += 4) { const f32x4 v = ld4(q + k); wih[k] = v[0]; wih[k + 1] = v[1]; wih[k + 2] = v[2]; wih[k + 3] = v[3]; }
  }
  const float bias = a.b_ih[d][r] + a.b_hh[d][r];

  // state: lane j of EVERY wave carries c_j; h lives in LDS (one copy per wave)
  float c = (d == 0 && a.c0) ? a.c0[(size_t)n * H + lane] : 0.f;
  float hcur = (d == 0 && a.h0) ? a.h0[(size_t)n * H + lane] : 0.f;
  h_l[w][lane] = hcur;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // LayerNorm helpers: 8 threads per step of the chunk, CC / 8 channels each
  constexpr int PER = CC / 8;
  const int ls = tid >> 3, lq = tid & 7;
  float lg[PER], lb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { lg[i] = a.ln_g[lq * PER + i]; lb[i] = a.ln_b[lq * PER + i]; }

  for (int s0 = 0; s0 < a.nsteps; s0 += CH) {
    const int ns = min(CH, a.nsteps - s0);
    __syncthreads();                                                 // the previous chunk's zin / u are no longer read
    // ---- u = LayerNorm_C(x) for the chunk's steps (two-pass variance, eps 1e-5: torch.nn.LayerNorm) ----
    if (ls < ns) {
      const int s = s0 + ls, sp = d ? a.nsteps - 1 - s : s;
      const float* xr = a.x + (pos0 + (int64_t)sp * a.p_step) * CC + lq * PER;
      float v[PER], sum = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) { v[i] = xr[i]; sum += v[i]; }
      sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64); sum += __shfl_xor(sum, 4, 64);
      const float mean = sum * (1.0f / CC);
      float sq = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) { const float t = v[i] - mean; sq += t * t; }
      sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
      const float rstd = 1.0f / sqrtf(sq * (1.0f / CC) + 1e-5f);
#pragma unroll
      for (int i = 0; i < PER; ++i) u_l[ls][lq * PER + i] = (v[i] - mean) * rstd * lg[i] + lb[i];
    }
    __syncthreads();
    // ---- zin[s][r] = b + W_ih[r] . u_s (off the serial chain) ----
    for (int s = 0; s < ns; ++s) {
      float acc0 = bias, acc1 = 0.f;
#pragma unroll
      for (int k = 0; k < CC; k += 8) {
        const f32x4 u0 = ld4(&u_l[s][k]), u1 = ld4(&u_l[s][k + 4]);
        acc0 = __builtin_fmaf(wih[k], u0[0], acc0);     acc1 = __builtin_fmaf(wih[k + 4], u1[0], acc1);
        acc0 = __builtin_fmaf(wih[k + 1], u0[1], acc0); acc1 = __builtin_fmaf(wih[k + 5], u1[1], acc1);
        acc0 = __builtin_fmaf(wih[k + 2], u0[2], acc0); acc1 = __builtin_fmaf(wih[k + 6], u1[2], acc1);
        acc0 = __builtin_fmaf(wih[k + 3], u0[3], acc0); acc1 = __builtin_fmaf(wih[k + 7], u1[3], acc1);
      }
      zin[s][r] = acc0 + acc1;
    }
    // (zin[s][r] is read back by thread r only; h_l[w] is wave-private: no barrier needed here)

    // ---- the recurrence over the chunk ----
    for (int s = 0; s < ns; ++s) {
      // packed fp32 fused multiply-adds (v_pk_fma_f32: two per lane and instruction): the 64-term row product is the bulk of
      // a step's vector-ALU time
      f32x2 a0 = {zin[s][r], 0.f}, a1 = {0.f, 0.f}, a2 = {0.f, 0.f}, a3 = {0.f, 0.f};
      const float* hp = h_l[w];
#pragma unroll
      for (int k = 0; k < H; k += 16) {
        const f32x4 h0 = ld4(hp + k), h1 = ld4(hp + k + 4), h2 = ld4(hp + k + 8), h3 = ld4(hp + k + 12);
        a0 = __builtin_elementwise_fma(whh2[k / 2], (f32x2){h0[0], h0[1]}, a0);
        a1 = __builtin_elementwise_fma(whh2[k / 2 + 1], (f32x2){h0[2], h0[3]}, a1);
        a2 = __builtin_elementwise_fma(whh2[k / 2 + 2], (f32x2){h1[0], h1[1]}, a2);
        a3 = __builtin_elementwise_fma(whh2[k / 2 + 3], (f32x2){h1[2], h1[3]}, a3);
        a0 = __builtin_elementwise_fma(whh2[k / 2 + 4], (f32x2){h2[0], h2[1]}, a0);
        a1 = __builtin_elementwise_fma(whh2[k / 2 + 5], (f32x2){h2[2], h2[3]}, a1);
        a2 = __builtin_elementwise_fma(whh2[k / 2 + 6], (f32x2){h3[0], h3[1]}, a2);
        a3 = __builtin_elementwise_fma(whh2[k / 2 + 7], (f32x2){h3[2], h3[3]}, a3);
      }
      const f32x2 zz = (a0 + a1) + (a2 + a3);
      const float z = zz[0] + zz[1];
      float* ab = act[s & 1];
      ab[r] = (w == 2) ? tanhf_fast(z) : sigmoidf_fast(z);
      // one barrier per step: a wave can be at most one step ahead of the slowest one (it waits here), and then it writes
      // the OTHER act buffer
      __syncthreads();
      const float gi = ab[lane], gf = ab[H + lane], gg = ab[2 * H + lane], go = ab[3 * H + lane];
      c = __builtin_fmaf(gf, c, gi * gg);
      hcur = go * tanhf_fast(c);
      h_l[w][lane] = hcur;                                           // this wave's copy: read back by this wave only
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");         // (LDS operations of one wave execute in order)
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (w == 0) hs_l[s][lane] = hcur;
    }
    // ---- the chunk's hs rows -> global, 16 threads x 16 bytes per row ----
    if (a.hs) {
      __syncthreads();
      for (int s = tid >> 4; s < ns; s += 16) {
        const int sg = s0 + s, sp = d ? a.nsteps - 1 - sg : sg;
        st4(a.hs + (pos0 + (int64_t)sp * a.p_step) * (a.ndir * H) + d * H + 4 * (tid & 15), ld4(&hs_l[s][4 * (tid & 15)]));
      }
    }
  }
  if (d == 0 && w == 0) {
    if (a.hN) a.hN[(size_t)n * H + lane] = hcur;
    if (a.cN) a.cN[(size_t)n * H + lane] = c;
  }
}

}  // namespace

// the calls sb_lstm_fwd hands to this kernel: hs (+ final state) only, at most 256 workgroups
bool sb_lstm_fwd_vec_ok(const sb_lstm_fwd_args& a) {
  return !a.no_vec && !a.save_gates && !a.save_u && !a.save_c && !a.lin_w && !a.x_part && !a.film_w && !a.slab_flags && !a.tile_order &&
         (int64_t)a.nseq * a.ndir <= 256;
}

int sb_launch_lstm_fwd_vec(const sb_lstm_fwd_args& a, hipStream_t st) {
  dim3 grid(a.nseq, a.ndir);
  if (a.C == 32) hipLaunchKernelGGL(lstm_fwd_vec_kernel<32>, grid, dim3(256), 0, st, a);
  else if (a.C == 16) hipLaunchKernelGGL(lstm_fwd_vec_kernel<16>, grid, dim3(256), 0, st, a);
  else return -1002;
  return 0;
}
