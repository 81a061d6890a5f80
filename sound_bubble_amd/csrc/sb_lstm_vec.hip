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
// Step s of a workgroup (256 threads; wave w owns hidden units 16 w .. 16 w + 15, lane = 16 g + u: gate g of i, f, g, o for
// unit j = 16 w + u, i.e. gate row r = 64 g + j):
//   z_r   = zin[s][r] + sum_k W_hh[r][k] h[k]        zin = W_ih . LN(x_s) + b_ih + b_hh, precomputed 32 steps at a time
//   act_r = sigmoid / tanh (z_r); the four gates of a unit sit in ONE wave: lanes u gather f, g, o with three ds_bpermutes
//   c_j = f c_j + i g, h_j = o tanh(c_j) -> LDS (double-buffered); ONE workgroup barrier; every thread reads all of h
// No BPTT records (training keeps the tile kernels), no fused Linear / FiLM epilogues: sb_lstm_fwd picks this kernel only for
// calls that ask for hs (+ final state) alone.
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

constexpr int H = 64;
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int CH = 32;         // steps whose input projections are staged in LDS at a time

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains the vector-memory counter, i.e. every step
// would wait for global stores / prefetch loads in flight (a microsecond each)
SB_DEVINL void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// lane l gets v of lane (l & ~3) + k, k encoded as CTRL = 0x55 k (DPP quad_perm [k, k, k, k])
template <int CTRL>
SB_DEVINL float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int CC>
__global__ __launch_bounds__(256) void lstm_fwd_vec_kernel(sb_lstm_fwd_args a) {
  __shared__ __attribute__((aligned(16))) float zin[CH][4 * H];      // 32 KB
  __shared__ __attribute__((aligned(16))) float u_l[CH][CC];         // LayerNorm output of the chunk's steps
  __shared__ __attribute__((aligned(16))) float h_l[2][H];           // double-buffered: one barrier per step
  __shared__ __attribute__((aligned(16))) float hs_l[CH][H];         // the chunk's hidden states: stored in bulk (no global
                                                                     // store -- and no wait for one -- inside the step)

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, j = 16 * w + (lane & 15);
  const int n = blockIdx.x, d = blockIdx.y;
  const int64_t pos0 = (int64_t)(n / a.n_inner) * a.p_outer + (int64_t)(n % a.n_inner) * a.p_inner;
  const int r = g * H + j;                                           // gate row
  const bool owner = g == 0;                                         // lanes 0..15 of a wave carry c_j / h_j of its 16 units

  // weights of this row: registers for the whole launch
  float wsc[H], wih[CC];
  {
    const float* p = a.w_hh[d] + (size_t)r * H;
#pragma unroll
    for (int k = 0; k < H; k += 4) { const f32x4 v = ld4(p + k); wsc[k] = v[0]; wsc[k + 1] = v[1]; wsc[k + 2] = v[2]; wsc[k + 3] = v[3]; }
    const float* q = a.w_ih[d] + (size_t)r * CC;
#pragma unroll
    for (int k = 0; k < CC; k += 4) { const f32x4 v = ld4(q + k); wih[k] = v[0]; wih[k + 1] = v[1]; wih[k + 2] = v[2]; wih[k + 3] = v[3]; }
  }
  const float bias = a.b_ih[d][r] + a.b_hh[d][r];
  // sigmoid(z) = 1 / (1 + 2^(-log2e z)); tanh(z) = 2 sigmoid(2 z) - 1 (gate g = 2, the cell candidate): one exp2 + one rcp per row
  const float act_scale = g == 2 ? 2.0f * SB_NLOG2E : SB_NLOG2E, act_mul = g == 2 ? 2.0f : 1.0f, act_add = g == 2 ? -1.0f : 0.0f;

  float c = (owner && d == 0 && a.c0) ? a.c0[(size_t)n * H + j] : 0.f;
  float hcur = (owner && d == 0 && a.h0) ? a.h0[(size_t)n * H + j] : 0.f;
  if (owner) h_l[0][j] = hcur;

  // LayerNorm helpers: 8 threads per step of the chunk, CC / 8 channels each
  constexpr int PER = CC / 8;
  const int ls = tid >> 3, lq = tid & 7;
  float lg[PER], lb[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) { lg[i] = a.ln_g[lq * PER + i]; lb[i] = a.ln_b[lq * PER + i]; }

  // x rows of a chunk are fetched one chunk ahead (in flight under the previous chunk's recurrence)
  float xv[PER];
  auto fetch = [&](int s0) {
    const int s = s0 + ls;
    if (s < a.nsteps) {
      const int sp = d ? a.nsteps - 1 - s : s;
      const float* xr = a.x + (pos0 + (int64_t)sp * a.p_step) * CC + lq * PER;
#pragma unroll
      for (int i = 0; i < PER; ++i) xv[i] = xr[i];
    }
  };
  fetch(0);

  int hb = 0;                                                        // h buffer the next step reads
  for (int s0 = 0; s0 < a.nsteps; s0 += CH) {
    const int ns = min(CH, a.nsteps - s0);
    lds_barrier();                                                   // the previous chunk's zin / u / hs rows are no longer read
    // ---- u = LayerNorm_C(x) for the chunk's steps (two-pass variance, eps 1e-5: torch.nn.LayerNorm) ----
    if (ls < ns) {
      float v[PER], sum = 0.f;
#pragma unroll
      for (int i = 0; i < PER; ++i) { v[i] = xv[i]; sum += v[i]; }
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
    fetch(s0 + CH);
    lds_barrier();
    // ---- zin[s][tid] = b + W_ih[r] . u_s (off the serial chain; written and read back by the same thread) ----
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
      zin[s][tid] = acc0 + acc1;
    }

    // ---- the recurrence over the chunk ----
    for (int s = 0; s < ns; ++s) {
      // h reaches the lanes through registers, not through 64-lane LDS broadcasts (a broadcast ds_read_b128 still returns
      // 1 KB per wave: 64 KB per step and workgroup through the 128 B/clk LDS port -- 512 clocks, more than everything else in
      // the step): lane L fetches the 16 floats h[16 i + 4 (L & 3) + c] (four ds_read_b128: 4 KB per wave) and every product
      // takes its h operand from the quad neighbour that holds it (DPP quad_perm broadcast folded into the multiply-add)
      const float* hp = h_l[hb] + 4 * (lane & 3);
      const f32x4 hq0 = ld4(hp), hq1 = ld4(hp + 16), hq2 = ld4(hp + 32), hq3 = ld4(hp + 48);
      float a0 = zin[s][tid], a1 = 0.f, a2 = 0.f, a3 = 0.f;
      // v_fmac_f32_dpp: acc += (h of quad lane k) * w -- one instruction per product (hipcc leaves update_dpp + fma as two)
#define SB_FMAC_Q(ACC, HV, WV, Q)                                                                                  \
      asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[" #Q "," #Q "," #Q "," #Q "] row_mask:0xf bank_mask:0xf"      \
                   : "+v"(ACC) : "v"(HV), "v"(WV))
#define SB_ROW16(HQ, I)                                                                   \
      _Pragma("unroll") for (int cc = 0; cc < 4; ++cc) {                                    \
        const float hv = HQ[cc];                                                           \
        SB_FMAC_Q(a0, hv, wsc[16 * I + cc], 0);                                            \
        SB_FMAC_Q(a1, hv, wsc[16 * I + 4 + cc], 1);                                        \
        SB_FMAC_Q(a2, hv, wsc[16 * I + 8 + cc], 2);                                        \
        SB_FMAC_Q(a3, hv, wsc[16 * I + 12 + cc], 3);                                       \
      }
      // (a DPP source register written by a VALU instruction needs two wait states; the operands here come straight from
      // the four LDS reads above -- checked in the ISA -- and the s_nop covers a copy the register allocator might insert)
      asm volatile("s_nop 1");
      SB_ROW16(hq0, 0) SB_ROW16(hq1, 1) SB_ROW16(hq2, 2) SB_ROW16(hq3, 3)
#undef SB_ROW16
#undef SB_FMAC_Q
      const float z = (a0 + a1) + (a2 + a3);
      const float av = __builtin_fmaf(act_mul, sigmoid_pre(act_scale * z), act_add);
      // the unit's other three gates: lanes u + 16, u + 32, u + 48 of this wave (no workgroup barrier for this exchange)
      const float gf = __shfl(av, (lane & 15) + 16, 64), gg = __shfl(av, (lane & 15) + 32, 64), go = __shfl(av, (lane & 15) + 48, 64);
      c = __builtin_fmaf(gf, c, av * gg);                            // (meaningful in the owner lanes, where av = i)
      hcur = go * tanhf_fast(c);
      hb ^= 1;
      if (owner) { h_l[hb][j] = hcur; hs_l[s][j] = hcur; }
      // one barrier per step; h is double-buffered: a wave can run at most one step ahead of the slowest one, and the buffer
      // it then writes is not the one still being read
      lds_barrier();
    }
    // ---- the chunk's hs rows -> global, 16 threads x 16 bytes per row ----
    if (a.hs) {
      for (int s = tid >> 4; s < ns; s += 16) {
        const int sg = s0 + s, sp = d ? a.nsteps - 1 - sg : sg;
        st4(a.hs + (pos0 + (int64_t)sp * a.p_step) * (a.ndir * H) + d * H + 4 * (tid & 15), ld4(&hs_l[s][4 * (tid & 15)]));
      }
    }
  }
  if (d == 0 && owner) {
    if (a.hN) a.hN[(size_t)n * H + j] = hcur;
    if (a.cN) a.cN[(size_t)n * H + j] = c;
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
