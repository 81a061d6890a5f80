// Streaming (non-recurrent) half of the LSTM backward pass for gfx950:
//   one pass over dgates produces   dW_ih, dW_hh, db_ih, db_hh   (TN GEMMs over positions)
//                           and     dU = W_ih^T dgates            (gradient w.r.t. the LayerNorm output),
// so dgates (1 KB / position / direction) is read from HBM exactly once after the recurrent kernel wrote it.
// A second, light kernel folds the per-direction dU partials through the LayerNorm (and PReLU) backward.
//
// Layout trick: the MFMA contraction index of the weight gradients is the POSITION, so the operands need
// "position on the k-lanes".  Instead of 4-byte loads, gate columns / hidden units are assigned to MFMA tiles
// as  gate = 64*w + 4*i + tile  (unit = 4*i + tile), so ONE 16-byte load per lane per position yields the
// operand of four tiles at once (4 x 256 B contiguous per wave-instruction instead of 16 x 64 B).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include "sb_common.h"
#include "../../include/sound_bubble_hip.h"

namespace {

constexpr int H = SB_H;
constexpr int STREAM_MAX_WG = 512;

// SMALLSEG: seg_len < 32 (tiny test shapes): segment index by modulo; otherwise by one/two conditional
// subtractions, which keeps the loader branch-free (no inner loops -> one schedulable basic block).
template <int C, bool SMALLSEG>
__global__ __launch_bounds__(256) void lstm_bwd_stream_kernel(sb_lstm_stream_args a) {
  constexpr int CK = C / 16;            // u blocks
  constexpr int KT = CK + 4;            // + 4 h_prev blocks
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = blockIdx.y, ndir = a.ndir;
  const int64_t P = a.P;
  __shared__ __attribute__((aligned(16))) float R[2][4][CK][64][4];

  // W_ih^T fragments for this wave's 64-gate slice: A[i = channel 16ct + j][k = gate 64w + 16m + 4q + r]
  const float* __restrict__ wih = a.w_ih[dir];
  f32x4 Awt[CK][4];
#pragma unroll
  for (int ct = 0; ct < CK; ++ct)
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) Awt[ct][m][r] = wih[(size_t)(64 * w + 16 * m + 4 * q + r) * C + 16 * ct + j];

  f32x4 acc[4][KT];
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = zero4();

  const float* __restrict__ dg = a.dgates + (size_t)dir * 4 * H + 64 * w;
  const float* __restrict__ hs = a.hs + (size_t)dir * H;
  const _Float16* __restrict__ hs16 = reinterpret_cast<const _Float16*>(a.hs) + (size_t)dir * H;
  const _Float16* __restrict__ u16 = reinterpret_cast<const _Float16*>(a.u);
  const int64_t ldg = (int64_t)ndir * 4 * H, ldh = (int64_t)ndir * H;
  const int64_t hshift = (dir == 0 ? -1 : 1) * a.shift_pos * ldh;
  const int skip_first = dir == 0 ? a.skip : 0, skip_last = dir == 1 ? a.skip : 0;

  // Operands of one 16-position tile.  Tiles are software-pipelined: the loads of tile t+1 are issued before the
  // 128 MFMAs of tile t, so HBM latency hides under the matrix work.
  struct Tile { f32x4 a4[4], h4[4], d4[4]; float uv[CK][4]; };
  const int ntiles = (int)((P + 15) / 16);
  const int Pi = (int)P;
  // branch-free loader: out-of-range / masked lanes read a valid (clamped) address and are zeroed by a select,
  // so the loop body stays one basic block (exec-masked loads would put a branch around every load)
  auto load_tile = [&](int tile, Tile& t) {
    const int p0 = tile * 16;
    const int idx0 = p0 % a.seg_len;                     // wave-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int p = p0 + 4 * q + r;
      const bool ok = p < Pi;
      const int pc = ok ? p : Pi - 1;
      const f32x4 av = ld4(dg + (int64_t)pc * ldg + 4 * j);
      t.a4[r] = ok ? av : zero4();
      int idx = idx0 + 4 * q + r;
      if constexpr (SMALLSEG) idx %= a.seg_len; else idx -= idx >= a.seg_len ? a.seg_len : 0;
      const bool ok2 = ok & (idx >= skip_first) & (idx < a.seg_len - skip_last);
      const f32x4 hv = ld4(hs + (int64_t)pc * ldh + (ok2 ? hshift : 0) + 4 * j);
      t.h4[r] = ok2 ? hv : zero4();
      if constexpr (CK == 2) {
        const float2 v = *reinterpret_cast<const float2*>(a.u + (int64_t)pc * C + 2 * j);
        t.uv[0][r] = ok ? v.x : 0.f; t.uv[1][r] = ok ? v.y : 0.f;
      } else {
        const float v = a.u[(int64_t)pc * C + j];
        t.uv[0][r] = ok ? v : 0.f;
      }
    }
    const int pj = p0 + j;
    const int pjc = pj < Pi ? pj : Pi - 1;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 dv = ld4(dg + (int64_t)pjc * ldg + 16 * m + 4 * q);
      t.d4[m] = pj < Pi ? dv : zero4();
    }
  };

  Tile cur;
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x, cur);
  int it = 0;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
    Tile nxt;
    const int tn = tile + gridDim.x;
    load_tile(tn < ntiles ? tn : tile, nxt);             // (re-loads the last tile at the tail: branch-free)
    const int pj = tile * 16 + j;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const float av = cur.a4[r][nt];
        csum[nt] += av;
#pragma unroll
        for (int kt = 0; kt < CK; ++kt) acc[nt][kt] = mfma16(av, cur.uv[kt][r], acc[nt][kt]);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) acc[nt][CK + kt] = mfma16(av, cur.h4[r][kt], acc[nt][CK + kt]);
      }
    f32x4 du[CK];
#pragma unroll
    for (int ct = 0; ct < CK; ++ct) du[ct] = zero4();
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) du[ct] = mfma16(Awt[ct][m][r], cur.d4[m][r], du[ct]);
    const int buf = it & 1;
#pragma unroll
    for (int ct = 0; ct < CK; ++ct) st4(&R[buf][w][ct][lane][0], du[ct]);
    __syncthreads();
    if (w < CK && pj < Pi) {
      const f32x4 s = ld4(&R[buf][0][w][lane][0]) + ld4(&R[buf][1][w][lane][0]) + ld4(&R[buf][2][w][lane][0]) +
                      ld4(&R[buf][3][w][lane][0]);
      st4(a.du_part + ((int64_t)pj * ndir + dir) * C + 16 * w + 4 * q, s);
    }
    cur = nxt;
  }

  // ---- partial results: [N*(C+64) + N] per workgroup, true (un-permuted) indices ----
  constexpr int Ktot = C + H;
  float* part = a.scratch + ((size_t)dir * gridDim.x + blockIdx.x) * ((size_t)4 * H * Ktot + 4 * H);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gate = 64 * w + 4 * (4 * q + r) + nt;
#pragma unroll
      for (int kt = 0; kt < CK; ++kt) part[(size_t)gate * Ktot + (CK == 2 ? 2 * j + kt : j)] = acc[nt][kt][r];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) part[(size_t)gate * Ktot + C + 4 * j + kt] = acc[nt][CK + kt][r];
    }
    const float cs = quad_sum(csum[nt]);
    if (q == 0) part[(size_t)4 * H * Ktot + 64 * w + 4 * j + nt] = cs;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Same computation on the bf16 matrix pipe with 3-term split products ("bf16x3"): every fp32 operand x is
// split as x = hi + lo (hi = bf16(x), lo = bf16(x - hi)) and a*b ~= ah*bh + ah*bl + al*bh, each product exact
// in the fp32 accumulator -- relative error ~2^-16 per term, i.e. fp32-class for a gradient reduction, at
// 1/5 of the fp32-MFMA cycles.  v_mfma_f32_16x16x32_bf16 contracts 32 positions per instruction (8 per lane:
// lane l holds k = 8*(l>>4)..+7), so a chunk is 32 positions; the tile-interleaved column assignment
// (gate = 64w + 4i + tile) still gives one 16-byte load per lane per position.  With the matrix work cut 4x
// the kernel becomes HBM-bound, so the next chunk's raw operands are prefetched into registers.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct SplitBf { bf16x8 hi, lo; };
SB_DEVINL SplitBf split8(const float (&x)[8]) {
  SplitBf s;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const __bf16 h = (__bf16)x[k];
    s.hi[k] = h;
    s.lo[k] = (__bf16)(x[k] - (float)h);
  }
  return s;
}
SB_DEVINL f32x4 mfma_bf3(const SplitBf& a, const SplitBf& b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.lo, b.hi, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.lo, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.hi, b.hi, c, 0, 0, 0);
  return c;
}

template <int C, bool SMALLSEG>
__global__ __launch_bounds__(256) void lstm_bwd_stream_bf16_kernel(sb_lstm_stream_args a) {
  constexpr int CK = C / 16, KT = CK + 4;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = blockIdx.y, ndir = a.ndir;
  const int Pi = (int)a.P;
  __shared__ __attribute__((aligned(16))) float R[2][4][2][CK][64][4];

  // W_ih^T, split once: A[i = channel 16ct + j][k = gate 64w + 32m + 8q + kk]
  const float* __restrict__ wih = a.w_ih[dir];
  SplitBf Awt[CK][2];
#pragma unroll
  for (int ct = 0; ct < CK; ++ct)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = wih[(size_t)(64 * w + 32 * m + 8 * q + kk) * C + 16 * ct + j];
      Awt[ct][m] = split8(t);
    }

  f32x4 acc[4][KT];
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = zero4();

  const float* __restrict__ dg = a.dgates + (size_t)dir * 4 * H + 64 * w;
  const float* __restrict__ hs = a.hs + (size_t)dir * H;
  const _Float16* __restrict__ hs16 = reinterpret_cast<const _Float16*>(a.hs) + (size_t)dir * H;
  const _Float16* __restrict__ u16 = reinterpret_cast<const _Float16*>(a.u);
  const int64_t ldg = (int64_t)ndir * 4 * H, ldh = (int64_t)ndir * H;
  const int64_t hshift = (dir == 0 ? -1 : 1) * a.shift_pos * ldh;
  const int skip_first = dir == 0 ? a.skip : 0, skip_last = dir == 1 ? a.skip : 0;

  struct Chunk { f32x4 a4[8], h4[8], d4[2][2][2]; float uv[CK][8]; };     // raw fp32 operands of 32 positions
  const int nchunks = (Pi + 31) / 32;
  auto load_chunk = [&](int ch, Chunk& t) {          // branch-free (clamped address + select), see the fp32 kernel
    const int p0 = ch * 32;
    const int idx0 = p0 % a.seg_len;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int p = p0 + 8 * q + kk;
      const bool ok = p < Pi;
      const int pc = ok ? p : Pi - 1;
      const f32x4 av = ld4(dg + (int64_t)pc * ldg + 4 * j);
      t.a4[kk] = ok ? av : zero4();
      int idx = idx0 + 8 * q + kk;
      if constexpr (SMALLSEG) idx %= a.seg_len; else idx -= idx >= a.seg_len ? a.seg_len : 0;
      const bool ok2 = ok & (idx >= skip_first) & (idx < a.seg_len - skip_last);
      const f32x4 hv = ld4(hs + (int64_t)pc * ldh + (ok2 ? hshift : 0) + 4 * j);
      t.h4[kk] = ok2 ? hv : zero4();
      if constexpr (CK == 2) {
        const float2 v = *reinterpret_cast<const float2*>(a.u + (int64_t)pc * C + 2 * j);
        t.uv[0][kk] = ok ? v.x : 0.f; t.uv[1][kk] = ok ? v.y : 0.f;
      } else {
        const float v = a.u[(int64_t)pc * C + j];
        t.uv[0][kk] = ok ? v : 0.f;
      }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {          // dU operand: position p0 + 16 sb + j, gates 32m + 8q .. +7
      const int pj = p0 + 16 * sb + j;
      const int pjc = pj < Pi ? pj : Pi - 1;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const f32x4 dv = ld4(dg + (int64_t)pjc * ldg + 32 * m + 8 * q + 4 * hh);
          t.d4[sb][m][hh] = pj < Pi ? dv : zero4();
        }
    }
  };

  Chunk cur;
  if ((int)blockIdx.x < nchunks) load_chunk(blockIdx.x, cur);
  int it = 0;
  for (int ch = blockIdx.x; ch < nchunks; ch += gridDim.x, ++it) {
    Chunk nxt;
    const int cn = ch + gridDim.x;
    load_chunk(cn < nchunks ? cn : ch, nxt);
    // ---- weight gradients: 4 gate tiles x (CK + 4) column tiles, K = 32 positions ----
    SplitBf Bop[KT];
#pragma unroll
    for (int kt = 0; kt < CK; ++kt) Bop[kt] = split8(cur.uv[kt]);
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = cur.h4[kk][kt];
      Bop[CK + kt] = split8(t);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) { t[kk] = cur.a4[kk][nt]; csum[nt] += t[kk]; }
      const SplitBf Aop = split8(t);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = mfma_bf3(Aop, Bop[kt], acc[nt][kt]);
    }
    // ---- dU = W_ih^T dgates for the two 16-position sub-tiles ----
    const int buf = it & 1;
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      f32x4 du[CK];
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) du[ct] = zero4();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float t[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) t[kk] = cur.d4[sb][m][kk >> 2][kk & 3];
        const SplitBf Dop = split8(t);
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) du[ct] = mfma_bf3(Awt[ct][m], Dop, du[ct]);
      }
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) st4(&R[buf][w][sb][ct][lane][0], du[ct]);
    }
    __syncthreads();
    {   // 2 sub-tiles x CK channel tiles = 2*CK (<= 4) reductions: one per wave
      const int sb = w / CK, ct = w % CK;
      const int pj = ch * 32 + 16 * sb + j;
      if (w < 2 * CK && pj < Pi) {
        const f32x4 s4 = ld4(&R[buf][0][sb][ct][lane][0]) + ld4(&R[buf][1][sb][ct][lane][0]) +
                         ld4(&R[buf][2][sb][ct][lane][0]) + ld4(&R[buf][3][sb][ct][lane][0]);
        st4(a.du_part + ((int64_t)pj * ndir + dir) * C + 16 * ct + 4 * q, s4);
      }
    }
    cur = nxt;
  }

  constexpr int Ktot = C + H;
  float* part = a.scratch + ((size_t)dir * gridDim.x + blockIdx.x) * ((size_t)4 * H * Ktot + 4 * H);
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gate = 64 * w + 4 * (4 * q + r) + nt;
#pragma unroll
      for (int kt = 0; kt < CK; ++kt) part[(size_t)gate * Ktot + (CK == 2 ? 2 * j + kt : j)] = acc[nt][kt][r];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) part[(size_t)gate * Ktot + C + 4 * j + kt] = acc[nt][CK + kt][r];
    }
    const float cs = quad_sum(csum[nt]);
    if (q == 0) part[(size_t)4 * H * Ktot + 64 * w + 4 * j + nt] = cs;
  }
}

// ---------------------------------------------------------------------------------------------------------
// fp16 variant for the compact dgates of sb_lstm_bwd_rec (fp16, scaled by S = 2^-ceil(log2 gmax)): dgates go to
// the matrix pipe as they are (exact), the fp32 operands (u, h_prev, W_ih) as fp16 hi + lo (22 mantissa bits), so
// every product needs 2 MFMAs instead of 3 and no dgates split; half the dgates bytes.  Outputs are multiplied by
// 1/S.  Tiling, prefetch and output layout are those of the bf16 kernel above.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

struct SplitH { h16x8 hi, lo; };
SB_DEVINL SplitH splith8(const float (&x)[8]) {
  SplitH s;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const _Float16 h = (_Float16)x[k];
    s.hi[k] = h;
    s.lo[k] = (_Float16)(x[k] - (float)h);
  }
  return s;
}
SB_DEVINL f32x4 mfma_h(h16x8 a, h16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// U16 / HS16: u / hs arrive as the fp16 tensors the forward kernel wrote for this purpose (sb_lstm_fwd_args.aux_f16)
// LNB (single direction): the LayerNorm backward that follows (sb_ln_bwd) rides in the flush of the du rows:
// dx[p] = LN-backward(du[p]; ln_x[p], ln_g) + ln_res[p] is stored instead of du, d(ln_g) / d(ln_b) join the workgroup's
// partial row, max |dx| goes to absmax_out.  After the barrier wave w takes sub-tile w >> 1, positions 8 (w & 1) + (lane & 7)
// and channel quad lane >> 3, so the 4 CK lanes that hold one position's channels differ in lane bits 3.. and the
// LayerNorm sums are three (two for C = 16) xor-shuffles.  Saves the du round trip (2 x 4C bytes per position) and a launch.
// LINW (with LNB): the weight gradient of the Linear in front of the residual rides along as well -- its output gradient IS
// ln_res, which the flush lanes hold; they drop it (scaled fp16) into a [channel][position] LDS tile, wave w multiplies
// it with the UNSHIFTED hs rows of the chunk (h columns 4j + w): d_lin_w [C, 64] += dy^T hs, d_lin_b [C] += sum dy.
// SLAB: consumer side of the overlapped inter-frame backward (sb_lstm_bwd_inter_overlapped).  Chunk c is chunk i of batch
// entry b of time slab k (slab_len steps, latest first -- the order the recurrence produces them in).  The workgroups of
// BOTH launches of the pass draw UNITS of kUnit consecutive chunks from one atomic counter (a draw per chunk was tried:
// device-scope atomics on one address retire at a few tens per microsecond, and whatever thread 0 does per chunk in front
// of the barrier is time the other three waves spend waiting there); inside a unit the chunk coordinates advance by
// carries and the next chunk is prefetched as in the plain kernel.  A unit is started once slab_flags[k] of its LAST chunk
// has reached slab_need (every tile of the recurrence has stored that slab's dgates, write-through).  A chunk never reads
// a dgates row outside its own (b, slab) range, so no line of an unfinished slab is ever brought into this XCD's L2.
// Guarded launch (the one that runs NEXT to the recurrence): a workgroup that does not find every recurrence workgroup
// started within ~50 us draws nothing -- should the dispatcher have placed this launch first, it must not sit on the CUs
// the recurrence needs.  Partial rows from row_base on.
// XPS (with U16, HS16, LNB, LINW, SLAB): the WIDE form of the overlapped inter-frame backward (sb_lstm_stream_args.wide).  A
// dgates row is [hi x 256 | lo' x 256] halves, x = hi + 2^-11 lo' (written by lstm_bwd_rec_bf_kernel<..., SLAB, XP>), u and hs
// are the forward kernel's fp16 (hi, lo) pair tensors: three products per MAC, the scaled low terms of du and of the bias
// sums accumulated apart and folded in with 2^-11, the low dgates terms of dW meeting 2^-11 * hi of u / h.  Only the
// dgates of the next chunk are prefetched (64 registers); u / h / x / residual rows are fetched at the top of the chunk
// and meet their first use after the 24 MFMAs of du -- this launch fills CUs the recurrence leaves idle, it has time.
template <int C, bool SMALLSEG, bool U16, bool HS16, bool LNB = false, bool LINW = false, bool SLAB = false, bool XPS = false>
__global__ __launch_bounds__(256) void lstm_bwd_stream_f16_kernel(sb_lstm_stream_args a) {
  static_assert(!XPS || (U16 && HS16 && LNB && LINW && SLAB), "wide form: the overlapped instantiation only");
  constexpr int CK = C / 16, KT = CK + 4;
  constexpr float kLoDn = 1.0f / 2048.0f, kLoUp = 2048.0f;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), q = lane >> 4, j = lane & 15;
  const int dir = blockIdx.y, ndir = a.ndir;
  const int Pi = (int)a.P;
  __shared__ __attribute__((aligned(16))) float R[2][4][2][CK][64][4];
  __shared__ __attribute__((aligned(16))) _Float16 DY[LINW ? 2 : 1][LINW ? C : 1][40];
  __shared__ __attribute__((aligned(16))) _Float16 DYL[XPS ? 2 : 1][XPS ? C : 1][40];      // XPS: scaled low terms of dy
  float invS = 1.0f;
  {
    const float m = a.gmax[0];
    if (m > 0.f && m < 3.0e38f) invS = exp2f(ceilf(log2f(m)));
  }

  // W_ih^T, split once: A[i = channel 16ct + j][k = gate 64w + 32m + 8q + kk]
  const float* __restrict__ wih = a.w_ih[dir];
  SplitH Awt[CK][2];
#pragma unroll
  for (int ct = 0; ct < CK; ++ct)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = wih[(size_t)(64 * w + 32 * m + 8 * q + kk) * C + 16 * ct + j];
      Awt[ct][m] = splith8(t);
    }

  f32x4 acc[4][KT];
  float csum[4] = {0.f, 0.f, 0.f, 0.f}, csumx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = zero4();

  const _Float16* __restrict__ dg = reinterpret_cast<const _Float16*>(a.dgates) + (size_t)dir * 4 * H + 64 * w;
  const float* __restrict__ hs = a.hs + (size_t)dir * H;
  const _Float16* __restrict__ hs16 = reinterpret_cast<const _Float16*>(a.hs) + (size_t)dir * H;
  const _Float16* __restrict__ u16 = reinterpret_cast<const _Float16*>(a.u);
  const int64_t ldg = XPS ? (int64_t)8 * H : (int64_t)ndir * 4 * H, ldh = XPS ? (int64_t)2 * H : (int64_t)ndir * H;
  const int64_t hshift = (dir == 0 ? -1 : 1) * a.shift_pos * ldh;
  const int skip_first = dir == 0 ? a.skip : 0, skip_last = dir == 1 ? a.skip : 0;

  struct Chunk {                                                                   // raw operands of 32 positions
    h16x4 a4[8]; f32x4 h4[HS16 ? 1 : 8]; h16x4 hh4[HS16 && !XPS ? 8 : 1]; h16x8 d8[2][2];
    h16x4 a4l[XPS ? 8 : 1]; h16x8 d8l[XPS ? 2 : 1][2];                            // XPS: the scaled low terms
    float uv[U16 ? 1 : CK][8]; _Float16 uh[U16 && !XPS ? CK : 1][8];
    f32x4 xq, rq;                                                                  // LNB: x and residual of the flush position
    h16x4 hu[LINW && !XPS ? 8 : 1];                                                // LINW: hs of the chunk's own positions
  };
  // XPS: the u / h_prev / own-hs pair rows and the flush rows (x, residual) of a chunk reach the waves through LDS: every
  // thread fetches a few 16-byte pieces of the chunk TWO iterations ahead into registers (28 instead of the 120 a private
  // copy per wave would take -- there is no room to prefetch those), drops them into the other LDS buffer one iteration
  // later, and all four waves read their operands from there.  One fetch per workgroup instead of one per wave as well.
  constexpr int HROW = 256 + 16, UROW = 4 * C + 16;            // padded LDS rows (bytes): hs pairs; u pairs / x / residual
  __shared__ __attribute__((aligned(16))) char SHP[XPS ? 2 : 1][XPS ? 32 * HROW : 16];
  __shared__ __attribute__((aligned(16))) char SHU[XPS ? 2 : 1][XPS ? 32 * HROW : 16];
  __shared__ __attribute__((aligned(16))) char SUP[XPS ? 2 : 1][XPS ? 32 * UROW : 16];
  __shared__ __attribute__((aligned(16))) char SX[XPS ? 2 : 1][XPS ? 32 * UROW : 16];
  __shared__ __attribute__((aligned(16))) char SR[XPS ? 2 : 1][XPS ? 32 * UROW : 16];
  struct Stage { f32x4 hp[2], hu[2], up, x, r; };
  // SLAB geometry: B sequences-of-slabs x nslabs, cpb chunks per (batch, slab)
  const int sF = (int)a.shift_pos, sT = SLAB ? a.seg_len / sF : 0;
  // every slab but the last (shorter) one has cpb chunks per batch entry
  const int cpb = SLAB ? (a.slab_len * sF + 31) / 32 : 1, cps = SLAB ? (Pi / a.seg_len) * cpb : 1;
  const int nsl = SLAB ? (sT + a.slab_len - 1) / a.slab_len : 1;
  const int cpl = SLAB ? ((sT - (nsl - 1) * a.slab_len) * sF + 31) / 32 : 1;
  // A workgroup's chunks are ch, ch + G, ch + 2G ...: the (slab, batch entry, chunk-in-range) coordinates are decoded
  // once and then advanced by carries -- no integer division per chunk (uniform, scalar unit)
  const int nbat = SLAB ? Pi / a.seg_len : 1;
  struct Span { int k, b, i, cpk, p0, pe, idx0; };
  auto span_fill = [&](Span& c) {
    const int t1 = sT - c.k * a.slab_len, t0 = max(0, t1 - a.slab_len);
    c.idx0 = t0 * sF + 32 * c.i;
    c.p0 = c.b * a.seg_len + c.idx0;
    c.pe = c.b * a.seg_len + t1 * sF;
  };
  auto span_at = [&](int ch) -> Span {
    Span c;
    if constexpr (SLAB) {
      c.k = min(ch / cps, nsl - 1);
      const int r = ch - c.k * cps;
      c.cpk = c.k == nsl - 1 ? cpl : cpb;
      c.b = r / c.cpk;
      c.i = r - c.b * c.cpk;
      span_fill(c);
    } else {
      c.k = c.b = c.i = c.cpk = 0;
      c.p0 = ch * 32; c.pe = Pi; c.idx0 = c.p0 % a.seg_len;
    }
    return c;
  };
  auto span_next = [&](const Span& o, int ch_next) -> Span {     // coordinates of chunk ch_next = o's chunk + cstride
    Span c = o;
    if constexpr (SLAB) {
      c.i += 1;
      if (c.i >= c.cpk) {
        c.i = 0;
        if (++c.b == nbat) { c.b = 0; ++c.k; c.cpk = c.k >= nsl - 1 ? cpl : cpb; }
      }
      span_fill(c);
    } else {
      c.p0 = ch_next * 32; c.idx0 = c.p0 % a.seg_len;
    }
    return c;
  };
  // Thread 0 polls (bounded, like the segment hand-off) and leaves the verdict in LDS; the workgroup reads it after its
  // next barrier -- inside the chunk loop that is the barrier the loop has anyway, one chunk ahead of the loads.
  __shared__ int slab_abort;
  int ready_k = -1;                                        // thread 0 only
  auto slab_poll = [&](int k) {
    if (tid == 0 && k > ready_k) {
      unsigned spins = 0;
      while (sb_poll(a.slab_flags + k) < a.slab_need) {
        ++spins;
        if (sb_wait_over(a.sched_status, spins, SB_TRIP_BWD_STREAM, k, a.slab_flags + k, a.slab_need)) { slab_abort = 1; break; }
        sb_poll_pause();
      }
      ready_k = k;
    }
  };
  // LNB flush role of this lane
  const int fsb = w >> 1, fjj = 8 * (w & 1) + (lane & 7), fqq = lane >> 3;
  const int fct = CK == 2 ? fqq >> 2 : 0, fqr = fqq & 3;
  const bool fact = CK == 2 || fqq < 4;
  const int fcol = 16 * fct + 4 * fqr;
  f32x4 lgam = zero4(), dgam = zero4(), dbet = zero4(), dlb = zero4();
  f32x4 lacc[CK];
#pragma unroll
  for (int ct = 0; ct < CK; ++ct) lacc[ct] = zero4();
  const float gS = 1.0f / invS;                      // exact: invS is a power of two
  float amax = 0.f;
  if constexpr (LNB) lgam = ld4(a.ln_g + fcol);
  auto grp_sum = [&](float v) {                      // over the lanes holding one position's channels
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    if constexpr (CK == 2) v += __shfl_xor(v, 32, 64);
    return v;
  };
  const int nchunks = SLAB ? a.nchunks : (Pi + 31) / 32;
  constexpr int kUnit = 16;                              // chunks per draw (SLAB; 8: +8 %, 32: +6 % on the pass)
  const int cstride = SLAB ? 1 : (int)gridDim.x;
  const h16x4 hz4 = {0, 0, 0, 0};
  const h16x8 hz8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto load_chunk = [&](const Span& sp, Chunk& t) {  // branch-free (clamped address + select)
    const int p0 = sp.p0, pe = sp.pe, idx0 = sp.idx0;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int p = p0 + 8 * q + kk;
      const bool ok = p < pe;
      const int pc = min(p, pe - 1);
      // positions beyond P are cancelled through their (zeroed) dgates alone; u / h_prev of the clamped row are finite
      const h16x4 av = *reinterpret_cast<const h16x4*>(dg + (int64_t)pc * ldg + 4 * j);
      t.a4[kk] = ok ? av : hz4;
      if constexpr (XPS) {
        const h16x4 avl = *reinterpret_cast<const h16x4*>(dg + (int64_t)pc * ldg + 4 * H + 4 * j);
        t.a4l[kk] = ok ? avl : hz4;            // everything else of the chunk is fetched late (load_late)
      } else {
      int idx = idx0 + 8 * q + kk;
      if constexpr (SMALLSEG) idx %= a.seg_len; else idx -= idx >= a.seg_len ? a.seg_len : 0;
      const bool ok2 = ok & (idx >= skip_first) & (idx < a.seg_len - skip_last);
      if constexpr (HS16) {
        const h16x4 hv = *reinterpret_cast<const h16x4*>(hs16 + (int64_t)pc * ldh + (ok2 ? hshift : 0) + 4 * j);
        t.hh4[kk] = ok2 ? hv : hz4;
        if constexpr (LINW) {
          const h16x4 hu = *reinterpret_cast<const h16x4*>(hs16 + (int64_t)pc * ldh + 4 * j);
          t.hu[kk] = ok ? hu : hz4;
        }
      } else {
        const f32x4 hv = ld4(hs + (int64_t)pc * ldh + (ok2 ? hshift : 0) + 4 * j);
        t.h4[kk] = ok2 ? hv : zero4();
      }
      if constexpr (U16) {
        if constexpr (CK == 2) {
          const h16x2 v = *reinterpret_cast<const h16x2*>(u16 + (int64_t)pc * C + 2 * j);
          t.uh[0][kk] = v[0]; t.uh[1][kk] = v[1];
        } else {
          t.uh[0][kk] = u16[(int64_t)pc * C + j];
        }
      } else if constexpr (CK == 2) {
        const float2 v = *reinterpret_cast<const float2*>(a.u + (int64_t)pc * C + 2 * j);
        t.uv[0][kk] = v.x; t.uv[1][kk] = v.y;
      } else {
        t.uv[0][kk] = a.u[(int64_t)pc * C + j];
      }
      }
    }
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {          // dU operand: position p0 + 16 sb + j, gates 32m + 8q .. +7 (one 16-byte load)
      const int pjc = min(p0 + 16 * sb + j, pe - 1);      // dU of positions beyond the range is computed but never stored
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        t.d8[sb][m] = *reinterpret_cast<const h16x8*>(dg + (int64_t)pjc * ldg + 32 * m + 8 * q);
        if constexpr (XPS) t.d8l[sb][m] = *reinterpret_cast<const h16x8*>(dg + (int64_t)pjc * ldg + 4 * H + 32 * m + 8 * q);
      }
    }
    if constexpr (LNB && !XPS) {
      const int64_t pf = min(p0 + 16 * fsb + fjj, pe - 1);
      t.xq = ld4(a.ln_x + pf * C + fcol);
      t.rq = ld4(a.ln_res + pf * C + fcol);
    }
  };

  const int srow = tid >> 4, scol = tid & 15;                    // hs rows: 16 pieces of 16 bytes; this thread: rows srow, 16 + srow
  constexpr int UPC = C / 4;                                     // 16-byte pieces per u / x / residual row (C = 32: 8)
  const int urow = tid / UPC, ucol = tid % UPC;                  // (threads < 32 UPC carry one piece each)
  const f32x4 z4 = zero4();
  auto stage_load = [&](const Span& sp, Stage& g) {
    const int p0 = sp.p0, pe = sp.pe, idx0 = sp.idx0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = srow + 16 * i, p = p0 + r;
      const bool ok = p < pe;
      const int pc = min(p, pe - 1);
      int idx = idx0 + r;
      if constexpr (SMALLSEG) idx %= a.seg_len; else idx -= idx >= a.seg_len ? a.seg_len : 0;
      const bool ok2 = ok & (idx >= skip_first) & (idx < a.seg_len - skip_last);
      const f32x4 hv = ld4(reinterpret_cast<const float*>(hs16 + (int64_t)pc * ldh + (ok2 ? hshift : 0) + 8 * scol));
      g.hp[i] = ok2 ? hv : z4;
      const f32x4 hu = ld4(reinterpret_cast<const float*>(hs16 + (int64_t)pc * ldh + 8 * scol));
      g.hu[i] = ok ? hu : z4;
    }
    if (tid < 32 * UPC) {
      const int pc = min(p0 + urow, pe - 1);
      g.up = ld4(reinterpret_cast<const float*>(u16 + ((int64_t)pc * C) * 2 + 8 * ucol));
      g.x = ld4(a.ln_x + (int64_t)pc * C + 4 * ucol);
      g.r = ld4(a.ln_res + (int64_t)pc * C + 4 * ucol);
    }
  };
  auto stage_store = [&](int sb_, const Stage& g) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      st4(reinterpret_cast<float*>(&SHP[sb_][(srow + 16 * i) * HROW + 16 * scol]), g.hp[i]);
      st4(reinterpret_cast<float*>(&SHU[sb_][(srow + 16 * i) * HROW + 16 * scol]), g.hu[i]);
    }
    if (tid < 32 * UPC) {
      st4(reinterpret_cast<float*>(&SUP[sb_][urow * UROW + 16 * ucol]), g.up);
      st4(reinterpret_cast<float*>(&SX[sb_][urow * UROW + 16 * ucol]), g.x);
      st4(reinterpret_cast<float*>(&SR[sb_][urow * UROW + 16 * ucol]), g.r);
    }
  };
  Stage stg;
  int sbuf = 0;                                                  // LDS buffer of the chunk in hand (XPS)

  const h16x2 ones = {(_Float16)1.0f, (_Float16)1.0f};
  Chunk cur;
  __shared__ int unit_box;
  bool gok = true;                                   // thread 0: the guard's verdict
  int it = 0;
  for (int round = 0;; ++round) {                    // SLAB: one unit per round; plain launches: one round
  int ch, ch_hi;
  if constexpr (SLAB) {
    if (tid == 0) {
      if (round == 0) {
        slab_abort = 0;
        if (a.guard) {                               // every recurrence workgroup started?
          gok = false;
          for (int i = 0; i < 200 && !gok; ++i) {
            gok = __hip_atomic_load(a.started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.slab_need;
            if (!gok) __builtin_amdgcn_s_sleep(8);
          }
        }
      }
      unit_box = gok ? __hip_atomic_fetch_add(a.chunk_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (1 << 26);
    }
    __syncthreads();
    const int u = __builtin_amdgcn_readfirstlane(unit_box);
    if (u >= (nchunks + kUnit - 1) / kUnit) break;
    ch = u * kUnit;
    ch_hi = min(ch + kUnit, nchunks);
    slab_poll(span_at(ch_hi - 1).k);                 // thread 0; the slabs only grow along a unit
    __syncthreads();
    if (slab_abort) break;
  } else {
    if (round) break;
    ch = blockIdx.x;
    ch_hi = nchunks;
  }
  Span sc = span_at(ch), sn = span_next(sc, ch + cstride);          // chunk in hand, the one being loaded
  if (ch < ch_hi) load_chunk(sc, cur);               // (a workgroup without chunks still writes its zero partial row)
  if constexpr (XPS) {                               // pipeline start of this unit: chunk ch into LDS, chunk ch + 1 in flight
    if (ch < ch_hi) {
      sbuf ^= 1;
      stage_load(sc, stg);
      stage_store(sbuf, stg);
      stage_load(ch + cstride < ch_hi ? sn : sc, stg);
      __syncthreads();
    }
  }
  for (; ch < ch_hi; ++it) {
    Chunk nxt;
    const int cn = ch + cstride;
    load_chunk(cn < ch_hi ? sn : sc, nxt);
    const int cp0 = sc.p0, cpe = sc.pe;              // positions [cp0, cpe) of the chunk in hand (uniform)
    const Span s2 = span_next(sn, cn + cstride);     // ... and of the chunk the NEXT iteration loads
    if constexpr (XPS) {
      // chunk ch + 1 (fetched one iteration ago) -> the other LDS buffer: every wave has passed the barrier of the previous
      // iteration, i.e. has finished reading that buffer; then chunk ch + 2 goes in flight
      stage_store(sbuf ^ 1, stg);
      stage_load(cn + cstride < ch_hi ? s2 : sc, stg);
    }
    const int buf = it & 1;
    f32x4 fxq, frq;                                  // x and residual of this lane's flush position
    h16x8 Bu, Bul, Bus;                              // LINW: h tile w of the chunk's own positions (XPS: hi, lo, 2^-11 hi)
    if constexpr (XPS) {
      const char* __restrict__ shp = SHP[sbuf];
      const char* __restrict__ shu = SHU[sbuf];
      const char* __restrict__ sup = SUP[sbuf];
      const h16x8 dn8 = {(_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn,
                         (_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn};
      // ---- dU first: it needs nothing but the (prefetched) dgates ----
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        f32x4 du[CK], dux[CK];
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) { du[ct] = zero4(); dux[ct] = zero4(); }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int ct = 0; ct < CK; ++ct) {
            dux[ct] = mfma_h(Awt[ct][m].hi, cur.d8l[sb][m], dux[ct]);
            du[ct] = mfma_h(Awt[ct][m].lo, cur.d8[sb][m], du[ct]);
            du[ct] = mfma_h(Awt[ct][m].hi, cur.d8[sb][m], du[ct]);
          }
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) {
#pragma unroll
          for (int r = 0; r < 4; ++r) du[ct][r] = __builtin_fmaf(dux[ct][r], kLoDn, du[ct][r]);
          st4(&R[buf][w][sb][ct][lane][0], du[ct]);
        }
      }
      // ---- weight gradients: 3 products per MAC ----
      h16x8 Aoh[4], Aol[4];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { Aoh[nt][kk] = cur.a4[kk][nt]; Aol[nt][kk] = cur.a4l[kk][nt]; }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          csum[nt] = __builtin_amdgcn_fdot2(h16x2{Aoh[nt][2 * pr], Aoh[nt][2 * pr + 1]}, ones, csum[nt], false);
          csumx[nt] = __builtin_amdgcn_fdot2(h16x2{Aol[nt][2 * pr], Aol[nt][2 * pr + 1]}, ones, csumx[nt], false);
        }
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) {
        h16x8 bh, bl;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          if (kt < CK) {
            if constexpr (CK == 2) {
              const h16x4 up = *reinterpret_cast<const h16x4*>(sup + (8 * q + kk) * UROW + 8 * j);      // (hi0, hi1, lo0, lo1)
              bh[kk] = up[kt < CK ? kt : 0]; bl[kk] = up[2 + (kt < CK ? kt : 0)];
            } else {
              const h16x2 up = *reinterpret_cast<const h16x2*>(sup + (8 * q + kk) * UROW + 4 * j);      // (hi, lo)
              bh[kk] = up[0]; bl[kk] = up[1];
            }
          } else {
            const h16x8 hp = *reinterpret_cast<const h16x8*>(shp + (8 * q + kk) * HROW + 16 * j);       // (hi x 4, lo x 4)
            bh[kk] = hp[kt >= CK ? kt - CK : 0]; bl[kk] = hp[4 + (kt >= CK ? kt - CK : 0)];
          }
        }
        const h16x8 bs = bh * dn8;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][kt] = mfma_h(Aol[nt], bs, acc[nt][kt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][kt] = mfma_h(Aoh[nt], bl, acc[nt][kt]);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][kt] = mfma_h(Aoh[nt], bh, acc[nt][kt]);
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {             // unit 4j + w of the chunk's own positions: (hi, lo) at halves w, 4 + w
        const _Float16* hu = reinterpret_cast<const _Float16*>(shu + (8 * q + kk) * HROW + 16 * j);
        Bu[kk] = hu[w];
        Bul[kk] = hu[4 + w];
      }
      Bus = Bu * dn8;
      fxq = ld4(reinterpret_cast<const float*>(&SX[sbuf][(16 * fsb + fjj) * UROW + 4 * fcol]));
      frq = ld4(reinterpret_cast<const float*>(&SR[sbuf][(16 * fsb + fjj) * UROW + 4 * fcol]));
      sbuf ^= 1;                                     // the next iteration's chunk sits in the other buffer
    } else {
    // ---- weight gradients: 4 gate tiles x (CK + 4) column tiles, K = 32 positions ----
    // u (LayerNorm output) and h_prev enter as single fp16 terms: like the dgates they multiply, they carry 2^-12
    // relative rounding noise, unbiased and averaged over millions of positions in these sums
    h16x8 Bop[KT];
#pragma unroll
    for (int kt = 0; kt < CK; ++kt)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if constexpr (U16) Bop[kt][kk] = cur.uh[kt][kk]; else Bop[kt][kk] = (_Float16)cur.uv[kt][kk];
      }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        if constexpr (HS16) Bop[CK + kt][kk] = cur.hh4[kk][kt]; else Bop[CK + kt][kk] = (_Float16)cur.h4[kk][kt];
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      h16x8 Aop;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) Aop[kk] = cur.a4[kk][nt];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr)         // bias gradient: running sum of the 8 dgates (2 per v_dot2)
        csum[nt] = __builtin_amdgcn_fdot2(h16x2{Aop[2 * pr], Aop[2 * pr + 1]}, ones, csum[nt], false);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) acc[nt][kt] = mfma_h(Aop, Bop[kt], acc[nt][kt]);
    }
    // ---- dU = W_ih^T dgates for the two 16-position sub-tiles ----
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      f32x4 du[CK];
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) du[ct] = zero4();
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) {
          du[ct] = mfma_h(Awt[ct][m].lo, cur.d8[sb][m], du[ct]);
          du[ct] = mfma_h(Awt[ct][m].hi, cur.d8[sb][m], du[ct]);
        }
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) st4(&R[buf][w][sb][ct][lane][0], du[ct]);
    }
    if constexpr (LINW) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) Bu[kk] = cur.hu[kk][w];
    }
    fxq = cur.xq; frq = cur.rq;
    }
    if constexpr (LINW) {
      const bool valid = fact && cp0 + 16 * fsb + fjj < cpe;
      if (fact) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = valid ? frq[r] : 0.f;
          dlb[r] += v;
          const _Float16 hh = (_Float16)(v * gS);
          DY[buf][fcol + r][16 * fsb + fjj] = hh;
          if constexpr (XPS) DYL[buf][fcol + r][16 * fsb + fjj] = (_Float16)__builtin_fmaf((float)hh, -kLoUp, v * gS * kLoUp);
        }
      }
    }
    __syncthreads();
    if constexpr (LINW) {
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) {
        const h16x8 dyh = *reinterpret_cast<const h16x8*>(&DY[buf][16 * ct + j][8 * q]);
        if constexpr (XPS) {
          lacc[ct] = mfma_h(*reinterpret_cast<const h16x8*>(&DYL[buf][16 * ct + j][8 * q]), Bus, lacc[ct]);
          lacc[ct] = mfma_h(dyh, Bul, lacc[ct]);
        }
        lacc[ct] = mfma_h(dyh, Bu, lacc[ct]);
      }
    }
    if constexpr (LNB) {
      const int rl = fqr * 16 + fjj;
      const int pj = cp0 + 16 * fsb + fjj;
      const bool valid = fact && pj < cpe;
      const f32x4 du4 = (ld4(&R[buf][0][fsb][fct][rl][0]) + ld4(&R[buf][1][fsb][fct][rl][0]) + ld4(&R[buf][2][fsb][fct][rl][0]) +
                         ld4(&R[buf][3][fsb][fct][rl][0])) * invS;
      const f32x4 x4 = fxq;
      const float mean = grp_sum(fact ? x4[0] + x4[1] + x4[2] + x4[3] : 0.f) * (1.0f / C);
      f32x4 xh;
      float sq = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { xh[r] = x4[r] - mean; sq += xh[r] * xh[r]; }
      const float rstd = 1.0f / sqrtf(grp_sum(fact ? sq : 0.f) * (1.0f / C) + 1e-5f);
      float m1 = 0.f, m2 = 0.f;
      f32x4 gg;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xh[r] *= rstd;
        const float g = valid ? du4[r] : 0.f;
        dgam[r] += g * xh[r];
        dbet[r] += g;
        gg[r] = g * lgam[r];
        m1 += gg[r];
        m2 += gg[r] * xh[r];
      }
      m1 = grp_sum(m1) * (1.0f / C);
      m2 = grp_sum(m2) * (1.0f / C);
      f32x4 dx4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dx4[r] = rstd * (gg[r] - m1 - xh[r] * m2) + frq[r];
        if (valid) amax = fmaxf(amax, fabsf(dx4[r]));
      }
      if (valid) st4(a.dx + (int64_t)pj * C + fcol, dx4);
    } else
    {   // 2 sub-tiles x CK channel tiles = 2*CK (<= 4) reductions: one per wave
      const int sb = w / CK, ct = w % CK;
      const int pj = cp0 + 16 * sb + j;
      if (w < 2 * CK && pj < cpe) {
        const f32x4 s4 = ld4(&R[buf][0][sb][ct][lane][0]) + ld4(&R[buf][1][sb][ct][lane][0]) +
                         ld4(&R[buf][2][sb][ct][lane][0]) + ld4(&R[buf][3][sb][ct][lane][0]);
        st4(a.du_part + ((int64_t)pj * ndir + dir) * C + 16 * ct + 4 * q, s4 * invS);
      }
    }
    cur = nxt;
    ch = cn;
    sc = sn; sn = s2;
  }
  }

  constexpr int Ktot = C + H;
  float* part = a.scratch + (SLAB ? (size_t)a.row_base + blockIdx.x : (size_t)dir * gridDim.x + blockIdx.x) *
                                ((size_t)4 * H * Ktot + 4 * H + (LNB ? 2 * C : 0) + (LINW ? C * H + C : 0));
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gate = 64 * w + 4 * (4 * q + r) + nt;
#pragma unroll
      for (int kt = 0; kt < CK; ++kt) part[(size_t)gate * Ktot + (CK == 2 ? 2 * j + kt : j)] = acc[nt][kt][r] * invS;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) part[(size_t)gate * Ktot + C + 4 * j + kt] = acc[nt][CK + kt][r] * invS;
    }
    const float cs = quad_sum(XPS ? __builtin_fmaf(csumx[nt], kLoDn, csum[nt]) : csum[nt]);
    if (q == 0) part[(size_t)4 * H * Ktot + 64 * w + 4 * j + nt] = cs * invS;
  }
  if constexpr (LNB) {
    // d(ln_g), d(ln_b): lanes with the same channel quad (lane >> 3) differ in lane bits 0..2 and in the wave
    __syncthreads();
    float* red = &R[0][0][0][0][0][0];                // [wave][2][C]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float g = dgam[r], b = dbet[r];
      g += __shfl_xor(g, 1, 64); g += __shfl_xor(g, 2, 64); g += __shfl_xor(g, 4, 64);
      b += __shfl_xor(b, 1, 64); b += __shfl_xor(b, 2, 64); b += __shfl_xor(b, 4, 64);
      if ((lane & 7) == 0 && fact) { red[(w * 3 + 0) * C + fcol + r] = g; red[(w * 3 + 1) * C + fcol + r] = b; }
      if constexpr (LINW) {
        float l = dlb[r];
        l += __shfl_xor(l, 1, 64); l += __shfl_xor(l, 2, 64); l += __shfl_xor(l, 4, 64);
        if ((lane & 7) == 0 && fact) red[(w * 3 + 2) * C + fcol + r] = l;
      }
    }
    __syncthreads();
    if (tid < (LINW ? 3 : 2) * C) {
      const int which = tid / C, c = tid % C;
      const float v = red[(0 * 3 + which) * C + c] + red[(1 * 3 + which) * C + c] + red[(2 * 3 + which) * C + c] +
                      red[(3 * 3 + which) * C + c];
      // layout of the extras: d(ln_g) [C], d(ln_b) [C], d(lin_w) [C, 64], d(lin_b) [C]
      part[(size_t)4 * H * Ktot + 4 * H + (which < 2 ? tid : 2 * C + C * H + c)] = v;
    }
    if constexpr (LINW) {
#pragma unroll
      for (int ct = 0; ct < CK; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          part[(size_t)4 * H * Ktot + 4 * H + 2 * C + (size_t)(16 * ct + 4 * q + r) * H + 4 * j + w] = lacc[ct][r] * invS;
    }
    if (a.absmax_out) {                              // one atomic per workgroup
      for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
      __shared__ float wm[4];
      if (lane == 0) wm[w] = amax;
      __syncthreads();
      if (tid == 0) atomicMax(reinterpret_cast<unsigned*>(a.absmax_out),
                              __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
    }
  }
}

// extra column ranges of the partial rows that ride along in the same launch (the fused backward's Linear / LayerNorm
// parameter gradients): columns [off, off + n) are summed into out[0 .. n)
struct ReduceExtras { int count; int off[4]; int n[4]; float* out[4]; };

__global__ __launch_bounds__(256) void stream_reduce_kernel(const float* __restrict__ partials_in, int rows, int C,
                                                            float* __restrict__ dW1, float* __restrict__ dW2,
                                                            float* __restrict__ db1, float* __restrict__ db2,
                                                            int64_t ld = 0, ReduceExtras ex = ReduceExtras{}) {
  const int Ktot = C + H, N = 4 * H;
  const int total_ = N * Ktot + N;
  // rows are `ld` floats apart (0: packed); the index arithmetic below uses `total` as the row pitch
  const int64_t total = ld > 0 ? ld : total_;
  const float* __restrict__ partials = partials_in;
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total_) {                                  // threads past the LSTM part: the extra ranges, back to back
    int e = gi - total_;
    for (int x = 0; x < ex.count; ++x) {
      if (e < ex.n[x]) {
        const int rper = (rows + gridDim.y - 1) / gridDim.y;
        const int r0 = blockIdx.y * rper, r1 = min(rows, r0 + rper);
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += partials[(size_t)r * total + ex.off[x] + e];
        atomicAdd(ex.out[x] + e, s);
        return;
      }
      e -= ex.n[x];
    }
    return;
  }
  const int i = gi;
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
    if (k < C) atomicAdd(dW1 + (size_t)n * C + k, s);
    else atomicAdd(dW2 + (size_t)n * H + (k - C), s);
  } else {
    atomicAdd(db1 + (i - N * Ktot), s);
    atomicAdd(db2 + (i - N * Ktot), s);
  }
}

// LayerNorm (+ optional PReLU) backward over C channels per position, 16 lanes per position.
//   g = sum_dir du_part[p, dir, :]   (gradient w.r.t. the LN output)
//   x = xin[p] (pre-LN input; PReLU(xin) when prelu_a)        out = LN-bwd(g) (* prelu') (+ res[p])
template <int C>
__global__ __launch_bounds__(256) void ln_bwd_kernel(sb_ln_bwd_args a) {
  constexpr int VPT = C / 16;
  const int tid = threadIdx.x, cpart = tid & 15;
  const float alpha = a.prelu_a ? a.prelu_a[0] : 0.f;
  float gam[VPT], dgam[VPT], dbet[VPT], dalpha = 0.f;
#pragma unroll
  for (int v = 0; v < VPT; ++v) { gam[v] = a.ln_g[cpart * VPT + v]; dgam[v] = 0.f; dbet[v] = 0.f; }
  const int64_t nrow = a.P;
  float amax = 0.f;                                  // max |out| (a.absmax_out)
  // (the rows of the NEXT position are requested before this one is worked on -- the sums below are dependent DPP chains; round 4:
  //  the same change took the fused LayerNorm + FiLM backward from 0.247 to 0.222 ms)
  const int64_t pstride = (int64_t)gridDim.x * 16;
  float ng[VPT], nraw[VPT], nres[VPT];
  auto fetch = [&](int64_t p) {
    const int64_t pc = p < nrow ? p : nrow - 1;
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int c = cpart * VPT + v;
      float s = 0.f;
      // (read-once streams: non-temporal loads, as in the fused LayerNorm + FiLM backward)
      for (int d = 0; d < a.ndir; ++d) s += ld1_once(a.du_part + (pc * a.ndir + d) * C + c);
      ng[v] = s;
      nraw[v] = ld1_once(a.xin + pc * C + c);
      nres[v] = a.res ? ld1_once(a.res + pc * C + c) : 0.f;
    }
  };
  fetch((int64_t)blockIdx.x * 16 + (tid >> 4));
  for (int64_t p = (int64_t)blockIdx.x * 16 + (tid >> 4); p < nrow; p += pstride) {
    float g[VPT], raw[VPT], x[VPT], rsd[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      g[v] = ng[v];
      raw[v] = nraw[v];
      rsd[v] = nres[v];
      x[v] = (a.prelu_a && raw[v] <= 0.f) ? alpha * raw[v] : raw[v];
    }
    fetch(p + pstride);
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) sum += x[v];
    const float mean = row16_sum(sum) * (1.0f / C);
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPT; ++v) { const float d = x[v] - mean; sq += d * d; }
    const float rstd = 1.0f / sqrtf(row16_sum(sq) * (1.0f / C) + 1e-5f);
    float m1 = 0.f, m2 = 0.f, xh[VPT], gg[VPT];
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      xh[v] = (x[v] - mean) * rstd;
      dgam[v] += g[v] * xh[v];
      dbet[v] += g[v];
      gg[v] = g[v] * gam[v];
      m1 += gg[v];
      m2 += gg[v] * xh[v];
    }
    m1 = row16_sum(m1) * (1.0f / C);
    m2 = row16_sum(m2) * (1.0f / C);
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int c = cpart * VPT + v;
      float dx = rstd * (gg[v] - m1 - xh[v] * m2);
      if (a.prelu_a && raw[v] <= 0.f) { dalpha += dx * raw[v]; dx *= alpha; }
      dx += rsd[v];
      a.out[p * C + c] = dx;
      amax = fmaxf(amax, fabsf(dx));
    }
  }
  // per-block partials: [2C + 1]
  __shared__ float red[16][2 * C + 1];
  const int rowi = tid >> 4;
#pragma unroll
  for (int v = 0; v < VPT; ++v) { red[rowi][cpart * VPT + v] = dgam[v]; red[rowi][C + cpart * VPT + v] = dbet[v]; }
  const float da = row16_sum(dalpha);
  if (cpart == 0) red[rowi][2 * C] = da;
  __syncthreads();
  if (tid < 2 * C + 1) {
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += red[r][tid];
    a.partials[(size_t)blockIdx.x * (2 * C + 1) + tid] = s;
  }
  if (a.absmax_out) {                                // one atomic per workgroup
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    __shared__ float wm[4];
    if ((tid & 63) == 0) wm[tid >> 6] = amax;
    __syncthreads();
    if (tid == 0) atomicMax(reinterpret_cast<unsigned*>(a.absmax_out),
                            __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
  }
}

}  // namespace

extern "C" int sb_lstm_stream_grid(int64_t positions) {
  const int64_t t = (positions + 15) / 16;
  return (int)(t < STREAM_MAX_WG ? (t < 1 ? 1 : t) : STREAM_MAX_WG);
}

int sb_launch_stream_reduce(const float* partials, int rows, int64_t ld, int C, float* dW_ih, float* dW_hh, float* db_ih,
                            float* db_hh, hipStream_t st, int n_extra, const int* ex_off, const int* ex_n,
                            float* const* ex_out);

extern "C" int sb_lstm_bwd_stream(const sb_lstm_stream_args* ap, void* stream) {
  if (!ap || ap->P <= 0 || (ap->ndir != 1 && ap->ndir != 2)) return -1001;
  if (ap->C != 16 && ap->C != 32) return -1002;
  hipStream_t st = (hipStream_t)stream;
  const int gx = sb_lstm_stream_grid(ap->P);
  dim3 grid(gx, ap->ndir), block(256);
  const bool sm = ap->seg_len < 32;
#define SB_S(K, CC) do { if (sm) hipLaunchKernelGGL((K<CC, true>), grid, block, 0, st, *ap); \
                         else hipLaunchKernelGGL((K<CC, false>), grid, block, 0, st, *ap); } while (0)
#define SB_SH(CC, U, HH) do { if (sm) hipLaunchKernelGGL((lstm_bwd_stream_f16_kernel<CC, true, U, HH>), grid, block, 0, st, *ap); \
                              else hipLaunchKernelGGL((lstm_bwd_stream_f16_kernel<CC, false, U, HH>), grid, block, 0, st, *ap); } while (0)
#define SB_SHC(CC) do { if (ap->u_f16 && ap->hs_f16) SB_SH(CC, true, true); else if (ap->u_f16) SB_SH(CC, true, false); \
                        else SB_SH(CC, false, false); } while (0)
  if ((ap->u_f16 || ap->hs_f16) && (!ap->gmax || !ap->u_f16)) return -1003;      // fp16 hs comes with fp16 u
  const bool lnb = ap->dx != nullptr;
  if (lnb) {                                         // fused LayerNorm backward: single direction, fp16 side outputs
    if (ap->ndir != 1 || !ap->gmax || !ap->u_f16 || !ap->hs_f16 || !ap->ln_x || !ap->ln_g || !ap->ln_res || !ap->d_ln_g ||
        !ap->d_ln_b)
      return -1003;
    const bool linw = ap->d_lin_w != nullptr;
    if (linw && !ap->d_lin_b) return -1003;
#define SB_SL(CC, LW) do { if (sm) hipLaunchKernelGGL((lstm_bwd_stream_f16_kernel<CC, true, true, true, true, LW>), grid, block, 0, st, *ap); \
                           else hipLaunchKernelGGL((lstm_bwd_stream_f16_kernel<CC, false, true, true, true, LW>), grid, block, 0, st, *ap); } while (0)
    if (linw) { if (ap->C == 32) SB_SL(32, true); else SB_SL(16, true); }
    else { if (ap->C == 32) SB_SL(32, false); else SB_SL(16, false); }
#undef SB_SL
    SB_CHECK_LAUNCH();
    const int tot = 4 * H * (ap->C + H) + 4 * H, Cc = ap->C;
    const int ex_off[4] = {tot, tot + Cc, tot + 2 * Cc, tot + 2 * Cc + Cc * H}, ex_n[4] = {Cc, Cc, Cc * H, Cc};
    float* const ex_out[4] = {ap->d_ln_g, ap->d_ln_b, ap->d_lin_w, ap->d_lin_b};
    return sb_launch_stream_reduce(ap->scratch, gx, (int64_t)tot + 2 * Cc + (linw ? Cc * H + Cc : 0), Cc, ap->dW_ih[0],
                                   ap->dW_hh[0], ap->db_ih[0], ap->db_hh[0], st, linw ? 4 : 2, ex_off, ex_n, ex_out);
  }
  if (ap->gmax) { if (ap->C == 32) SB_SHC(32); else SB_SHC(16); }
  else if (ap->split_bf16) { if (ap->C == 32) SB_S(lstm_bwd_stream_bf16_kernel, 32); else SB_S(lstm_bwd_stream_bf16_kernel, 16); }
  else { if (ap->C == 32) SB_S(lstm_bwd_stream_kernel, 32); else SB_S(lstm_bwd_stream_kernel, 16); }
#undef SB_SHC
#undef SB_SH
#undef SB_S
  SB_CHECK_LAUNCH();
  const int total = 4 * H * (ap->C + H) + 4 * H;
  for (int d = 0; d < ap->ndir; ++d) {
    hipLaunchKernelGGL(stream_reduce_kernel, dim3((total + 255) / 256, gx >= 64 ? 16 : 1), dim3(256), 0, st,
                       ap->scratch + (size_t)d * gx * total, gx, ap->C, ap->dW_ih[d], ap->dW_hh[d], ap->db_ih[d],
                       ap->db_hh[d]);
  }
  SB_CHECK_LAUNCH();
  return 0;
}

// ---- overlapped schedules: the side stream (see the header) ----
namespace {
// The runtime multiplexes HIP streams over a few hardware queues, and not every pair of queues gives real concurrency:
// measured on MI355X / ROCm 7.2, a side stream created after torch had used another stream of its pool (graph capture is
// enough) landed on a queue whose kernels made no progress until the main stream's kernel had drained -- the "overlapped"
// step ran at 434-552 instead of 560 (plain) / 590 (overlapped, fresh process) utterances/s, although spin-wait probe
// kernels on the same pair of streams did see each other.  So a candidate side stream is TIMED against the caller's
// stream with the shape of the real thing and the choreography of the real calls (fork event, main launch, side launch,
// join): a fixed amount of dependent arithmetic in 9/16 of the CUs' worth of one-per-CU workgroups (96 KB of LDS each) on
// the caller's stream and in 3/8 of the CUs' worth on the candidate; the pair must take less than 0.7 of the two solo
// times added up.  Up to 8 candidates, each created while the rejected ones are still alive so that it lands on the next
// queue; none passing = no overlap on this stream (sb_overlap_available() == 0, the overlapped calls return -1009).
__global__ __launch_bounds__(256) void probe_busy_kernel(float* sink, int iters) {
  __shared__ float pad[24 * 1024];
  pad[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  float x = pad[(threadIdx.x * 7) & 255] * 1e-3f + 1.0f;
  for (int i = 0; i < iters; ++i) x = __builtin_fmaf(x, 0.999999f, 1e-7f);
  if (x == 123.456f) sink[0] = x + pad[3];       // never true: keeps the chain
}

// One entry per (device, caller stream), created by sb_overlap_init and kept until sb_overlap_shutdown: a side stream that
// passed the probe is never destroyed or re-created behind the caller's back, every entry owns its fork / join events, and
// the table is guarded by a mutex (the library is called from one thread per process in this product, but the header
// promises thread safety).  The overlapped entry points only LOOK UP: no allocation, no synchronisation, no probe inside a
// data-path call.
struct SideStream { int dev = -1; hipStream_t main = nullptr, s = nullptr; hipEvent_t fork = nullptr, join = nullptr;
                    hipEvent_t dfork = nullptr, djoin = nullptr;   // deferred small launches (sb_overlap_side_fork / sb_overlap_join)
                    bool ok = false; };
constexpr int kMaxSide = 64;
SideStream g_side[kMaxSide];
int g_nside = 0;
std::mutex g_side_mu;

SideStream* side_lookup_locked(int dev, hipStream_t main_st) {
  for (int i = 0; i < g_nside; ++i)
    if (g_side[i].dev == dev && g_side[i].main == main_st) return &g_side[i];
  return nullptr;
}
SideStream* side_stream(hipStream_t main_st) {       // data path: look-up only
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream* t = side_lookup_locked(dev, main_st);
  return (t && t->ok && t->s) ? t : nullptr;
}
// Measurement aid (sb_overlap_time_next_side_launch): caller-owned events recorded on the side stream straight in front of and
// behind the NEXT kernel an overlapped entry point places there -- the only way to time such a launch with HIP events on the
// stream it runs on (bench.py's live roofline; the caller's own stream never sees it).  One-shot, process-wide, data-path cost
// when disarmed: one relaxed load.
std::atomic<int> g_side_timer_armed{0};
hipEvent_t g_side_timer_ev[2] = {nullptr, nullptr};
inline void side_timer_mark(hipStream_t side, int which) {
  if (!g_side_timer_armed.load(std::memory_order_relaxed)) return;
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (!g_side_timer_armed.load(std::memory_order_relaxed) || !g_side_timer_ev[which]) return;
  (void)hipEventRecord(g_side_timer_ev[which], side);
  if (which == 1) { g_side_timer_armed.store(0, std::memory_order_relaxed); g_side_timer_ev[0] = g_side_timer_ev[1] = nullptr; }
}
// the timed pair: `a` on the caller's stream, `b` on `side` (fork / join choreography of the real calls) or, with
// side == nullptr, behind `a` on the caller's stream.  ms, or < 0 on error.  Synchronises the caller's stream.
float probe_timed(hipStream_t main_st, hipStream_t side, hipEvent_t fork, hipEvent_t join, hipEvent_t e0, hipEvent_t e1,
                  float* buf, int ga, int gb, int iters) {
  if (hipEventRecord(e0, main_st) != hipSuccess) return -1.f;
  if (side && hipEventRecord(fork, main_st) != hipSuccess) return -1.f;
  hipLaunchKernelGGL(probe_busy_kernel, dim3(ga), dim3(256), 0, main_st, buf, iters);
  if (side) {
    if (hipStreamWaitEvent(side, fork, 0) != hipSuccess) return -1.f;
    hipLaunchKernelGGL(probe_busy_kernel, dim3(gb), dim3(256), 0, side, buf, iters);
    if (hipEventRecord(join, side) != hipSuccess || hipStreamWaitEvent(main_st, join, 0) != hipSuccess) return -1.f;
  } else {
    hipLaunchKernelGGL(probe_busy_kernel, dim3(gb), dim3(256), 0, main_st, buf, iters);
  }
  float ms = -1.f;
  if (hipEventRecord(e1, main_st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
      hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
    return -1.f;
  return ms;
}
int device_cus() {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return n;
}
}  // namespace

// 1 when sb_overlap_init found (and sb_overlap_reprobe has not since lost) a concurrent side stream for `stream`
extern "C" int sb_overlap_available(void* stream) { return side_stream((hipStream_t)stream) != nullptr ? 1 : 0; }

extern "C" int sb_overlap_init(void* stream, float* scratch, float* timings_ms) {
  hipStream_t main_st = (hipStream_t)stream;
  if (!scratch) return -1001;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1009;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream* t = side_lookup_locked(dev, main_st);
  if (t) return t->ok ? 1 : 0;                        // probed before: the verdict stands until sb_overlap_reprobe
  if (g_nside >= kMaxSide) return 0;
  t = &g_side[g_nside];
  *t = SideStream{};
  t->dev = dev; t->main = main_st;
  ++g_nside;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (cus < 32 || hipEventCreateWithFlags(&t->fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&t->join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&t->dfork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&t->djoin, hipEventDisableTiming) != hipSuccess || hipEventCreate(&e0) != hipSuccess ||
      hipEventCreate(&e1) != hipSuccess)
    return 0;
  const int ga = cus * 9 / 16, gb = cus * 3 / 8, iters = 12000;          // ~0.2 ms each
  (void)probe_timed(main_st, nullptr, t->fork, t->join, e0, e1, scratch, ga, gb, iters);      // warm-up (code object load)
  const float solo = probe_timed(main_st, nullptr, t->fork, t->join, e0, e1, scratch, ga, gb, iters);
  float best_pair = -1.f;
  hipStream_t rejected[8];
  int nrej = 0;
  for (int c = 0; c < 8 && !t->s && solo > 0.f; ++c) {
    hipStream_t cand = nullptr;
    // developer experiment for the stalled forward producer: SB_SIDE_STREAM_PRIORITY=low|high creates the
    // side stream -- which carries the POLLING launches of the overlapped schedules -- at the device's least / greatest stream
    // priority instead of the default.  Unset (the product): no priority, as in every measurement so far.
    const char* pr = getenv("SB_SIDE_STREAM_PRIORITY");
    int least = 0, greatest = 0;
    if (pr && (pr[0] == 'l' || pr[0] == 'h') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) {
      if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, pr[0] == 'l' ? least : greatest) != hipSuccess) break;
    } else if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
    (void)probe_timed(main_st, cand, t->fork, t->join, e0, e1, scratch, ga, gb, iters);
    const float pair = probe_timed(main_st, cand, t->fork, t->join, e0, e1, scratch, ga, gb, iters);
    if (best_pair < 0.f || (pair > 0.f && pair < best_pair)) best_pair = pair;
    if (pair > 0.f && pair < 0.7f * solo) t->s = cand; else rejected[nrej++] = cand;
  }
  for (int i = 0; i < nrej; ++i) (void)hipStreamDestroy(rejected[i]);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (timings_ms) { timings_ms[0] = solo; timings_ms[1] = best_pair; }
  t->ok = t->s != nullptr;
  return t->ok ? 1 : 0;
}

extern "C" int sb_overlap_reprobe(void* stream, float* scratch, float* timings_ms) {
  hipStream_t main_st = (hipStream_t)stream;
  if (!scratch) return -1001;
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1009;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream* t = side_lookup_locked(dev, main_st);
  if (!t || !t->s) return 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1009;
  const int ga = cus * 9 / 16, gb = cus * 3 / 8, iters = 12000;
  // One measurement is two 0.2 ms kernels: a host hiccup of 0.1 ms between the two launches (the call comes straight out of an
  // epoch's Python) reads as pair = 0.29-0.33 ms against 0.40 back-to-back -- a FALSE loss, seen about once per 3 000 calls in
  // round 5's stress runs, after which the rest of the run silently took the plain order.  So a failing measurement is repeated
  // (up to 4 in all, the first doubling as the warm-up of the others) and the best pair counts: a side stream that really
  // serialises fails every time (pair >= back-to-back), a hiccup does not repeat.
  float solo = -1.f, pair = -1.f;
  bool ok = false;
  for (int attempt = 0; attempt < 4 && !ok; ++attempt) {
    const float s1 = probe_timed(main_st, nullptr, t->fork, t->join, e0, e1, scratch, ga, gb, iters);
    const float p1 = probe_timed(main_st, t->s, t->fork, t->join, e0, e1, scratch, ga, gb, iters);
    if (s1 <= 0.f || p1 <= 0.f) { solo = s1; pair = p1; break; }
    if (solo < 0.f || s1 < solo) solo = s1;
    if (pair < 0.f || p1 < pair) pair = p1;
    ok = pair < 0.7f * solo;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (timings_ms) { timings_ms[0] = solo; timings_ms[1] = pair; }
  t->ok = ok;
  return t->ok ? 1 : 0;
}

// Measurement aid: make the overlapped entry points take their overlapped code paths on `stream` WHATEVER the timed probe said
// -- under `rocprofv3 --pmc` every dispatch is serialised, the probe fails and the shipped kernels (cross-pass producer /
// consumer, ordered forward consumer, slab pair) would never be seen by the counters.  Serialised, they still complete: every
// producer is enqueued in front of its consumers, whose waits are bounded (watchdog word).  Creates the side stream if the
// probe left none.  1 on success.  Never used by the product.
extern "C" int sb_overlap_force(void* stream) {
  hipStream_t main_st = (hipStream_t)stream;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1009;
  std::lock_guard<std::mutex> lk(g_side_mu);
  SideStream* t = side_lookup_locked(dev, main_st);
  if (!t || !t->fork || !t->join || !t->dfork || !t->djoin) return 0;          // sb_overlap_init first
  if (!t->s && hipStreamCreateWithFlags(&t->s, hipStreamNonBlocking) != hipSuccess) { t->s = nullptr; return 0; }
  t->ok = true;
  return 1;
}

extern "C" int sb_overlap_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_side_mu);
  for (int i = 0; i < g_nside; ++i) {
    if (g_side[i].s) (void)hipStreamDestroy(g_side[i].s);
    if (g_side[i].fork) (void)hipEventDestroy(g_side[i].fork);
    if (g_side[i].join) (void)hipEventDestroy(g_side[i].join);
    if (g_side[i].dfork) (void)hipEventDestroy(g_side[i].dfork);
    if (g_side[i].djoin) (void)hipEventDestroy(g_side[i].djoin);
    g_side[i] = SideStream{};
  }
  g_nside = 0;
  return 0;
}

// Small launches off the critical path (the partial-row reductions between two blocks' backward kernels: their results are
// only read by the optimiser).  sb_overlap_side_fork: the side stream of `stream` waits for everything enqueued on `stream` so
// far; *side receives its handle -- launches the caller then enqueues there run beside what follows on `stream`.
// sb_overlap_join: `stream` waits for everything enqueued on its side stream so far (a no-op without one).  Memory such a
// launch touches must stay allocated until the join has been enqueued.
extern "C" int sb_overlap_side_fork(void* stream, void** side) {
  if (!side) return -1001;
  hipStream_t main_st = (hipStream_t)stream;
  SideStream* ss = side_stream(main_st);
  if (!ss || !ss->dfork) return -1009;
  if (hipEventRecord(ss->dfork, main_st) != hipSuccess || hipStreamWaitEvent(ss->s, ss->dfork, 0) != hipSuccess) return -1009;
  *side = (void*)ss->s;
  return 0;
}
extern "C" int sb_overlap_join(void* stream) {
  hipStream_t main_st = (hipStream_t)stream;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1009;
  SideStream* t;
  {
    std::lock_guard<std::mutex> lk(g_side_mu);
    t = side_lookup_locked(dev, main_st);
  }
  if (!t || !t->s || !t->djoin) return 0;              // (also when the probe has since failed: pending work is still joined)
  if (hipEventRecord(t->djoin, t->s) != hipSuccess || hipStreamWaitEvent(main_st, t->djoin, 0) != hipSuccess) return -1009;
  return 0;
}

// ---- flag memory of the guarded schedules (round 5) ----
// Every word one workgroup raises and another polls -- slab counters, item counters, per-tile claim / done words, segment flags,
// the watchdog word -- lives in memory the L2s do NOT cache (hipDeviceMallocUncached: MTYPE UC).  The per-XCD L2s are not
// coherent with each other; agent-scope atomics and sc1 loads go to memory only while the line is not resident in the issuing
// XCD's L2, and a line the zero-fill (plain stores) or an earlier poll left there was seen to serve stale counts for as long as
// it stayed: in a 9 600-step training run one overlapped forward in a few thousand had a consumer poll a slab counter that never
// moved (52 of 82) while memory held the full count, until the watchdog gave up -- and the watchdog word itself came back from a
// stale line two epochs after the host had cleared it.  Uncached words have one home; they are a few KB per step, touched by one
// lane per workgroup.  kind: 3 uncached, 1 fine-grained (fallback), 0 ordinary device memory (last resort; the caller is told).
extern "C" int sb_flags_alloc(int64_t bytes, void** ptr, int* kind) {
  if (!ptr || bytes <= 0) return -1001;
  *ptr = nullptr;
  const unsigned tries[2] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained};
  for (int i = 0; i < 2; ++i) {
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, (size_t)bytes, tries[i]) == hipSuccess && p) { *ptr = p; if (kind) *kind = (int)tries[i]; return 0; }
    (void)hipGetLastError();
  }
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)bytes) != hipSuccess || !p) { (void)hipGetLastError(); return -1009; }
  *ptr = p; if (kind) *kind = 0;
  return 0;
}
extern "C" int sb_flags_free(void* ptr) { return (!ptr || hipFree(ptr) == hipSuccess) ? 0 : -1009; }
namespace {
__global__ void flags_zero_kernel(int* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __hip_atomic_store(p + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (write-through: no line stays behind in this XCD's L2)
}
}  // namespace
extern "C" int sb_flags_zero(void* ptr, int64_t n_ints, void* stream) {
  if (!ptr || n_ints <= 0) return -1001;
  hipLaunchKernelGGL(flags_zero_kernel, dim3((unsigned)((n_ints + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int*)ptr, n_ints);
  SB_CHECK_LAUNCH();
  return 0;
}
// host read-back (synchronises `stream`): the watchdog word, and flag arrays for diagnostics
extern "C" int sb_flags_read(const void* ptr, int64_t n_ints, int* host_out, void* stream) {
  if (!ptr || !host_out || n_ints <= 0) return -1001;
  if (hipMemcpyAsync(host_out, ptr, (size_t)n_ints * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return -1009;
  return 0;
}

extern "C" int sb_overlap_time_next_side_launch(void* ev_start, void* ev_stop) {
  std::lock_guard<std::mutex> lk(g_side_mu);
  const int was_armed = g_side_timer_armed.load(std::memory_order_relaxed);
  g_side_timer_ev[0] = (hipEvent_t)ev_start; g_side_timer_ev[1] = (hipEvent_t)ev_stop;
  g_side_timer_armed.store((ev_start && ev_stop) ? 1 : 0, std::memory_order_relaxed);
  return was_armed;                                      // 1: the previous timer never fired (no side launch since it was armed)
}

extern "C" int sb_lstm_overlap_rows(int64_t positions, int nseq) {
  (void)positions;
  const int cus = device_cus(), idle = cus - (nseq + 15) / 16;
  return (idle > 0 ? idle : 0) + cus;
}

// flags layout: [0] recurrence workgroups started, [1] the unit counter, [2], [3] spare, [4 ..] the slab flags
// serial: the same two kernels in plain order on `stream` (recurrence, then ONE stream-kernel launch that finds every flag
// up) -- sb_lstm_bwd_inter_pair_serial, a measurement aid: profilers that serialise kernels (rocprofv3 --pmc) cannot see
// the overlapped pair, and its HBM traffic is the sum of these two launches
static int inter_pair(const sb_lstm_bwd_args* rec_in, const sb_lstm_stream_args* st_in, int* flags, int slab_len, void* stream,
                      bool serial) {
  if (!rec_in || !st_in || !flags) return -1001;
  sb_lstm_bwd_args rec = *rec_in;
  sb_lstm_stream_args sa = *st_in;
  hipStream_t main_st = (hipStream_t)stream;
  const int T = rec.nsteps, ntiles = (rec.nseq + 15) / 16, C = sa.C;
  if (rec.ndir != 1 || sa.ndir != 1 || !sa.dx || !sa.d_lin_w || !sa.d_lin_b || !sa.gmax || !sa.u_f16 || !sa.hs_f16 ||
      !sa.ln_x || !sa.ln_g || !sa.ln_res || !sa.d_ln_g || !sa.d_ln_b || !sa.sched_status || (C != 16 && C != 32) ||
      slab_len < 2 || (slab_len & 1) || sa.shift_pos <= 0 || sa.seg_len != T * sa.shift_pos || sa.P % sa.seg_len != 0 ||
      sa.seg_len < 32 || (sa.wide != 0) != (rec.wide != 0))
    return -1003;
  const int cus = device_cus(), idle = cus - ntiles;
  if (idle < 16) return -1003;
  SideStream* ss = serial ? nullptr : side_stream(main_st);
  if (!ss && !serial) return -1009;
  const int nslabs = (T + slab_len - 1) / slab_len;
  const int nb = (int)(sa.P / sa.seg_len);
  const int cpb = (int)((slab_len * sa.shift_pos + 31) / 32), cps = nb * cpb;
  const int cpl = (int)(((T - (nslabs - 1) * slab_len) * sa.shift_pos + 31) / 32);     // the last slab is shorter
  const int nch = (nslabs - 1) * cps + nb * cpl;
  // next to the recurrence: one workgroup per idle CU (register budget: none fits on a recurrence CU; one that cannot be
  // placed at once starts later and draws fewer units); behind it: one per CU
  const int g1 = serial ? 0 : idle, g2 = cus;

  if (sb_flags_zero(flags, nslabs + 4, main_st) != 0) return -1009;
  if (!serial && hipEventRecord(ss->fork, main_st) != hipSuccess) return -1009;
  rec.slab_flags = flags + 4; rec.slab_len = slab_len; rec.slab_started = flags;
  int rc = sb_lstm_bwd_rec(&rec, stream);
  if (rc) return rc;
  sa.slab_flags = flags + 4; sa.slab_len = slab_len; sa.slab_need = ntiles;
  sa.started = flags; sa.chunk_counter = flags + 1; sa.nchunks = nch;
#define SB_SO(CC, ST, G) do { \
    if (sa.wide) hipLaunchKernelGGL((lstm_bwd_stream_f16_kernel<CC, false, true, true, true, true, true, true>), dim3(G), dim3(256), 0, ST, sa); \
    else hipLaunchKernelGGL((lstm_bwd_stream_f16_kernel<CC, false, true, true, true, true, true>), dim3(G), dim3(256), 0, ST, sa); } while (0)
  if (!serial) {
    if (hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) return -1009;
    sa.guard = 1; sa.row_base = 0;
    side_timer_mark(ss->s, 0);
    if (C == 32) SB_SO(32, ss->s, g1); else SB_SO(16, ss->s, g1);
    SB_CHECK_LAUNCH();
    side_timer_mark(ss->s, 1);
    if (hipEventRecord(ss->join, ss->s) != hipSuccess) return -1009;
  }
  // behind the recurrence on `stream` (all flags up): the same kernel drawing what is left; then the join
  sa.guard = 0; sa.row_base = g1;
  if (C == 32) SB_SO(32, main_st, g2); else SB_SO(16, main_st, g2);
#undef SB_SO
  SB_CHECK_LAUNCH();
  if (!serial && hipStreamWaitEvent(main_st, ss->join, 0) != hipSuccess) return -1009;
  const int tot = 4 * H * (C + H) + 4 * H;
  const int ex_off[4] = {tot, tot + C, tot + 2 * C, tot + 2 * C + C * H}, ex_n[4] = {C, C, C * H, C};
  float* const ex_out[4] = {sa.d_ln_g, sa.d_ln_b, sa.d_lin_w, sa.d_lin_b};
  return sb_launch_stream_reduce(sa.scratch, g1 + g2, (int64_t)tot + 2 * C + C * H + C, C, sa.dW_ih[0], sa.dW_hh[0],
                                 sa.db_ih[0], sa.db_hh[0], main_st, 4, ex_off, ex_n, ex_out);
}

extern "C" int sb_lstm_bwd_inter_overlapped(const sb_lstm_bwd_args* rec_in, const sb_lstm_stream_args* st_in, int* flags,
                                            int slab_len, void* stream) {
  return inter_pair(rec_in, st_in, flags, slab_len, stream, false);
}
extern "C" int sb_lstm_bwd_inter_pair_serial(const sb_lstm_bwd_args* rec_in, const sb_lstm_stream_args* st_in, int* flags,
                                             int slab_len, void* stream) {
  return inter_pair(rec_in, st_in, flags, slab_len, stream, true);
}

// ---- overlapped forward (see the header) ----
// flags layout: [0] producer workgroups started, [1], [2] the consumer's item counters (one per direction), [3] spare,
// [4 .. 4 + 520) the consumer's hand-back block (sb_lstm_fwd_flag_ints), then the slab flags
extern "C" int sb_lstm_fwd_flag_ints(int producer_steps, int slab_len) {
  return slab_len > 0 ? 4 + kOrdCtl + (producer_steps + slab_len - 1) / slab_len : -1001;
}
extern "C" int sb_lstm_fwd_produce(const sb_lstm_fwd_args* a_in, int* flags, int slab_len, void* stream) {
  return sb_lstm_fwd_produce_ex(a_in, flags, slab_len, 0, stream);
}
// flags_zeroed != 0: the caller hands over flags it has zeroed itself, in stream order before the call
extern "C" int sb_lstm_fwd_produce_ex(const sb_lstm_fwd_args* a_in, int* flags, int slab_len, int flags_zeroed, void* stream) {
  if (!a_in || !flags) return -1001;
  sb_lstm_fwd_args a = *a_in;
  hipStream_t main_st = (hipStream_t)stream;
  const int ntiles = (a.nseq + 15) / 16;
  if (a.ndir != 1 || !a.lin_w || slab_len < 4 || (slab_len & 3) || device_cus() - ntiles < 16) return -1003;
  SideStream* ss = side_stream(main_st);
  if (!ss) return -1009;
  const int nslabs = (a.nsteps + slab_len - 1) / slab_len;
  if (!flags_zeroed && sb_flags_zero(flags, 4 + kOrdCtl + nslabs, main_st) != 0) return -1009;
  if (hipEventRecord(ss->fork, main_st) != hipSuccess) return -1009;      // the side stream starts from here
  a.slab_flags = flags + 4 + kOrdCtl; a.slab_len = slab_len; a.tile_order = nullptr; a.tile_need = nullptr;
  a.ord_started = flags;
  return sb_lstm_fwd(&a, stream);
}

extern "C" int sb_lstm_fwd_consume(const sb_lstm_fwd_args* a_in, int* flags, int slab_len, int producer_tiles,
                                   const int* order, const int* need, void* stream) {
  if (!a_in || !flags || !order || !need) return -1001;
  sb_lstm_fwd_args a = *a_in;
  hipStream_t main_st = (hipStream_t)stream;
  const int ntiles = (a.nseq + 15) / 16;
  const int idle = device_cus() - producer_tiles;
  if (a.ndir != 2 || !a.lin_w || !a.sched_status || idle < 16 || slab_len < 4) return -1003;
  SideStream* ss = side_stream(main_st);
  if (!ss) return -1009;
  a.slab_flags = flags + 4 + kOrdCtl; a.slab_len = slab_len; a.slab_need = producer_tiles;
  a.tile_order = order; a.tile_need = need;
  a.ord_started = flags; a.ord_counter = flags + 1; a.ord_ret = flags + 4;
  // next to the producer: two persistent workgroups per idle CU (254 registers each; none fits on a producer's CU); one
  // that cannot be placed at once simply starts later and draws fewer items
  int g1 = 2 * idle;                                   // (1 per CU: -10 % forward-only, 3: no better)
  // developer experiment (unmeasured): workgroups are dealt round the 8 XCDs, the producer's ceil(tiles / 8) on
  // the fullest XCD leave 2 x (CUs per XCD - that) slots there -- SB_FWD_GUARD_XCD_EXACT=1 launches 8 x that many, so that no
  // guarded workgroup ever waits for a slot (82 tiles: 336 instead of 348).  Unset (the product): as measured all round.
  static const bool xcd_exact = [] { const char* e = getenv("SB_FWD_GUARD_XCD_EXACT"); return e && e[0] == '1'; }();
  if (xcd_exact) {
    const int per_xcd = device_cus() / 8 - (producer_tiles + 7) / 8;
    if (per_xcd > 0 && 16 * per_xcd < g1) g1 = 16 * per_xcd;
  }
  if (g1 > 2 * ntiles) g1 = 2 * ntiles;
  if (hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) return -1009;
  a.ord_guard = 1; a.ord_grid = g1;
  side_timer_mark(ss->s, 0);
  int rc = sb_lstm_fwd(&a, ss->s);
  side_timer_mark(ss->s, 1);
  if (rc) return rc;
  if (hipEventRecord(ss->join, ss->s) != hipSuccess) return -1009;
  // behind the producer on `stream` (all flags up): one workgroup per item, each taking what is left; then the join
  a.ord_guard = 0; a.ord_grid = 2 * ntiles;
  rc = sb_lstm_fwd(&a, stream);
  if (hipStreamWaitEvent(main_st, ss->join, 0) != hipSuccess) return -1009;
  return rc;
}

// Test hook: the hand-back path of the overlapped forward staged on ONE stream, no concurrency needed -- (1) the producer has
// "started" but no slab is complete: the guarded launch draws its items, every wait runs out (~2 ms), every item is handed back;
// (2) all `nslabs` slab flags are raised; (3) the launch behind the producer drains the counter and the return stacks.  The
// caller compares y with the plain call's and reads the control block (flags[4 ..]).
namespace {
__global__ void flags_fill_kernel(int* p, int64_t n, int v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) __hip_atomic_store(p + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
}  // namespace
extern "C" int sb_lstm_fwd_consume_staged_test(const sb_lstm_fwd_args* a_in, int* flags, int slab_len, int producer_tiles,
                                               int nslabs, const int* order, const int* need, void* stream) {
  if (!a_in || !flags || !order || !need || nslabs < 1) return -1001;
  sb_lstm_fwd_args a = *a_in;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = (a.nseq + 15) / 16;
  if (a.ndir != 2 || !a.lin_w || !a.sched_status || slab_len < 4) return -1003;
  if (sb_flags_zero(flags, 4 + kOrdCtl + nslabs, st) != 0) return -1009;
  hipLaunchKernelGGL(flags_fill_kernel, dim3(1), dim3(64), 0, st, flags, (int64_t)1, producer_tiles);      // every producer workgroup "started"
  a.slab_flags = flags + 4 + kOrdCtl; a.slab_len = slab_len; a.slab_need = producer_tiles;
  a.tile_order = order; a.tile_need = need;
  a.ord_started = flags; a.ord_counter = flags + 1; a.ord_ret = flags + 4;
  a.ord_guard = 1; a.ord_grid = 2 * ntiles;
  int rc = sb_lstm_fwd(&a, st);
  if (rc) return rc;
  hipLaunchKernelGGL(flags_fill_kernel, dim3((nslabs + 63) / 64), dim3(64), 0, st, flags + 4 + kOrdCtl, (int64_t)nslabs, producer_tiles);
  a.ord_guard = 0; a.ord_grid = 2 * ntiles;
  rc = sb_lstm_fwd(&a, st);
  SB_CHECK_LAUNCH();
  return rc;
}

// ---- backward overlapped across the two passes of a block (see the header) ----
// flags layout as for the forward: [0] producer workgroups started, [1], [2] the consumer's item counters (one per
// direction), [3] spare, [4 ..] the slab flags
static void cross_grids(int nseq, int producer_tiles, int* g1, int* g2) {
  const int ntiles = (nseq + 15) / 16, cus = device_cus();
  int a = (cus - producer_tiles) / 2, b = cus / 2;     // workgroups PER DIRECTION: next to the producer / behind it
  if (a < 1) a = 1;
  if (a > ntiles) a = ntiles;
  if (b < 1) b = 1;
  if (b > ntiles) b = ntiles;
  // multiples of 8 per direction: the consumer's direction is bit 3 of the workgroup index (see the kernel), so that the 16
  // workgroups the dispatcher deals round the XCDs carry both directions to every XCD; a workgroup or two beyond the idle CUs
  // just starts late and draws less
  a = (a + 7) & ~7; b = (b + 7) & ~7;
  *g1 = a; *g2 = b;
}
extern "C" int sb_lstm_bwd_cross_rows(int nseq, int producer_tiles) {
  int g1, g2;
  cross_grids(nseq, producer_tiles, &g1, &g2);
  return 2 * (g1 + g2);
}

extern "C" int sb_lstm_bwd_cross_produce(const sb_lstm_bwd_args* a_in, int* flags, int n_flags, int slab_len, void* stream) {
  return sb_lstm_bwd_cross_produce_ex(a_in, flags, n_flags, slab_len, 0, stream);
}
// flags_zeroed != 0: the caller hands over flags it has zeroed itself (on `stream`, e.g. one fill for all blocks of a step)
extern "C" int sb_lstm_bwd_cross_produce_ex(const sb_lstm_bwd_args* a_in, int* flags, int n_flags, int slab_len, int flags_zeroed,
                                            void* stream) {
  if (!a_in || !flags) return -1001;
  sb_lstm_bwd_args a = *a_in;
  hipStream_t main_st = (hipStream_t)stream;
  const int ntiles = (a.nseq + 15) / 16;
  if (a.ndir != 1 || !a.wide || !a.split || !a.wpart || !a.du || a.dx || slab_len < 2 || (slab_len & 1) ||
      device_cus() - ntiles < 16)
    return -1003;
  SideStream* ss = side_stream(main_st);
  if (!ss) return -1009;
  // flags: [0] producer workgroups started, [1 .. 3] spare, [4 + tile] slabs completed by tile, then the consumer's 16 item
  // counters and three words per CONSUMER tile (prologue claimed / done / max |dy1|)
  if (n_flags < ntiles + 4) return -1003;
  if (!flags_zeroed && sb_flags_zero(flags, n_flags, main_st) != 0) return -1009;
  if (hipEventRecord(ss->fork, main_st) != hipSuccess) return -1009;      // the side stream starts from here
  a.slab_flags = flags + 4; a.slab_len = slab_len; a.slab_started = flags;
  a.seg_state = nullptr; a.seg_flags = nullptr;                           // (no time segments under the producer)
  return sb_lstm_bwd_rec(&a, stream);
}

extern "C" int sb_lstm_bwd_cross_consume(const sb_lstm_bwd_args* a_in, int* flags, int slab_len, int producer_tiles,
                                         const int* order, const int* need, void* stream) {
  return sb_lstm_bwd_cross_consume_ex(a_in, flags, slab_len, producer_tiles, order, need, 0, stream);
}
// reduce_on_side != 0: the partial-row reductions run on the library's side stream behind both consumer launches instead of on
// `stream` (they sit between this block's backward and the next one's otherwise; nothing on the critical path reads their
// results) -- the caller keeps wpart allocated and calls sb_overlap_join(stream) before anything on `stream` reads a gradient target
extern "C" int sb_lstm_bwd_cross_consume_ex(const sb_lstm_bwd_args* a_in, int* flags, int slab_len, int producer_tiles,
                                            const int* order, const int* need, int reduce_on_side, void* stream) {
  if (!a_in || !flags || !order || !need) return -1001;
  sb_lstm_bwd_args a = *a_in;
  hipStream_t main_st = (hipStream_t)stream;
  if (a.ndir != 2 || !a.wide || !a.split || a.C != 32 || !a.dy || a.C_lin != 32 || !a.sched_status || !a.wpart ||
      !a.d_ln_g || !a.d_ln_b || device_cus() - producer_tiles < 16 || slab_len < 2)
    return -1003;
  SideStream* ss = side_stream(main_st);
  if (!ss) return -1009;
  int g1, g2;
  cross_grids(a.nseq, producer_tiles, &g1, &g2);
  a.slab_flags = flags + 4; a.slab_len = slab_len; a.slab_need = producer_tiles; a.slab_started = flags;
  // behind the producer's progress words: 16 item counters (8 XCD queues x 2 directions), then three words per consumer tile
  a.tile_order = order; a.tile_need = need; a.ord_counter = flags + 4 + producer_tiles;
  a.seg_state = nullptr; a.seg_flags = flags + 4 + producer_tiles + 16;  // (no time segments here: the per-tile prologue words)
  // next to the producer: one persistent workgroup (8 waves, 256 registers each: none fits on a producer's CU) per idle CU;
  // one that cannot be placed at once starts later and draws fewer items
  if (hipStreamWaitEvent(ss->s, ss->fork, 0) != hipSuccess) return -1009;
  a.ord_guard = 1; a.ord_grid = g1; a.row_base = 0;
  side_timer_mark(ss->s, 0);
  int rc = sb_lstm_bwd_rec(&a, ss->s);
  side_timer_mark(ss->s, 1);
  if (rc) return rc;
  if (hipEventRecord(ss->join, ss->s) != hipSuccess) return -1009;
  // behind the producer on `stream` (every flag up): the same kernel taking what is left; then the join
  a.ord_guard = 0; a.ord_grid = g2; a.row_base = 2 * g1;
  rc = sb_lstm_bwd_rec(&a, stream);
  if (hipStreamWaitEvent(main_st, ss->join, 0) != hipSuccess) return -1009;
  if (rc) return rc;
  // partial rows: [launch][direction][workgroup]; row = LSTM part, dW_lin [32][128] + db_lin [32], d(ln gamma / beta) [64]
  const int C = a.C;
  const int64_t ld = (int64_t)4 * H * (C + H) + 4 * H + C * 2 * H + C + 2 * C;
  const float* base[2] = {a.wpart, a.wpart + (size_t)2 * g1 * ld};
  const int gx[2] = {g1, g2};
  // the riders' column ranges (dW_lin [32][128], db_lin, d(ln gamma), d(ln beta): sums over the rows of BOTH directions) go along
  // with the four LSTM-part reductions -- each launch adds its rows' share -- instead of four launches of their own: the
  // reductions sit between the consumer and the next block's producer (kernel trace: 8 x ~9 us per block)
  const int o0 = 4 * H * (C + H) + 4 * H;
  const int ex_off[4] = {o0, o0 + C * 2 * H, o0 + C * 2 * H + C, o0 + C * 2 * H + 2 * C};
  const int ex_n[4] = {C * 2 * H, C, C, C};
  float* const ex_out[4] = {a.dW_lin, a.db_lin, a.d_ln_g, a.d_ln_b};
  hipStream_t red_st = main_st;
  if (reduce_on_side && ss->dfork) {                   // behind the second launch too (the first one is on the side stream itself)
    if (hipEventRecord(ss->dfork, main_st) != hipSuccess || hipStreamWaitEvent(ss->s, ss->dfork, 0) != hipSuccess) return -1009;
    red_st = ss->s;
  }
  for (int l = 0; l < 2 && !rc; ++l) {
    rc = sb_launch_stream_reduce(base[l], gx[l], ld, C, a.dW_ih, a.dW_hh, a.db_ih, a.db_hh, red_st, 4, ex_off, ex_n, ex_out);
    if (!rc) rc = sb_launch_stream_reduce(base[l] + (size_t)gx[l] * ld, gx[l], ld, C, a.dW_ih1, a.dW_hh1, a.db_ih1, a.db_hh1, red_st,
                                          4, ex_off, ex_n, ex_out);
  }
  return rc;
}

// shared with the fused backward recurrence (sb_lstm_bf.hip), which emits the same partial rows
int sb_launch_stream_reduce(const float* partials, int rows, int64_t ld, int C, float* dW_ih, float* dW_hh, float* db_ih,
                            float* db_hh, hipStream_t st, int n_extra, const int* ex_off, const int* ex_n,
                            float* const* ex_out) {
  const int total = 4 * H * (C + H) + 4 * H;
  ReduceExtras ex{};
  int extra_cols = 0;
  for (int x = 0; x < n_extra && ex.count < 4; ++x) {
    if (!ex_out[x]) continue;
    ex.off[ex.count] = ex_off[x]; ex.n[ex.count] = ex_n[x]; ex.out[ex.count] = ex_out[x];
    extra_cols += ex_n[x];
    ++ex.count;
  }
  hipLaunchKernelGGL(stream_reduce_kernel, dim3((total + extra_cols + 255) / 256, rows >= 64 ? 16 : 1), dim3(256), 0, st,
                     partials, rows, C, dW_ih, dW_hh, db_ih, db_hh, ld, ex);
  SB_CHECK_LAUNCH();
  return 0;
}

extern "C" int sb_ln_bwd_grid(int64_t positions) {
  const int64_t t = (positions + 15) / 16;
  return (int)(t < 2048 ? (t < 1 ? 1 : t) : 2048);
}

extern "C" int sb_ln_bwd(const sb_ln_bwd_args* ap, void* stream) {
  if (!ap || ap->P <= 0 || ap->ndir < 1) return -1001;
  dim3 grid(sb_ln_bwd_grid(ap->P)), block(256);
  if (ap->C == 32) hipLaunchKernelGGL(ln_bwd_kernel<32>, grid, block, 0, (hipStream_t)stream, *ap);
  else if (ap->C == 16) hipLaunchKernelGGL(ln_bwd_kernel<16>, grid, block, 0, (hipStream_t)stream, *ap);
  else if (ap->C == 64) hipLaunchKernelGGL(ln_bwd_kernel<64>, grid, block, 0, (hipStream_t)stream, *ap);
  else return -1002;
  SB_CHECK_LAUNCH();
  return 0;
}
