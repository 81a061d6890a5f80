// Backward (BPTT) recurrence of the LSTM passes on the 16-bit matrix pipes, with the streaming part of the backward fused in
// (see sb_lstm_bf_common.h for the arithmetic): lstm_bwd_rec_bf_kernel and its launcher.  Forward: sb_lstm_bf_fwd.hip.
#include "sb_lstm_bf_common.h"
#ifndef SB_EXP_CONS          // developer experiments on the cross-pass overlap (0 in the shipped library): bit 0 consumer without its
#define SB_EXP_CONS 0        // prologue loop, bit 1 producer with plain (not write-through) du stores, bit 2 producer without slab signals
#endif

// Phase timing (developer tool): build with -DSB_PHASE_TIMING and pass a scratch buffer (dhs with dy == NULL).
#ifdef SB_PHASE_TIMING
// chunk role of the role-split backward (wave 4 of workgroup 0): ticks per period of [work before the hand-over barrier, wait
// at it, work after it, wait at the second barrier]; read with sb_debug_phase_bwd_split()
__device__ float g_phase_bwd_split[8];
// ... and of the recurrence role's four waves, ticks per STEP: [wait for this step's records, issue of the next records' loads,
// records consumed .. dgates formed, LDS dgates + MFMA, partial sums out, barrier + reduction]; read with sb_debug_phase_bwd_rec()
__device__ float g_phase_bwd_rec[4][6];
extern "C" int sb_debug_phase_bwd_rec(float* host_out) {
  return -(int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_bwd_rec), sizeof(g_phase_bwd_rec));
}
extern "C" int sb_debug_phase_bwd_split(float* host_out) {
  return -(int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_phase_bwd_split), sizeof(g_phase_bwd_split));
}
#endif

namespace {

// ---------------------------------------------------------------------------------------------------------
// BPTT recurrence on the bf16 pipe.  Wave w owns gate rows {g*64 + 16w + 4q + r}; its 16 dgates per lane are the
// B operand straight from registers: K-chunk c covers gate types (2c, 2c+1), k = 8q + kk <-> gate 2c + (kk >> 2),
// unit 16w + 4q + (kk & 3).
// FUSE_LIN (C = channels of dy): the gradient w.r.t. the hidden sequence is not read from memory but formed on
// the fly as dh_ext = W_lin[:, dir*64 + unit]^T dy[pos] (the backward of the Linear that follows the LSTM), one
// 16x16x32 tile per wave with a 2-term split (hi, lo) -- it enters the recurrence additively, un-amplified.
// DG16: dgates leave as fp16 of S * dgates, S = 2^-ceil(log2 max|incoming gradient|) (see sb_lstm_bwd_args.gmax): the
// recurrence is linear in the incoming gradient, so scaling it once at the entry scales everything consistently.
// {(fp16) fma((float) h[0], m, c0), (fp16) fma((float) h[1], m, c1)} of a packed fp16 pair h: one v_fma_mixlo_f16 / v_fma_mixhi_f16
// each -- the fp16 -> fp32 conversion of h, the fp32 fma and the rounding of its result to fp16 in ONE instruction (what hipcc
// emits for the scalar expression; after its SLP pass packed the fp32 halves it wrote two conversions, a v_pk_fma_f32 and a
// v_cvt_pk instead: two instructions more per pair, ~28 a step in the recurrence role, whose own instruction stream is what
// bounds the role-split kernels).  -DSB_FMA_MIX=0: the plain expressions.
#ifndef SB_FMA_MIX
#define SB_FMA_MIX 1
#endif
SB_DEVINL unsigned mix_pair(unsigned h, float m, float c0, float c1) {
#if SB_FMA_MIX
  unsigned d;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h), "s"(m), "v"(c0));
  asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(d) : "v"(h), "s"(m), "v"(c1));
  return d;
#else
  const h16x2 hv = __builtin_bit_cast(h16x2, h);
  const h16x2 r = {(_Float16)__builtin_fmaf((float)hv[0], m, c0), (_Float16)__builtin_fmaf((float)hv[1], m, c1)};
  return __builtin_bit_cast(unsigned, r);
#endif
}
// lo terms of 8 values v with their fp16 heads hi: (fp16)(s v[k] - s (float) hi[k])
SB_DEVINL h16x8 mix_lo8(const h16x8 hi, const float (&v)[8], float s) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 hp = __builtin_bit_cast(u32x4, hi);
  u32x4 d;
#pragma unroll
  for (int p2 = 0; p2 < 4; ++p2) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 c = f32x2{v[2 * p2], v[2 * p2 + 1]} * s;       // (one packed multiply: pinned as a pair, or hipcc splits it for the asm operands)
    asm("" : "+v"(c));
    d[p2] = mix_pair(hp[p2], -s, c[0], c[1]);
  }
  return __builtin_bit_cast(h16x8, d);
}

SB_DEVINL float grad_scale(const float* gmax) {
  const float m = gmax[0];
  return (m > 0.f && m < 3.0e38f) ? exp2f(-ceilf(log2f(m))) : 1.0f;
}

// SEG: (tile, time-segment) work items as in the forward kernel; here a tile is walked from its last step down and
// the state handed from segment to segment is (dc, dh_rec).
// FST (= channels C of u; single-direction passes): the streaming part of the backward (sb_lstm_stream.hip) runs inside
// this kernel.  The fp16 dgates of a step go to an LDS tile instead of HBM; after every second step the workgroup
// multiplies the 32 (step, sequence) rows it holds -- wave w = gate type w, exactly the chunk arithmetic of
// lstm_bwd_stream_f16_kernel -- into running dW_ih / dW_hh / db sums (registers, one partial row per workgroup at the
// end) and into du = W_ih^T dgates (partial sums over the four waves through LDS, stored one step later).  u and
// h_prev arrive as the fp16 side outputs of the forward kernel.  Saves the 512 B/position dgates round trip (the
// store alone was 17-45 % of this kernel) and the whole streaming launch.
#ifndef SB_TR_DGATES
#define SB_TR_DGATES 1     // 0: row-major LDS reads + register shuffles for the chunk role's dgates fragments (rounds 3-5)
#endif
#ifndef SB_EXP_IDX64
#define SB_EXP_IDX64 0      // 1: the 64-bit record / dy index arithmetic of rounds 3-5 (A/B switch)
#endif
constexpr int DGP = 260;      // halves per dgates row in LDS (520 B: the four position groups of a load hit distinct banks)
// LNB (FST == 16): the LayerNorm backward of the block runs in the flush of the du rows (one wave holds all 16 channels
// of its 16 positions), dx = LN-backward(du) + dy goes out instead of du.
// BI (FST > 0, two directions): hs is the fp32 [P, 128] tensor, du goes to [P, 2, C], no Linear / LayerNorm riders;
// persistent workgroups (gridDim.x <= tiles) walk several tiles.
// HS16B (BI): hs is the fp16 [P, 128] side output of the forward kernel's partial-Linear mode (sb_lstm_bwd_args.hs_f16)
// RECOMP (BI + HS16B): the forward kept no gate records; the four gates of a step are recomputed here as
// act(W_ih u_s + W_hh h_{s-1} + b) -- 12 MFMAs per step on the matrix pipe (13 % busy in this kernel) with the weights as
// single fp16 terms in LDS (48 KB, lane order) and u_s / h_{s-1} fetched as fp16 B operands with the other records.
// Halves the bytes the intra-frame forward writes (it is store-bound) and the record bytes read here.
// SLAB (DG16, no FST / SEG, single direction): producer side of the overlapped inter-frame backward
// (sb_lstm_bwd_inter_overlapped).  The dgates rows are stored write-through at agent scope (sc1) and after every
// slab_len steps the workgroup counts itself into slab_flags[k]; the stream kernel, running at the same time on the CUs
// this launch leaves idle, starts on slab k when all tiles have.  Same protocol as the segment hand-off: sc1 accesses
// ordered by s_waitcnt + barrier, no L2-wide release fences.
// XP (FST > 0, DG16, !REC16): WIDE BPTT state (sb_lstm_bwd_args.wide) -- the fused forms at the reference's own precision.
// Records are the blocked fp32 ones of the forward kernel's SAVE == 4, u and hs are fp32, and every gradient quantity that
// meets the fp16 matrix pipe is TWO fp16 terms x = hi + 2^-11 lo', lo' = fp16((x - hi) * 2^11): the low term is scaled
// up so that it cannot underflow however small x is against the power-of-two scale S (which is derived from the LARGEST
// incoming gradient), and its products are accumulated apart and folded in with the factor 2^-11 (recurrence, du) or
// meet an operand that carries the factor (dW: the hi terms of u / h_prev times 2^-11).  u, h_prev and the weights are
// fp16 hi + lo.  Three products per MAC instead of one, dgates tile in LDS twice the size, records twice the bytes.
// SPLIT (FST > 0, no SEG / LNB): ROLE-SPLIT workgroup of 8 waves, two per SIMD.  Waves 0..3 run the recurrence (the serial
// chain: records in, cell backward, W_hh^T dgates, the partial-sum exchange) and leave the dgates of a step in the LDS
// tiles; waves 4..7 run the chunk arithmetic of the PREVIOUS pair of steps (dW_ih / dW_hh / db, du, the Linear's weight
// gradient) from those tiles, u / h_prev from HBM.  Measured on the one-role kernel: the chunk arithmetic is ~70 % of its
// instructions and none of it is on the serial chain, yet a single wave per SIMD issues it in line (SQ: 40 % VALU-active,
// 50 % stalled, nothing to hide behind), and its 500+ registers force ~500 AGPR copies per pair.  Split, each role fits
// 256 registers, the chunk role fills the recurrence role's stalls, and the copies are gone.  Both roles pass the same
// two workgroup barriers per pair: the recurrence's partial-sum exchange barriers double as the hand-over points.
// GREC (SLAB + XP, single direction, C = 32; sb_lstm_bwd_args.recompute with `wide`): the forward pass stored NO gate records
// for this layer -- c_prev and the (hi, lo) pair tensors u / hs only, 512 + 128 of the 1536 bytes per position -- and the four
// gates of a step are recomputed here exactly as the forward kernel formed them: the same scaled weight and bias terms, the
// same three products per MAC in the same order on the same operand terms (u_s and h_{s-1} ARE the forward's own fp16 pairs),
// so the recomputed gates are the forward's bit for bit.  36 MFMAs + 16 activations per step and wave, forward weights in
// registers (96: this kernel runs one wave per SIMD and has them to spare).  The inter-frame forward is store-bound in
// training: without its 1 KB of gates per position it runs 1.10 -> 0.84 ms, and the backward pair reads 1 KB less.
// Backward overlapped ACROSS the two passes of a block (sb_lstm_bwd_cross_produce / _consume; round 4).  The inter-frame
// backward has 145 serial chains for 256 CUs; run as the fused role-split kernel (dgates in LDS: no 2 KB per position of dgates
// traffic, and half the CU time of the recurrence + stream-kernel pair) it leaves 111 CUs idle for 1.3 ms -- which the NEXT
// kernel of the backward, the intra-frame bidirectional pass of the same block, can use, because a tile of 16 frames only
// needs the inter-frame pass's result for those frames:
// PROD (single direction, SPLIT + XP, C = 32): the chunk role stores its du rows write-through (sc1) and the workgroup counts
//   itself into slab_flags[k] when the du rows of time slab k (slab_len steps, latest first) are on their way -- the protocol of
//   the overlapped forward's producer.
// CONS (bidirectional, SPLIT + XP, C = 32): persistent workgroups draw (tile, direction) items in the order tile_order (tiles
//   sorted by the slab that completes their 16 frames) from one atomic counter per direction -- two launches, one on the
//   library's side stream next to the producer (guarded) and one behind it, like the overlapped forward's consumer -- wait
//   (bounded) for that slab, and then run a PROLOGUE over the tile's 16 F positions: the block's inter-frame LayerNorm
//   backward + residual, dy1 = LN-backward(du; x) + res, which IS this pass's incoming gradient (written to pro_dy for this
//   workgroup's own readers and for the LayerNorm backward that follows; both directions of a tile write the same values).
//   Its scale S for the fp16 terms is derived PER TILE from max |dy1| (the running dW sums are rescaled by the power-of-two
//   ratio when a workgroup moves to its next tile), so no gradient maximum has to be known before the launch.
// Variant switches of the kernel as ONE bit set (round 4; they were thirteen positional bools): instantiations read
// lstm_bwd_rec_bf_kernel<K_FULL | K_DG16 | K_XP | K_SPLIT | K_BI | K_CONS, 32, 32>.  What each stands for is described above.
enum : unsigned { K_FULL = 1u << 0, K_REC16 = 1u << 1, K_DG16 = 1u << 2, K_SEG = 1u << 3, K_LNB = 1u << 4, K_BI = 1u << 5,
                  K_HS16B = 1u << 6, K_RECOMP = 1u << 7, K_SLAB = 1u << 8, K_XP = 1u << 9, K_SPLIT = 1u << 10, K_GREC = 1u << 11,
                  K_PROD = 1u << 12, K_CONS = 1u << 13 };
template <unsigned KF, int FUSE_C, int FST = 0>
__global__ __launch_bounds__((KF & K_SPLIT) ? 512 : 256) void lstm_bwd_rec_bf_kernel(sb_lstm_bwd_args a) {
  constexpr bool FULL = (KF & K_FULL) != 0, REC16 = (KF & K_REC16) != 0, DG16 = (KF & K_DG16) != 0, SEG = (KF & K_SEG) != 0,
                 LNB = (KF & K_LNB) != 0, BI = (KF & K_BI) != 0, HS16B = (KF & K_HS16B) != 0, RECOMP = (KF & K_RECOMP) != 0,
                 SLAB = (KF & K_SLAB) != 0, XP = (KF & K_XP) != 0, SPLIT = (KF & K_SPLIT) != 0, GREC = (KF & K_GREC) != 0,
                 PROD = (KF & K_PROD) != 0, CONS = (KF & K_CONS) != 0;
  static_assert(!PROD || (SPLIT && XP && !BI && FST == 32 && !SEG && !LNB), "cross-pass producer: role-split wide single-direction fused form");
  static_assert(!CONS || (SPLIT && XP && BI && FST == 32 && FUSE_C == 32), "cross-pass consumer: role-split wide bidirectional fused form");
  static_assert(!SPLIT || (FST > 0 && !RECOMP && !SLAB), "role split: fused forms");
  static_assert(!GREC || (SLAB && XP && FUSE_C == 32 && !SPLIT), "wide gate recomputation: overlapped inter-frame recurrence, C = 32");
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6) & 3, q = lane >> 4, j = lane & 15;
  const bool crole = SPLIT && __builtin_amdgcn_readfirstlane(tid >> 8) != 0;       // chunk role (waves 4..7)
  // CONS: 1-D grid, direction = bit 3 of the workgroup index -- whatever subset of a launch the dispatcher has resident, it serves
  // both directions evenly (with a (workgroups, 2) grid the x index runs first: next to 110 resident side-stream workgroups only
  // 128 + 18 of the main launch fit, direction 1 was served by 73 workgroups against 183 and set the pace: +50 %), and every XCD
  // gets both: the dispatcher deals consecutive workgroups round the 8 XCDs, so with direction = parity an XCD only ever saw one
  // direction (which defeats the per-XCD sharing of a tile's rows: see the prologue).  The grid is a multiple of 16.
#ifndef SB_EXP_DD
#define SB_EXP_DD 0
#endif
  const int dir = CONS ? ((SB_EXP_DD & 128) ? (int)(blockIdx.x & 1) : (int)((blockIdx.x >> 3) & 1)) : (int)blockIdx.y;
  const int S = a.nsteps, ndir = a.ndir;
  const bool rev = dir == 1;
  if constexpr (SLAB || PROD) { if (tid == 0) __hip_atomic_fetch_add(a.slab_started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  // CONS: wait until the producer tiles lo .. hi (the sequences of this tile's batch entries) have all completed `need` slabs;
  // bounded like seg_wait (watchdog word), uniform result
  auto cross_wait = [&](int packed) -> bool {
    __shared__ int cw_abort;
    if (tid == 0) {
      const int need = (packed & 0xFFF) + 1, lo = (packed >> 12) & 0x3FF, hi = (packed >> 22) & 0x3FF;
      int bad = 0;
      unsigned spins = 0;
      for (int t = lo; t <= hi && !bad; ++t) {
        while (sb_poll(a.slab_flags + t) < need) {
          ++spins;
          if (sb_wait_over(a.sched_status, spins, SB_TRIP_CROSS_SLABS, t, a.slab_flags + t, need)) { bad = 1; break; }
          sb_poll_pause();
        }
      }
      cw_abort = bad;
    }
    __syncthreads();
    return cw_abort == 0;
  };
  // CONS: the item draw (uniform over the workgroup) and the guard of the launch that runs NEXT to the producer
  __shared__ int ord_item;
  __shared__ float pro_red[8];
  // CONS: EIGHT queues per direction, one per XCD: item i of the slab-ordered list belongs to queue i mod 8, and a workgroup
  // draws from the queue of the XCD it runs on (HW_REG_XCC_ID) -- both directions of a tile are then walked by CUs behind ONE
  // L2, which is what lets the tile's incoming-gradient rows be computed once and shared through it (see the prologue) -- and
  // steals from the other queues when its own is empty (any number of XCDs in the partition, and no idle tail).
  const int xcd = CONS ? (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7) : 0;      // hwreg(HW_REG_XCC_ID, 0, 4)
  auto ord_next = [&]() -> int {
    if (tid == 0) {
      const int nt = (a.nseq + 15) / 16;
      int it = nt;
      if constexpr ((SB_EXP_DD & 64) != 0) {
        it = __hip_atomic_fetch_add(a.ord_counter + dir, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else
      for (int qq = 0; qq < 8 && it >= nt; ++qq) {
        const int qx = (xcd + qq) & 7;
        if (__hip_atomic_load(a.ord_counter + 2 * qx + dir, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 8 + qx >= nt) continue;
        const int k = __hip_atomic_fetch_add(a.ord_counter + 2 * qx + dir, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (8 * k + qx < nt) it = 8 * k + qx;
      }
      ord_item = it;
    }
    __syncthreads();
    const int v = ord_item;
    __syncthreads();                                 // (the next draw may not overwrite it before everybody has read it)
    return v;
  };
  int ord_first = 0;
  if constexpr (CONS) {
    bool take = true;
    if (a.ord_guard) {
      if (tid == 0) {
        int ok = 0;
        for (int i = 0; i < 200 && !ok; ++i) {
          ok = __hip_atomic_load(a.slab_started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.slab_need;
          if (!ok) __builtin_amdgcn_s_sleep(8);
        }
        ord_item = ok;
      }
      __syncthreads();
      take = ord_item != 0;
      __syncthreads();
    }
    // a workgroup that may not (guard) or need not (counter exhausted) work still writes its zero partial row at the end
    ord_first = take ? ord_next() : (a.nseq + 15) / 16;
  }
  __shared__ __attribute__((aligned(16))) float P[2][4][4][64][4];
  constexpr int CK = FST > 0 ? FST / 16 : 1, KT = CK + 4;
  __shared__ __attribute__((aligned(16))) _Float16 DG[FST > 0 ? 4 : 1][FST > 0 ? 16 : 1][FST > 0 ? DGP : 8];
  __shared__ __attribute__((aligned(16))) _Float16 DGL[XP ? 4 : 1][XP ? 16 : 1][XP ? DGP : 8];      // XP: the scaled low terms
  // STG (role split, wide form with hs pairs: C = 32): the chunk role would be 40 registers over its 256 with the u /
  // h_prev / dy rows of a chunk held in registers across the hand-over barrier (64 of them), and its spill traffic cost more
  // than the split gained.  There the four chunk waves fetch those rows ONCE, cooperatively, at the top of the period (16
  // registers per thread, dead before the barrier), park them in LDS and read their operands from there afterwards; rows
  // that must not count (h_prev / dy of walk index 0, a missing second step, sequences beyond nseq) are zeroed on the way
  // in, so the mask blocks are gone too.  The du exchange buffer needs no double buffering in the split kernel (its
  // reduction and the next write are two barriers apart), which pays for the LDS.
  constexpr bool STG = SPLIT && XP && FUSE_C > 0 && FST == 32;
  // The rows travel by async global -> LDS copies (global_load_lds_dwordx4: no registers at all; destination = wave-uniform
  // base + lane x 16 bytes, hence unpadded rows), issued one period ahead into the other buffer and waited for (vmcnt) just
  // before the hand-over barrier of the period that reads them.
  // HREC (STG without time segments): the h_prev rows do not come from memory at all.  The recurrence role recomputes h of its step
  // from the records it holds anyway (h = o tanh(f c_prev + i g): the forward kernel's own expression on the same fp32
  // values, so the same bits), splits it like the forward kernel and leaves the (hi, lo) row in a four-step ring in LDS
  // (the SHP buffers); the chunk of steps (s, s - 1) runs one period later and reads h of steps s - 1 (written a period
  // ago) and s - 2 (written before this period's first barrier; the chunk needs it after that barrier).  The forward pass
  // then stores no hs for these layers: 512 of its 3456 bytes per position, and 512 fewer read here.
  // Single-direction passes too (round 4: the inter-frame backward of the cross-pass schedule is this role-split kernel): the
  // forward then stores no hs pairs for them either, 256 of its 2 048 bytes per position.  Not with time segments: a segment's
  // last chunk needs h of a step that another workgroup walks.
  constexpr bool HREC = STG && (BI || !SEG);
  constexpr int SHROW = 256, SUROW = 4 * (FST > 0 ? FST : 16);                  // LDS rows (bytes)
  __shared__ __attribute__((aligned(16))) float R[FST > 0 ? (STG ? 1 : 2) : 1][4][2][CK][FST > 0 ? 64 : 1][4];
  __shared__ __attribute__((aligned(16))) char SHP[STG ? 2 : 1][STG ? 32 * SHROW : 16];   // h_prev pair rows of the chunk's 32 slots
  __shared__ __attribute__((aligned(16))) char SUP[STG ? 2 : 1][STG ? 32 * SUROW : 16];   // u pair rows
  __shared__ __attribute__((aligned(16))) char SDY[STG ? 2 : 1][STG ? 32 * SUROW : 16];   // dy rows of the h_prev positions (fp32)
  auto hring = [&](int step, int seq) -> char* {   // HREC: row of sequence seq of walk index step (two's complement & 3: step -1 is slot 3)
    return &SHP[0][0] + ((step & 3) * 16 + seq) * SHROW;
  };
  if constexpr (HREC) {                            // finite contents from the start: rows of steps that do not exist meet zero dgates
    for (int i = tid; i < 2 * 32 * SHROW / 16; i += 512) reinterpret_cast<f32x4*>(&SHP[0][0])[i] = zero4();
    __syncthreads();
  }
  // SLAB: the step's 16 dgates rows are assembled here so that they leave as whole 512-byte rows (write-through stores of
  // the 32-byte pieces each lane holds would reach HBM as partial lines)
  __shared__ __attribute__((aligned(16))) _Float16 DS[SLAB ? 2 : 1][SLAB ? 16 : 1][SLAB ? 4 * H + 8 : 8];
  __shared__ __attribute__((aligned(16))) _Float16 DSL[SLAB && XP ? 2 : 1][SLAB && XP ? 16 : 1][SLAB && XP ? 4 * H + 8 : 8];
  static_assert(FST == 0 || (DG16 && (REC16 || XP)), "fused streaming part: compact fp16 records or the wide form");
  static_assert(!XP || ((FST > 0 || SLAB) && DG16 && !REC16 && !HS16B && !RECOMP), "wide form: fused or overlapped kernels, fp32 records");
  constexpr float kLoUp = 2048.0f, kLoDn = 1.0f / 2048.0f;      // scale of the low fp16 term (XP)
  static_assert(!RECOMP || (BI && HS16B && FST == 32), "gate recomputation: bidirectional C = 32 form with fp16 hs");
  // forward weights of this direction as MFMA A operands: WR[gate][chunk][wave][lane] = rows gate*64 + 16 wave + (lane & 15),
  // k = 8 (lane >> 4) .. + 7 of chunk 0 (W_ih) / 1, 2 (W_hh), scaled like the forward kernel's (activations as rcp(1 + 2^z));
  // BR[gate][unit]: the scaled bias sums
  __shared__ __attribute__((aligned(16))) h16x8 WR[RECOMP ? 4 : 1][RECOMP ? 3 : 1][RECOMP ? 4 : 1][RECOMP ? 64 : 1];
  __shared__ __attribute__((aligned(16))) float BR[RECOMP || GREC ? 4 : 1][RECOMP || GREC ? H : 1];
  const float* __restrict__ whh = a.w_hh[dir];
  // [out tile ot][chunk]: A[i = out unit 16ot + j][k] = W_hh[gate row(k)][16ot + j].  DG16: the dgates enter the
  // product as the fp16 values that are stored (one term), W_hh as fp16 hi + lo -> 2 MFMAs per tile and chunk
  // instead of 6 and no operand split on the critical path.
  Split3 At[4][2];
  SplitH Ah[4][2];
#pragma unroll
  for (int ot = 0; ot < 4; ++ot)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
        t[kk] = whh[(size_t)((2 * c + (kk >> 2)) * H + 16 * w + 4 * q + (kk & 3)) * H + 16 * ot + j];
      if constexpr (DG16) Ah[ot][c] = splith8(t); else At[ot][c] = split8(t);
    }

  if constexpr (RECOMP) {
    const float* __restrict__ wih = dir == 0 ? a.w_ih : a.w_ih1;
    for (int e = tid; e < 4 * 3 * 4 * 64; e += 256) {
      const int ln = e & 63, wv = (e >> 6) & 3, ck = (e >> 8) % 3, g = (e >> 8) / 3;
      const int row = g * H + 16 * wv + (ln & 15), k0 = 8 * (ln >> 4);
      const float gsc = (g == 2 ? 2.0f : 1.0f) * SB_NLOG2E;
      h16x8 v;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
        v[kk] = (_Float16)(gsc * (ck == 0 ? wih[(size_t)row * FST + k0 + kk] : whh[(size_t)row * H + 32 * (ck - 1) + k0 + kk]));
      WR[g][ck][wv][ln] = v;
    }
    if (tid < 4 * H) BR[tid >> 6][tid & 63] = ((tid >> 6) == 2 ? 2.0f : 1.0f) * SB_NLOG2E * (a.b_ih[dir][tid] + a.b_hh[dir][tid]);
    __syncthreads();
  }

  // GREC: the forward kernel's weight terms (its Wt[gate][chunk]: rows gate * 64 + 16 w + j, k = 8q .. 8q + 7 of chunk 0 = W_ih,
  // chunks 1, 2 = W_hh; activation scale folded in) and bias sums, formed by the same expressions
  SplitH Wg[GREC ? 4 : 1][GREC ? 3 : 1];
  if constexpr (GREC) {
    const float* __restrict__ wih = a.w_ih;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = g * H + 16 * w + j;
      const float gsc = (g == 2 ? 2.0f : 1.0f) * SB_NLOG2E;
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = gsc * wih[(size_t)row * FUSE_C + 8 * q + kk];
      Wg[g][0] = splith8(t);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) t[kk] = gsc * whh[(size_t)row * H + 32 * c + 8 * q + kk];
        Wg[g][1 + c] = splith8(t);
      }
    }
    if (tid < 4 * H) BR[tid >> 6][tid & 63] = ((tid >> 6) == 2 ? 2.0f : 1.0f) * SB_NLOG2E * (a.b_ih[0][tid] + a.b_hh[0][tid]);
    __syncthreads();
  }
  f32x4 H0h[GREC ? 2 : 1], H0l[GREC ? 2 : 1];     // GREC: the initial hidden state (h_prev of step 0) as operand terms, per tile

  // STG: this lane's pieces of a chunk's rows.  Wave w of the chunk role fetches slot rows 8w .. 8w + 7 (step sa - (w >> 1),
  // sequences 8 (w & 1) ..): h rows 8w + (lane >> 4) and 8w + 4 + (lane >> 4), piece lane & 15; u / dy row 8w + (lane >> 3),
  // piece lane & 7 -- so that one wave instruction fills one contiguous KB of LDS
  // Round 6: these per-lane, per-tile constants (and `base` for the chunk role's flush) are PARKED IN LDS, 4 ints per chunk-role
  // thread.  Kept in registers through the period loop they were what hipcc spilled in the register-starved instantiations
  // (cross-pass consumer: 111 spilled registers), and a scratch reload inside the loop is a vector-memory load that misses the
  // L2 the record stream flows through: ~1 500 ticks each, twice per period (profiles/r06_phase_table.txt: "stage issue" 1 490 +
  // "flush" 1 286 of a 8 900-tick period for four copy instructions and one store).  An LDS read costs ~100 and no registers.
  __shared__ int PARK[(STG && SPLIT) ? 4 : 1][(STG && SPLIT) ? 256 : 1];
  int64_t stg_b[3] = {0, 0, 0};                   // step-0 positions of those three sequences (set_tile)
  bool stg_v[3] = {false, false, false};
  int stg_sel = 0;                                // LDS staging buffer of the chunk in hand
  bool valid = false;                            // per work item (tile): set_tile()
  int64_t base = 0, rec_tile = 0;
  int posb[FST > 0 ? 8 : 1];                      // FST: step-0 position of sequence 8 (q & 1) + kk (chunk slot 8q + kk)
  unsigned slotv = 0;                             // ... and whether that sequence exists (bit kk)
  int posq[2] = {0, 0};                           // LNB: step-0 position of this lane's two flush sequences
  bool fvalid[2] = {false, false};
  int64_t sbase[4] = {0, 0, 0, 0};                // SLAB: step-0 position of row 4w + i (uniform; -1: no such sequence)
  auto set_tile = [&](int tile) {
    rec_tile = (int64_t)tile * S;
    const int nc = tile * 16 + j;
    valid = FULL || nc < a.nseq;
    base = valid ? ((int64_t)(nc / a.n_inner) * a.p_outer + (int64_t)(nc % a.n_inner) * a.p_inner) : 0;
    if constexpr (GREC) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float t[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) t[kk] = (a.h0 && valid) ? a.h0[(size_t)nc * H + 32 * c + 8 * q + kk] : 0.f;
        const SplitH sp = splith8(t);
        H0h[c] = __builtin_bit_cast(f32x4, sp.hi);
        H0l[c] = __builtin_bit_cast(f32x4, sp.lo);
      }
    }
    if constexpr (SLAB) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int n4 = tile * 16 + 4 * w + i;
        sbase[i] = (FULL || n4 < a.nseq) ? (int64_t)(n4 / a.n_inner) * a.p_outer + (int64_t)(n4 % a.n_inner) * a.p_inner : -1;
      }
    }
    if constexpr (LNB) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int n3 = tile * 16 + 8 * (w & 1) + 2 * q + r;
        fvalid[r] = FULL || n3 < a.nseq;
        posq[r] = fvalid[r] ? (int)((int64_t)(n3 / a.n_inner) * a.p_outer + (int64_t)(n3 % a.n_inner) * a.p_inner) : 0;
      }
    }
    if constexpr (STG) {
      const int sq[3] = {8 * (w & 1) + (lane >> 4), 8 * (w & 1) + 4 + (lane >> 4), 8 * (w & 1) + (lane >> 3)};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int n = tile * 16 + sq[i];
        stg_v[i] = FULL || n < a.nseq;
        stg_b[i] = stg_v[i] ? (int64_t)(n / a.n_inner) * a.p_outer + (int64_t)(n % a.n_inner) * a.p_inner : 0;
      }
      if constexpr (SPLIT) {                       // (positions are 32-bit: the launchers check P < 2^31)
        if (crole) {
#pragma unroll
          for (int i = 0; i < 3; ++i) PARK[i][tid & 255] = (int)stg_b[i];
          PARK[3][tid & 255] = (int)base;
        }
      }
    }
    if constexpr (FST > 0) {
      slotv = 0;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int n2 = tile * 16 + 8 * (q & 1) + kk;
        if (FULL || n2 < a.nseq) slotv |= 1u << kk;
        // sequences beyond nseq: any readable position (their dgates rows are zero)
        posb[kk] = (FULL || n2 < a.nseq) ? (int)((int64_t)(n2 / a.n_inner) * a.p_outer + (int64_t)(n2 % a.n_inner) * a.p_inner) : 0;
      }
    }
  };
  const int uoff = 16 * w + 4 * q;

  // (uniform values, kept in scalar registers: as per-lane values the compiler holds them -- and {x, x} pairs for the packed
  //  multiplies -- in vector registers through the period loops, where the role-split kernels have none to spare)
  auto uniform_f = [](float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); };
  float gS = DG16 ? uniform_f(grad_scale(a.gmax)) : 1.0f;       // (CONS: re-derived per tile by the prologue)
  // W_lin^T tile of this wave's units (FUSE): A[i = unit 16w + j][k = channel 8q + kk], 2-term split
  bf16x8 Lh, Ll;
  h16x8 Lxh, Lxl;                                   // XP: fp16 hi + lo of the UNSCALED weights (dy carries the scale)
  if constexpr (FUSE_C > 0) {
    const int ldw = a.ndir * H;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int ch = 8 * q + kk;
      if constexpr (XP) {
        const float v = ch < FUSE_C ? a.w_lin[(size_t)ch * ldw + dir * H + 16 * w + j] : 0.f;
        const _Float16 hh = (_Float16)v;
        Lxh[kk] = hh;
        Lxl[kk] = (_Float16)(v - (float)hh);
      } else {
        const float v = ch < FUSE_C ? gS * a.w_lin[(size_t)ch * ldw + dir * H + 16 * w + j] : 0.f;
        const __bf16 hh = (__bf16)v;
        Lh[kk] = hh;
        Ll[kk] = (__bf16)(v - (float)hh);
      }
    }
  }

  // ---- fused streaming part: state and helpers (see lstm_bwd_stream_f16_kernel for the chunk arithmetic) ----
  SplitH Awt[CK][2];                               // W_ih^T: A[i = channel 16ct + j][k = gate 64w + 32m + 8q + kk]
  f32x4 wacc[4][KT];                               // dW rows of gate type w: [nt][u tiles | h_prev tiles]
  float csum[4] = {0.f, 0.f, 0.f, 0.f};
  float csumx[4] = {0.f, 0.f, 0.f, 0.f};           // XP: sums of the scaled low terms
  if constexpr (FST > 0) {
#pragma unroll
    for (int ct = 0; ct < CK; ++ct)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float t[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
          t[kk] = (dir == 0 ? a.w_ih : a.w_ih1)[(size_t)(64 * w + 32 * m + 8 * q + kk) * FST + 16 * ct + j];
        Awt[ct][m] = splith8(t);
      }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) wacc[nt][kt] = zero4();
  }
  // u / h_prev rows of the 32 slots of a chunk; dyv: dy of the h_prev rows' own positions (Linear weight gradient)
  // RAW loaded registers only: any arithmetic on them here would pin an s_waitcnt behind the loads at the top of the
  // loop body (the select of the first version did, and doubled the kernel time once the dy loads joined)
  static_assert(!BI || (!LNB && !SEG), "bidirectional fused form: no LayerNorm rider, no time segments");
  constexpr bool LINW = !BI || FUSE_C > 0;        // the fused Linear's weight gradient rides along (needs dy)
  // XP: u is the forward kernel's fp16 (hi, lo) pair tensor; so is hs when the Linear was applied in the forward kernel
  // (HSP: every form with the fused Linear backward), else hs is fp32 (the conv-LSTM flavour, whose ConvTranspose reads it)
  constexpr bool HSP = XP && FUSE_C > 0;
  // TRA: the chunk's dgates A fragments come from transposing LDS reads; M-tile nt of wave w then covers gate columns
  // 64 w + 16 nt + (0 .. 15) instead of 64 w + 4 (0 .. 15) + nt (the write-out below follows)
  constexpr bool TRA = XP && SB_TR_DGATES;
  constexpr bool H32 = (XP && !HSP) || (BI && !HS16B && !XP);   // hs as fp32 rows
  struct PairOps { h16x4 hh4[H32 || HSP ? 1 : 8]; f32x4 hh32[H32 ? 8 : 1]; h16x2 uh2[CK == 2 && !XP ? 8 : 1]; _Float16 uh1[CK == 2 || XP ? 1 : 8];
                   h16x8 hp8[HSP ? 8 : 1];                    // XP + HSP: h_prev units 4j .. 4j + 3 as (hi x 4, lo x 4)
                   h16x4 up4[XP && CK == 2 ? 8 : 1]; h16x2 up2[XP && CK == 1 ? 8 : 1];   // XP: u channels (2j, 2j + 1) / j as (hi.., lo..)
                   float dyv[CK][LINW && !STG ? 8 : 1];
                   float xq[2], rq[2]; };           // LNB: x and dy (channel j) of this lane's two flush positions
  static_assert(!LNB || FST == 16, "fused LayerNorm backward: C = 16");
  // LNB: the du tile is formed TRANSPOSED (positions as rows, channels as columns: the two MFMA operands swapped), so a
  // lane holds channel j of positions 4q..4q+3 and the LayerNorm sums over the 16 channels are DPP sums within a
  // 16-lane row.  (With channels along the rows they were __shfl chains across rows: ~1000 exposed cycles per pair.)
  // The 32 positions of a chunk are split over all four waves: wave w takes step sa - (w >> 1), sequences
  // 8 (w & 1) + 2q + r, r = 0, 1.
  float lng = 0.f, dgam = 0.f, dbet = 0.f;
  if constexpr (LNB) lng = a.ln_g[j];
  f32x4 lacc[CK];                                  // dW_lin tile: channels 16ct + 4q + r x units 4j + w
  float lbs[CK];                                   // db_lin: channel 16ct + j, this lane's positions
#pragma unroll
  for (int ct = 0; ct < CK; ++ct) { lacc[ct] = zero4(); lbs[ct] = 0.f; }
  const float* __restrict__ dyj = a.dy + j;
  const _Float16* __restrict__ hs16 = reinterpret_cast<const _Float16*>(a.hs);
  const float* __restrict__ hs32 = reinterpret_cast<const float*>(a.hs) + (BI ? dir * H : 0);
  constexpr int LDH = BI ? 2 * H : H;               // row length of hs
  // walk index s of this direction <-> time step; the step before walk index s (its h is h_prev) is walk index s - 1
  auto st_of = [&](int s) { return rev ? S - 1 - s : s; };
  const _Float16* __restrict__ u16 = reinterpret_cast<const _Float16*>(a.u);
  const float* __restrict__ u32 = reinterpret_cast<const float*>(a.u);
  // slot 8q + kk of the chunk of steps (sa, sa - 1): step sa - (q >> 1), sequence 8 (q & 1) + kk.  `two` = false: the
  // second step does not exist (its dgates rows are zero; addresses clamped to the first)
  auto pair_loads = [&](int sa, bool two) {
    PairOps o;
    const int sw = (q >> 1) == 0 ? sa : (two ? sa - 1 : sa);
    const bool hp = sw > 0;                        // h_prev of the first step is the (zero) initial state: masked in chunk()
    const int st = st_of(sw), sth = st_of(hp ? sw - 1 : sw);
    // positions in 32 bits (the fused launchers check nseq nsteps < 2^31): one add per position and a widening shift per address --
    // as 64-bit sums these 28 addresses were ~100 vector instructions a period, a quarter of the chunk role's arithmetic in the
    // C = 16 kernels, where the chunk role sets the pace
    typedef std::conditional_t<SB_EXP_IDX64 != 0, int64_t, size_t> pidx_t;
    const unsigned ps = (unsigned)st * (unsigned)a.p_step, psh = (unsigned)sth * (unsigned)a.p_step;      // uniform
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const pidx_t pos = SB_EXP_IDX64 ? (pidx_t)((int64_t)posb[kk] + (int64_t)st * a.p_step) : (pidx_t)((unsigned)posb[kk] + ps);
      const pidx_t posh = SB_EXP_IDX64 ? (pidx_t)((int64_t)posb[kk] + (int64_t)sth * a.p_step) : (pidx_t)((unsigned)posb[kk] + psh);
      if constexpr (HSP) o.hp8[kk] = *reinterpret_cast<const h16x8*>(hs16 + (posh * LDH + (BI ? dir * H : 0) + 4 * j) * 2);
      else if constexpr (H32) o.hh32[kk] = ld4(hs32 + posh * LDH + 4 * j);
      else if constexpr (BI) o.hh4[kk] = *reinterpret_cast<const h16x4*>(hs16 + posh * (2 * H) + dir * H + 4 * j);
      else o.hh4[kk] = *reinterpret_cast<const h16x4*>(hs16 + posh * H + 4 * j);
      if constexpr (XP) {
        if constexpr (CK == 2) o.up4[kk] = *reinterpret_cast<const h16x4*>(u16 + (pos * FST + 2 * j) * 2);
        else o.up2[kk] = *reinterpret_cast<const h16x2*>(u16 + (pos * FST + j) * 2);
      } else if constexpr (CK == 2) o.uh2[kk] = *reinterpret_cast<const h16x2*>(u16 + pos * FST + 2 * j);
      else o.uh1[kk] = u16[pos * FST + j];
      // the Linear's weight gradient pairs h of a position with dy of the SAME position (the h_prev row's)
      if constexpr (LINW && !STG) {
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) o.dyv[ct][kk] = dyj[posh * FST + 16 * ct];
      }
    }
    if constexpr (LNB) {                           // channel j of (step sa - (w >> 1), sequences 8 (w & 1) + 2q + r)
      const int stf = sa - ((w >> 1) && two ? 1 : 0);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const pidx_t posf = SB_EXP_IDX64 ? (pidx_t)((int64_t)posq[r] + (int64_t)stf * a.p_step)
                                         : (pidx_t)((unsigned)posq[r] + (unsigned)stf * (unsigned)a.p_step);
        o.xq[r] = a.ln_x[posf * FST + j];
        o.rq[r] = a.dy[posf * FST + j];
      }
    }
    return o;
  };
  // rows of the chunk of steps (sa, sa - 1) -> staging buffer sel.  Direct (async) copies when every row counts; else -- the
  // last pair of a tile (h_prev / dy of walk index 0 are zero), a missing second step, a ragged tile -- through registers
  // with the rows that must not count zeroed (rare: the stall does not matter)
  auto stage_issue = [&](int sa, bool two, int sel, int tile, auto&& between) {
    const int i = w >> 1;
    const int sw = i == 0 ? sa : (two ? sa - 1 : sa);
    const bool cnt = sw > 0 && (i == 0 || two);                    // uniform over the wave
    const int64_t sp = (int64_t)st_of(sw) * a.p_step, sph = (int64_t)st_of(sw > 0 ? sw - 1 : sw) * a.p_step;
    int64_t b0, b1, b2;
    if constexpr (SPLIT) { b0 = PARK[0][tid & 255]; b1 = PARK[1][tid & 255]; b2 = PARK[2][tid & 255]; }
    else { b0 = stg_b[0]; b1 = stg_b[1]; b2 = stg_b[2]; }
    // (the lane's column pieces go into the INDEX, through an opaque copy: `pointer + lane piece` is loop-invariant, and hipcc
    //  would hoist that 64-bit per-lane sum out of the period loop -- into a register it then has to spill)
    // (... and the lane id itself is re-derived from the execution mask by a volatile asm -- every lane is active here --, or the
    //  invariant `lane & 7` would be the value that gets spilled)
    int ln = lane;
    if constexpr (SPLIT) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
    const int l15 = ln & 15, l7 = ln & 7;
    const float* gh0 = reinterpret_cast<const float*>(hs16 + (((b0 + sph) * LDH + (BI ? dir * H : 0)) * 2 + 8 * l15));
    const float* gh1 = reinterpret_cast<const float*>(hs16 + (((b1 + sph) * LDH + (BI ? dir * H : 0)) * 2 + 8 * l15));
    const float* gu = reinterpret_cast<const float*>(u16 + (((b2 + sp) * FST) * 2 + 8 * l7));
    const float* gd = a.dy + ((b2 + sph) * FST + 4 * l7);
    char* lh = &SHP[sel][(8 * w) * SHROW];
    char* lu = &SUP[sel][(8 * w) * SUROW];
    char* ld = &SDY[sel][(8 * w) * SUROW];
    const bool whole = FULL || tile * 16 + 16 <= a.nseq;
    // Round 6: every source address is complete BEFORE the first copy is issued.  Some of the per-lane 64-bit bases live in
    // scratch in the register-starved instantiations (the cross-pass consumer: 111 spilled registers), a scratch reload is a
    // vector-memory load, and hipcc waits for it with vmcnt(0): placed between two copies, that wait sat out the full latency of
    // the copy issued just before it -- once per period (the phase table's "stage issue": 1 490 ticks for four instructions).
    // `between` (the flush of this period's du rows, whose store address may be a third reload) runs HERE: after the copies'
    // addresses are in registers -- so that nothing is reloaded behind its store or behind the copies -- and before the copies.
#ifndef SB_EXP_NO_STAGE_PIN
    asm volatile("" : "+v"(gh0), "+v"(gh1), "+v"(gu), "+v"(gd));
#endif
    between();
#ifndef SB_EXP_NO_STAGE_PIN
    asm volatile("" : "+v"(gh0), "+v"(gh1), "+v"(gu), "+v"(gd));
#endif
    if (cnt && whole) {
      if constexpr (!HREC) {
        __builtin_amdgcn_global_load_lds(gh0, (__attribute__((address_space(3))) void*)(lh), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(gh1, (__attribute__((address_space(3))) void*)(lh + 4 * SHROW), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds(gu, (__attribute__((address_space(3))) void*)(lu), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(gd, (__attribute__((address_space(3))) void*)(ld), 16, 0, 0);
    } else {
      const f32x4 z4 = zero4();
      if constexpr (!HREC) {
        const f32x4 h0 = ld4(gh0), h1 = ld4(gh1);
        st4(reinterpret_cast<float*>(lh + 16 * lane), (cnt && stg_v[0]) ? h0 : z4);
        st4(reinterpret_cast<float*>(lh + 4 * SHROW + 16 * lane), (cnt && stg_v[1]) ? h1 : z4);
      }
      const f32x4 uu = ld4(gu), dd = ld4(gd);
      st4(reinterpret_cast<float*>(lu + 16 * lane), uu);
      st4(reinterpret_cast<float*>(ld + 16 * lane), (cnt && stg_v[2]) ? dd : z4);
    }
  };
  const h16x2 ones2 = {(_Float16)1.0f, (_Float16)1.0f};
  // chunk arithmetic on the dgates rows in LDS slots (sl, sl + 1); du partial sums -> R[buf]
  // steady (uniform): a full pair of steps away from walk index 0 -- every h_prev row exists, and the per-lane masks
  // (~50-90 v_cndmask per chunk) sit in blocks behind a uniform branch (the empty asm keeps hipcc from turning the branch
  // back into selects)
  // phase: -1 everything (one-role kernel); SPLIT: 0 = du only (before the hand-over barrier), 1 = the rest
  auto chunk = [&](int sl, int buf, const PairOps& o, int sa, bool two, int phase = -1) {
    const bool steady = two && sa >= 2;
    const int sw = (q >> 1) == 0 ? sa : (two ? sa - 1 : sa);
    const bool hp = sw > 0;
    const h16x4 hz4 = {0, 0, 0, 0};
    if constexpr (XP) {
      const h16x8 dn8 = {(_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn,
                         (_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn, (_Float16)kLoDn};
      auto split3 = [&](const float (&v)[8], h16x8& bh, h16x8& bl, h16x8& bs) {   // v = bh + bl; bs = 2^-11 bh
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const _Float16 hh = (_Float16)v[kk];
          bh[kk] = hh;
          bl[kk] = (_Float16)(v[kk] - (float)hh);
        }
        bs = bh * dn8;
      };
      auto split2x = [&](const float (&v)[8], h16x8& ah, h16x8& al) {              // v = ah + 2^-11 al
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const _Float16 hh = (_Float16)v[kk];
          ah[kk] = hh;
          al[kk] = (_Float16)__builtin_fmaf((float)hh, -kLoUp, v[kk] * kLoUp);
        }
      };
      // dgates of this lane's 8 k-slots, gate columns 64w + 4j .. + 3: hi and scaled low terms
      auto build_A = [&](h16x8 (&Aoh)[4], h16x8 (&Aol)[4]) __attribute__((always_inline)) {
        if constexpr (TRA) {
          // transposing LDS reads (ds_read_b64_tr_b16, gfx950): a 16-lane group fetches a [4 rows][16 columns] block as 8-byte row
          // pieces (lane i: row i >> 2, columns 4 (i & 3) .. + 3) and lane c receives COLUMN c of it -- the A fragment itself.
          // Row-major reads + the 4 x 8 -> 8 x 4 shuffle in registers were 32 v_perm_b32 per call, two calls a period.
          // (role-split kernels: the lane id is re-derived from the execution mask -- every lane is active here -- by a volatile
          //  asm, or the loop-invariant per-lane address pieces are hoisted out of the period loop into registers that spill: see
          //  stage_issue)
          int ln = lane;
          if constexpr (SPLIT) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
          const int i16 = ln & 15, qq = ln >> 4;
#pragma unroll
          for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
              const int row = 8 * (qq & 1) + 4 * hf + (i16 >> 2), col = 64 * w + 16 * nt + 4 * (i16 & 3);
              const s16x4 vh = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) s16x4*)(&DG[sl + (qq >> 1)][row][col]));
              const s16x4 vl = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                  (__attribute__((address_space(3))) s16x4*)(&DGL[sl + (qq >> 1)][row][col]));
              const h16x4 fh = __builtin_bit_cast(h16x4, vh), fl = __builtin_bit_cast(h16x4, vl);
#pragma unroll
              for (int r = 0; r < 4; ++r) { Aoh[nt][4 * hf + r] = fh[r]; Aol[nt][4 * hf + r] = fl[r]; }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const h16x4 th = *reinterpret_cast<const h16x4*>(&DG[sl + (q >> 1)][8 * (q & 1) + kk][64 * w + 4 * j]);
          const h16x4 tl = *reinterpret_cast<const h16x4*>(&DGL[sl + (q >> 1)][8 * (q & 1) + kk][64 * w + 4 * j]);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) { Aoh[nt][kk] = th[nt]; Aol[nt][kk] = tl[nt]; }
        }
        }
      };
      auto col_sums = [&](const h16x8 (&Aoh)[4], const h16x8 (&Aol)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            csum[nt] = __builtin_amdgcn_fdot2(h16x2{Aoh[nt][2 * pr], Aoh[nt][2 * pr + 1]}, ones2, csum[nt], false);
            csumx[nt] = __builtin_amdgcn_fdot2(h16x2{Aol[nt][2 * pr], Aol[nt][2 * pr + 1]}, ones2, csumx[nt], false);
          }
      };
      auto w_prod = [&](int kt, const h16x8 (&Aoh)[4], const h16x8 (&Aol)[4], const h16x8& bh, const h16x8& bl, const h16x8& bs)
          __attribute__((always_inline)) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wacc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aol[nt], bs, wacc[nt][kt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wacc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aoh[nt], bl, wacc[nt][kt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) wacc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aoh[nt], bh, wacc[nt][kt], 0, 0, 0);
      };
      // column tiles of u (dW_ih) and the bias sums: they need no h_prev, so in the role-split kernel they run BEFORE the
      // hand-over barrier (phase 0), where the chunk role used to wait ~1 350 ticks per period for the recurrence role's step
      auto u_tiles = [&](const h16x8 (&Aoh)[4], const h16x8 (&Aol)[4]) __attribute__((always_inline)) {
        col_sums(Aoh, Aol);
#pragma unroll
        for (int kt = 0; kt < CK; ++kt) {
          h16x8 bh, bl, bs;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            if constexpr (STG) {
              const h16x4 up = *reinterpret_cast<const h16x4*>(&SUP[stg_sel][(8 * q + kk) * SUROW + 8 * j]);
              bh[kk] = up[kt]; bl[kk] = up[2 + kt];
            } else if constexpr (CK == 2) { bh[kk] = o.up4[kk][kt]; bl[kk] = o.up4[kk][2 + kt]; }
            else { bh[kk] = o.up2[kk][0]; bl[kk] = o.up2[kk][1]; }
          }
          bs = bh * dn8;
          w_prod(kt, Aoh, Aol, bh, bl, bs);
        }
      };
      if (phase != 0) {
      // h_prev tile kt (units 4j + kt of the 8 k-slots) as matrix operands hi, lo, 2^-11 hi -- built on demand from the
      // (masked) pairs: materialising all four tiles up front costs 48 registers the role-split kernel does not have
      h16x8 hpm[HSP ? 8 : 1];
      if constexpr (HSP) {
        const h16x8 hz8 = {0, 0, 0, 0, 0, 0, 0, 0};
        if constexpr (HREC) {
          const char* hr = hring(sw - 1, 8 * (q & 1)) + 16 * j;
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) hpm[kk] = *reinterpret_cast<const h16x8*>(hr + kk * SHROW);
          if (!steady) {                               // walk index 0 has no h_prev
            asm volatile("");
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) hpm[kk] = hp ? hpm[kk] : hz8;
          }
        } else if constexpr (STG) {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) hpm[kk] = *reinterpret_cast<const h16x8*>(&SHP[stg_sel][(8 * q + kk) * SHROW + 16 * j]);
        } else {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) hpm[kk] = o.hp8[kk];
          if (!steady) {
            asm volatile("");
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) hpm[kk] = hp ? hpm[kk] : hz8;
          }
        }
      }
      auto htile = [&](auto kt_tag, h16x8& bh, h16x8& bl, h16x8& bs) {
        constexpr int kt = decltype(kt_tag)::value;
        if constexpr (HSP) {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) { bh[kk] = hpm[kk][kt]; bl[kk] = hpm[kk][4 + kt]; }
          bs = bh * dn8;
        } else {
          float v[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) v[kk] = hp ? o.hh32[kk][kt] : 0.f;
          split3(v, bh, bl, bs);
        }
      };
      float dyl[CK][LINW ? 8 : 1];
      if constexpr (LINW) {
#pragma unroll
        for (int ct = 0; ct < CK; ++ct)
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            if constexpr (STG) dyl[ct][kk] = *reinterpret_cast<const float*>(&SDY[stg_sel][(8 * q + kk) * SUROW + 4 * (16 * ct + j)]);
            else dyl[ct][kk] = o.dyv[ct][kk];
          }
      }
      auto do_linw = [&]() {                         // dW_lin: A = dy^T (channel 16ct + j x 8 positions), B = h tile w
        h16x8 bwh, bwl, bws;
        if (w == 0) htile(std::integral_constant<int, 0>{}, bwh, bwl, bws);          // (uniform branches)
        else if (w == 1) htile(std::integral_constant<int, 1>{}, bwh, bwl, bws);
        else if (w == 2) htile(std::integral_constant<int, 2>{}, bwh, bwl, bws);
        else htile(std::integral_constant<int, 3>{}, bwh, bwl, bws);
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) {
          float dv8[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) dv8[kk] = dyl[ct][kk] * gS;
          if (!STG && (!FULL || !steady)) {          // slots of a missing second step / of sequences beyond nseq must not count
            asm volatile("");                        // (STG: zeroed on the way into LDS)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              const bool dv = hp && (two || q < 2) && ((slotv >> kk) & 1u);
              dv8[kk] = dv ? dv8[kk] : 0.f;
            }
          }
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) lbs[ct] += dv8[kk];
          h16x8 adh, adl;
          split2x(dv8, adh, adl);
          lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(adl, bws, lacc[ct], 0, 0, 0);
          lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(adh, bwl, lacc[ct], 0, 0, 0);
          lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(adh, bwh, lacc[ct], 0, 0, 0);
        }
      };
      if constexpr (LINW) do_linw();
      h16x8 Aoh[4], Aol[4];
      build_A(Aoh, Aol);
      if (phase != 1 || !STG) u_tiles(Aoh, Aol);     // (one-role kernel, and the split kernels that hold the rows in registers: here)
#pragma unroll
      for (int kt = CK; kt < KT; ++kt) {             // column tiles of h_prev
        h16x8 bh, bl, bs;
        if (kt == CK) htile(std::integral_constant<int, 0>{}, bh, bl, bs);
        else if (kt == CK + 1) htile(std::integral_constant<int, 1>{}, bh, bl, bs);
        else if (kt == CK + 2) htile(std::integral_constant<int, 2>{}, bh, bl, bs);
        else htile(std::integral_constant<int, 3>{}, bh, bl, bs);
        w_prod(kt, Aoh, Aol, bh, bl, bs);
      }
      }
      if (phase != 1)
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        f32x4 du[CK], dux[CK];
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) { du[ct] = zero4(); dux[ct] = zero4(); }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const h16x8 d8 = *reinterpret_cast<const h16x8*>(&DG[sl + sb][j][64 * w + 32 * m + 8 * q]);
          const h16x8 d8l = *reinterpret_cast<const h16x8*>(&DGL[sl + sb][j][64 * w + 32 * m + 8 * q]);
#pragma unroll
          for (int ct = 0; ct < CK; ++ct) {
            if constexpr (LNB) {                     // du^T: rows = positions, columns = channels
              dux[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d8l, Awt[ct][m].hi, dux[ct], 0, 0, 0);
              du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d8, Awt[ct][m].lo, du[ct], 0, 0, 0);
              du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d8, Awt[ct][m].hi, du[ct], 0, 0, 0);
            } else {
              dux[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Awt[ct][m].hi, d8l, dux[ct], 0, 0, 0);
              du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Awt[ct][m].lo, d8, du[ct], 0, 0, 0);
              du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Awt[ct][m].hi, d8, du[ct], 0, 0, 0);
            }
          }
        }
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) {
#pragma unroll
          for (int r = 0; r < 4; ++r) du[ct][r] = __builtin_fmaf(dux[ct][r], kLoDn, du[ct][r]);
          st4(&R[buf][w][sb][ct][lane][0], du[ct]);
        }
      }
      if (phase == 0 && STG) {                       // (rows in LDS: nothing is held across the barrier for it)
        h16x8 Aoh[4], Aol[4];
        build_A(Aoh, Aol);
        u_tiles(Aoh, Aol);
      }
    } else {
    if (phase != 0) {
    h16x8 Bop[KT];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if constexpr (CK == 2) { Bop[0][kk] = o.uh2[kk][0]; Bop[1][kk] = o.uh2[kk][1]; }
      else Bop[0][kk] = o.uh1[kk];
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      if constexpr (H32) {
        const f32x4 hm = hp ? o.hh32[kk] : zero4();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) Bop[CK + kt][kk] = (_Float16)hm[kt];
      } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) Bop[CK + kt][kk] = o.hh4[kk][kt];
      }
    }
    if constexpr (!H32) {
      if (!steady) {                               // h_prev of walk index 0 is the (zero) initial state
        asm volatile("");
        const _Float16 z = (_Float16)0.f;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) Bop[CK + kt][kk] = hp ? Bop[CK + kt][kk] : z;
      }
    }
    if constexpr (LINW)
#pragma unroll
    for (int ct = 0; ct < CK; ++ct) {              // dW_lin: A = dy^T (channel 16ct + j x 8 positions), B = h tile w
      h16x8 Ad;
      float dv8[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) dv8[kk] = o.dyv[ct][kk] * gS;
      if (!FULL || !steady) {                      // slots of a missing second step and of sequences beyond nseq must not count
        asm volatile("");
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const bool dv = hp && (two || q < 2) && ((slotv >> kk) & 1u);
          dv8[kk] = dv ? dv8[kk] : 0.f;
        }
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        Ad[kk] = (_Float16)dv8[kk];
        lbs[ct] += dv8[kk];
      }
      const h16x8 Bw = w == 0 ? Bop[CK] : (w == 1 ? Bop[CK + 1] : (w == 2 ? Bop[CK + 2] : Bop[CK + 3]));
      lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ad, Bw, lacc[ct], 0, 0, 0);
    }
    h16x4 a4[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
      a4[kk] = *reinterpret_cast<const h16x4*>(&DG[sl + (q >> 1)][8 * (q & 1) + kk][64 * w + 4 * j]);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      h16x8 Aop;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) Aop[kk] = a4[kk][nt];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr)
        csum[nt] = __builtin_amdgcn_fdot2(h16x2{Aop[2 * pr], Aop[2 * pr + 1]}, ones2, csum[nt], false);
#pragma unroll
      for (int kt = 0; kt < KT; ++kt) wacc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Aop, Bop[kt], wacc[nt][kt], 0, 0, 0);
    }
    }
    if (phase != 1)
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) {
      f32x4 du[CK];
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) du[ct] = zero4();
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const h16x8 d8 = *reinterpret_cast<const h16x8*>(&DG[sl + sb][j][64 * w + 32 * m + 8 * q]);
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) {
          if constexpr (LNB) {                     // du^T: rows = positions, columns = channels
            du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d8, Awt[ct][m].lo, du[ct], 0, 0, 0);
            du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d8, Awt[ct][m].hi, du[ct], 0, 0, 0);
          } else {
            du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Awt[ct][m].lo, d8, du[ct], 0, 0, 0);
            du[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Awt[ct][m].hi, d8, du[ct], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) st4(&R[buf][w][sb][ct][lane][0], du[ct]);
    }
    }
  };
  // the chunks pair dy with h at steps 0 .. S-2 (the h_prev rows); the last step of a tile is added here
  auto lin_top = [&]() {
    if constexpr (XP) {
      float hw[8], dv8[CK][8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int64_t pos = (int64_t)posb[kk] + (int64_t)st_of(S - 1) * a.p_step;       // walk index S - 1
        float hv;
        if constexpr (HSP) {
          const h16x8 hp8 = HREC ? *reinterpret_cast<const h16x8*>(hring(S - 1, 8 * (q & 1) + kk) + 16 * j)
                                 : *reinterpret_cast<const h16x8*>(hs16 + (pos * LDH + (BI ? dir * H : 0) + 4 * j) * 2);
          hv = w == 0 ? (float)hp8[0] + (float)hp8[4] : (w == 1 ? (float)hp8[1] + (float)hp8[5]
               : (w == 2 ? (float)hp8[2] + (float)hp8[6] : (float)hp8[3] + (float)hp8[7]));
        } else {
          const f32x4 h32 = ld4(hs32 + pos * LDH + 4 * j);
          hv = w == 0 ? h32[0] : (w == 1 ? h32[1] : (w == 2 ? h32[2] : h32[3]));
        }
        hw[kk] = q < 2 ? hv : 0.f;
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) {
          const float v = dyj[pos * FST + 16 * ct];
          dv8[ct][kk] = (q < 2 && ((slotv >> kk) & 1u)) ? v * gS : 0.f;
        }
      }
      h16x8 bwh, bwl, bws;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const _Float16 hh = (_Float16)hw[kk];
        bwh[kk] = hh;
        bwl[kk] = (_Float16)(hw[kk] - (float)hh);
        bws[kk] = hh * (_Float16)kLoDn;
      }
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) {
        h16x8 adh, adl;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const _Float16 hh = (_Float16)dv8[ct][kk];
          adh[kk] = hh;
          adl[kk] = (_Float16)((dv8[ct][kk] - (float)hh) * kLoUp);
          lbs[ct] += dv8[ct][kk];
        }
        lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(adl, bws, lacc[ct], 0, 0, 0);
        lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(adh, bwl, lacc[ct], 0, 0, 0);
        lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(adh, bwh, lacc[ct], 0, 0, 0);
      }
    } else {
    h16x8 Bw;
    float lin_top_dy[CK][8];
    const h16x4 hz4 = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int64_t pos = (int64_t)posb[kk] + (int64_t)st_of(S - 1) * a.p_step;       // walk index S - 1
      h16x4 hv;
      if constexpr (H32) {
        const f32x4 h32 = ld4(hs32 + pos * LDH + 4 * j);
#pragma unroll
        for (int r = 0; r < 4; ++r) hv[r] = (_Float16)h32[r];
      } else if constexpr (BI) {
        hv = *reinterpret_cast<const h16x4*>(hs16 + pos * (2 * H) + dir * H + 4 * j);
      } else {
        hv = *reinterpret_cast<const h16x4*>(hs16 + pos * H + 4 * j);
      }
      const h16x4 hm = q < 2 ? hv : hz4;
      Bw[kk] = w == 0 ? hm[0] : (w == 1 ? hm[1] : (w == 2 ? hm[2] : hm[3]));
#pragma unroll
      for (int ct = 0; ct < CK; ++ct) {
        const float v = dyj[pos * FST + 16 * ct];
        lin_top_dy[ct][kk] = (q < 2 && ((slotv >> kk) & 1u)) ? v * gS : 0.f;
      }
    }
#pragma unroll
    for (int ct = 0; ct < CK; ++ct) {
      h16x8 Ad;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) { Ad[kk] = (_Float16)lin_top_dy[ct][kk]; lbs[ct] += lin_top_dy[ct][kk]; }
      lacc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ad, Bw, lacc[ct], 0, 0, 0);
    }
    }
  };
  float invS = uniform_f(1.0f / gS);               // gS is a power of two
  // du rows of a finished chunk (its R[buf] is complete after the barrier that followed it): wave w < 2 CK reduces
  // sub-tile sb = w / CK (step sa - sb), channel tile ct = w % CK
  // where this lane's du row of the chunk (sa, ..) goes (!LNB): computed -- and, where the base lives in scratch, reloaded -- at
  // the TOP of the chunk role's second phase, in front of the next period's copies (see stage_issue)
  auto flush_ptr = [&](int sa) -> float* {
    if constexpr (LNB) return nullptr;
    else {
      const int sb = w / CK, ct = w % CK;
      int64_t bb = base;
      int qq = q;
      if constexpr (STG && SPLIT) {                  // (parked: see PARK; the lane's row re-derived, not kept: see stage_issue)
        bb = PARK[3][tid & 255];
        int ln;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(ln));
        qq = ln >> 4;
      }
      const int64_t pos = bb + (int64_t)st_of(sa - sb) * a.p_step;
      return a.du + ((pos * ndir + dir) * FST + 16 * ct + 4 * qq);
    }
  };
  auto flush = [&](int sa, int nsteps_in_chunk, int buf, const float (&xq)[2], const float (&rq)[2], float* dst = nullptr) {
    if constexpr (LNB) {
      const int sbf = w >> 1;
      if (sbf < nsteps_in_chunk) {
        const int st = sa - sbf;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          // du of (sequence n0, channel j): element n0 & 3 of lane (n0 >> 2, j) of the transposed tile, summed over waves
          const int n0 = 8 * (w & 1) + 2 * q + r;
          const int sl = (n0 >> 2) * 16 + j, el = n0 & 3;
          const float du = (R[buf][0][sbf][0][sl][el] + R[buf][1][sbf][0][sl][el] + R[buf][2][sbf][0][sl][el] +
                            R[buf][3][sbf][0][sl][el]) * invS;
          const float x = xq[r];
          const float mean = row16_sum(x) * (1.0f / 16);
          const float d = x - mean;
          const float rstd = __builtin_amdgcn_rsqf(__builtin_fmaf(row16_sum(d * d), 1.0f / 16, 1e-5f));   // v_rsq_f32, 1 ulp
          const float xh = d * rstd, gg = du * lng;
          dgam = __builtin_fmaf(du, xh, dgam);
          dbet += du;
          const float m1 = row16_sum(gg) * (1.0f / 16), m2 = row16_sum(gg * xh) * (1.0f / 16);
          const float dxv = (gg - m1 - xh * m2) * rstd + rq[r];
          const int64_t pos = (int64_t)posq[r] + (int64_t)st * a.p_step;
          if (fvalid[r]) a.dx[pos * FST + j] = dxv;
        }
      }
    } else {
      const int sb = w / CK, ct = w % CK;
      if (w < 2 * CK && sb < nsteps_in_chunk && valid) {
        const f32x4 s4 = ld4(&R[buf][0][sb][ct][lane][0]) + ld4(&R[buf][1][sb][ct][lane][0]) +
                         ld4(&R[buf][2][sb][ct][lane][0]) + ld4(&R[buf][3][sb][ct][lane][0]);
        float* const o = dst ? dst : flush_ptr(sa);
        if constexpr (PROD && !(SB_EXP_CONS & 2)) st4_sc1(o, s4 * invS);
        else st4(o, s4 * invS);
      }
    }
  };

  // RECOMP: u_s, h_{s-1} as B operands; GREC: the same as raw 32-byte pieces of the pair tensors (gu: channels 8q .. 8q + 7 as
  // [hi0 hi1 lo0 lo1] x 4; gh: units 8q .. + 7 of K-chunks 1 and 2 as [hi x 4, lo x 4] x 2 each)
  struct Raw { f32x4 r0, r1, r2, r3, cp, dh, dy1; h16x4 cp16; h16x8 ub, hb0, hb1; f32x4 gu[2], gh[4]; };
  auto load_raw = [&](int s) {
    Raw r;
    const int st = rev ? S - 1 - s : s;
    const int64_t pos = base + (int64_t)st * a.p_step;
    if (valid) {
      if constexpr (REC16) {
        // blocked lane-order layout of the forward kernel: one contiguous KB per load instruction
        const int64_t blk = (rec_tile + st) * ndir + dir;
        const float* rec = a.save_gates + blk * (16 * 4 * H / 2) + (w * 128 + lane) * 4;
        if constexpr (RECOMP) {
          // B operands of the gate recomputation: lane (q, j) = sequence j, k = 8q .. 8q+7 of the chunk
          r.r0 = r.r1 = zero4();
          r.ub = *reinterpret_cast<const h16x8*>(u16 + pos * FST + 8 * q);
          const int sp = s > 0 ? s - 1 : 0;                              // walk index of the previous step (masked at s == 0)
          const int64_t posp = base + (int64_t)(rev ? S - 1 - sp : sp) * a.p_step;
          const _Float16* hp = hs16 + posp * (2 * H) + dir * H + 8 * q;
          const h16x8 h0v = *reinterpret_cast<const h16x8*>(hp), h1v = *reinterpret_cast<const h16x8*>(hp + 32);
          const h16x8 hz = {0, 0, 0, 0, 0, 0, 0, 0};
          r.hb0 = s > 0 ? h0v : hz;
          r.hb1 = s > 0 ? h1v : hz;
        } else
        if (SB_EXP_RECOMPUTE & 2) { const float cv = __builtin_bit_cast(float, (unsigned)(0x38003800u + (s & 1))); r.r0 = r.r1 = f32x4{cv, cv, cv, cv}; }
        else { r.r0 = ld4(rec); r.r1 = ld4(rec + 256); }
        r.r2 = r.r3 = r.cp = zero4();
        r.cp16 = *reinterpret_cast<const h16x4*>(reinterpret_cast<const _Float16*>(a.save_c) + blk * (16 * H) + (w * 64 + lane) * 4);
      } else if constexpr (XP) {           // blocked fp32 records of the forward kernel's SAVE == 4
        // (FST > 0: block and position indices in 32 bits -- the fused launchers check nseq nsteps < 2^31 -- and ONE widening
        //  multiply per pointer: the 64-bit index arithmetic was ~25 scalar instructions of the recurrence role's step)
        typedef std::conditional_t<(FST > 0 && !SB_EXP_IDX64), unsigned, int64_t> blk_t;
        const blk_t blk = (FST > 0 && !SB_EXP_IDX64) ? (blk_t)(((unsigned)rec_tile + (unsigned)st) * (unsigned)ndir + (unsigned)dir)
                                    : (blk_t)((rec_tile + st) * ndir + dir);
        if constexpr (GREC) {              // no gate records: the operands of their recomputation instead
          r.r0 = r.r1 = r.r2 = r.r3 = zero4();
          const float* up = reinterpret_cast<const float*>(u16 + (pos * FUSE_C + 8 * q) * 2);
          r.gu[0] = ld4(up); r.gu[1] = ld4(up + 4);
          const int64_t posp = base + (int64_t)(s > 0 ? st - 1 : st) * a.p_step;      // h_prev: the row of the step before (masked at step 0)
          const float* hp = reinterpret_cast<const float*>(hs16 + (posp * H + 8 * q) * 2);
          r.gh[0] = ld4(hp); r.gh[1] = ld4(hp + 4); r.gh[2] = ld4(hp + 32); r.gh[3] = ld4(hp + 36);
        } else {
#if SB_REC_Q24
          const float* rec = a.save_gates + (size_t)blk * (16 * kWideGateDwords) + (w * 192 + lane) * 4;      // three packed pieces (sb_lstm_bf_common.h)
          r.r0 = ld4_rec(rec); r.r1 = ld4_rec(rec + 256); r.r2 = ld4_rec(rec + 512); r.r3 = zero4();
#else
          const float* rec = a.save_gates + (size_t)blk * (16 * 4 * H) + (w * 256 + lane) * 4;
          r.r0 = ld4_rec(rec); r.r1 = ld4_rec(rec + 256); r.r2 = ld4_rec(rec + 512); r.r3 = ld4_rec(rec + 768);
#endif
        }
        r.cp = ld4_rec(a.save_c + (size_t)blk * (16 * H) + (w * 64 + lane) * 4);
      } else {
        const float* rec = a.save_gates + (pos * ndir + dir) * (5 * H) + uoff;
        r.r0 = ld4(rec); r.r1 = ld4(rec + H); r.r2 = ld4(rec + 2 * H); r.r3 = ld4(rec + 3 * H); r.cp = ld4(rec + 4 * H);
      }
      if constexpr (FUSE_C > 0) {          // dy[pos][8q .. 8q+7] (channels beyond C are zero)
        const bool okc = 8 * q < FUSE_C;
        const float* dyp = (XP && !SB_EXP_IDX64) ? a.dy + (size_t)((unsigned)base + (unsigned)st * (unsigned)a.p_step) * FUSE_C + (okc ? 8 * q : 0)
                              : a.dy + pos * FUSE_C + (okc ? 8 * q : 0);
        const f32x4 d0 = ld4(dyp), d1 = ld4(dyp + 4);
        r.dh = okc ? d0 : zero4();
        r.dy1 = okc ? d1 : zero4();
      } else {
        r.dh = ld4(a.dhs + (pos * ndir + dir) * H + uoff);
        r.dy1 = zero4();
      }
    } else {
      r.r0 = r.r1 = r.r2 = r.r3 = r.cp = r.dh = r.dy1 = zero4();
      r.cp16 = h16x4{0, 0, 0, 0};
      r.ub = r.hb0 = r.hb1 = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    if constexpr (GREC) { if (!valid) { r.gu[0] = r.gu[1] = r.gh[0] = r.gh[1] = r.gh[2] = r.gh[3] = zero4(); } }
    return r;
  };

  f32x4 dc = zero4(), dhrec = zero4();
#ifdef SB_PHASE_TIMING
  unsigned long long tph[5] = {0, 0, 0, 0, 0};
  unsigned tph_n = 0;
  unsigned long long tpc[2] = {0, 0};
#endif
  // Records are fetched TWO steps ahead: along the inter-frame walk consecutive steps are F positions (tens of KB,
  // a new page) apart and the measured load-to-use latency there exceeds one step.  The loop body covers a PAIR of
  // steps and issues both records of the next pair at its top, so the register copies hipcc places at the end of
  // the body (loop-carried values) only touch loads that are two steps old.  (Four steps ahead with a 4-step body:
  // -4 % on the small inter-frame pass, +2..6 % on the others -- the big intra-frame variant crosses 256 VGPRs.)
  auto consume = [&](Raw& raw) {      // pins the s_waitcnt of this record here
    if constexpr (RECOMP) asm volatile("" : "+v"(raw.ub), "+v"(raw.hb0), "+v"(raw.hb1), "+v"(raw.dh));
    else if constexpr (GREC) asm volatile("" : "+v"(raw.gu[0]), "+v"(raw.gu[1]), "+v"(raw.gh[0]), "+v"(raw.gh[1]), "+v"(raw.gh[2]), "+v"(raw.gh[3]), "+v"(raw.dh));
    else asm volatile("" : "+v"(raw.r0), "+v"(raw.r1), "+v"(raw.dh));
    if constexpr (REC16) asm volatile("" : "+v"(raw.cp16)); else asm volatile("" : "+v"(raw.cp));
    if constexpr (!REC16 && !GREC) {
      if constexpr (XP && SB_REC_Q24 != 0) asm volatile("" : "+v"(raw.r2)); else asm volatile("" : "+v"(raw.r2), "+v"(raw.r3));
    }
    if constexpr (FUSE_C > 0) asm volatile("" : "+v"(raw.dy1));
  };
  // SLAB: a step's dgates rows leave one step late -- whole rows (one instruction = 64 lanes x 8 bytes = one row),
  // write-through (sc1), issued where the LDS reads that fetched them have long returned
  // XP: a dgates row is [hi x 256 | scaled lo x 256] halves (1 KB): two planes, each stored as one whole 512-byte piece
  unsigned long long prow[4] = {0, 0, 0, 0}, prowl[4] = {0, 0, 0, 0};
  int prow_st = -1;
  constexpr int DGROW = XP ? 8 * H : 4 * H;         // halves per dgates row in HBM
  auto rows_out = [&]() {
    if (prow_st >= 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (sbase[i] >= 0) {
          _Float16* row = reinterpret_cast<_Float16*>(a.dgates) + (sbase[i] + (int64_t)prow_st * a.p_step) * DGROW + 4 * lane;
          __hip_atomic_store(reinterpret_cast<unsigned long long*>(row), prow[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if constexpr (XP)
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(row + 4 * H), prowl[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      prow_st = -1;
    }
  };
  auto step = [&](int s, const Raw& raw, int slot = 0) {    // slot: LDS dgates tile of this step (FST)
    const int cur = s & 1;
    SB_TICK(c0);
    SB_TICK(c1);
    f32x4 gi, gf, gg, go;
    if constexpr (RECOMP) {
      f32x4 z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = ld4(&BR[g][uoff]);
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WR[g][0][w][lane], raw.ub, z[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WR[g][1][w][lane], raw.hb0, z[g], 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(WR[g][2][w][lane], raw.hb1, z[g], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = sigmoid_pre(z[0][r]);
        gf[r] = sigmoid_pre(z[1][r]);
        gg[r] = tanh_pre(z[2][r]);
        go[r] = sigmoid_pre(z[3][r]);
      }
    } else if constexpr (GREC) {
      // B operands from the pair pieces: u = dwords (0, 2, 4, 6) hi / (1, 3, 5, 7) lo; an h chunk = dwords (0, 1, 4, 5) hi /
      // (2, 3, 6, 7) lo; step 0 sees the initial state
      h16x8 bh[3], bl[3];
      bh[0] = __builtin_bit_cast(h16x8, f32x4{raw.gu[0][0], raw.gu[0][2], raw.gu[1][0], raw.gu[1][2]});
      bl[0] = __builtin_bit_cast(h16x8, f32x4{raw.gu[0][1], raw.gu[0][3], raw.gu[1][1], raw.gu[1][3]});
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const f32x4 vh = {raw.gh[2 * c][0], raw.gh[2 * c][1], raw.gh[2 * c + 1][0], raw.gh[2 * c + 1][1]};
        const f32x4 vl = {raw.gh[2 * c][2], raw.gh[2 * c][3], raw.gh[2 * c + 1][2], raw.gh[2 * c + 1][3]};
        f32x4 sh, sl;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sh[r] = s > 0 ? vh[r] : H0h[c][r]; sl[r] = s > 0 ? vl[r] : H0l[c][r]; }
        bh[1 + c] = __builtin_bit_cast(h16x8, sh);
        bl[1 + c] = __builtin_bit_cast(h16x8, sl);
      }
      f32x4 z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) z[g] = ld4(&BR[g][uoff]);
      // the forward kernel's product order (its mma6): per K-chunk  W.lo x.hi, W.hi x.lo, W.hi x.hi, gates round-robin
#pragma unroll
      for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wg[g][c].lo, bh[c], z[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x7F6);
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wg[g][c].hi, bl[c], z[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x7F6);
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wg[g][c].hi, bh[c], z[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x7F6);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = sigmoid_pre(z[0][r]);
        gf[r] = sigmoid_pre(z[1][r]);
        gg[r] = tanh_pre(z[2][r]);
        go[r] = sigmoid_pre(z[3][r]);
      }
    } else if constexpr (REC16) {
      const h16x8 lo = __builtin_bit_cast(h16x8, raw.r0), hi = __builtin_bit_cast(h16x8, raw.r1);
#pragma unroll
      for (int k = 0; k < 4; ++k) { gi[k] = (float)lo[k]; gf[k] = (float)lo[4 + k]; gg[k] = (float)hi[k]; go[k] = (float)hi[4 + k]; }
    } else if constexpr (XP && SB_REC_Q24 != 0) {
      q24_unpack(raw.r0, raw.r1, raw.r2, gi, gf, gg, go);
    } else {
      gi = raw.r0; gf = raw.r1; gg = raw.r2; go = raw.r3;
    }
#if SB_EXP_RECOMPUTE & 1
    {
      // operands that depend on loaded data (so nothing is hoisted out of the time loop)
      const h16x8 xb = __builtin_bit_cast(h16x8, raw.dh), wa = __builtin_bit_cast(h16x8, f32x4{gi[0], gf[1], gg[2], go[3]});
      f32x4 z[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
      for (int c3 = 0; c3 < 3; ++c3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xb, z[g], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0x7F6);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = __builtin_fmaf(1e-30f, sigmoid_pre(z[0][r]), gi[r]);
        gf[r] = __builtin_fmaf(1e-30f, sigmoid_pre(z[1][r]), gf[r]);
        gg[r] = __builtin_fmaf(1e-30f, tanh_pre(z[2][r]), gg[r]);
        go[r] = __builtin_fmaf(1e-30f, sigmoid_pre(z[3][r]), go[r]);
      }
    }
#endif
    f32x4 dhext = raw.dh;
    if constexpr (DG16 && FUSE_C == 0) dhext *= gS;
    if constexpr (FUSE_C > 0 && XP) {
      h16x8 bh, bl;
      float dv[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        dv[kk] = (kk < 4 ? raw.dh[kk] : raw.dy1[kk - 4]) * gS;
        bh[kk] = (_Float16)dv[kk];
      }
      bl = mix_lo8(bh, dv, kLoUp);
      const f32x4 dxl = __builtin_amdgcn_mfma_f32_16x16x32_f16(Lxh, bl, zero4(), 0, 0, 0);
      dhext = __builtin_amdgcn_mfma_f32_16x16x32_f16(Lxl, bh, zero4(), 0, 0, 0);
      dhext = __builtin_amdgcn_mfma_f32_16x16x32_f16(Lxh, bh, dhext, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) dhext[r] = __builtin_fmaf(dxl[r], kLoDn, dhext[r]);
    } else if constexpr (FUSE_C > 0) {
      bf16x8 bh, bl;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float v = kk < 4 ? raw.dh[kk] : raw.dy1[kk - 4];
        const __bf16 hh = (__bf16)v;
        bh[kk] = hh;
        bl[kk] = (__bf16)(v - (float)hh);
      }
      dhext = mma(Ll, bh, zero4());
      dhext = mma(Lh, bl, dhext);
      dhext = mma(Lh, bh, dhext);
    }
    f32x4 dG[4];
    float hrow[4];                                   // HREC: h of this step, as the forward kernel formed it
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = dhext[r] + dhrec[r];
      const float cpr = REC16 ? (float)raw.cp16[r] : raw.cp[r];
      const float cc = __builtin_fmaf(gf[r], cpr, gi[r] * gg[r]);
      const float tc = tanhf_fast(cc);
      hrow[r] = go[r] * tc;
      const float dO = dh * tc;
      const float dct = __builtin_fmaf(dh * go[r], __builtin_fmaf(-tc, tc, 1.0f), dc[r]);
      dG[0][r] = dct * gg[r] * gi[r] * (1.0f - gi[r]);
      dG[1][r] = dct * cpr * gf[r] * (1.0f - gf[r]);
      dG[2][r] = dct * gi[r] * (1.0f - gg[r] * gg[r]);
      dG[3][r] = dO * go[r] * (1.0f - go[r]);
      dc[r] = dct * gf[r];
    }
    // the 16 dgates of this lane are the two K-chunks of the B operand (3-way bf16 split, or the fp16 values)
    Split3 Bop[2];
    h16x8 Bh[2], Bl[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = dG[2 * c + (kk >> 2)][kk & 3];
      if constexpr (DG16) {
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) Bh[c][kk] = (_Float16)t[kk];
        if constexpr (XP) Bl[c] = mix_lo8(Bh[c], t, kLoUp);
      } else {
        Bop[c] = split8(t);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    SB_TICK(c2);
    if constexpr (FST > 0) {
      // rows of sequences beyond nseq carry zeros (their records are zero), so every lane writes
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h16x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = Bh[g >> 1][4 * (g & 1) + r];
        *reinterpret_cast<h16x4*>(&DG[slot][j][g * H + uoff]) = t;
        if constexpr (XP) {
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = Bl[g >> 1][4 * (g & 1) + r];
          *reinterpret_cast<h16x4*>(&DGL[slot][j][g * H + uoff]) = t;
        }
      }
      if constexpr (HREC) {                            // (hi x 4, lo x 4) of units 16w + 4q .. + 3: the forward kernel's hs pair
        const h16x4 hh4 = {(_Float16)hrow[0], (_Float16)hrow[1], (_Float16)hrow[2], (_Float16)hrow[3]};
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x2 hi2 = __builtin_bit_cast(u32x2, hh4);
        const u32x4 hp8 = {hi2[0], hi2[1], mix_pair(hi2[0], -1.0f, hrow[0], hrow[1]), mix_pair(hi2[1], -1.0f, hrow[2], hrow[3])};
        *reinterpret_cast<u32x4*>(hring(s, j) + 16 * (4 * w + q)) = hp8;
      }
    } else if constexpr (SLAB) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        h16x4 t;
#pragma unroll
        for (int r = 0; r < 4; ++r) t[r] = Bh[g >> 1][4 * (g & 1) + r];
        *reinterpret_cast<h16x4*>(&DS[cur][j][g * H + uoff]) = t;
        if constexpr (XP) {
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = Bl[g >> 1][4 * (g & 1) + r];
          *reinterpret_cast<h16x4*>(&DSL[cur][j][g * H + uoff]) = t;
        }
      }
    } else
    if (valid && !(SB_EXP_SKIP & 256)) {
      const int st = rev ? S - 1 - s : s;
      const int64_t pos = base + (int64_t)st * a.p_step;
      if constexpr (DG16) {
        _Float16* dg = reinterpret_cast<_Float16*>(a.dgates) + (pos * ndir + dir) * (4 * H) + uoff;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          h16x4 t;
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] = Bh[g >> 1][4 * (g & 1) + r];
          *reinterpret_cast<h16x4*>(dg + g * H) = t;
        }
      } else {
        float* dg = a.dgates + (pos * ndir + dir) * (4 * H) + uoff;
        st4(dg, dG[0]); st4(dg + H, dG[1]); st4(dg + 2 * H, dG[2]); st4(dg + 3 * H, dG[3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 part[4] = {zero4(), zero4(), zero4(), zero4()};
    if constexpr (DG16) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) part[ot] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ot][c].lo, Bh[c], part[ot], 0, 0, 0);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) part[ot] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ot][c].hi, Bh[c], part[ot], 0, 0, 0);
      }
      if constexpr (XP) {                              // the scaled low terms of the dgates: accumulated apart, folded in
        f32x4 partx[4] = {zero4(), zero4(), zero4(), zero4()};
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int ot = 0; ot < 4; ++ot) partx[ot] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[ot][c].hi, Bl[c], partx[ot], 0, 0, 0);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[ot][r] = __builtin_fmaf(partx[ot][r], kLoDn, part[ot][r]);
      }
    } else
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) part[ot] = mma(At[ot][c].l, Bop[c].h, part[ot]);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) part[ot] = mma(At[ot][c].h, Bop[c].l, part[ot]);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) part[ot] = mma(At[ot][c].m, Bop[c].m, part[ot]);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) part[ot] = mma(At[ot][c].m, Bop[c].h, part[ot]);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) part[ot] = mma(At[ot][c].h, Bop[c].m, part[ot]);
#pragma unroll
      for (int ot = 0; ot < 4; ++ot) part[ot] = mma(At[ot][c].h, Bop[c].h, part[ot]);
    }
    SB_TICK(c3);
    if constexpr (SLAB) rows_out();                    // the previous step's rows, in the shadow of the MFMAs above
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) st4(&P[cur][w][ot][lane][0], part[ot]);
    SB_TICK(c4);
    __syncthreads();
    dhrec = ld4(&P[cur][0][w][lane][0]) + ld4(&P[cur][1][w][lane][0]) + ld4(&P[cur][2][w][lane][0]) +
            ld4(&P[cur][3][w][lane][0]);
    if constexpr (SLAB) {                              // rows 4w .. 4w + 3 of the step: picked up now, stored by rows_out()
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        prow[i] = *reinterpret_cast<const unsigned long long*>(&DS[cur][4 * w + i][4 * lane]);
        if constexpr (XP) prowl[i] = *reinterpret_cast<const unsigned long long*>(&DSL[cur][4 * w + i][4 * lane]);
      }
      prow_st = rev ? S - 1 - s : s;
    }
#ifdef SB_PHASE_TIMING
    asm volatile("" : "+v"(dhrec));
    SB_TICK(c5);
    tph[0] += c1 - c0; tph[1] += c2 - c1; tph[2] += c3 - c2; tph[3] += c4 - c3; tph[4] += c5 - c4;
    ++tph_n;
#endif
  };
  const int ntiles = (a.nseq + 15) / 16;
  const int nitems = SEG ? ntiles * a.seg_count : ntiles;          // !SEG: gridDim.x == ntiles, one item each
  // CONS prologue (all 512 threads): dy1 = LN-backward(du; x, gamma) + res over the tile's positions [16 tile nsteps, ..) --
  // the arithmetic of ln_bwd_kernel<32> (16 lanes per position, two channels per lane) -- its LayerNorm parameter gradients
  // (direction 0 only: both directions run the same rows), max |dy1| -> this tile's scale S
  // The LayerNorm parameter sums of a tile go through the (idle between tiles) du exchange buffer R into a 64-float LDS
  // accumulator: no registers carried through the tile loops of either role (the first version kept 8 per thread and pushed
  // the kernel from 6 to 74 spilled registers).
  __shared__ float pro_ln[CONS ? 64 : 1];
  if constexpr (CONS) { if (tid < 64) pro_ln[tid] = 0.f; }
  // (chunk_tag: compile-time role of the caller -- the rescaling of the chunk role's running sums must not even be instantiated
  // in the recurrence role's loop, or its ~110 accumulator registers become live through that loop: 74 spilled registers and a
  // 45 % slower tile in the first version)
  // ONE prologue per tile and XCD (round 4, second step): both directions of a tile need the same dy1 rows, and run by both the
  // prologue moved 1.5 GB per block -- 0.36 ms of a 2.9 ms block at the chip's HBM rate, against 0.2 ms for the plain order's
  // LayerNorm-backward kernel (scripts/exp_cross_consume.py with -DSB_EXP_CONS=1).  Per consumer tile three words (pst: claim,
  // done, max |dy1|; zeroed by the produce call): whoever claims a tile first computes its rows, publishes the maximum and raises
  // `done`; the other direction's item waits for `done` (bounded) and reads the maximum -- IF it runs behind the same L2.  The
  // rows are plain (write-back) stores: CUs of one XCD share them through their L2, another XCD's L2 would fetch the lines from
  // memory, where they may not have arrived (write-through stores were tried: a reader in the SAME workgroup then saw unwritten
  // memory).  The claim and done words therefore carry 1 + the XCD, and an item that finds a tile claimed from another XCD (it
  // stole the item, or the partition has fewer XCDs than queues) computes the rows again for itself -- identical values, and it
  // leaves the LayerNorm parameter sums to the owner.  So that a wait is rarely a wait, every item first LOOKS AHEAD: the tile
  // kLook8 draws further down ITS queue is claimed and computed now if its slabs are in and nobody has it.
  int* const pst = CONS ? a.seg_flags : nullptr;
#ifndef SB_EXP_DD
#define SB_EXP_DD 0            // developer experiments: bit 0 no look-ahead, bit 1 recompute instead of waiting for `done`
#endif
  #ifndef SB_EXP_LOOK
#define SB_EXP_LOOK 20
#endif
  constexpr int kLook8 = (SB_EXP_DD & 1) ? (1 << 26) : SB_EXP_LOOK;
  // developer statistics (bit 5): behind the per-tile words -- [0..15] workgroups per (XCD, direction), [16] look-ahead rows computed,
  // [17] own, [18] recomputed (owner behind another L2), [19] waited for, [20] items taken from another XCD's queue
  // [21] ticks / 64 in the row computation, [22] waiting for `done`, [23] waiting for the producer's slabs
  auto dbg_add = [&](int i, int v = 1) {
    if constexpr ((SB_EXP_DD & 32) != 0) { if (tid == 0) __hip_atomic_fetch_add(pst + 3 * ntiles + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  };
  auto dbg_now = [&]() -> unsigned long long { return (SB_EXP_DD & 32) ? __builtin_readcyclecounter() : 0ull; };
  auto pro_claim = [&](int t) __attribute__((always_inline)) -> int {               // 0: this workgroup owns tile t's rows; else 1 + the owner's XCD  (uniform)
    if (tid == 0) {
      int expected = 0;
      __hip_atomic_compare_exchange_strong(pst + t, &expected, 1 + xcd, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ord_item = expected;                           // (the old value: 0 = exchanged)
    }
    __syncthreads();
    const int v = ord_item;
    __syncthreads();
    return v;
  };
  auto pro_ready = [&](int packed) __attribute__((always_inline)) -> bool {         // non-blocking cross_wait: are the tile's producer slabs complete?  (uniform)
    const int need = (packed & 0xFFF) + 1, lo = (packed >> 12) & 0x3FF, hi = (packed >> 22) & 0x3FF;
    if (tid == 0) ord_item = 1;
    __syncthreads();
    for (int t = lo + tid; t <= hi; t += 512)
      if (__hip_atomic_load(a.slab_flags + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) ord_item = 0;
    __syncthreads();
    const int v = ord_item;
    __syncthreads();
    return v != 0;
  };
  auto pro_done_wait = [&](int t) __attribute__((always_inline)) -> bool {          // bounded like cross_wait; uniform
    __shared__ int dw_abort;
    if (tid == 0) {
      int bad = 0;
      unsigned spins = 0;
      while (sb_poll(pst + ntiles + t) == 0) {
        ++spins;
        if (sb_wait_over(a.sched_status, spins, SB_TRIP_CROSS_ROWS, t, pst + t, 1 + xcd)) { bad = 1; break; }
        sb_poll_pause();
      }
      dw_abort = bad;
    }
    __syncthreads();
    return dw_abort == 0;
  };
  // own: this workgroup owns the tile (it reports the LayerNorm sums and publishes); -> max |dy1| of the tile (uniform)
  // (always_inline: called from four places, hipcc otherwise makes it a real function -- calls, a stack, 1 KB of scratch per lane
  //  and a consumer 0.5 ms slower)
  auto pro_compute = [&](int tile, bool own) __attribute__((always_inline)) -> float {
    constexpr int CC = FST > 0 ? FST : 32;
    static_assert(!CONS || CC == 32, "prologue: 8 lanes x 4 channels per position");
    const int cp8 = tid & 7;                         // this lane's channels 4 cp8 .. 4 cp8 + 3
    const int64_t p_lo = (int64_t)tile * 16 * S, p_hi = (int64_t)min(tile * 16 + 16, a.nseq) * S;
    const f32x4 gam = ld4(a.pro_ln_g + 4 * cp8);
    float amax = 0.f;
    f32x4 pdg = zero4(), pdb = zero4();
    // 64 positions per pass and group of four passes: the 12 sixteen-byte loads of a lane are in flight before the first use
    constexpr int U = 4;
    for (int64_t pb = p_lo + (tid >> 3); pb < ((SB_EXP_CONS & 1) ? p_lo : p_hi); pb += 64 * U) {
      f32x4 g[U], x[U], rs[U];
#pragma unroll
      for (int e = 0; e < U; ++e) {
        const int64_t pp = pb + 64 * e;
        const int64_t pc = (pp < p_hi ? pp : pb) * CC + 4 * cp8;             // (clamped: rows past the tile are masked below)
        g[e] = ld4(a.pro_du + pc);
        x[e] = ld4(a.pro_x + pc);
        rs[e] = ld4(a.pro_res + pc);
      }
#pragma unroll
      for (int e = 0; e < U; ++e) {
        const int64_t pp = pb + 64 * e;
        const bool ok = pp < p_hi;                                           // uniform over the 8 lanes of a position
        const float mean = row8_sum((x[e][0] + x[e][1]) + (x[e][2] + x[e][3])) * (1.0f / CC);
        f32x4 d, xh, gg, o;
        float sq = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) { d[v] = x[e][v] - mean; sq += d[v] * d[v]; }
        const float rstd = 1.0f / sqrtf(row8_sum(sq) * (1.0f / CC) + 1e-5f);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          xh[v] = d[v] * rstd;
          gg[v] = g[e][v] * gam[v];
          s1 += gg[v];
          s2 += gg[v] * xh[v];
          if (ok) { pdg[v] += g[e][v] * xh[v]; pdb[v] += g[e][v]; }
        }
        const float m1 = row8_sum(s1) * (1.0f / CC), m2 = row8_sum(s2) * (1.0f / CC);
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = rstd * (gg[v] - m1 - xh[v] * m2) + rs[e][v];
        if (ok) {
          st4(a.pro_dy + pp * CC + 4 * cp8, o);
          amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
        }
      }
    }
    {                                                // this tile's LayerNorm parameter sums: 64 lane groups -> R
      float* red = &R[0][0][0][0][0][0];
      st4(red + (tid >> 3) * 64 + 4 * cp8, pdg);
      st4(red + (tid >> 3) * 64 + 32 + 4 * cp8, pdb);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) pro_red[tid >> 6] = amax;
    __builtin_amdgcn_s_waitcnt(0);                   // this wave's dy1 rows are out ...
    __syncthreads();                                 // ... and so are everybody's: the tile's readers may start
    if (own && tid < 64) {                           // (the tile's owner reports its sums)
      const float* red = &R[0][0][0][0][0][0];
      float sum = 0.f;
      for (int g = 0; g < 64; ++g) sum += red[g * 64 + tid];
      pro_ln[tid] += sum;
    }
    float m = pro_red[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) m = fmaxf(m, pro_red[i]);
    if (own && tid == 0) {                           // publish: the maximum, then (acknowledged) the flag
      __hip_atomic_store(pst + 2 * ntiles + tile, __float_as_int(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_s_waitcnt(0);
      __hip_atomic_store(pst + ntiles + tile, 1 + xcd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();                                 // (pro_red and R are rewritten: the next prologue, this tile's chunks)
    return m;
  };
  auto pro_scale = [&](float m, auto chunk_tag) __attribute__((always_inline)) {
    constexpr bool kChunkRole = decltype(chunk_tag)::value;
    const float Sn = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(
        (m > 0.f && m < 3.0e38f) ? exp2f(-ceilf(log2f(m))) : 1.0f)));
    const float ratio = Sn * invS;                   // a power of two: the running sums move to the new scale exactly
    gS = Sn;
    invS = uniform_f(1.0f / Sn);
    if constexpr (FST > 0 && kChunkRole) {
      if (ratio != 1.0f) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
          for (int kt = 0; kt < KT; ++kt) wacc[nt][kt] *= ratio;
          csum[nt] *= ratio; csumx[nt] *= ratio;
        }
#pragma unroll
        for (int ct = 0; ct < CK; ++ct) { lacc[ct] *= ratio; lbs[ct] *= ratio; }
      }
    }
  };
  // an item's entry: look ahead, then this tile's rows -- computed here or awaited -- and its scale; false: watchdog
  auto prologue = [&](int item, int tile, auto chunk_tag) __attribute__((always_inline)) -> bool {
    if ((item & 7) != xcd) dbg_add(20);
    const int la = item + 8 * kLook8;                // same queue (same residue mod 8): a tile this XCD is the home of
    if ((item & 7) == xcd && la < nitems) {
      const int tl = a.tile_order[la];
      if (pro_ready(a.tile_need[la]) && pro_claim(tl) == 0) { const auto t0 = dbg_now(); pro_compute(tl, true); dbg_add(16); dbg_add(21, (int)((dbg_now() - t0) >> 6)); }
    }
    float m;
    const int owner = (SB_EXP_DD & 2) ? 9 : pro_claim(tile);
    if (owner == 0 || owner != 1 + xcd) {            // mine, or computed behind another L2: (re)compute the rows here
      const auto t0 = dbg_now();
      if (!cross_wait(a.tile_need[item])) return false;
      const auto t1 = dbg_now();
      m = pro_compute(tile, owner == 0);
      dbg_add(owner == 0 ? 17 : 18);
      dbg_add(23, (int)((t1 - t0) >> 6)); dbg_add(21, (int)((dbg_now() - t1) >> 6));
    } else {
      dbg_add(19);
      const auto t0 = dbg_now();
      if (!pro_done_wait(tile)) return false;
      dbg_add(22, (int)((dbg_now() - t0) >> 6));
      m = __int_as_float(__builtin_amdgcn_readfirstlane(
          __hip_atomic_load(pst + 2 * ntiles + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
    }
    pro_scale(m, chunk_tag);
    return true;
  };
  if constexpr (CONS) { if (ord_first < nitems) dbg_add(2 * xcd + dir); }
  if constexpr (SPLIT) {
    // Periods of two barriers.  Recurrence role, period k: the two steps of pair k (dgates -> LDS slots 2 (k & 1), + 1); after
    // the last pair one empty period.  Chunk role, period 0: the Linear's top row; period k >= 1: the chunk of pair k - 1
    // (du before the first barrier, its 4-wave reduction + dW + dW_lin after it).  The slots of pair k - 1 are rewritten in
    // period k + 1, behind the second barrier of period k.  The role branch sits OUTSIDE the tile loop: inside it the
    // compiler would keep each role's loop-carried registers (dW sums; weights) alive through the other role's body.
    // Time segments (SEG): an item is (tile, segment) and walks steps s_hi .. s_lo; the hand-off wait and the state publish
    // carry workgroup barriers of their own, which the chunk role mirrors.
    // Issue priority for the chunk role's waves (round 6).  The phase table (profiles/r06_phase_table.txt) shows them busy 92 % of
    // a period with the recurrence waves parked at its barriers, and every SIMD hosts one wave of each role: when both are
    // ready the chunk wave should issue.  Same-box A/B (profiles/r06_ab_prio.txt): C = 32 (big) +1.2 .. +1.4 % on the train step
    // at priority 1, 2 or 3 alike; C = 16 (small) -1 % -- so only the C = 32 instantiations raise it.  -DSB_CHUNK_PRIO=n overrides
    // (n = 0: off) for every width.
#ifdef SB_REC_PRIO
    if (!crole) __builtin_amdgcn_s_setprio(SB_REC_PRIO);
#endif
#ifdef SB_CHUNK_PRIO
    if (crole && SB_CHUNK_PRIO > 0) __builtin_amdgcn_s_setprio(SB_CHUNK_PRIO);
#else
    if constexpr (FST == 32) { if (crole) __builtin_amdgcn_s_setprio(1); }
#endif
    if (!crole) {
      for (int item = CONS ? ord_first : (int)blockIdx.x; item < nitems; item = CONS ? ord_next() : item + (int)gridDim.x) {
        const int seg = SEG ? item / ntiles : 0;
        const int tile = CONS ? a.tile_order[item] : SEG ? item - seg * ntiles : item;
        const int s_hi = SEG ? S - 1 - seg * a.seg_len : S - 1;
        const int s_lo = SEG ? max(0, s_hi - a.seg_len + 1) : 0;
        const int npairs = (s_hi - s_lo + 2) / 2;                  // the last pair may be a single step
        set_tile(tile);
        if constexpr (CONS) {                                      // this tile's incoming gradient rows (and the look-ahead)
          if (!prologue(item, tile, std::false_type{})) return;
        }
        dc = zero4();
        dhrec = zero4();
        if constexpr (SEG) {
          if (seg > 0) {
            if (!seg_wait(a.seg_flags, tile, seg, a.sched_status, SB_TRIP_BWD_SEGMENT)) return;
            const float* st = a.seg_state + ((size_t)tile * 2 * 16 + j) * H + uoff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              dc[r] = __hip_atomic_load(st + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              dhrec[r] = __hip_atomic_load(st + 16 * H + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
        // records ONE step ahead (-DSB_SPLIT_LOOK=2: TWO, as in the one-role kernel -- round 6 A/B, see DESIGN.md 9: with the
        // 3-piece Q24 records a Raw is 24 registers and the recurrence role's loop holds no spill)
#ifndef SB_SPLIT_LOOK
#define SB_SPLIT_LOOK 1
#endif
        Raw nxt = load_raw(s_hi);
#if SB_SPLIT_LOOK == 2
        Raw nxt2 = load_raw(max(s_hi - 1, 0));
#endif
        int s = s_hi;
        for (int k = 0; k < npairs; ++k, s -= 2) {
          Raw curA = nxt;
          SB_TICK(p0);
          consume(curA);
          __builtin_amdgcn_sched_barrier(0);
          SB_TICK(p1);
#if SB_SPLIT_LOOK == 2
          nxt = load_raw(max(s - 2, 0));
#else
          nxt = load_raw(max(s - 1, 0));
#endif
          __builtin_amdgcn_sched_barrier(0);
#ifdef SB_PHASE_TIMING
          SB_TICK(p2);
          tpc[0] += p1 - p0; tpc[1] += p2 - p1;
#endif
          step(s, curA, 2 * (k & 1));
          if (s - 1 >= s_lo) {
#if SB_SPLIT_LOOK == 2
            Raw curB = nxt2;
            consume(curB);
            __builtin_amdgcn_sched_barrier(0);
            nxt2 = load_raw(max(s - 3, 0));
#else
            Raw curB = nxt;
            consume(curB);
            __builtin_amdgcn_sched_barrier(0);
            nxt = load_raw(max(s - 2, 0));
#endif
            __builtin_amdgcn_sched_barrier(0);
            step(s - 1, curB, 2 * (k & 1) + 1);
          } else {                                                 // odd step count: an empty second half
            const h16x4 hz = {0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              *reinterpret_cast<h16x4*>(&DG[2 * (k & 1) + 1][j][g * H + uoff]) = hz;
              if constexpr (XP) *reinterpret_cast<h16x4*>(&DGL[2 * (k & 1) + 1][j][g * H + uoff]) = hz;
            }
            __syncthreads();
          }
        }
        __syncthreads();                                           // the chunk role's last period
        __syncthreads();
        if constexpr (SEG) {
          if (s_lo > 0) {
            float* st = a.seg_state + ((size_t)tile * 2 * 16 + j) * H + uoff;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              __hip_atomic_store(st + r, dc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              __hip_atomic_store(st + 16 * H + r, dhrec[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) __hip_atomic_store(a.seg_flags + tile, seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        __syncthreads();                                           // between items
      }
#ifdef SB_PHASE_TIMING
      if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && tph_n > 0)
      {
        // (the first step of a pair is the timed one for the two leading columns: per pair -> per step as measured)
        g_phase_bwd_rec[w][0] = (float)tpc[0] * 2 / tph_n; g_phase_bwd_rec[w][1] = (float)tpc[1] * 2 / tph_n;
        for (int i = 0; i < 4; ++i) g_phase_bwd_rec[w][2 + i] = (float)tph[i + 1] / tph_n;
      }
#endif
      if constexpr (!CONS) return;                                 // (CONS: the LayerNorm partial sums of this role's threads follow)
    } else
    for (int item = CONS ? ord_first : (int)blockIdx.x; item < nitems; item = CONS ? ord_next() : item + (int)gridDim.x) {
      const int seg = SEG ? item / ntiles : 0;
      const int tile = CONS ? a.tile_order[item] : SEG ? item - seg * ntiles : item;
      const int s_hi = SEG ? S - 1 - seg * a.seg_len : S - 1;
      const int s_lo = SEG ? max(0, s_hi - a.seg_len + 1) : 0;
      const int npairs = (s_hi - s_lo + 2) / 2;
      set_tile(tile);
      if constexpr (CONS) {
        if (!prologue(item, tile, std::true_type{})) return;
      }
      if constexpr (SEG) { if (seg > 0) { if (!seg_wait(a.seg_flags, tile, seg, a.sched_status, SB_TRIP_BWD_SEGMENT)) return; } }
      if constexpr (LINW && !HREC) { if (s_hi == S - 1) lin_top(); }
      if constexpr (STG) stage_issue(s_hi, s_hi - 1 >= s_lo, 1, tile, [] {});     // the first chunk's rows: read in period 1 (buffer 1)
      __syncthreads();
      if constexpr (LINW && HREC) lin_top();                               // h of walk index S - 1 is in the ring by now
      if constexpr (STG) __builtin_amdgcn_s_waitcnt(0);                    // the first chunk's rows have landed
      __syncthreads();
      int s = s_hi;
      int slab_fill = 0, slab_idx = 0;                             // PROD: steps flushed into the current slab, its index
#ifdef SB_PHASE_TIMING
      unsigned long long cph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
      for (int k = 1; k <= npairs; ++k, s -= 2) {                  // chunk of pair k - 1: steps (s, s - 1)
        SB_TICK(q0);
        const bool two = s - 1 >= s_lo;
        const int pb = (k - 1) & 1, rb = STG ? 0 : pb;
        PairOps ops2;
        if constexpr (STG) {
          stg_sel = k & 1;                                         // this period's rows: landed and published before the last barrier
          chunk(2 * pb, rb, ops2, s, two, 0);
        } else {
          ops2 = pair_loads(s, two);
          chunk(2 * pb, rb, ops2, s, two, 0);                      // du partial sums -> R[rb]
        }
        __builtin_amdgcn_sched_barrier(0);
        SB_TICK(q1);
        __syncthreads();
        SB_TICK(q2);
        // (round 6: the flush's du store address may be a scratch reload too: it runs INSIDE stage_issue, between the address
        //  reloads and the copies -- behind the copies its vmcnt(0) waited for them to land, a second full memory latency per period)
#ifndef SB_EXP_NO_STAGE_PIN
        if constexpr (STG) {
          if (k < npairs) stage_issue(s - 2, s - 3 >= s_lo, (k + 1) & 1, tile, [&] { flush(s, two ? 2 : 1, rb, ops2.xq, ops2.rq); });
          else flush(s, two ? 2 : 1, rb, ops2.xq, ops2.rq);
        } else flush(s, two ? 2 : 1, rb, ops2.xq, ops2.rq);
#else
        if constexpr (STG) {
          if (k < npairs) stage_issue(s - 2, s - 3 >= s_lo, (k + 1) & 1, tile, [] {});  // the next period's rows, a period ahead
        }
#endif
#ifdef SB_PHASE_TIMING
        __builtin_amdgcn_sched_barrier(0);
#endif
        SB_TICK(qa);
#ifdef SB_EXP_NO_STAGE_PIN
        flush(s, two ? 2 : 1, rb, ops2.xq, ops2.rq);
#endif
#ifdef SB_PHASE_TIMING
        __builtin_amdgcn_sched_barrier(0);
#endif
        SB_TICK(qb);
        chunk(2 * pb, rb, ops2, s, two, 1);
#ifdef SB_PHASE_TIMING
        __builtin_amdgcn_sched_barrier(0);
#endif
        SB_TICK(qc);
        if constexpr (STG) __builtin_amdgcn_s_waitcnt(0);          // the next period's rows (issued at the top of this phase) have landed
        __builtin_amdgcn_sched_barrier(0);
        SB_TICK(q3);
        __syncthreads();
        if constexpr (PROD) {
          // every du row down to step (two ? s - 1 : s) is out (the s_waitcnt above drained this role's stores before the
          // barrier): count the workgroup into the slab that just became complete (slab_len is even; the last one may be short)
          // One PROGRESS WORD per producer tile (slab_flags[tile] = slabs complete), written through by every lane of the chunk
          // role in EVERY period, branch-free (same address, same value: one 4-byte write).  Per-slab counters as in the
          // overlapped forward cost +0.17 ms here, and so did a store behind `if (slab complete)`: the uniform branch cut the
          // period loop into several basic blocks, and hipcc only pipelines loads / counts its waits inside one.
          slab_fill += two ? 2 : 1;
          const bool full = slab_fill >= a.slab_len || k == npairs;
          slab_idx += full ? 1 : 0;
          slab_fill = full ? 0 : slab_fill;
          if constexpr (!(SB_EXP_CONS & 4))
            __hip_atomic_store(a.slab_flags + tile, slab_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#ifdef SB_PHASE_TIMING
        SB_TICK(q4);
        cph[0] += q1 - q0; cph[1] += q2 - q1; cph[2] += q3 - q2; cph[3] += q4 - q3;
        cph[4] += qa - q2; cph[5] += qb - qa; cph[6] += qc - qb; cph[7] += q3 - qc;      // phase B split: stage issue, flush, dW chunk, wait for the staged rows
#endif
      }
#ifdef SB_PHASE_TIMING
      if (blockIdx.x == 0 && blockIdx.y == 0 && w == 0 && lane == 0 && item == (int)blockIdx.x)
        for (int i = 0; i < 8; ++i) g_phase_bwd_split[i] = (float)cph[i] / npairs;
#endif
      if constexpr (SEG) { if (s_lo > 0) __syncthreads(); }        // (the recurrence role publishes its state)
      __syncthreads();                                             // between items
    }
  } else
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int seg = SEG ? item / ntiles : 0;
    const int tile = SEG ? item - seg * ntiles : item;
    const int s_hi = SEG ? S - 1 - seg * a.seg_len : S - 1;        // this item walks steps s_hi .. s_lo
    const int s_lo = SEG ? max(0, s_hi - a.seg_len + 1) : 0;
    set_tile(tile);
    dc = zero4();
    dhrec = zero4();
    if constexpr (SEG) {
      float* st = a.seg_state + ((size_t)tile * 2 * 16 + j) * H + uoff;
      if (seg > 0) {                                   // see the forward kernel for the hand-off protocol
        if (!seg_wait(a.seg_flags, tile, seg, a.sched_status, SB_TRIP_BWD_SEGMENT)) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dc[r] = __hip_atomic_load(st + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dhrec[r] = __hip_atomic_load(st + 16 * H + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    Raw rA = load_raw(s_hi), rB = load_raw(max(s_hi - 1, 0));
    int s = s_hi;
    if constexpr (FST > 0) {
      // pair k: dgates of its two steps -> LDS slots 2 (k & 1), +1; chunk after the second step's barrier (every wave
      // has written both rows by then); its du partial sums are reduced after the NEXT barrier (flush).  The slots of
      // pair k are rewritten by pair k + 2, two barriers after every wave has finished chunk k.
      int pk = 0, pend_s = 0, pend_n = 0;
      float pend_x[2] = {0.f, 0.f}, pend_r[2] = {0.f, 0.f};
      if constexpr (LINW) { if (s_hi == S - 1) lin_top(); }
      for (; s >= s_lo + 1; s -= 2, pk ^= 1) {
        Raw curA = rA, curB = rB;
        consume(curA);
        __builtin_amdgcn_sched_barrier(0);
        rA = load_raw(max(s - 2, 0));
        rB = load_raw(max(s - 3, 0));
        const PairOps ops2 = pair_loads(s, true);
        __builtin_amdgcn_sched_barrier(0);
        step(s, curA, 2 * pk);
        if (pend_n) flush(pend_s, pend_n, pk ^ 1, pend_x, pend_r);
        consume(curB);
        step(s - 1, curB, 2 * pk + 1);
        chunk(2 * pk, pk, ops2, s, true);
        pend_s = s; pend_n = 2;
        pend_x[0] = ops2.xq[0]; pend_x[1] = ops2.xq[1]; pend_r[0] = ops2.rq[0]; pend_r[1] = ops2.rq[1];
      }
      if (s == s_lo) {                                   // odd step count: a chunk with an empty second half
        consume(rA);
        const PairOps ops1 = pair_loads(s_lo, false);
        step(s_lo, rA, 2 * pk);
        if (pend_n) flush(pend_s, pend_n, pk ^ 1, pend_x, pend_r);
        const h16x4 hz = {0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *reinterpret_cast<h16x4*>(&DG[2 * pk + 1][j][g * H + uoff]) = hz;
          if constexpr (XP) *reinterpret_cast<h16x4*>(&DGL[2 * pk + 1][j][g * H + uoff]) = hz;
        }
        __syncthreads();
        chunk(2 * pk, pk, ops1, s_lo, false);
        pend_s = s_lo; pend_n = 1;
        pend_x[0] = ops1.xq[0]; pend_x[1] = ops1.xq[1]; pend_r[0] = ops1.rq[0]; pend_r[1] = ops1.rq[1];
        pk ^= 1;
      }
      __syncthreads();                                   // R of the last chunk complete
      if (pend_n) flush(pend_s, pend_n, pk ^ 1, pend_x, pend_r);
    } else {
    auto slab_signal = [&](int k) {                    // every dgates row of slab k of this tile is on its way: count in
      rows_out();
      __builtin_amdgcn_s_waitcnt(0);
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(a.slab_flags + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    for (; s >= s_lo + 1; s -= 2) {
      Raw curA = rA, curB = rB;
      consume(curA);
      __builtin_amdgcn_sched_barrier(0);
      rA = load_raw(max(s - 2, 0));
      rB = load_raw(max(s - 3, 0));
      __builtin_amdgcn_sched_barrier(0);
      step(s, curA);
      consume(curB);
      step(s - 1, curB);
      if constexpr (SLAB) {                            // slab_len is even: slabs end on pair boundaries
        const int done = S - (s - 1);
        if (done % a.slab_len == 0) slab_signal(done / a.slab_len - 1);
      }
    }
    if (s == s_lo) { consume(rA); step(s_lo, rA); }
    if constexpr (SLAB) { if (S % a.slab_len != 0) slab_signal(S / a.slab_len); }
    }
    if constexpr (SEG) {
      if (s_lo > 0) {
        float* st = a.seg_state + ((size_t)tile * 2 * 16 + j) * H + uoff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          __hip_atomic_store(st + r, dc[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(st + 16 * H + r, dhrec[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(a.seg_flags + tile, seg + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();                                 // the LDS exchange buffers are reused by the next item
    } else if constexpr (FST > 0) {
      __syncthreads();                                 // persistent workgroups: same, between tiles
    }
  }
  if constexpr (FST > 0) {                           // this workgroup's partial row of the weight / bias gradients
    constexpr int Ktot = FST + H;
    constexpr int LW = BI ? 2 * H : H;               // row length of the dW_lin partial: both directions' columns
    if constexpr (CONS) { if (!crole) return; }      // (the recurrence role came along for the barriers of the draws)
    float* part = a.wpart + (CONS ? (size_t)a.row_base + (size_t)dir * (gridDim.x >> 1) + ((SB_EXP_DD & 128) ? (blockIdx.x >> 1) : (((blockIdx.x >> 4) << 3) | (blockIdx.x & 7)))
                                  : (size_t)dir * gridDim.x + blockIdx.x) *
                  ((size_t)4 * H * Ktot + 4 * H + (LINW ? FST * LW + FST : 0) + (LNB || CONS ? 2 * FST : 0));
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gate = TRA ? 64 * w + 16 * nt + 4 * q + r : 64 * w + 4 * (4 * q + r) + nt;
#pragma unroll
        for (int kt = 0; kt < CK; ++kt) part[(size_t)gate * Ktot + (CK == 2 ? 2 * j + kt : j)] = wacc[nt][kt][r] * invS;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) part[(size_t)gate * Ktot + FST + 4 * j + kt] = wacc[nt][CK + kt][r] * invS;
      }
      const float cs = quad_sum(XP ? __builtin_fmaf(csumx[nt], kLoDn, csum[nt]) : csum[nt]);
      if (q == 0) part[(size_t)4 * H * Ktot + 64 * w + (TRA ? 16 * nt + j : 4 * j + nt)] = cs * invS;
    }
    float* plin = part + (size_t)4 * H * Ktot + 4 * H;           // [C][64] dW_lin, then [C] db_lin
    if constexpr (LINW)
#pragma unroll
    for (int ct = 0; ct < CK; ++ct) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        plin[(size_t)(16 * ct + 4 * q + r) * LW + dir * H + 4 * j + w] = lacc[ct][r] * invS;
        if constexpr (BI) plin[(size_t)(16 * ct + 4 * q + r) * LW + (1 - dir) * H + 4 * j + w] = 0.f;   // other direction's columns
      }
      const float bs = quad_sum(lbs[ct]);              // both directions see every dy row: direction 0 reports the sum
      if (w == 0 && q == 0) plin[(size_t)FST * LW + 16 * ct + j] = dir == 0 ? bs * invS : 0.f;
    }
    if constexpr (CONS) {                            // d(ln gamma) [32], d(ln beta) [32] of the prologue's LayerNorm backward
      const int te = tid & 255;
      if (te < 64) plin[(size_t)FST * LW + FST + te] = pro_ln[te];
    }
    if constexpr (LNB) {                             // LayerNorm parameter gradients: waves 0 / 1 hold the two steps' sums
      __syncthreads();
      float* red = &R[0][0][0][0][0][0];
      const float g = quad_sum(dgam), b = quad_sum(dbet);      // over the four lane rows
      if (q == 0) { red[w * 32 + j] = g; red[w * 32 + 16 + j] = b; }
      __syncthreads();
      const int te = SPLIT ? (tid & 255) : tid;
      if (te < 32) plin[(size_t)FST * H + FST + te] = red[te] + red[32 + te] + red[64 + te] + red[96 + te];
    }
  }
#ifdef SB_PHASE_TIMING
  if (a.dhs && !a.dy && lane == 0 && blockIdx.x < 4) {
    float* d = const_cast<float*>(a.dhs) + (blockIdx.x * 4 + w) * 8;
    for (int i = 0; i < 5; ++i) d[i] = (float)tph[i] / S;
  }
#endif
}


}  // namespace

int sb_launch_lstm_bwd_bf(const sb_lstm_bwd_args& a_in, hipStream_t st) {
  sb_lstm_bwd_args a = a_in;
  const int ntiles = (a.nseq + 15) / 16;
  dim3 grid(ntiles, a.ndir), block(256);
  const bool full = a.nseq % 16 == 0, r16 = a.save_c != nullptr, dg16 = a.gmax != nullptr;
  if (dg16 && !r16) return -1003;
  const int fc = a.dy ? a.C_lin : 0;
  const int cus = device_cu_count();
  if (a.sched_workers < 0 || a.sched_segments < 0 || a.sched_workers > cus) return -1003;
  const int W = a.sched_workers > 0 ? a.sched_workers : cus, kforce = a.sched_segments;
  bool seg = dg16 && a.ndir == 1 && a.seg_state && a.seg_flags && a.sched_status && ntiles >= W &&
             ((ntiles > W && ntiles <= 2 * W) || kforce > 0);
  if (seg) {
    double cost = 0.0;
    const int k = kforce > 0 ? kforce : choose_segments(ntiles, W, a.nsteps, &cost);
    if (k < 2 || (kforce == 0 && cost > 1.30)) seg = false;
    else {
      a.seg_len = ((a.nsteps + k - 1) / k + 3) & ~3;              // as in the forward launcher (pair-unrolled loop here)
      a.seg_count = (a.nsteps + a.seg_len - 1) / a.seg_len;
      grid.x = W;
      (void)sb_flags_zero(a.seg_flags, ntiles, st);
    }
  }
  // fused streaming part (see the kernel): single direction, fused Linear backward with the same channel count
  const bool fst = a.wpart != nullptr;
  const bool wide = a.wide != 0;                    // fp32 records / u / hs, two-term gradients (see the kernel: XP)
  if (wide && ((!fst && !a.slab_flags) || !dg16 || a.hs_f16 || (a.recompute && !a.slab_flags))) return -1003;
  if (fst && a.ndir == 2) {                          // bidirectional fused form: persistent workgroups, one per CU
    // hs: not read by the wide role-split C = 32 form (its recurrence role recomputes h: see HREC in the kernel)
    const bool hrec = wide && a.split && a.C == 32 && fc == 32;
    if (!dg16 || !a.u || (!a.hs && !hrec) || !a.w_ih || !a.w_ih1 || !a.du || !a.dW_ih || !a.dW_hh || !a.db_ih || !a.db_hh ||
        !a.dW_ih1 || !a.dW_hh1 || !a.db_ih1 || !a.db_hh1 || (int64_t)a.nseq * a.nsteps * H >= (1ll << 31))
      return -1003;
    int gx = device_cu_count() / 2;
    if (gx < 1) gx = 1;
    if (gx > ntiles) gx = ntiles;
    dim3 g2(gx, 2);
#define SB_FB(FL, FC_, CC, H16_) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_REC16 | K_DG16 | K_BI | (H16_) * K_HS16B, FC_, CC>), g2, block, 0, st, a)
#define SB_FBX(FL, FC_, CC) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_DG16 | K_BI | K_XP, FC_, CC>), g2, block, 0, st, a)
#define SB_FBXS(FL, FC_, CC) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_DG16 | K_BI | K_XP | K_SPLIT, FC_, CC>), g2, dim3(512), 0, st, a)
#define SB_FBS(FL, FC_, CC, H16_) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_REC16 | K_DG16 | K_BI | (H16_) * K_HS16B | K_SPLIT, FC_, CC>), g2, dim3(512), 0, st, a)
    if (a.split && a.recompute) return -1003;
    if (a.tile_order) {
      // cross-pass consumer (sb_lstm_bwd_cross_consume launches it twice and runs the partial-row reductions itself)
      if (!wide || !a.split || a.C != 32 || fc != 32 || !a.tile_need || !a.ord_counter || !a.slab_flags || !a.slab_started ||
          !a.sched_status || a.ord_grid < 1 || !a.pro_du || !a.pro_x || !a.pro_res || !a.pro_ln_g || !a.pro_dy ||
          a.pro_dy != a.dy || a.p_step != 1 || a.p_inner != a.nsteps || a.row_base < 0)
        return -1003;
      dim3 gc(2 * a.ord_grid);                       // direction = workgroup parity (see the kernel)
      if (full) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_FULL | K_DG16 | K_BI | K_XP | K_SPLIT | K_CONS, 32, 32>), gc, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_DG16 | K_BI | K_XP | K_SPLIT | K_CONS, 32, 32>), gc, dim3(512), 0, st, a);
      SB_CHECK_LAUNCH();
      return 0;
    }
    if (wide && a.split) {                            // role-split workgroups (8 waves): see the kernel
      if (a.C == 16 && fc == 0) { if (full) SB_FBXS(true, 0, 16); else SB_FBXS(false, 0, 16); }
      else if (a.C == 32 && fc == 32) { if (full) SB_FBXS(true, 32, 32); else SB_FBXS(false, 32, 32); }
      else return -1003;
    } else if (wide) {
      if (a.C == 16 && fc == 0) { if (full) SB_FBX(true, 0, 16); else SB_FBX(false, 0, 16); }
      else if (a.C == 32 && fc == 32) { if (full) SB_FBX(true, 32, 32); else SB_FBX(false, 32, 32); }
      else return -1003;
    } else if (a.split) {
      if (a.C == 16 && fc == 0 && !a.hs_f16) { if (full) SB_FBS(true, 0, 16, false); else SB_FBS(false, 0, 16, false); }
      else if (a.C == 32 && fc == 32 && !a.hs_f16) { if (full) SB_FBS(true, 32, 32, false); else SB_FBS(false, 32, 32, false); }
      else if (a.C == 32 && fc == 32) { if (full) SB_FBS(true, 32, 32, true); else SB_FBS(false, 32, 32, true); }
      else return -1003;
    } else
    if (a.C == 16 && fc == 0 && !a.hs_f16) { if (full) SB_FB(true, 0, 16, false); else SB_FB(false, 0, 16, false); }
    else if (a.C == 32 && fc == 32 && !a.hs_f16) { if (full) SB_FB(true, 32, 32, false); else SB_FB(false, 32, 32, false); }
    else if (a.C == 32 && fc == 32 && !a.recompute) { if (full) SB_FB(true, 32, 32, true); else SB_FB(false, 32, 32, true); }
    else if (a.C == 32 && fc == 32) {               // gate recomputation: no gate records, forward weights + biases needed
      if (!a.b_ih[0] || !a.b_hh[0] || !a.b_ih[1] || !a.b_hh[1]) return -1003;
      if (full) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_FULL | K_REC16 | K_DG16 | K_BI | K_HS16B | K_RECOMP, 32, 32>), g2, block, 0, st, a);
      else hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_REC16 | K_DG16 | K_BI | K_HS16B | K_RECOMP, 32, 32>), g2, block, 0, st, a);
    }
    else return -1003;
#undef SB_FBS
#undef SB_FBXS
#undef SB_FBX
#undef SB_FB
    const int64_t ld = (int64_t)4 * H * (a.C + H) + 4 * H + (fc > 0 ? a.C * 2 * H + a.C : 0);
    int rc = sb_launch_stream_reduce(a.wpart, gx, ld, a.C, a.dW_ih, a.dW_hh, a.db_ih, a.db_hh, st);
    if (!rc) rc = sb_launch_stream_reduce(a.wpart + (size_t)gx * ld, gx, ld, a.C, a.dW_ih1, a.dW_hh1, a.db_ih1, a.db_hh1, st);
    if (fc > 0) {                                     // dW_lin [C, 128] / db_lin over the rows of both directions
      const float* plin = a.wpart + (size_t)4 * H * (a.C + H) + 4 * H;
      if (!rc && a.dW_lin) rc = sb_reduce_rows(plin, 2 * gx, ld, a.C * 2 * H, a.dW_lin, st);
      if (!rc && a.db_lin) rc = sb_reduce_rows(plin + a.C * 2 * H, 2 * gx, ld, a.C, a.db_lin, st);
    }
    return rc;
  }
  if (fst) {
    // hs: not read by the wide role-split C = 32 form without time segments (HREC in the kernel: h recomputed from the records)
    const bool hrec1 = wide && a.split && a.C == 32 && !seg;
    if (!dg16 || a.ndir != 1 || !a.u || (!a.hs && !hrec1) || !a.w_ih || !a.dW_ih || !a.dW_hh || !a.db_ih || !a.db_hh ||
        (a.C != 16 && a.C != 32) || fc != a.C || (int64_t)a.nseq * a.nsteps >= (1ll << 31))
      return -1003;
    const bool lnb = a.dx != nullptr;
    if (lnb && (a.C != 16 || !a.ln_x || !a.ln_g)) return -1003;
    if (!lnb && !a.du) return -1003;
#define SB_F(FL, CC, SG, LB) do { \
    if (SG && !fits_one_per_cu<lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_REC16 | K_DG16 | (SG) * K_SEG | (LB) * K_LNB, CC, CC>>()) return -1008; \
    hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_REC16 | K_DG16 | (SG) * K_SEG | (LB) * K_LNB, CC, CC>), grid, block, 0, st, a); } while (0)
#define SB_FC(CC, LB) do { if (full) { if (seg) SB_F(true, CC, true, LB); else SB_F(true, CC, false, LB); } \
                           else { if (seg) SB_F(false, CC, true, LB); else SB_F(false, CC, false, LB); } } while (0)
#define SB_FX(FL, CC, SG, LB) do { \
    if (SG && !fits_one_per_cu<lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_DG16 | (SG) * K_SEG | (LB) * K_LNB | K_XP, CC, CC>>()) return -1008; \
    hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_DG16 | (SG) * K_SEG | (LB) * K_LNB | K_XP, CC, CC>), grid, block, 0, st, a); } while (0)
#define SB_FXC(CC, LB) do { if (full) { if (seg) SB_FX(true, CC, true, LB); else SB_FX(true, CC, false, LB); } \
                            else { if (seg) SB_FX(false, CC, true, LB); else SB_FX(false, CC, false, LB); } } while (0)
#define SB_FS(FL, R16_, CC, SG, LB, XP_) do { \
    if (SG && !fits_one_per_cu<lstm_bwd_rec_bf_kernel<(FL) * K_FULL | (R16_) * K_REC16 | K_DG16 | (SG) * K_SEG | (LB) * K_LNB | (XP_) * K_XP | K_SPLIT, CC, CC>, 512>()) return -1008; \
    hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | (R16_) * K_REC16 | K_DG16 | (SG) * K_SEG | (LB) * K_LNB | (XP_) * K_XP | K_SPLIT, CC, CC>), grid, dim3(512), 0, st, a); } while (0)
#define SB_FSC(R16_, CC, LB, XP_) do { if (full) { if (seg) SB_FS(true, R16_, CC, true, LB, XP_); else SB_FS(true, R16_, CC, false, LB, XP_); } \
                                       else { if (seg) SB_FS(false, R16_, CC, true, LB, XP_); else SB_FS(false, R16_, CC, false, LB, XP_); } } while (0)
    if (a.slab_flags) {                               // cross-pass producer (sb_lstm_bwd_cross_produce)
      if (!a.split || !wide || a.C != 32 || lnb || seg || !a.slab_started || a.slab_len < 2 || (a.slab_len & 1)) return -1003;
      if (full) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_FULL | K_DG16 | K_XP | K_SPLIT | K_PROD, 32, 32>), grid, dim3(512), 0, st, a);
      else hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_DG16 | K_XP | K_SPLIT | K_PROD, 32, 32>), grid, dim3(512), 0, st, a);
    } else
    if (a.split && wide) { if (a.C == 16) { if (lnb) SB_FSC(false, 16, true, true); else SB_FSC(false, 16, false, true); } else SB_FSC(false, 32, false, true); }
    else if (a.split) { if (a.C == 16) { if (lnb) SB_FSC(true, 16, true, false); else SB_FSC(true, 16, false, false); } else SB_FSC(true, 32, false, false); }
    else
    if (wide) { if (a.C == 16) { if (lnb) SB_FXC(16, true); else SB_FXC(16, false); } else SB_FXC(32, false); }
    else
    if (a.C == 16) { if (lnb) SB_FC(16, true); else SB_FC(16, false); } else SB_FC(32, false);
#undef SB_FSC
#undef SB_FS
#undef SB_FXC
#undef SB_FX
#undef SB_FC
#undef SB_F
    const int64_t ld = (int64_t)4 * H * (a.C + H) + 4 * H + a.C * H + a.C + (lnb ? 2 * a.C : 0);
    // one reduction launch for the LSTM part and the riders' column ranges
    const int o0 = 4 * H * (a.C + H) + 4 * H;
    const int ex_off[4] = {o0, o0 + a.C * H, o0 + a.C * H + a.C, o0 + a.C * H + 2 * a.C};
    const int ex_n[4] = {a.C * H, a.C, a.C, a.C};
    float* const ex_out[4] = {a.dW_lin, a.db_lin, lnb ? a.d_ln_g : nullptr, lnb ? a.d_ln_b : nullptr};
    return sb_launch_stream_reduce(a.wpart, (int)grid.x, ld, a.C, a.dW_ih, a.dW_hh, a.db_ih, a.db_hh, st, 4, ex_off, ex_n,
                                   ex_out);
  }
  if (a.slab_flags) {                               // overlapped form: see sb_lstm_bwd_inter_overlapped
    if (seg || !dg16 || a.ndir != 1 || a.slab_len < 2 || (a.slab_len & 1) || (fc != 16 && fc != 32) || !a.slab_started)
      return -1003;
#define SB_SLB(FL, FC) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_REC16 | K_DG16 | K_SLAB, FC, 0>), grid, block, 0, st, a)
#define SB_SLBX(FL, FC) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | K_DG16 | K_SLAB | K_XP, FC, 0>), grid, block, 0, st, a)
    if (wide && a.recompute) {                      // no gate records: recomputed from the u / hs pairs (GREC in the kernel)
      if (fc != 32 || !a.u || !a.hs || !a.w_ih || !a.b_ih[0] || !a.b_hh[0] || !a.save_c) return -1003;
      if (full) hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_FULL | K_DG16 | K_SLAB | K_XP | K_GREC, 32, 0>), grid, block, 0, st, a);
      else hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<K_DG16 | K_SLAB | K_XP | K_GREC, 32, 0>), grid, block, 0, st, a);
    } else if (wide) {
      if (fc == 32) { if (full) SB_SLBX(true, 32); else SB_SLBX(false, 32); }
      else { if (full) SB_SLBX(true, 16); else SB_SLBX(false, 16); }
    } else
    if (fc == 32) { if (full) SB_SLB(true, 32); else SB_SLB(false, 32); }
    else { if (full) SB_SLB(true, 16); else SB_SLB(false, 16); }
#undef SB_SLBX
#undef SB_SLB
    SB_CHECK_LAUNCH();
    return 0;
  }
#define SB_B(FL, R16, FC, D16, SG) do { \
    if (SG && !fits_one_per_cu<lstm_bwd_rec_bf_kernel<(FL) * K_FULL | (R16) * K_REC16 | (D16) * K_DG16 | (SG) * K_SEG, FC, 0>>()) return -1008; \
    hipLaunchKernelGGL((lstm_bwd_rec_bf_kernel<(FL) * K_FULL | (R16) * K_REC16 | (D16) * K_DG16 | (SG) * K_SEG, FC, 0>), grid, block, 0, st, a); } while (0)
#define SB_BR(FL, FC) do { if (seg) SB_B(FL, true, FC, true, true); else if (dg16) SB_B(FL, true, FC, true, false); \
                           else if (r16) SB_B(FL, true, FC, false, false); else SB_B(FL, false, FC, false, false); } while (0)
#define SB_BF(FC) do { if (full) SB_BR(true, FC); else SB_BR(false, FC); } while (0)
  if (fc == 0) SB_BF(0); else if (fc == 16) SB_BF(16); else if (fc == 32) SB_BF(32); else return -1002;
#undef SB_BF
#undef SB_BR
#undef SB_B
  return 0;
}

