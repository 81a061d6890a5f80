/* sound_bubble_hip.h -- C ABI of libsoundbubble_hip.so (gfx950 / MI355X).
 *
 * The reference (chentuochao/Sound_Bubble) has no FFI: its hot path is Python
 * calling torch.nn modules.  Each entry point below replaces the torch op(s)
 * the reference launches for one stage of that path; the reference line each
 * one stands in for is cited.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the
 *     library never allocates or frees device memory and the data-path calls
 *     never synchronise; scratch is passed in.  The one piece of process state
 *     is the side-stream table of the overlapped schedules, managed by the
 *     explicit sb_overlap_init / _reprobe / _shutdown calls (mutex-guarded);
 *   - fp32, contiguous unless a stride is given (strides are in floats);
 *   - asynchronous on `stream` (a hipStream_t passed as void*);
 *   - returns 0 on success, -(hipError_t) on a launch error, -1000-x on a
 *     bad argument; never throws; thread-safe and, the side-stream table
 *     aside, stateless.
 * Activations are channels-last [B, T, F, C]; a "position" p is the dense
 * index over that (b, t, f) grid (or (b, t, k) on the down-sampled intra grid).
 */
#ifndef SOUND_BUBBLE_HIP_H
#define SOUND_BUBBLE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SB_H 64 /* LSTM hidden size the recurrent kernels are built for */

/* The watchdog word (sched_status) after a bounded wait gave up:  site << 28 | timed_out << 27 | index << 14 | seen << 7 | wanted
   (index: tile / slab the wait was for, seen / wanted: the flag's value and the value waited for, both mod 128; timed_out = 0: the
   waiter left because it found the word already set).  Only the FIRST wait to give up writes; decode with
   sound_bubble_amd.ops.decode_trip.  A caller that has read a non-zero word should move to a FRESH word rather than clear it in
   place (sound_bubble_amd.ops does). */
enum {
  SB_TRIP_FWD_SEGMENT = 1,    /* time-segmented forward: the previous segment of the tile never published its state */
  SB_TRIP_FWD_CONSUMER = 2,   /* overlapped forward, intra-frame consumer: the producer's time slab never completed */
  SB_TRIP_BWD_SEGMENT = 3,    /* time-segmented backward recurrence */
  SB_TRIP_BWD_STREAM = 4,     /* overlapped inter-frame backward, stream kernel: the recurrence's dgates slab never completed */
  SB_TRIP_CROSS_SLABS = 5,    /* cross-pass backward, consumer: a producer tile never reached the slab count waited for */
  SB_TRIP_CROSS_ROWS = 6      /* cross-pass backward, consumer: the owner of a tile's prologue rows never raised `done` */
};

/* ---- flag memory of the guarded schedules ----
 * Every word one workgroup raises and another polls (slab_flags / seg_flags / the flags arrays of the overlapped entry points,
 * sched_status) should live in memory the per-XCD L2s do not cache: those L2s are not coherent with each other, and a line that a
 * zero-fill or a poll left resident was seen to serve stale counts (a consumer polling 52 of 82 for seconds while memory held 82).
 * sb_flags_alloc: `bytes` of such memory on the current device (*kind: 3 = uncached, 1 = fine-grained fallback, 0 = ordinary device
 * memory as the last resort); sb_flags_zero: n_ints words zeroed on `stream` with write-through stores (the library's own zeroing
 * of caller flags uses it too); sb_flags_read: n_ints words to the host, synchronising `stream`.  Ordinary device memory still
 * WORKS for every flags argument (rounds 2-4 ran on it); it carries the hazard above. */
int sb_flags_alloc(int64_t bytes, void** ptr, int* kind);
int sb_flags_free(void* ptr);
int sb_flags_zero(void* ptr, int64_t n_ints, void* stream);
int sb_flags_read(const void* ptr, int64_t n_ints, int* host_out, void* stream);

/* ---- recurrent LSTM (forward) -------------------------------------------
 * Replaces LayerNorm(C) + nn.LSTM forward of
 *   intra: dis_embd3/tfgridnet_causal.py:819-823 (plain), :804-808 (conv-LSTM)
 *          optim/tfgridnet_causal.py:690-695,700-703
 *   inter: dis_embd3/tfgridnet_causal.py:832-843 ; optim :711-722
 * Sequence n, step s live at position  (n / n_inner) * p_outer + (n % n_inner)
 * * p_inner + s * p_step.  Direction 1 (bidirectional intra) walks s backwards.
 * Gate order i,f,g,o (torch).  x is the PRE-LayerNorm input [P, C]; the
 * kernel normalises over C (eps 1e-5) with ln_g/ln_b on the fly.
 * hs: [P, ndir*64].  BPTT record (training; nullable): either
 *   save_gates [P, ndir, 5, 64] fp32 = i,f,g,o (post-activation), c_prev        (save_c == NULL), or
 *   save_gates [R, ndir, 256] fp16 gates + save_c [R, ndir, 64] c_prev (compact: 768 / 640 instead of 1280 B per
 *   step; gate values lie in [-1,1], fp16 rounding 2^-12 relative).  Opaque to the caller, to be handed unchanged to
 *   sb_lstm_bwd_rec with the same mma: with mma == 0 R = P, position-major, c_prev fp32; with mma >= 1 c_prev is
 *   fp16 and both are blocked per (16-sequence tile, step, direction) in the kernels' lane order, so every store /
 *   load instruction moves one contiguous KB:  R = ceil(nseq / 16) * 16 * nsteps rows.
 * save_u ([P, C] LayerNorm output) is required whenever save_gates is given.
 * h0/c0 (nullable = zeros), hN/cN (nullable): [nseq, 64], direction 0 only. */
typedef struct {
  int nseq, nsteps, n_inner, ndir, C;
  int64_t p_outer, p_inner, p_step;
  const float* x;
  const float* ln_g; const float* ln_b;
  const float* w_ih[2]; const float* w_hh[2]; const float* b_ih[2]; const float* b_hh[2];
  const float* h0; const float* c0; float* hN; float* cN;
  float* hs; float* save_gates; float* save_u; float* save_c;
  int mma;   /* 0: fp32-input MFMA (exact fma chain); 1: fp16 matrix pipe, operands split hi+lo (11+11 bits), 3 products
                per MAC (dropped term <= 2^-22; default); 2: bf16 matrix pipe, 3-way split (8+8+8 bits), 6 products
                (dropped terms <= 2^-24) */
  /* optional fusion of the Linear(64 -> C) + residual that follows a single-direction LSTM (mma == 1, ndir == 1):
     when lin_w != NULL, y[p, :] = x[p, :] + lin_w[C, 64] . hs[p, :] + lin_b is written as well, and hs may be NULL
     (inference).  tfgridnet_causal.py:843-849.
     With ndir == 2 (intra-frame pass, lin_w [C, 128]; :824-827) the kernel works in PARTIAL mode: y is [P, 2, C] and
     y[p, d, :] = lin_w[:, 64 d .. 64 d + 63] . h_d[p] (+ lin_b for d == 0), without the residual -- the caller finishes
     with sb_add3 (x + y[:, 0] + y[:, 1]); hs [P, 128] is then only needed by the backward kernels (fp16 with aux_f16).
     Records without gates: save_c != NULL with save_gates == NULL (aux_f16 only) stores c_prev, u and hs but not the four
     gates -- for a backward that recomputes them (sb_lstm_bwd_args.recompute). */
  const float* lin_w; const float* lin_b; float* y;
  /* ... and the consumer side of the partial mode (ndir == 1, lin_w != NULL, C == 32): when x_part != NULL ([P, 2, C], the
     y of a preceding bidirectional call) the input row of position p is x[p] + x_part[p, 0] + x_part[p, 1] (summed as
     sb_add3 does); x_sum (nullable, [P, C]) receives that sum -- the backward kernels' pre-LayerNorm input. */
  const float* x_part; float* x_sum;
  /* ... and FiLM in the same epilogue (ndir == 1, lin_w != NULL): when film_w != NULL, film_w / film_b [nseq, C] (the scale /
     shift planes of the FilmLayer that follows, tfgridnet_causal.py:59-68,509-513; constant along the time walk) are applied
     to y before it is stored, y = (x + lin_w . hs + lin_b) * film_w[n] + film_b[n]; y_pre (nullable, [P, C]) receives the
     pre-FiLM value the FiLM backward needs. */
  const float* film_w; const float* film_b; float* y_pre;
  /* optional scratch for time-segmented scheduling of single-direction passes with more 16-sequence tiles than the
     chip has CUs (mma == 1): seg_state [ceil(nseq/16) * 2 * 16 * 64] floats, seg_flags [ceil(nseq/16)] ints (zeroed by
     the call).  seg_count / seg_len are filled in by the library; pass 0. */
  float* seg_state; int* seg_flags; int seg_count, seg_len;
  /* Time-segmented scheduling needs its `sched_workers` workgroups co-resident (one per CU): a segment waits for the
     state its predecessor publishes.  sched_status (one int, zeroed once by the caller and then left alone; REQUIRED for
     the segmented schedule, which is otherwise not used) is the watchdog word: a wait that exceeds ~2^22 polls (seconds)
     sets *sched_status != 0 (a code naming the wait: SB_TRIP_* below) and every workgroup of the launch (and of later launches that see the word set) bails out
     instead of hanging -- the outputs of such a launch are garbage, and the caller must check the word after its next
     synchronisation (sound_bubble_amd.ops.check_sched_status).  sched_workers / sched_segments: 0 = automatic (CU count
     of the device, checked against the kernel's occupancy; segment count minimising the makespan); > 0 overrides them,
     e.g. for a CU-masked / partitioned device where fewer workgroups are guaranteed co-resident, or to force the
     schedule on a small problem (the parity tests do). */
  int* sched_status; int sched_workers, sched_segments;
  /* training, compact records (save_c != NULL) on the 16-bit matrix path (mma >= 1): the two tensors that only the
     backward kernels read are written as fp16 -- save_u [P, C], and hs [P, 64] when the fused Linear is on (lin_w !=
     NULL; y carries the fp32 result forward).  sb_lstm_bwd_stream consumes them as single fp16 terms anyway (u_f16 /
     hs_f16 there), sb_wgrad takes the fp16 hs through in_f16. */
  int aux_f16;
  /* overlapped forward (set by sb_lstm_fwd_produce / sb_lstm_fwd_consume; leave NULL / 0 otherwise) */
  int* slab_flags; int slab_len, slab_need;
  const int* tile_order; const int* tile_need; int* ord_counter; int* ord_started; int ord_guard, ord_grid;
  /* WIDE BPTT state (mma == 1, save_gates and save_c given, aux_f16 == 0): rec_f32 != 0 keeps the reference's own
     precision in everything the backward reads -- save_c [R, ndir, 64] floats (c_prev) and save_gates
     [R, ndir, sb_lstm_wide_rec_dwords()] dwords: the four post-activation gates, which lie in [0, 1] / [-1, 1], as 24-BIT FIXED
     POINT (step 2^-24 / 2^-23: as fine as fp32 at the top of that range), four values in three dwords -- 192 dwords per
     sequence, step and direction (256 = plain fp32 in a -DSB_REC_Q24=0 build); opaque, private to the forward / backward
     kernel pair.  R as for the compact records, both blocked per (16-sequence tile, step, direction) in the kernels' lane
     order like them: 1024 B per step and direction, one contiguous KB per store instruction.  The two side
     outputs only the backward kernels read keep their BYTE SIZE (4 bytes per element) but are OPAQUE pair-form buffers, not
     fp32 arrays: save_u [P][2 C halves] holds the fp16 (hi, lo) terms the forward kernel multiplies with
     ([P][C/2][hi0, hi1, lo0, lo1] for C = 32, [P][C][hi, lo] for C = 16), and hs -- when the Linear is applied in the kernel
     (lin_w != NULL) -- [P][ndir][16 unit quads][hi x 4, lo x 4] halves; with lin_w == NULL hs is plain fp32 [P, ndir*64].
     An integrator passes both UNCHANGED from sb_lstm_fwd to sb_lstm_bwd_rec / sb_lstm_bwd_stream with `wide` set and never
     reads them as floats.  hs may be NULL for the bidirectional pass whose backward recomputes h (sb_lstm_bwd_args.split).
     save_gates == NULL with rec_f32 (single direction, lin_w != NULL, hs given): records WITHOUT the four gates -- c_prev
     and the u / hs pairs only -- for a backward that recomputes the gates from them (sb_lstm_bwd_args.recompute + wide). */
  int rec_f32;
  /* Calls that ask for hs (+ hN / cN) alone -- no records, no fused Linear / summed input / FiLM -- with at most 256
     (sequence, direction) chains are served by a one-workgroup-per-chain kernel on the vector ALU (exact fp32 matrix-vector
     products, weights in registers, one barrier per step; sb_lstm_vec.hip): the streaming chunk step's intra-frame pass is ONE
     sequence per direction, for which a 16-sequence MFMA tile pays a 4x longer step.  no_vec != 0 keeps the tile kernels. */
  int no_vec;
  /* products (mma == 1, inference calls only: no records, no side outputs): 0 / 3 = the default three products per MAC (weights
     AND activations as fp16 hi + lo, the hi x lo cross terms kept: 22 mantissa bits, fp32-class); 2 = OPT-IN reduced-product
     mode -- activations as ONE fp16 term against hi + lo weights, two products per MAC (11-bit activations: NOT fp32-class; the
     bench reports its rel-L2 against the default beside its speed).  -1003 with records / side outputs requested. */
  int products;
  /* overlapped forward, consumer launch next to the producer (ord_guard != 0): a workgroup whose wait for its item's slab runs
     out (~2 ms) hands the item back and stops helping -- the launch behind the producer does it, the outputs stay correct.
     ord_giveups (nullable): counts such waits, for the caller's health report. */
  int* ord_giveups;
  int* ord_ret;              /* the consumer's hand-back block inside the flags (set by sb_lstm_fwd_consume; leave NULL) */
} sb_lstm_fwd_args;
int sb_lstm_fwd(const sb_lstm_fwd_args* a, void* stream);
/* dwords per (sequence, step, direction) of the wide gate records (save_gates with rec_f32): 192 (24-bit fixed point), or 256
   in a fp32-record build.  The caller sizes save_gates with it. */
int sb_lstm_wide_rec_dwords(void);
/* test hook: n floats (a multiple of 16: gates i, f, g, o x 4 units per group, values in [0, 1] / [-1, 1] for g) through the
   record packing and back; packed (nullable): the 12 dwords of every group */
int sb_rec_q24_roundtrip(const float* in, float* out, uint32_t* packed, int64_t n, void* stream);

/* ---- inter-frame forward of block k OVERLAPPED with the intra-frame forward of block k + 1 ------------------------
 * The inter-frame pass of the BASELINE big configuration has 145 serial chains on 256 CUs; the bidirectional intra-frame
 * pass that follows it needs, for a tile of 16 frames (b, t .. t + 15), only the y rows of those frames.  Two calls:
 *   sb_lstm_fwd_produce(a, flags, slab_len, stream): sb_lstm_fwd of a single-direction pass with the fused Linear (y
 *     written; fewer tiles than CUs, no time segments) on `stream`; y rows are stored write-through and after every
 *     slab_len steps (multiple of 4) each tile counts itself into its slab's flag.  flags: sb_lstm_fwd_flag_ints(nsteps,
 *     slab_len) ints, zeroed by the call ([0] producer workgroups started, [1], [2] the consumer's item counters, [3] spare,
 *     520 ints of the consumer's hand-back block, then one flag per slab); uncached memory recommended (sb_flags_alloc).
 *   sb_lstm_fwd_consume(a, flags, slab_len, producer_tiles, order, need, stream): sb_lstm_fwd of the bidirectional
 *     partial-Linear pass (ndir == 2, lin_w != NULL, C == 32; a->x is the producer's y) whose tiles are taken in the
 *     order order[ntiles] (a permutation sorted by need[], need[i] = time slab of the producer that completes the frames
 *     of tile order[i]) as (tile, direction) items drawn from one atomic counter per direction by TWO launches: persistent
 *     workgroups on a side stream of the library (two per CU the producer leaves idle; guarded: a workgroup that does not
 *     see all producer workgroups started within ~50 us leaves), each item waiting for its slab's flag == producer_tiles --
 *     a wait that runs out (~2 ms) hands the item BACK and ends that workgroup's help -- and one workgroup per item on `stream`
 *     behind the producer, taking what is left and what was handed back: every item is processed exactly once whatever the
 *     timing (a->sched_status required: the draining launch's own bounded waits).
 * The consume call must be the next library call after its produce call on that device.  Memory the producer reads or
 * writes must stay allocated until the consume call has returned (the side stream is not ordered after `stream`).
 * -1003 when fewer than 16 CUs stay idle, -1009 without a concurrent side stream (sb_overlap_available). */
int sb_lstm_fwd_flag_ints(int producer_steps, int slab_len);
int sb_lstm_fwd_produce(const sb_lstm_fwd_args* a, int* flags, int slab_len, void* stream);
/* ... with flags_zeroed != 0 the caller has zeroed all sb_lstm_fwd_flag_ints(...) flags itself, in stream order before the
 * call (one fill for many blocks instead of a memset in front of every producer) */
int sb_lstm_fwd_produce_ex(const sb_lstm_fwd_args* a, int* flags, int slab_len, int flags_zeroed, void* stream);
int sb_lstm_fwd_consume(const sb_lstm_fwd_args* a, int* flags, int slab_len, int producer_tiles, const int* order,
                        const int* need, void* stream);
/* test hook: the consumer's hand-back path staged on one stream (guarded launch against a producer that has started but completed
   no slab -> every item handed back; slab flags raised; the launch behind the producer drains counter and return stacks).  flags:
   sb_lstm_fwd_flag_ints(...) ints with nslabs slab flags, zeroed by the call. */
int sb_lstm_fwd_consume_staged_test(const sb_lstm_fwd_args* a, int* flags, int slab_len, int producer_tiles, int nslabs,
                                    const int* order, const int* need, void* stream);

/* ---- recurrent LSTM (backward through time, recurrent part) --------------
 * Autograd of the nn.LSTM calls above (loss.backward(), tain_val.py:75).
 * Reads save_gates and dhs ([P, ndir*64], gradient w.r.t. hs), walks the
 * steps in reverse and writes dgates [P, ndir, 4, 64] (gradient w.r.t. the
 * pre-activation gates).  Input/weight gradients are then position-wise
 * GEMMs (sb_linear_fwd / sb_wgrad). Initial state is assumed zero-grad. */
typedef struct {
  int nseq, nsteps, n_inner, ndir;
  int64_t p_outer, p_inner, p_step;
  const float* w_hh[2];
  const float* save_gates; const float* dhs; float* dgates;
  const float* save_c;        /* non-NULL: compact fp16 record (see sb_lstm_fwd_args) */
  int mma;                    /* as sb_lstm_fwd_args.mma */
  /* optional fusion of the following Linear's backward (mma == 1 only): when dy != NULL, dhs is ignored and
     d(hs)[p, dir*64 + u] = sum_c w_lin[c, dir*64 + u] * dy[p, c] is formed on the fly.  dy [P, C_lin] dense,
     w_lin [C_lin, ndir*64] (nn.Linear weight), C_lin = 16 or 32. */
  const float* dy; const float* w_lin; int C_lin;
  /* compact dgates (mma == 1, compact records only): when gmax != NULL, dgates is written as fp16 [P, ndir, 4, 64]
     holding S * dgates with S = 2^-ceil(log2(*gmax)) -- *gmax = max |incoming gradient| from sb_absmax -- the whole
     backward recurrence is linear in that gradient, so it simply runs on the scaled values. */
  const float* gmax;
  /* optional scratch for time-segmented scheduling, as in sb_lstm_fwd_args (needs gmax != NULL) */
  float* seg_state; int* seg_flags; int seg_count, seg_len;
  int* sched_status; int sched_workers, sched_segments;     /* watchdog word + overrides, as in sb_lstm_fwd_args */
  /* optional fusion of the streaming part (sb_lstm_bwd_stream) into the recurrence: single direction, gmax != NULL,
     u [P, C] and hs [P, 64] the fp16 side outputs of sb_lstm_fwd (aux_f16).  When wpart != NULL the dgates never leave
     the chip: every two steps the workgroup multiplies the 32 (step, sequence) dgates rows it holds in LDS into
     running dW_ih / dW_hh / db sums and forms du [P, C] = dgates . W_ih (gradient w.r.t. the LayerNorm output);
     `dgates` is not written.  h_prev of (sequence, step) is hs at step - 1 (zero at step 0).  The weight gradient of
     the fused Linear rides along (dW_lin [C, 64] += dy^T hs, db_lin [C] += column sums of dy; dy enters as a single
     scaled fp16 term like the dgates).  wpart: one row of 256*(C+64)+256 + C*64 + C (+ 2*C with dx) floats per workgroup, at most
     ceil(nseq/16) rows; they are ADDED into dW_ih [256, C], dW_hh [256, 64], db_ih / db_hh [256] (and dW_lin / db_lin
     when non-NULL) by reductions the call launches itself. */
  const void* u; const void* hs; const float* w_ih; int C;
  float* du; float* wpart;
  float* dW_ih; float* dW_hh; float* db_ih; float* db_hh;
  float* dW_lin; float* db_lin;
  /* ... and (C == 16) the LayerNorm backward that follows: when dx != NULL, du is not written; instead
     dx [P, C] = LN-backward(du; ln_x, ln_g) + dy (the residual branch), ln_x [P, C] the pre-LayerNorm input, and
     d_ln_g / d_ln_b [C] += the parameter gradients (wpart rows grow by 2*C floats). */
  const float* ln_x; const float* ln_g; float* dx; float* d_ln_g; float* d_ln_b;
  /* Bidirectional passes (ndir == 2) fuse the same way with hs [P, 128] in fp32 (it feeds the following Linear), u fp16,
     du [P, 2, C] per direction, no Linear / LayerNorm riders; direction 1 takes w_ih1 and accumulates into the *1
     targets.  wpart: 2 * min(ceil(nseq/16), CUs/2) rows of 256*(C+64)+256 floats (persistent workgroups). */
  const float* w_ih1; float* dW_ih1; float* dW_hh1; float* db_ih1; float* db_hh1;
  int hs_f16;   /* bidirectional fused form only: hs is the fp16 [P, 128] tensor written by sb_lstm_fwd in its partial-Linear
                   mode (lin_w != NULL with ndir == 2); C == 32 with the fused Linear backward (dy form) */
  /* ... and, with hs_f16: recompute != 0 says the forward wrote NO gate records (sb_lstm_fwd with save_gates == NULL,
     save_c != NULL, aux_f16): save_gates is ignored and the four gates of every step are recomputed from u, the fp16
     hs and the forward weights w_ih / w_ih1, w_hh[2], b_ih[2], b_hh[2] (single fp16 terms on the matrix pipe). */
  int recompute; const float* b_ih[2]; const float* b_hh[2];
  /* producer side of sb_lstm_bwd_inter_overlapped (set by that call; leave NULL / 0 otherwise) */
  int* slab_flags; int slab_len; int* slab_started;
  /* WIDE BPTT state: the fused forms above (wpart != NULL; single direction or bidirectional) at the reference's own
     precision.  wide != 0: save_gates / save_c are the blocked fp32 records of sb_lstm_fwd_args.rec_f32, u and hs are the
     forward call's pair-form side outputs (opaque, 4 bytes per element: see rec_f32; hs plain fp32 [P, ndir * 64] when the
     forward did not apply the Linear); gmax is still required (the recurrence runs on gradients scaled by the power of two S so
     that 16-bit terms cannot overflow) and every gradient quantity that meets the fp16 matrix pipe does so as TWO fp16
     terms x = hi + 2^-11 lo' (lo' = fp16((x - hi) * 2^11): 22 mantissa bits with no underflow of the low term), against
     fp16 hi + lo splits of u / h_prev / the weights -- three products per MAC, as in the forward kernels.  hs_f16 and
     recompute must be 0. */
  int wide;
  /* Fused forms (wpart != NULL; single direction or bidirectional, compact or wide): split != 0 launches ROLE-SPLIT workgroups of 8 waves
     (two per SIMD): four run the recurrence, four the chunk arithmetic (dW, du, Linear weight gradient) of the previous pair
     of steps from the dgates tiles in LDS.  Same arithmetic, other summation order.  The wide bidirectional form with the
     fused Linear backward (C = C_lin = 32) does not read hs in this mode (hs may be NULL, and the forward pass need not
     store it -- sb_lstm_fwd_args.hs == NULL with the Linear fused): its recurrence waves recompute h of a step from the
     records, bit for bit as the forward kernel formed it, and hand it to the chunk waves through LDS. */
  int split;
  /* WIDE gate recomputation (wide != 0, recompute != 0; the overlapped inter-frame pair sb_lstm_bwd_inter_overlapped /
     _pair_serial only, single direction, C = C_lin = 32): the forward stored no gate records (sb_lstm_fwd_args.rec_f32 with
     save_gates == NULL) -- save_gates is ignored and the gates of every step are recomputed from the forward call's u / hs
     pair tensors (u, hs here), w_ih, w_hh[0], b_ih[0], b_hh[0] with the forward kernel's own three-product arithmetic, bit
     for bit.  h0 [nseq, 64] (nullable = zeros): the initial hidden state the forward call was given (h_prev of step 0). */
  const float* h0;
  /* backward overlapped ACROSS the two passes of a block (set by sb_lstm_bwd_cross_produce / _consume below; leave NULL / 0
     otherwise).  Consumer side: the item order / slab tables, the per-direction item counters, the guard flag, the number
     of producer workgroups, the first partial row of this launch and its workgroups per direction; pro_*: the prologue's
     operands -- dy1 = LN-backward(pro_du; pro_x, pro_ln_g) + pro_res -> pro_dy (which `dy` must point to as well). */
  const int* tile_order; const int* tile_need; int* ord_counter; int ord_guard, slab_need, row_base, ord_grid;
  const float* pro_du; const float* pro_x; const float* pro_res; const float* pro_ln_g; float* pro_dy;
} sb_lstm_bwd_args;
int sb_lstm_bwd_rec(const sb_lstm_bwd_args* a, void* stream);

/* ---- generic-shape recurrence (round 6) -------------------------------------------------------------------------
 * The reference constructors' OWN defaults -- Net(D = 64, H = 128, ...): src/models/tfgridnet_realtime_clean_dis_embd3/net.py:21-26,
 * src/models/tfgridnet_realtime_clean_optim/net.py:21-26 -- and any other width the tuned kernels above (C in {16, 32}, H = 64) are
 * not built for.  Same operation as sb_lstm_fwd / sb_lstm_bwd_rec with mma == 0 (LayerNorm(C) + nn.LSTM forward and its BPTT:
 * tfgridnet_causal.py:804-808,819-823,832-843), same position addressing (n_inner / p_outer / p_inner / p_step), for
 * C in {16, 32, 64} and H in {64, 128} (sb_lstm_gen_supported): H / 16 waves per 16-sequence tile, exact fp32 products
 * (v_mfma_f32_16x16x4_f32), W_hh in registers, W_ih in LDS.
 *   hs [P, ndir * H];  save_gates (nullable) [P, ndir, 5, H] fp32 = i, f, g, o (post-activation), c_prev;  save_u [P, C]
 *   (LayerNorm output; required with save_gates);  h0 / c0 (nullable = zeros), hN / cN (nullable): [nseq, H], direction 0.
 * sb_lstm_gen_bwd_rec: reads save_gates and dhs [P, ndir * H] (gradient w.r.t. hs), writes dgates [P, ndir, 4 H] (gradient
 *   w.r.t. the pre-activation gates, rows i, f, g, o).  The input / weight gradients are then position-wise GEMMs:
 *   du = dgates . W_ih (sb_linear_fwd), dW_ih / dW_hh / db (sb_wgrad, two sources: u, and hs shifted by one step with the
 *   first step of every sequence masked).  Initial-state gradients are not produced (training starts from zero state). */
typedef struct {
  int nseq, nsteps, n_inner, ndir, C, H;
  int64_t p_outer, p_inner, p_step;
  const float* x;
  const float* ln_g; const float* ln_b;
  const float* w_ih[2]; const float* w_hh[2]; const float* b_ih[2]; const float* b_hh[2];
  const float* h0; const float* c0; float* hN; float* cN;
  float* hs; float* save_gates; float* save_u;
} sb_lstm_gen_fwd_args;
int sb_lstm_gen_fwd(const sb_lstm_gen_fwd_args* a, void* stream);
typedef struct {
  int nseq, nsteps, n_inner, ndir, H;
  int64_t p_outer, p_inner, p_step;
  const float* w_hh[2];
  const float* save_gates; const float* dhs; float* dgates;
  const float* gmax;          /* device scalar: max |dhs| (sb_absmax), the fp16 scale of the dgates operand; NULL: scale 1 */
} sb_lstm_gen_bwd_args;
int sb_lstm_gen_bwd_rec(const sb_lstm_gen_bwd_args* a, void* stream);
int sb_lstm_gen_supported(int C, int H);      /* 1 when the two calls above are built for (C, H) */

/* ---- inter-frame backward of a block OVERLAPPED with the intra-frame backward of the same block (round 4) -----------
 * The wide (sb_lstm_bwd_args.wide) inter-frame backward of the BASELINE big configuration has 145 serial chains on 256 CUs.
 * As ONE fused role-split launch (wpart != NULL, split != 0: dgates never leave the chip) it needs half the CU time of the
 * recurrence + stream-kernel pair of sb_lstm_bwd_inter_overlapped, but leaves 111 CUs idle -- and the kernel that follows it
 * in the backward pass, the bidirectional intra-frame backward of the same block, needs for a tile of 16 frames (b, t .. t +
 * 15) only the inter-frame result of those frames.  Two calls, the mirror image of sb_lstm_fwd_produce / _consume:
 *   sb_lstm_bwd_cross_produce(a, flags, n_flags, slab_len, stream): sb_lstm_bwd_rec of the single-direction fused form (wide, split,
 *     C = C_lin = 32, du written, no LayerNorm rider, fewer tiles than CUs - 16) on `stream`; its du rows are stored
 *     write-through and after every slab_len steps (even), latest steps first, each tile t publishes its progress,
 *     flags[4 + t] = slabs completed (a plain write-through store per tile, not a contended counter per slab).
 *     flags: n_flags = 4 + ceil(nseq / 16) + 16 + 3 * (the consumer's tiles) ints, zeroed by the call ([0] producer workgroups
 *     started, [1 .. 3] spare, the producer tiles' progress words, the consumer's 16 item counters -- one queue per XCD and
 *     direction -- and three words per consumer tile: prologue claimed / done / max |dy1|).
 *   sb_lstm_bwd_cross_consume(a, flags, slab_len, producer_tiles, order, need, stream): sb_lstm_bwd_rec of the bidirectional
 *     fused form (wide, split, C = C_lin = 32, hs == NULL: h recomputed from the records) whose incoming gradient does not
 *     exist yet: the block's inter-frame LayerNorm backward + residual runs per TILE of 16 nsteps positions, pro_dy =
 *     LN-backward(pro_du; pro_x, pro_ln_g) + pro_res (pro_du = the producer's du, a->dy must equal pro_dy), ONCE per tile and XCD
 *     -- by whichever item claims the tile first (its own, or one looking ahead down its queue); the other direction's item,
 *     drawn from the same XCD's queue, waits for the tile's `done` word and reads the rows through the shared L2 -- and every
 *     item derives its fp16 scale per tile (gmax is not read; pass any valid scalar), and d_ln_g / d_ln_b [32] receive
 *     that LayerNorm's parameter gradients.  Items are taken in the order order[ntiles] (need[i] packs, for tile order[i], the
 *     producer slab that completes its frames -- bits 0..11 -- and the range lo..hi of producer tiles that hold the sequences of
 *     its batch entries -- bits 12..21, 22..31) from eight queues per direction (item i in queue i mod 8; a workgroup draws from
 *     the queue of the XCD it runs on and steals from the others when that is empty) by TWO launches: persistent workgroups on the
 *     library's side stream (one per CU the producer leaves idle, guarded as in sb_lstm_fwd_consume) and one per CU on
 *     `stream` behind the producer.  wpart: sb_lstm_bwd_cross_rows(a->nseq, producer_tiles) rows of
 *     256 * (32 + 64) + 256 + 32 * 128 + 32 + 64 floats.  Geometry: p_step == 1, p_inner == nsteps (rows of a tile contiguous).
 * The consume call must be the next library call after its produce call on that device; memory the producer touches must stay
 * allocated until it has returned.  -1003 bad geometry / arguments, -1009 without a concurrent side stream. */
int sb_lstm_bwd_cross_rows(int nseq, int producer_tiles);
int sb_lstm_bwd_cross_produce(const sb_lstm_bwd_args* a, int* flags, int n_flags, int slab_len, void* stream);
int sb_lstm_bwd_cross_consume(const sb_lstm_bwd_args* a, int* flags, int slab_len, int producer_tiles, const int* order,
                              const int* need, void* stream);
/* The same two calls with the small launches around them taken off the critical path between two blocks' backward kernels:
 *   flags_zeroed != 0: the caller has zeroed flags[0 .. n_flags) itself, in stream order before the call (one fill for all the
 *     blocks of a step instead of a memset in front of every producer);
 *   reduce_on_side != 0: the partial-row reductions into dW_ih .. d_ln_b run on the library's side stream behind both consumer
 *     launches instead of on `stream`.  Nothing but the optimiser reads their results: the caller keeps wpart allocated and
 *     calls sb_overlap_join(stream) before anything on `stream` reads (or another stream's work adds into) a gradient target. */
int sb_lstm_bwd_cross_produce_ex(const sb_lstm_bwd_args* a, int* flags, int n_flags, int slab_len, int flags_zeroed, void* stream);
int sb_lstm_bwd_cross_consume_ex(const sb_lstm_bwd_args* a, int* flags, int slab_len, int producer_tiles, const int* order,
                                 const int* need, int reduce_on_side, void* stream);

/* ---- position-wise linear (MFMA, weights staged in LDS) -------------------
 * out[p, n] = epi( sum_k in(p, k) * W[n, k] + bias[n] )  for every position
 * p = (b, t, f) of a B x T x F grid.  in(p,k) = in[b*is_b + t*is_t + f*is_f +
 * (k / kseg) * is_seg + k % kseg]  (kseg | 16; overlapping rows are allowed,
 * which is how the STFT frames and the 3x3 convolutions are expressed).
 * out row at b*os_b + t*os_t + f*os_f, features contiguous.  N, K multiples
 * of 16; N <= 128 per call, or any multiple of 16 with the plain / residual
 * epilogues (served as column slices, one workgroup row per slice -- also what a
 * call with <= 256 positions gets, so that the streaming chunk step's one-frame
 * GEMMs read their weights through many CUs).  n_valid <= N features are stored.
 * Replaces nn.Linear / Conv1d(k=s) / ConvTranspose1d(k=s) / Conv2d(3x3) /
 * asteroid Encoder+Decoder GEMMs: tfgridnet_causal.py:475,507,520,537,803-824,845. */
/* A logical [N, K] weight matrix laid over a parameter tensor in its NATIVE (torch) layout, so that no transposed /
 * permuted / zero-padded copy of a weight is ever made on the host side of a training step:
 *   element (n, k) = base[off + (n % nmod) * sn_lo + (n / nmod) * sn_hi + (k % kmod) * sk_lo + (k / kmod) * sk_hi]
 * and elements with (k % kmod) >= kvalid or n >= nvalid are zero when read (sb_linear_fwd) / dropped when written
 * (sb_wgrad).  nmod == 0 means "no view": a dense row-major [N, K] matrix.  Examples (C channels):
 *   Conv1d weight [co, ci, j] as W[co][j*C + ci]              : nmod 1<<30, sn_lo C*s, kmod C, sk_lo s, sk_hi 1
 *   ConvTranspose1d weight [h, c, j] as W[j*C + c][h]         : nmod C, sn_lo s, sn_hi 1, kmod 1<<30, sk_lo C*s
 *   Conv2d weight [co, 27, 3, 3] as W[co][(a*3+d)*32 + ci]    : sn_lo 243, kmod 32, sk_lo 9, sk_hi 1, kvalid 27
 *   ConvTranspose2d weight [c, 2, 3, 3] as W[o][(a*3+d)*C + c] (flipped taps): off 8, sn_lo 9, kmod C, sk_lo 18, sk_hi -1,
 *                                                                nvalid 2
 *   a transposed Linear weight [K, N] as W[n][k]              : sn_lo 1, sk_lo N                                     */
typedef struct {
  int64_t off;
  int nmod; int64_t sn_lo, sn_hi;
  int kmod; int64_t sk_lo, sk_hi;
  int kvalid, nvalid;
} sb_wview;
/* Kernel-layout weight forms.  The GEMM kernels stage dense row-major [N, K] weights with 16-byte loads; the forms of
 * ALL layers of a model (transposes, tap-major conv weights, zero-padded rows, biases repeated over taps) are refreshed
 * from the parameters by ONE launch per forward pass: jobs[i] (a DEVICE array, built once by the caller) copies the
 * logical matrix of view v over src into dst [N, K].  max_elems = max N*K over the jobs.  (Staging through the view inside
 * every GEMM launch was measured slower: 10-20 us of index arithmetic per launch, four waves per SIMD each.) */
typedef struct { const float* src; float* dst; sb_wview v; int N, K; } sb_wview_job;
int sb_wview_gather(const sb_wview_job* jobs, int njobs, int max_elems, void* stream);

enum {
  SB_EPI_NONE = 0,   /* + bias                                              */
  SB_EPI_RES = 1,    /* + bias + res[p]  (res addressed like out: rs_*)     */
  SB_EPI_PRELU = 2,  /* prelu(+bias) with scalar slope *prelu_a; aux_out (nullable) gets the pre-activation */
  SB_EPI_LN = 3,     /* LayerNorm over the N features of (+bias); aux_out (nullable) gets the pre-LN value */
  SB_EPI_LNBWD = 4   /* acc = d(LN out); x = aux_in[p] (pre-LN input, optional PReLU of it when prelu_a);
                        out = LN-backward (+ res[p]) (* prelu' when prelu_a); dgamma/dbeta/dprelu -> partials */
};
typedef struct {
  int B, T, F, N, K, n_valid, kseg, epi;
  const float* in; int64_t is_b, is_t, is_f, is_seg;
  const float* w; const float* bias;
  float* out; int64_t os_b, os_t, os_f;
  const float* res; int64_t rs_b, rs_t, rs_f;
  const float* prelu_a; const float* ln_g; const float* ln_b;
  const float* aux_in; float* aux_out;      /* dense [P, N] */
  float* partials;                          /* SB_EPI_LNBWD: [grid, 2N+1] floats */
  int accumulate;                           /* out += instead of out = (EPI_NONE/RES only) */
  float* absmax_out;                        /* optional (EPI_NONE/RES): *absmax_out = max(*absmax_out, max |out|), the
                                               scalar sb_lstm_bwd_rec wants as gmax; zero it before the call */
  int mma;   /* 0: fp32-input MFMA (exact fma chain); 1: fp16 matrix pipe with hi+lo split operands, 3 products per MAC
                (fp32-class, dropped term <= 2^-22) -- for the long-K narrow layers: N <= 32, K = 288 or 144, EPI_NONE /
                EPI_RES / EPI_LN (the 3x3 front-end convolution, the output transposed convolution) */
} sb_linear_args;
int sb_linear_fwd(const sb_linear_args* a, void* stream);
/* number of workgroups sb_linear_fwd launches for P positions (size of `partials`) */
int sb_linear_grid(int64_t positions);

/* ---- weight gradient (TN GEMM over positions) ----------------------------
 * dW[n, k] += sum_p g(p, n) * in(p, k)            k <  K   (source 1, addressed like sb_linear_args)
 * dW2[n, k] += sum_p g(p, n) * in2[p*ld2 + shift2 + k]  k < K2 (source 2, dense rows; optional)
 * dbias[n] (+ dbias2[n]) += sum_p g(p, n)                      (optional)
 * g: dense-strided rows [P, ldg].  Source-2 rows whose index within a segment of seg_len positions is
 * < skip_first or >= seg_len - skip_last are excluded (h_{t-1} of the first step of a sequence).
 * One pass over g serves both sources and the bias (dW_ih, dW_hh, db_ih, db_hh of an LSTM in one go).
 * If transpose_out, dW is stored [K, N].  Two-stage: per-workgroup partials in `scratch`
 * ([4*sb_wgrad_grid(P), N*(K+K2)+N] floats: one row per wave, fully written by the kernel), then a reduction
 * that ADDS into the outputs. */
typedef struct {
  int B, T, F, N, K, kseg;
  const float* g; int64_t ldg;
  const float* in; int64_t is_b, is_t, is_f, is_seg;
  const float* in2; int64_t ld2, shift2; int K2;
  int seg_len, skip_first, skip_last;
  int transpose_out;
  float* dW; float* dW2; float* dbias; float* dbias2; float* scratch;
  int in_f16;   /* 1: `in` points at fp16 elements (strides in elements), e.g. the fp16 hs of sb_lstm_fwd (aux_f16);
                   supported for K == 64, K2 == 0, N <= 32 */
  /* output index remapping of the reduction, so that gradients land in a parameter's native layout (no transposed
     copies on the host): with perm_k = c > 0 column k = j*c + i is written as i*(K/c) + j; with perm_n = c > 0 row
     n = j*c + i as i*(N/c) + j (Conv1d / ConvTranspose1d weights [out, in, k] addressed as [out][k*in + i]);
     bias_mod = c > 0 folds the bias gradient: dbias[n % c] += ... */
  int perm_k, perm_n, bias_mod;
  /* general form of the above (wv.nmod != 0; perm_k / perm_n / transpose_out must then be 0): dW[n, k] is ADDED at
     dW[wv-address(n, k)] of the parameter's gradient in its native layout (see sb_wview), and dbias[n % bias_mod] only
     for n < wv.nvalid. */
  sb_wview wv;
  const float* gmax;
  int mma;   /* 0: fp32-input MFMA; 1: fp16 matrix pipe, both operands fp16 hi + lo, 3 products per MAC (fp32-class) --
                gmax (nullable; a device scalar max |g|, e.g. from sb_absmax) makes the kernel scale g by the power of two
                2^-ceil(log2 *gmax) before the fp16 split and the sums back afterwards: gradients of 1e-7 would otherwise
                underflow fp16.  Single fp32 source, N <= 32 padded to whole 16-row tiles in g (ldg >= 16 * ceil(N/16)), K = 3 segments of
                96 or 48 columns (the 3x3 convolutions);
                2: a hint for the GENERIC tiled form only (the shapes no register-accumulator kernel covers: any N, K, K2 multiples
                of 16, fp32 sources): the same fp16 hi + lo arithmetic with the same gmax scaling; shapes that have a tuned
                kernel ignore it */
} sb_wgrad_args;
int sb_wgrad(const sb_wgrad_args* a, void* stream);
int sb_wgrad_grid(int64_t positions);
/* partial rows sb_wgrad will write for these arguments (the row count `scratch` must hold, rows of N*(K+K2)+N floats): 4 *
   sb_wgrad_grid(P) for the shapes with a register-accumulator kernel; for every other shape (any N, any K, K2 % 16 == 0,
   fp32 sources, mma == 0) sb_wgrad runs a generic tiled form -- 64 x 64 tiles of dW per workgroup over ranges of positions,
   one partial row per range -- and this returns its (much smaller) row count.  Negative: the error sb_wgrad would return. */
int sb_wgrad_scratch_rows(const sb_wgrad_args* a);

/* column sums: out[n] += sum_p g[p*ldg + n], n < N (bias gradients) */
int sb_colsum(const float* g, int64_t P, int64_t ldg, int N, float* out, float* scratch, void* stream);
/* out[i] += sum_r partials[r*ld + i] for i < n */
int sb_reduce_rows(const float* partials, int rows, int64_t ld, int n, float* out, void* stream);


/* ---- LSTM backward, streaming part ---------------------------------------
 * One pass over dgates [P, ndir, 4, 64] (written by sb_lstm_bwd_rec) per direction:
 *   dW_ih[d] [256, C] += dgates_d^T u          (u [P, C]: saved LayerNorm output)
 *   dW_hh[d] [256, 64] += dgates_d^T h_prev    (h_prev(p) = hs[p -/+ shift_pos], direction 0 / 1; positions whose
 *                                               index in a segment of seg_len is < skip (dir 0) / >= seg_len-skip
 *                                               (dir 1) are the first step of a sequence and are excluded)
 *   db_ih[d], db_hh[d] [256] += column sums of dgates_d
 *   du_part [P, ndir, C]   = dgates_d W_ih[d]  (gradient w.r.t. the LayerNorm output, per direction)
 * scratch: [ndir * sb_lstm_stream_grid(P) * (256*(C+64) + 256)] floats.  Autograd of nn.LSTM + LayerNorm,
 * tfgridnet_causal.py:819-823,832-840. */
typedef struct {
  int64_t P; int ndir, C;
  int64_t shift_pos; int seg_len, skip;
  const float* dgates; const float* u; const float* hs;
  const float* w_ih[2];
  float* dW_ih[2]; float* dW_hh[2]; float* db_ih[2]; float* db_hh[2];
  float* du_part; float* scratch;
  int split_bf16;     /* 1: bf16 matrix pipe with 3-term split products (fp32-class accuracy, ~4x fewer MFMA cycles) */
  const float* gmax;  /* != NULL: dgates is the scaled fp16 tensor written by sb_lstm_bwd_rec with the same gmax; the
                         fp16 matrix pipe is used (dgates exact, the fp32 operands as fp16 hi + lo) and every output
                         is multiplied by 1/S */
  int u_f16, hs_f16;  /* gmax != NULL only: u / hs are the fp16 tensors written by sb_lstm_fwd with aux_f16 */
  /* optional fusion of the LayerNorm backward that follows (single direction, gmax != NULL, u_f16 and hs_f16): when
     dx != NULL, du_part is not written; dx [P, C] = LN-backward(du; ln_x, ln_g) + ln_res (ln_x the pre-LayerNorm
     input, ln_res the residual branch's gradient), d_ln_g / d_ln_b [C] += the parameter gradients (the scratch rows
     grow by 2C floats), *absmax_out = max(*absmax_out, max |dx|) when non-NULL (zero it before the call). */
  const float* ln_x; const float* ln_g; const float* ln_res; float* dx; float* d_ln_g; float* d_ln_b; float* absmax_out;
  /* ... and of the weight gradient of the Linear in front of that residual (tfgridnet_causal.py:844-845), whose output
     gradient is ln_res: when d_lin_w != NULL (dx != NULL required), d_lin_w [C, 64] += ln_res^T hs and d_lin_b [C] +=
     column sums of ln_res (the scratch rows grow by another 64C + C floats; gmax must be max |ln_res| or above). */
  float* d_lin_w; float* d_lin_b;
  /* consumer side of sb_lstm_bwd_inter_overlapped (set by that call; leave NULL / 0 otherwise) */
  int* slab_flags; int slab_len, slab_need; int* chunk_counter; int* started; int nchunks, guard, row_base;
  int* sched_status;
  /* WIDE form (sb_lstm_bwd_inter_overlapped only; rec->wide must be set as well): dgates rows are [hi x 256 | lo' x 256]
     halves, x = hi + 2^-11 lo' (1 KB per position), u [P, 2C] and hs [P, 128] halves are the fp16 (hi, lo) pair tensors of
     sb_lstm_fwd_args.rec_f32 (u_f16 = hs_f16 = 1): three products per MAC, as in the fused wide kernels. */
  int wide;
} sb_lstm_stream_args;
int sb_lstm_bwd_stream(const sb_lstm_stream_args* a, void* stream);
int sb_lstm_stream_grid(int64_t positions);

/* ---- inter-frame LSTM backward, recurrence and streaming part OVERLAPPED ---------------------------------------
 * The inter-frame recurrence has nseq/16 serial chains (145 at the BASELINE big configuration) -- fewer than the chip has
 * CUs -- and the streaming part that follows is throughput work.  This call runs both at once: `rec` (dy form, compact
 * fp16 dgates, single direction) on `stream`, publishing the dgates per slab of `slab_len` time steps, and the streaming
 * part (fused LayerNorm backward + Linear weight gradient form: st->dx and st->d_lin_w set) as TWO launches whose
 * workgroups draw units of 16 consecutive 32-position chunks from one atomic counter in production order: one on a side
 * stream of the library with a workgroup per CU the recurrence leaves idle (guarded: a workgroup that does not see every
 * recurrence workgroup started within ~50 us draws nothing), each unit waiting (bounded) for its slab, and one on
 * `stream` behind the recurrence with a workgroup per CU, taking what is left.  st->sched_status: watchdog word as in
 * sb_lstm_fwd_args, required.
 *   flags [4 + ceil(nsteps / slab_len)] ints (zeroed by the call); slab_len even, >= 2
 *   st->scratch: sb_lstm_overlap_rows(P, nseq) partial rows of 256*(C+64) + 256 + 2C + 64C + C floats
 * returns -1003 when the geometry leaves fewer than 16 idle CUs, -1009 without a concurrent side stream
 * (sb_overlap_available): use the two plain calls. */
int sb_lstm_bwd_inter_overlapped(const sb_lstm_bwd_args* rec, const sb_lstm_stream_args* st, int* flags, int slab_len,
                                 void* stream);
/* Measurement aid: the same two kernels in plain order on `stream` (the recurrence, then one stream-kernel launch that
 * finds every slab flag up), no side stream.  Slower than the fused single launch -- it exists so that profilers that
 * serialise kernels (rocprofv3 --pmc) can count the HBM traffic of the overlapped pair: the sum of these two launches. */
int sb_lstm_bwd_inter_pair_serial(const sb_lstm_bwd_args* rec, const sb_lstm_stream_args* st, int* flags, int slab_len,
                                  void* stream);
int sb_lstm_overlap_rows(int64_t positions, int nseq);
/* The overlapped calls need a side stream whose kernels really run at the same time as those of `stream` (the runtime
 * multiplexes streams over a few hardware queues; two streams on one queue serialise).  These four calls are the ONLY
 * ones in the library that create / destroy HIP objects or synchronise; the data-path entry points above only look the
 * side stream up (-1009 when there is none: use the plain calls).
 *   sb_overlap_init(stream, scratch, timings): finds a side stream for (current device, stream) with a TIMED probe -- a
 *     0.2 ms pair of one-per-CU arithmetic kernels with the fork / join choreography of the real calls must finish in
 *     < 0.7 of the back-to-back time; up to 8 candidate streams -- and keeps it, with its own fork / join events, until
 *     sb_overlap_shutdown.  scratch: 4 floats of device memory owned by the caller (the probe kernels' sink); timings
 *     (nullable, HOST pointer to 2 floats): back-to-back and best pair time in ms.  Synchronises `stream` several times;
 *     not to be called while it is capturing.  Returns 1 (found) / 0 (none: overlapped calls unavailable on this stream);
 *     a second call for the same (device, stream) returns the stored verdict.
 *   sb_overlap_reprobe: re-times the stored pair and updates the verdict -- a stream that was concurrent at start-up can
 *     lose that (another process on the GPU, more streams alive); call it now and then (the harness does once per epoch)
 *     and fall back to the plain calls when it returns 0.  A failing measurement is repeated (up to 4, best pair counts:
 *     a 0.1 ms host hiccup between the two launches of one measurement otherwise reads as a loss); a later call may find
 *     the pair concurrent again and returns 1 then.
 *   sb_overlap_available: the stored verdict, no side effects.
 *   sb_overlap_shutdown: destroys every side stream and event of the process.
 * The table behind them is mutex-guarded; everything else in the library is stateless.
 * Small launches whose results nothing on the critical path reads (partial-row reductions into gradient targets) can ride on
 * the same side stream (data path: event record / wait only, no allocation, no synchronisation):
 *   sb_overlap_side_fork(stream, &side): the side stream of `stream` waits for everything enqueued on `stream` so far; *side
 *     receives its handle, to be passed as the `stream` argument of e.g. sb_reduce_rows.  -1009 without a side stream.
 *   sb_overlap_join(stream): `stream` waits for everything enqueued on its side stream so far (0 and no effect without one).
 *     Memory such launches touch must stay allocated until the join has been enqueued. */
int sb_overlap_init(void* stream, float* scratch, float* timings_ms);
int sb_overlap_reprobe(void* stream, float* scratch, float* timings_ms);
int sb_overlap_available(void* stream);
int sb_overlap_shutdown(void);
int sb_overlap_side_fork(void* stream, void** side);
int sb_overlap_join(void* stream);
/* Measurement aid: arms a ONE-SHOT timer -- the next kernel an overlapped entry point (sb_lstm_fwd_consume,
 * sb_lstm_bwd_inter_pair, sb_lstm_bwd_cross_consume[_ex]) places on a library side stream has `ev_start` recorded straight in
 * front of it and `ev_stop` straight behind it ON THAT SIDE STREAM (caller-owned hipEvent_t with timing enabled; the caller reads
 * hipEventElapsedTime after synchronising).  nullptrs disarm.  Returns 1 when a timer armed earlier was still waiting (it never
 * fired: no side launch happened), else 0.  Never needed for results: bench.py's live roofline uses it to
 * time launches its own stream never sees. */
int sb_overlap_time_next_side_launch(void* ev_start, void* ev_stop);
/* Measurement aid: after sb_overlap_init(stream, ...), take the overlapped code paths on `stream` whatever the timed probe said
 * (a side stream is created if the probe left none) -- for counter collection under a profiler that serialises dispatches
 * (rocprofv3 --pmc): the shipped producer / consumer kernels then run one after the other, which their protocols tolerate
 * (producers are enqueued first; consumer waits are bounded).  1 on success, 0 without a probed entry.  Not a product path. */
int sb_overlap_force(void* stream);

/* ---- LayerNorm (+PReLU) backward over C channels ---------------------------
 * g = sum_d du_part[p, d, :];  x = xin[p] (PReLU(xin[p]) with slope *prelu_a when prelu_a != NULL);
 * out[p] = LN_backward(g; x, ln_g) (* PReLU'(xin) when prelu_a) (+ res[p] when res != NULL).
 * partials [sb_ln_bwd_grid(P), 2C+1]: per-workgroup sums of d(ln_g), d(ln_b), d(prelu_a) (reduce with
 * sb_reduce_rows).  All tensors dense [P, C].  nn.LayerNorm / nn.PReLU backward of tfgridnet_causal.py:803-804,819,832. */
typedef struct {
  int64_t P; int ndir, C;
  const float* du_part; const float* xin; const float* ln_g; const float* prelu_a; const float* res;
  float* out; float* partials;
  float* absmax_out;     /* optional: *absmax_out = max(*absmax_out, max |out|) (see sb_linear_args.absmax_out) */
} sb_ln_bwd_args;
int sb_ln_bwd(const sb_ln_bwd_args* a, void* stream);
int sb_ln_bwd_grid(int64_t positions);

/* ---- full-band local self-attention ----------------------------------------
 * tfgridnet_causal.py:639-684,856-898 (off in every shipped config).
 * sb_head_ln: per (b, t) LayerNorm over the (f, d) elements of each head of in [B, T, F, Hh*D] (eps 1e-5, gamma/beta
 *   [F*D] shared by the heads), written head-major into out[(b*Hh + h) * rows + t_off + t][0..ldo) (columns beyond
 *   F*D zeroed).  With Hh == 1 and res != NULL: out = res[b,t,:] + LN(in[b,t,:])  (the attn_concat_proj LayerNorm).
 * sb_attn_core: Q [BH, T, ldk], K [BH, L-1+T, ldk], V [BH, L-1+T, ldv] (rows zero-padded to ldk / ldv, multiples of
 *   16) -> out [B, T, F, Hh*Cv]: softmax_l(q_t . k_{t+l} * scale) over l in [0, L) of the concatenated rows, times V;
 *   both contractions on the fp32-input MFMA.  NRp = 16*ceil((L+15)/16). */
int sb_head_ln(const float* in, const float* gamma, const float* beta, float* out, const float* res, int B, int T, int F,
               int Hh, int D, int rows, int t_off, int ldo, int ldi /* row stride of in, >= Hh*D */,
               const float* prelu_a /* nullable: PReLU(in) is what gets normalised */, void* stream);
typedef struct {
  int BH, Hh, T, F, Cv, L, NRp, ldk, ldv;
  float scale;
  const float* Q; const float* K; const float* V; float* out;
  float* lse;                 /* nullable: [BH, T] log-sum-exp of each query's scaled scores (kept for the backward) */
} sb_attn_args;
int sb_attn_core(const sb_attn_args* a, void* stream);

/* ---- attention backward (training with use_attn=True) ----------------------
 * sb_head_ln_bwd: backward of sb_head_ln incl. the PReLU applied on load.  dout is in the forward's head-major output
 *   layout; din [B,T,F,ldi] (gradient w.r.t. the pre-activation); partials [sb_head_ln_bwd_grid(B,T), 2n+1] with
 *   n = F*Hh*D: per-workgroup [dgamma_i][dbeta_i][dalpha] indexed by the flat (f, h, d) element -- reduce the rows with
 *   sb_reduce_rows and fold the heads on the host.  The residual branch (res) passes dout through unchanged.
 * sb_attn_core_bwd: dO [BH, T, ldv] head-major (zero-padded like V) -> dQ [BH, T, ldk], dK [BH, T, ldk], dV [BH, T, ldv]
 *   (current-frame rows only; the carried K/V buffer rows are not differentiated), delta [BH, T] scratch.  Two launches
 *   (per-query tile, per-key tile), probabilities recomputed from lse; no atomics on dQ/dK/dV. */
int sb_head_ln_bwd_grid(int B, int T);
int sb_head_ln_bwd(const float* in, const float* gamma, const float* dout, float* din, float* partials, int B, int T,
                   int F, int Hh, int D, int rows, int t_off, int ldo, int ldi, const float* prelu_a, void* stream);
typedef struct {
  int BH, Hh, T, F, Cv, L, NRp, ldk, ldv;
  float scale;
  const float* Q; const float* K; const float* V; const float* dO; const float* lse;
  float* delta; float* dQ; float* dK; float* dV;
} sb_attn_bwd_args;
int sb_attn_core_bwd(const sb_attn_bwd_args* a, void* stream);

/* ---- front-end features --------------------------------------------------
 * spec [B*M, T, ld_spec] (cols 0..F-1 real, F..2F-1 imag: asteroid Encoder
 * layout) -> zp [B, T+2, F+2, 32] channels-last, written at time offset 2 and
 * frequency offset 1 (zero borders for the 3x3 conv; channels 5M-3 .. 31 zero: 27.. for the shipped M = 6).
 * Channel order: re x M, im x M, ILD x (M-1), (sin, cos) x (M-1).  2 <= M <= 7 (-1002 otherwise).
 * Replaces tfgridnet_causal.py:482-500 + MC_features_OMNX :72-93, IPD_OMNX :32-48. */
int sb_features(const float* spec, int64_t ld_spec, float* zp, int B, int M, int T, int F, void* stream);

/* ---- FiLM (distance conditioning between blocks) --------------------------
 * y[b,t,f,c] = x[b,t,f,c] * w[b,f,c] + bias[b,f,c]   (tfgridnet_causal.py:59-68,509-513)
 * backward: dx = dy * w ; dw[b,f,c] = sum_t dy * x ; dbias[b,f,c] = sum_t dy. */
int sb_film_fwd(const float* x, const float* w, const float* bias, float* y, int B, int T, int F, int C, void* stream);
int sb_film_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* dbias,
                int B, int T, int F, int C, float* absmax_out /* optional: max |dx|, as sb_linear_args.absmax_out */,
                void* stream);

/* ---- distance embedding -> FiLM plane bank ---------------------------------------------------------------------------------
 * Dis_Embed_Conv (dis_embd3/tfgridnet_causal.py:150-173: Linear(K = 3 -> 4F, no bias) -> view [B, F, 4] -> LayerNorm(4)) and the
 * two Conv1d(4 -> C, k = 1) of every FilmLayer (:51-68), evaluated once per forward for all n = n_layers - 1 layers (:509-513):
 *   e[b, f, :] = LN_4(W_e[4f .. 4f+3, :] . dis[b, :])   (4 = d_in of dis_type "conv3"; 1 / 2 / 8 for conv1 / conv2 / conv4);   planes[j, b, f, c] = conv_b[j][c] + sum_i conv_w[j][c, i] e[b, f, i],
 *   j = 2 layer + which (0: the scale plane `weight`, 1: the shift plane `bias`); planes dense [2n, B, F, C].
 * sb_film_bank_bwd: G [2n, B, F, C] = d loss / d planes (the buffer sb_film_bwd / sb_ln_film_bwd accumulated into);
 *   d_conv_w[j] [C, d_in], d_conv_b[j] [C], d_ln_w [d_in], d_ln_b [d_in], dW_e [d_in F, K] are ACCUMULATED into (+=, fixed summation order:
 *   deterministic); partials: sb_film_bank_bwd_scratch(F, C, n, d_in) floats of caller scratch.  Two launches.
 * Limits: n <= SB_FILM_BANK_MAX_LAYERS, 2 n C <= 1024, K <= 8 (-1002 otherwise). */
#define SB_FILM_BANK_MAX_LAYERS 16
typedef struct {
  int B, F, C, n, K, d_in;   /* d_in = 1 / 2 / 4 / 8: dis_type conv1 .. conv4 (every shipped JSON: conv3 = 4) */
  const float* dis;        /* [B, K] */
  const float* W_e;        /* [d_in F, K] */
  const float* ln_w; const float* ln_b;                       /* [d_in] */
  const float* conv_w[2 * SB_FILM_BANK_MAX_LAYERS];           /* [C, d_in] each */
  const float* conv_b[2 * SB_FILM_BANK_MAX_LAYERS];           /* [C] each */
  float* planes;           /* forward out */
  const float* G;          /* backward in */
  float* dW_e; float* d_ln_w; float* d_ln_b;
  float* d_conv_w[2 * SB_FILM_BANK_MAX_LAYERS];
  float* d_conv_b[2 * SB_FILM_BANK_MAX_LAYERS];
  float* partials;
} sb_film_bank_args;
int sb_film_bank_fwd(const sb_film_bank_args* a, void* stream);
int sb_film_bank_bwd_scratch(int F, int C, int n, int d_in);
int sb_film_bank_bwd(const sb_film_bank_args* a, void* stream);

/* sb_ln_film_bwd (C = 32): the LayerNorm backward of a block's intra-frame pass (sb_ln_bwd with ndir = 2 and a residual) and the
 * FiLM backward of the block in front of it in one pass -- dx = LN-bwd(du[p, 0, :] + du[p, 1, :]; xin, ln_g) + res never leaves
 * the registers: out = dx * film_w[b, f, :];  dw[b, f, :] += sum_t dx * film_x;  dbias[b, f, :] += sum_t dx (atomics, pre-zeroed
 * or accumulated into, as sb_film_bwd);  partials: sb_ln_film_bwd_rows(B, T, F) rows of 64 floats, (sum g xhat [32], sum g [32])
 * per workgroup -- the caller reduces them (sb_reduce_rows) into the LayerNorm parameter gradients.  Replaces the pair
 * sb_ln_bwd + sb_film_bwd between two blocks (tfgridnet_causal.py:818-827 backward, :59-68 backward): 768 instead of 1 024 bytes
 * per position.  absmax_out optional: max |out|. */
int sb_ln_film_bwd_rows(int B, int T, int F);
int sb_ln_film_bwd(const float* du, const float* xin, const float* ln_g, const float* res, const float* film_x,
                   const float* film_w, float* out, float* dw, float* dbias, float* partials, int B, int T, int F, int C,
                   float* absmax_out, void* stream);

/* y[p, :] = x[p, :] + part[p, 0, :] + part[p, 1, :]  (x, y [P, C]; part [P, 2, C]): the residual + the two directions'
 * partial products of the intra-frame Linear written by sb_lstm_fwd in partial mode (tfgridnet_causal.py:824-827). */
int sb_add3(const float* x, const float* part, float* y, int64_t P, int C, void* stream);

/* ---- staging of the 3x3 convolutions' inputs and of the iSTFT's spectrum rows (round 6; rounds 1-5: ATen fills and strided copies) ----
 * sb_stage_frames: dst [B, Tp, F + 2, Cd] channels-last with zero frequency borders (columns 0 and F + 1).  Frame rows 0, 1 <- the
 *   carried context `state` [B, Cs, 2, F] (conv_buf / deconv_buf of tfgridnet_causal.py:403-421; channels Cs .. Cd - 1 zero);
 *   with src != NULL ([B, Tp - 2, F, Cd]: the last block's output, :517-520) frame rows 2 .. <- src as well (borders zeroed);
 *   with src == NULL only the two carried rows are written (the feature kernel sb_features fills the rest, :482-493).
 * sb_frames_to_state: the new carried context [B, Cs, 2, F] <- frame rows r0, r0 + 1 of such a tensor (interior, channels < Cs).
 * sb_spec_rows: spectrum rows [B, T + 1, ld] (interleaved re / im, ld >= 2 F).  mode 0: zero the padding columns 2 F .. ld - 1
 *   of every row and fill row 0 from the carried istft_buf [B, 2, F] (:533-536); mode 1: the new istft_buf <- row T. */
int sb_stage_frames(const float* state, const float* src, float* dst, int B, int Tp, int F, int Cs, int Cd, void* stream);
int sb_frames_to_state(const float* rows, float* state, int B, int Tp, int F, int Cs, int Cd, int r0, void* stream);
int sb_spec_rows(float* rows, float* buf, int B, int T, int F, int ld, int mode, void* stream);

/* out[r, f, :] = in[r, f, :] (+ bias[:] when bias != NULL) for the tail frequencies Fm <= f < F of every row r (in, out: [rows, F, C]):
 * the residual of the conv-LSTM intra path at the frequencies beyond down * floor(F / down), which the k = s = down
 * ConvTranspose1d does not reach (bias: only the `optim` flavour, whose deconvolution has output_padding -- optim/
 * tfgridnet_causal.py:706-707; dis_embd3 :811-813 crops instead), and, with bias == NULL, the same rows of its backward. */
int sb_tail_rows(const float* in, const float* bias, float* out, int64_t rows, int F, int Fm, int C, void* stream);

/* ---- iSTFT overlap-add ---------------------------------------------------
 * frames [B, T+1, 288] (row 0 = carried istft_buf frame) -> wave [B, hop*T]:
 * conv_transpose1d overlap-add, drop the first hop and the last (win-hop)
 * samples (tfgridnet_causal.py:533-542).  bwd: dframes from dwave. */
int sb_overlap_add(const float* frames, float* wave, int B, int T, int win, int hop, void* stream);
int sb_overlap_add_bwd(const float* dwave, float* dframes, int B, int T, int win, int hop, void* stream);

/* ---- several small dense copies in one launch ------------------------------
 * dst[i][0 .. n[i]) = src[i][0 .. n[i]) (floats) for i < njobs <= 16.  The streaming chunk step (edge/causal_infer.py:15-26:
 * `self.internal_state = next_state`) writes the carried state back into the static buffers its hipGraph reads: one graph
 * node instead of one copy node per state tensor.  The pointers travel as kernel arguments (no device-side table), so the
 * call can be stream-captured. */
#define SB_MULTI_COPY_MAX 16
typedef struct {
  const float* src[SB_MULTI_COPY_MAX];
  float* dst[SB_MULTI_COPY_MAX];
  int64_t n[SB_MULTI_COPY_MAX];
  int njobs;
} sb_multi_copy_args;
int sb_multi_copy(const sb_multi_copy_args* a, void* stream);

/* ---- output transposed-conv, data gradient --------------------------------
 * dy [B, T, F, C] (gradient w.r.t. the last block's output) from dspec
 * [B, T, F, 2] (interleaved re/im) and the ConvTranspose2d weight
 * w[C, 2, 3, 3] (tfgridnet_causal.py:401,520).  The forward and the weight
 * gradient of this layer are sb_linear_fwd / sb_wgrad over the padded grid.
 * absmax_out (nullable, one float zeroed by the caller): receives max |dy| (atomic max of the bit patterns, as the other
 * absmax_out arguments) -- the fp16 scale of the backward recurrence that reads dy next, without a pass of its own. */
int sb_deconv_bwd_data(const float* dspec, const float* w, float* dy, int B, int T, int F, int C, float* absmax_out, void* stream);

/* ---- SNRLP loss (src/losses/SNRLP.py:17-42, asteroid SingleSrcNegSDR('snr')) ----
 * est, gt [B, N].  stats [B, 12] scratch.  loss_vec [B].  Negative (all-zero gt)
 * samples get neg_weight * mean|est| over ALL negative samples' elements.
 * dest (nullable) = d mean_b(loss_vec) / d est.
 * sb_snrlp_loss_ex: `mode` = the positive samples' term, SNRLosses(name) of src/losses/SNRLosses.py:10-52 --
 *   0 'snr' (every shipped config), 1 'sisdr', 2 'fused' = (sisdr + snr) / 2, 3 'max_fused' = max(sisdr, snr),
 *   4 'sdsdr' = max(snr, sdsdr), 5 'full' = sisdr / 2 + max(snr, sdsdr) / 2; each term asteroid's SingleSrcNegSDR (zero-mean,
 *   EPS 1e-8), evaluated from the per-sample zero-mean moments S_tt, S_et, sum ((e - me) - (t - mt))^2.  -1002: unknown mode. */
int sb_snrlp_loss_ex(const float* est, const float* gt, int B, int64_t N, float neg_weight, int mode,
                     float* stats, float* loss_vec, float* dest, void* stream);
/* The same in two calls (round 6): _fwd fills stats and loss_vec and, when loss_mean != NULL, *loss_mean = mean_b loss_vec[b]
 * (hl_module:321's loss.mean(), no reduction launch of the caller's); _bwd forms dest = gout[0] * d mean_b(loss_vec) / d est from
 * the SAME stats (gout: device scalar, the incoming gradient of the mean; NULL = 1) -- the gradient is then computed in the
 * backward pass, scaled in the kernel, and never materialised for a forward that is not followed by one.
 * (rounds 1-5 exported sb_snrlp_loss, mode 0 with stats [B, 8]: REMOVED in round 6 -- stats grew to [B, 12] with the other modes, and
 * a caller built against the old size would have overrun its scratch silently; it now fails at load time.  Use _ex with mode 0.) */
int sb_snrlp_loss_fwd(const float* est, const float* gt, int B, int64_t N, float neg_weight, int mode,
                      float* stats, float* loss_vec, float* loss_mean, void* stream);
int sb_snrlp_loss_bwd(const float* est, const float* gt, int B, int64_t N, float neg_weight, int mode,
                      const float* stats, const float* gout, float* dest, void* stream);

/* ---- metric moments (src/metrics/metrics.py:44-55, hl_module:326-373) -------
 * One pass over est, gt [B, N] and the reference mixture channel (row b at mix + b*mix_stride):
 * out[b, 0..7] = sum e, t, m, e*e, t*t, m*m, e*t, m*t.  SNR / SI-SNR / SI-SDR (and their improvements) and the
 * decay metric follow in closed form on the host from ONE small D2H copy instead of O(batch x metrics) .item() syncs. */
int sb_signal_stats(const float* est, const float* gt, const float* mix, int B, int64_t N, int64_t mix_stride,
                    float* out, void* stream);

/* ---- optimiser ------------------------------------------------------------
 * sumsq[0] += sum g^2 (grad-norm for clip_grad_norm_, hl_module:437-441). */
/* max |x| over n (multiple of 4) floats -> out[0] (device scalar, set by the call) */
int sb_absmax(const float* x, int64_t n, float* out, void* stream);
/* ---- fine-tune loss: multi-resolution STFT magnitude L1 + waveform L1 -------------------------------------------
 * src/losses/MultiResoLoss.py:6-31 = auraloss.freq.MultiResolutionSTFTLoss(perceptual_weighting, w_lin_mag) +
 * l1_ratio * nn.L1Loss (selected by syn_experiments/finetune_stage.json:34, real_experiments/<x>_finetune.json).  The STFTs
 * themselves are sb_linear_fwd GEMMs over overlapping rows of the reflect-padded signal (frame t = samples
 * [t * hop + off, t * hop + off + K) of the padded row, K = window support); these calls are the rest of the chain.
 *   sb_fir: y[b, n] = sum_k taps[k] x[b, n + k - ntaps/2], zero padded (torch conv1d; auraloss FIRFilter "aw", 101 taps).
 *     Its input gradient is the same call with the taps reversed.  ntaps odd, <= 257.
 *   sb_reflect_pad: xp[b, i] = x[b, reflect(i - pad)], i < N + 2 pad (torch.stft center=True, pad_mode="reflect"); the
 *     rest of the row (ldp >= N + 2 pad floats) is zeroed.
 *   sb_stft_mag_l1: spectra rows [rows, ld] of interleaved (re, im) pairs, bins k < nbins.  |X| = sqrt(max(re^2 + im^2,
 *     eps)) (auraloss STFTLoss.stft); *loss (+)= loss_scale * sum | |X| - |Y| | (fixed summation tree);
 *     dspec_x (nullable) [rows, ld] = gscale * sign(|X| - |Y|) * X / |X| (zero on the clamp, zero in the padding columns).
 *     partial: sb_stft_mag_l1_grid(rows, ld) floats of scratch.
 *   sb_frames_fold: backward of framing + reflect padding: dx[b, m] (+)= sum of dframes[b, t, k] over all (t, k) whose
 *     padded sample t * hop + off + k maps onto m; dframes [B, nframes, ldk], k < K.
 *   sb_l1_grad: *loss (+)= loss_scale * sum |x - y|, dx (nullable) (+)= gscale * sign(x - y); partial: ceil(n / 256) floats. */
int sb_fir(const float* x, const float* taps, float* y, int B, int64_t N, int ntaps, void* stream);
int sb_reflect_pad(const float* x, float* xp, int B, int64_t N, int pad, int64_t ldp, void* stream);
int sb_stft_mag_l1_grid(int64_t rows, int ld);
int sb_stft_mag_l1(const float* spec_x, const float* spec_y, int64_t rows, int nbins, int ld, float eps, float gscale,
                   float* dspec_x, float* partial, float loss_scale, float* loss, int accumulate, void* stream);
/* All three auraloss STFTLoss terms of one resolution (auraloss/freq.py STFTLoss.forward; reached through the **kwargs of
 * src/losses/MultiResoLoss.py:12, whose defaults are w_sc = w_log_mag = 1):
 *   *loss += scale * ( w_lin * mean | |X| - |Y| |  +  w_log * mean | log|X| - log|Y| |  +  w_sc * || |Y| - |X| ||_F / || |Y| ||_F )
 * (means / norms over the rows x nbins magnitudes; fixed summation trees) and, when dspec_x != NULL, dspec_x [rows, ld] =
 * d(that) / d spec_x (zero on the clamp and in the padding columns).  Two passes over the spectra: the spectral-convergence
 * gradient needs the global norms.  partial: 4 * sb_stft_mag_l1_grid(rows, ld) floats of scratch; sums: 4 floats (out:
 * sum |e|, sum |log ratio|, sum e^2, sum |Y|^2). */
int sb_stft_mag_terms(const float* spec_x, const float* spec_y, int64_t rows, int nbins, int ld, float eps, float w_lin,
                      float w_log, float w_sc, float scale, float* dspec_x, float* partial, float* sums, float* loss,
                      void* stream);
/* sb_stft_f64acc: spec[(b, t), n] = sum_k xp[b * ldp + t * hop + off + k] * w[n * K + k] for b < B, t < nframes, n < N,
 * every fp32 product accumulated in double and the sum rounded once (rows of spec are dense, N floats).  The STFT the
 * log-magnitude term of auraloss's MultiResolutionSTFTLoss is evaluated from (src/losses/MultiResoLoss.py:12 with w_log_mag != 0):
 * its gradient weighs a bin by 1 / |X|^2, and the fp32 accumulation error of a K = 240 .. 1200 term dot product in the
 * near-cancelled bins would otherwise dominate it.  With perceptual weighting the signal itself is the pair of planes sb_fir_pair
 * returns (the A-weighted low bands lie below the fp32 rounding floor of the filtered signal). */
int sb_stft_f64acc(const float* xp, const float* w, float* spec, int B, int nframes, int64_t ldp, int hop, int off, int K,
                   int N, int64_t lo_off, void* stream);
/* sb_fir_pair: sb_fir with the sum formed in double and returned as two fp32 planes, y + y_lo (sb_stft_f64acc adds them back
 * up through lo_off, the element distance from xp to the low plane's padded copy; lo_off = 0: a single fp32 signal). */
int sb_fir_pair(const float* x, const float* taps, float* y, float* y_lo, int B, int64_t N, int ntaps, void* stream);
int sb_frames_fold(const float* dframes, float* dx, int B, int64_t N, int nframes, int K, int ldk, int hop, int off,
                   int pad, int accumulate, void* stream);
int sb_l1_grad(const float* x, const float* y, int64_t n, float gscale, float* dx, int accumulate, float* partial,
               float loss_scale, float* loss, int accumulate_loss, void* stream);

int sb_sumsq(const float* g, int64_t n, float* sumsq, void* stream);                       /* sumsq[0] += sum g^2 */
int sb_sumsq_ex(const float* g, int64_t n, float* sumsq, int accumulate, void* stream);      /* accumulate == 0: sumsq[0] = sum g^2 (no zero-fill in front) */
/* Adam step (torch.optim.Adam, no weight decay / amsgrad) over a flat bucket.
 * grad is first scaled by gscale * min(1, clip / (sqrt(sumsq[0]) * gscale + 1e-6))
 * when clip > 0 (clip_grad_norm_ semantics), else by gscale. */
int sb_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                 float eps, int step, float gscale, float clip, const float* sumsq, void* stream);
/* ... guarded (round 6): `guard` (nullable) is the watchdog word of the guarded schedules (sched_status of the step's launches).
 * When *guard != 0 -- a bounded wait of this step gave up, the gradients are garbage -- the update is a NO-OP: parameters and
 * moments keep their last good values without a host synchronisation, and *skipped (nullable, int) counts the skipped steps
 * for the report the caller raises at its next look at the word (tain_val.py:51-88 never takes a non-finite step silently). */
int sb_adam_step_guarded(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                         float eps, int step, float gscale, float clip, const float* sumsq, const int* guard, int* skipped,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif
