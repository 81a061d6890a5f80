"""ctypes binding of libsoundbubble_hip.so (the C ABI in include/sound_bubble_hip.h).

The product path has NO fallback: if the library is missing or a call returns a
non-zero status this module raises.  Nothing here imports `oracle/`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SB_LIB_PATH: a developer override (A/B runs against an experiment build under lib/exp/); the product default is the in-tree library
LIB_PATH = os.environ.get("SB_LIB_PATH") or os.path.join(_HERE, "lib", "libsoundbubble_hip.so")
if os.environ.get("SB_LIB_VARIANT"):      # developer A/B builds (scripts/build_variant.py): lib/exp/lib_<name>.so
    LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "exp", f"lib_{os.environ['SB_LIB_VARIANT']}.so")

c_fp = C.c_void_p          # device float*
i64 = C.c_int64


class LstmFwdArgs(C.Structure):
    _fields_ = [("nseq", C.c_int), ("nsteps", C.c_int), ("n_inner", C.c_int), ("ndir", C.c_int), ("C", C.c_int),
                ("p_outer", i64), ("p_inner", i64), ("p_step", i64),
                ("x", c_fp), ("ln_g", c_fp), ("ln_b", c_fp),
                ("w_ih", c_fp * 2), ("w_hh", c_fp * 2), ("b_ih", c_fp * 2), ("b_hh", c_fp * 2),
                ("h0", c_fp), ("c0", c_fp), ("hN", c_fp), ("cN", c_fp),
                ("hs", c_fp), ("save_gates", c_fp), ("save_u", c_fp), ("save_c", c_fp), ("mma", C.c_int),
                ("lin_w", c_fp), ("lin_b", c_fp), ("y", c_fp), ("x_part", c_fp), ("x_sum", c_fp),
                ("film_w", c_fp), ("film_b", c_fp), ("y_pre", c_fp),
                ("seg_state", c_fp), ("seg_flags", c_fp), ("seg_count", C.c_int), ("seg_len", C.c_int),
                ("sched_status", c_fp), ("sched_workers", C.c_int), ("sched_segments", C.c_int),
                ("aux_f16", C.c_int),
                ("slab_flags", C.c_void_p), ("slab_len", C.c_int), ("slab_need", C.c_int),
                ("tile_order", C.c_void_p), ("tile_need", C.c_void_p), ("ord_counter", C.c_void_p),
                ("ord_started", C.c_void_p), ("ord_guard", C.c_int), ("ord_grid", C.c_int), ("rec_f32", C.c_int),
                ("no_vec", C.c_int), ("products", C.c_int), ("ord_giveups", C.c_void_p), ("ord_ret", C.c_void_p)]


class LstmBwdArgs(C.Structure):
    _fields_ = [("nseq", C.c_int), ("nsteps", C.c_int), ("n_inner", C.c_int), ("ndir", C.c_int),
                ("p_outer", i64), ("p_inner", i64), ("p_step", i64),
                ("w_hh", c_fp * 2), ("save_gates", c_fp), ("dhs", c_fp), ("dgates", c_fp), ("save_c", c_fp), ("mma", C.c_int),
                ("dy", c_fp), ("w_lin", c_fp), ("C_lin", C.c_int), ("gmax", c_fp),
                ("seg_state", c_fp), ("seg_flags", c_fp), ("seg_count", C.c_int), ("seg_len", C.c_int),
                ("sched_status", c_fp), ("sched_workers", C.c_int), ("sched_segments", C.c_int),
                ("u", c_fp), ("hs", c_fp), ("w_ih", c_fp), ("C", C.c_int), ("du", c_fp), ("wpart", c_fp),
                ("dW_ih", c_fp), ("dW_hh", c_fp), ("db_ih", c_fp), ("db_hh", c_fp), ("dW_lin", c_fp), ("db_lin", c_fp),
                ("ln_x", c_fp), ("ln_g", c_fp), ("dx", c_fp), ("d_ln_g", c_fp), ("d_ln_b", c_fp),
                ("w_ih1", c_fp), ("dW_ih1", c_fp), ("dW_hh1", c_fp), ("db_ih1", c_fp), ("db_hh1", c_fp),
                ("hs_f16", C.c_int), ("recompute", C.c_int), ("b_ih", c_fp * 2), ("b_hh", c_fp * 2),
                ("slab_flags", C.c_void_p), ("slab_len", C.c_int), ("slab_started", C.c_void_p), ("wide", C.c_int), ("split", C.c_int),
                ("h0", C.c_void_p),
                ("tile_order", C.c_void_p), ("tile_need", C.c_void_p), ("ord_counter", C.c_void_p), ("ord_guard", C.c_int),
                ("slab_need", C.c_int), ("row_base", C.c_int), ("ord_grid", C.c_int),
                ("pro_du", c_fp), ("pro_x", c_fp), ("pro_res", c_fp), ("pro_ln_g", c_fp), ("pro_dy", c_fp)]


class WView(C.Structure):
    """sb_wview: a logical [N, K] matrix over a parameter's native layout (see the header)"""
    _fields_ = [("off", i64), ("nmod", C.c_int), ("sn_lo", i64), ("sn_hi", i64),
                ("kmod", C.c_int), ("sk_lo", i64), ("sk_hi", i64), ("kvalid", C.c_int), ("nvalid", C.c_int)]

    BIG = 1 << 30

    @classmethod
    def make(cls, sn_lo, sk_lo, *, off=0, nmod=None, sn_hi=0, kmod=None, sk_hi=0, kvalid=None, nvalid=None):
        v = cls()
        v.off, v.nmod, v.sn_lo, v.sn_hi = off, nmod or cls.BIG, sn_lo, sn_hi
        v.kmod, v.sk_lo, v.sk_hi = kmod or cls.BIG, sk_lo, sk_hi
        v.kvalid, v.nvalid = kvalid if kvalid is not None else cls.BIG, nvalid if nvalid is not None else cls.BIG
        return v


class LinearArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("F", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("n_valid", C.c_int), ("kseg", C.c_int), ("epi", C.c_int),
                ("inp", c_fp), ("is_b", i64), ("is_t", i64), ("is_f", i64), ("is_seg", i64),
                ("w", c_fp), ("bias", c_fp),
                ("out", c_fp), ("os_b", i64), ("os_t", i64), ("os_f", i64),
                ("res", c_fp), ("rs_b", i64), ("rs_t", i64), ("rs_f", i64),
                ("prelu_a", c_fp), ("ln_g", c_fp), ("ln_b", c_fp),
                ("aux_in", c_fp), ("aux_out", c_fp), ("partials", c_fp), ("accumulate", C.c_int),
                ("absmax_out", c_fp), ("mma", C.c_int)]


class WViewJob(C.Structure):
    _fields_ = [("src", c_fp), ("dst", c_fp), ("v", WView), ("N", C.c_int), ("K", C.c_int)]


class WgradArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("T", C.c_int), ("F", C.c_int), ("N", C.c_int), ("K", C.c_int), ("kseg", C.c_int),
                ("g", c_fp), ("ldg", i64),
                ("inp", c_fp), ("is_b", i64), ("is_t", i64), ("is_f", i64), ("is_seg", i64),
                ("in2", c_fp), ("ld2", i64), ("shift2", i64), ("K2", C.c_int),
                ("seg_len", C.c_int), ("skip_first", C.c_int), ("skip_last", C.c_int),
                ("transpose_out", C.c_int), ("dW", c_fp), ("dW2", c_fp), ("dbias", c_fp), ("dbias2", c_fp),
                ("scratch", c_fp), ("in_f16", C.c_int), ("perm_k", C.c_int), ("perm_n", C.c_int), ("bias_mod", C.c_int), ("wv", WView), ("gmax", c_fp), ("mma", C.c_int)]


class LstmGenFwdArgs(C.Structure):
    """sb_lstm_gen_fwd_args: the generic-shape recurrence (reference constructor defaults D = 64 / H = 128)"""
    _fields_ = [("nseq", C.c_int), ("nsteps", C.c_int), ("n_inner", C.c_int), ("ndir", C.c_int), ("C", C.c_int), ("H", C.c_int),
                ("p_outer", i64), ("p_inner", i64), ("p_step", i64),
                ("x", c_fp), ("ln_g", c_fp), ("ln_b", c_fp),
                ("w_ih", c_fp * 2), ("w_hh", c_fp * 2), ("b_ih", c_fp * 2), ("b_hh", c_fp * 2),
                ("h0", c_fp), ("c0", c_fp), ("hN", c_fp), ("cN", c_fp),
                ("hs", c_fp), ("save_gates", c_fp), ("save_u", c_fp)]


class LstmGenBwdArgs(C.Structure):
    _fields_ = [("nseq", C.c_int), ("nsteps", C.c_int), ("n_inner", C.c_int), ("ndir", C.c_int), ("H", C.c_int),
                ("p_outer", i64), ("p_inner", i64), ("p_step", i64),
                ("w_hh", c_fp * 2), ("save_gates", c_fp), ("dhs", c_fp), ("dgates", c_fp), ("gmax", c_fp)]


class LstmStreamArgs(C.Structure):
    _fields_ = [("P", i64), ("ndir", C.c_int), ("C", C.c_int),
                ("shift_pos", i64), ("seg_len", C.c_int), ("skip", C.c_int),
                ("dgates", c_fp), ("u", c_fp), ("hs", c_fp),
                ("w_ih", c_fp * 2),
                ("dW_ih", c_fp * 2), ("dW_hh", c_fp * 2), ("db_ih", c_fp * 2), ("db_hh", c_fp * 2),
                ("du_part", c_fp), ("scratch", c_fp), ("split_bf16", C.c_int), ("gmax", c_fp),
                ("u_f16", C.c_int), ("hs_f16", C.c_int),
                ("ln_x", c_fp), ("ln_g", c_fp), ("ln_res", c_fp), ("dx", c_fp), ("d_ln_g", c_fp), ("d_ln_b", c_fp),
                ("absmax_out", c_fp), ("d_lin_w", c_fp), ("d_lin_b", c_fp),
                ("slab_flags", C.c_void_p), ("slab_len", C.c_int), ("slab_need", C.c_int),
                ("chunk_counter", C.c_void_p), ("started", C.c_void_p), ("nchunks", C.c_int), ("guard", C.c_int),
                ("row_base", C.c_int), ("sched_status", C.c_void_p), ("wide", C.c_int)]


class LnBwdArgs(C.Structure):
    _fields_ = [("P", i64), ("ndir", C.c_int), ("C", C.c_int),
                ("du_part", c_fp), ("xin", c_fp), ("ln_g", c_fp), ("prelu_a", c_fp), ("res", c_fp),
                ("out", c_fp), ("partials", c_fp), ("absmax_out", c_fp)]


class AttnArgs(C.Structure):
    _fields_ = [("BH", C.c_int), ("Hh", C.c_int), ("T", C.c_int), ("F", C.c_int), ("Cv", C.c_int), ("L", C.c_int),
                ("NRp", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("scale", C.c_float),
                ("Q", c_fp), ("K", c_fp), ("V", c_fp), ("out", c_fp), ("lse", c_fp)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("BH", C.c_int), ("Hh", C.c_int), ("T", C.c_int), ("F", C.c_int), ("Cv", C.c_int), ("L", C.c_int),
                ("NRp", C.c_int), ("ldk", C.c_int), ("ldv", C.c_int), ("scale", C.c_float),
                ("Q", c_fp), ("K", c_fp), ("V", c_fp), ("dO", c_fp), ("lse", c_fp),
                ("delta", c_fp), ("dQ", c_fp), ("dK", c_fp), ("dV", c_fp)]


FILM_BANK_MAX_LAYERS = 16


class FilmBankArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("F", C.c_int), ("C", C.c_int), ("n", C.c_int), ("K", C.c_int), ("d_in", C.c_int),
                ("dis", c_fp), ("W_e", c_fp), ("ln_w", c_fp), ("ln_b", c_fp),
                ("conv_w", c_fp * (2 * FILM_BANK_MAX_LAYERS)), ("conv_b", c_fp * (2 * FILM_BANK_MAX_LAYERS)),
                ("planes", c_fp), ("G", c_fp), ("dW_e", c_fp), ("d_ln_w", c_fp), ("d_ln_b", c_fp),
                ("d_conv_w", c_fp * (2 * FILM_BANK_MAX_LAYERS)), ("d_conv_b", c_fp * (2 * FILM_BANK_MAX_LAYERS)),
                ("partials", c_fp)]


EPI_NONE, EPI_RES, EPI_PRELU, EPI_LN, EPI_LNBWD = range(5)

# every symbol include/sound_bubble_hip.h declares: name -> (restype, argtypes)
_vp, _ci, _cf = C.c_void_p, C.c_int, C.c_float
MULTI_COPY_MAX = 16


class MultiCopyArgs(C.Structure):
    _fields_ = [("src", C.c_void_p * MULTI_COPY_MAX), ("dst", C.c_void_p * MULTI_COPY_MAX), ("n", C.c_int64 * MULTI_COPY_MAX),
                ("njobs", C.c_int)]


SYMBOLS = {
    "sb_multi_copy": (_ci, [C.POINTER(MultiCopyArgs), _vp]),
    "sb_lstm_fwd": (_ci, [C.POINTER(LstmFwdArgs), _vp]),
    "sb_lstm_wide_rec_dwords": (_ci, []),
    "sb_flags_alloc": (_ci, [i64, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "sb_flags_free": (_ci, [_vp]),
    "sb_flags_zero": (_ci, [_vp, i64, _vp]),
    "sb_flags_read": (_ci, [_vp, i64, C.POINTER(C.c_int), _vp]),
    "sb_rec_q24_roundtrip": (_ci, [c_fp, c_fp, _vp, i64, _vp]),
    "sb_lstm_fwd_flag_ints": (_ci, [_ci, _ci]),
    "sb_lstm_fwd_produce": (_ci, [C.POINTER(LstmFwdArgs), _vp, _ci, _vp]),
    "sb_lstm_fwd_produce_ex": (_ci, [C.POINTER(LstmFwdArgs), _vp, _ci, _ci, _vp]),
    "sb_lstm_fwd_consume": (_ci, [C.POINTER(LstmFwdArgs), _vp, _ci, _ci, _vp, _vp, _vp]),
    "sb_lstm_fwd_consume_staged_test": (_ci, [C.POINTER(LstmFwdArgs), _vp, _ci, _ci, _ci, _vp, _vp, _vp]),
    "sb_lstm_bwd_cross_rows": (_ci, [_ci, _ci]),
    "sb_lstm_bwd_cross_produce": (_ci, [C.POINTER(LstmBwdArgs), _vp, _ci, _ci, _vp]),
    "sb_lstm_bwd_cross_consume": (_ci, [C.POINTER(LstmBwdArgs), _vp, _ci, _ci, _vp, _vp, _vp]),
    "sb_lstm_bwd_cross_produce_ex": (_ci, [C.POINTER(LstmBwdArgs), _vp, _ci, _ci, _ci, _vp]),
    "sb_lstm_bwd_cross_consume_ex": (_ci, [C.POINTER(LstmBwdArgs), _vp, _ci, _ci, _vp, _vp, _ci, _vp]),
    "sb_lstm_bwd_rec": (_ci, [C.POINTER(LstmBwdArgs), _vp]),
    "sb_linear_fwd": (_ci, [C.POINTER(LinearArgs), _vp]),
    "sb_linear_grid": (_ci, [i64]),
    "sb_wview_gather": (_ci, [c_fp, _ci, _ci, _vp]),
    "sb_wgrad": (_ci, [C.POINTER(WgradArgs), _vp]),
    "sb_wgrad_grid": (_ci, [i64]),
    "sb_wgrad_scratch_rows": (_ci, [C.POINTER(WgradArgs)]),
    "sb_lstm_gen_fwd": (_ci, [C.POINTER(LstmGenFwdArgs), _vp]),
    "sb_lstm_gen_bwd_rec": (_ci, [C.POINTER(LstmGenBwdArgs), _vp]),
    "sb_lstm_gen_supported": (_ci, [_ci, _ci]),
    "sb_lstm_bwd_stream": (_ci, [C.POINTER(LstmStreamArgs), _vp]),
    "sb_lstm_stream_grid": (_ci, [i64]),
    "sb_lstm_bwd_inter_overlapped": (_ci, [C.POINTER(LstmBwdArgs), C.POINTER(LstmStreamArgs), _vp, _ci, _vp]),
    "sb_lstm_bwd_inter_pair_serial": (_ci, [C.POINTER(LstmBwdArgs), C.POINTER(LstmStreamArgs), _vp, _ci, _vp]),
    "sb_lstm_overlap_rows": (_ci, [i64, _ci]),
    "sb_overlap_available": (_ci, [_vp]),
    "sb_overlap_init": (_ci, [_vp, c_fp, C.POINTER(C.c_float)]),
    "sb_overlap_reprobe": (_ci, [_vp, c_fp, C.POINTER(C.c_float)]),
    "sb_overlap_shutdown": (_ci, []),
    "sb_overlap_side_fork": (_ci, [_vp, C.POINTER(C.c_void_p)]),
    "sb_overlap_join": (_ci, [_vp]),
    "sb_overlap_time_next_side_launch": (_ci, [_vp, _vp]),
    "sb_overlap_force": (_ci, [_vp]),
    "sb_ln_bwd": (_ci, [C.POINTER(LnBwdArgs), _vp]),
    "sb_ln_bwd_grid": (_ci, [i64]),
    "sb_head_ln": (_ci, [c_fp, c_fp, c_fp, c_fp, c_fp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, c_fp, _vp]),
    "sb_attn_core": (_ci, [C.POINTER(AttnArgs), _vp]),
    "sb_head_ln_bwd_grid": (_ci, [_ci, _ci]),
    "sb_head_ln_bwd": (_ci, [c_fp, c_fp, c_fp, c_fp, c_fp, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _ci, c_fp, _vp]),
    "sb_attn_core_bwd": (_ci, [C.POINTER(AttnBwdArgs), _vp]),
    "sb_colsum": (_ci, [c_fp, i64, i64, _ci, c_fp, c_fp, _vp]),
    "sb_reduce_rows": (_ci, [c_fp, _ci, i64, _ci, c_fp, _vp]),
    "sb_features": (_ci, [c_fp, i64, c_fp, _ci, _ci, _ci, _ci, _vp]),
    "sb_film_fwd": (_ci, [c_fp, c_fp, c_fp, c_fp, _ci, _ci, _ci, _ci, _vp]),
    "sb_film_bwd": (_ci, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, _ci, _ci, _ci, _ci, c_fp, _vp]),
    "sb_ln_film_bwd_rows": (_ci, [_ci, _ci, _ci]),
    "sb_ln_film_bwd": (_ci, [c_fp] * 10 + [_ci, _ci, _ci, _ci, c_fp, _vp]),
    "sb_add3": (_ci, [c_fp, c_fp, c_fp, i64, _ci, _vp]),
    "sb_film_bank_fwd": (_ci, [C.POINTER(FilmBankArgs), _vp]),
    "sb_film_bank_bwd_scratch": (_ci, [_ci, _ci, _ci, _ci]),
    "sb_film_bank_bwd": (_ci, [C.POINTER(FilmBankArgs), _vp]),
    "sb_tail_rows": (_ci, [c_fp, c_fp, c_fp, i64, _ci, _ci, _ci, _vp]),
    "sb_overlap_add": (_ci, [c_fp, c_fp, _ci, _ci, _ci, _ci, _vp]),
    "sb_overlap_add_bwd": (_ci, [c_fp, c_fp, _ci, _ci, _ci, _ci, _vp]),
    "sb_deconv_bwd_data": (_ci, [c_fp, c_fp, c_fp, _ci, _ci, _ci, _ci, c_fp, _vp]),
    "sb_snrlp_loss_fwd": (_ci, [c_fp, c_fp, _ci, i64, _cf, _ci, c_fp, c_fp, c_fp, _vp]),
    "sb_snrlp_loss_bwd": (_ci, [c_fp, c_fp, _ci, i64, _cf, _ci, c_fp, c_fp, c_fp, _vp]),
    "sb_stage_frames": (_ci, [c_fp, c_fp, c_fp, _ci, _ci, _ci, _ci, _ci, _vp]),
    "sb_frames_to_state": (_ci, [c_fp, c_fp, _ci, _ci, _ci, _ci, _ci, _ci, _vp]),
    "sb_spec_rows": (_ci, [c_fp, c_fp, _ci, _ci, _ci, _ci, _ci, _vp]),
    "sb_sumsq_ex": (_ci, [c_fp, i64, c_fp, _ci, _vp]),
    "sb_snrlp_loss_ex": (_ci, [c_fp, c_fp, _ci, i64, _cf, _ci, c_fp, c_fp, c_fp, _vp]),
    "sb_signal_stats": (_ci, [c_fp, c_fp, c_fp, _ci, i64, i64, c_fp, _vp]),
    "sb_sumsq": (_ci, [c_fp, i64, c_fp, _vp]),
    "sb_absmax": (_ci, [c_fp, i64, c_fp, _vp]),
    "sb_adam_step": (_ci, [c_fp, c_fp, c_fp, c_fp, i64, _cf, _cf, _cf, _cf, _ci, _cf, _cf, c_fp, _vp]),
    "sb_adam_step_guarded": (_ci, [c_fp, c_fp, c_fp, c_fp, i64, _cf, _cf, _cf, _cf, _ci, _cf, _cf, c_fp, _vp, _vp, _vp]),
    "sb_fir": (_ci, [c_fp, c_fp, c_fp, _ci, i64, _ci, _vp]),
    "sb_reflect_pad": (_ci, [c_fp, c_fp, _ci, i64, _ci, i64, _vp]),
    "sb_stft_mag_l1_grid": (_ci, [i64, _ci]),
    "sb_stft_mag_l1": (_ci, [c_fp, c_fp, i64, _ci, _ci, _cf, _cf, c_fp, c_fp, _cf, c_fp, _ci, _vp]),
    "sb_stft_mag_terms": (_ci, [c_fp, c_fp, i64, _ci, _ci, _cf, _cf, _cf, _cf, _cf, c_fp, c_fp, c_fp, c_fp, _vp]),
    "sb_stft_f64acc": (_ci, [c_fp, c_fp, c_fp, _ci, _ci, i64, _ci, _ci, _ci, _ci, i64, _vp]),
    "sb_fir_pair": (_ci, [c_fp, c_fp, c_fp, c_fp, _ci, i64, _ci, _vp]),
    "sb_frames_fold": (_ci, [c_fp, c_fp, _ci, i64, _ci, _ci, _ci, _ci, _ci, _ci, _ci, _vp]),
    "sb_l1_grad": (_ci, [c_fp, c_fp, i64, _cf, c_fp, _ci, c_fp, _cf, c_fp, _ci, _vp]),
}

_lib = None


class SoundBubbleHipError(RuntimeError):
    pass


def load():
    """dlopen the HIP library (building nothing: see sound_bubble_amd/build.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SoundBubbleHipError(
            f"{LIB_PATH} is missing. Build it with `python -m sound_bubble_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)       # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise SoundBubbleHipError(f"{what} failed with status {rc}")
