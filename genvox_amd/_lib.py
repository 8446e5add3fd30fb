"""ctypes binding of libgenvox_amd.so (C ABI declared in include/genvox_amd.h).

There is no fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("GVX_LIB", "libgenvox_amd.so"))  # GVX_LIB: diagnostic builds only


class GvxError(RuntimeError):
    pass


class gvx_dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_tokens", "embed_dim", "enc_kernel", "enc_n_conv", "prenet_dim", "att_rnn_dim", "dec_rnn_dim", "att_dim",
        "att_loc_filters", "att_loc_kernel", "postnet_dim", "postnet_kernel", "postnet_n_conv", "n_mels")]


class gvx_weight_desc(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


_vp, _i, _sz, _f, _l = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_long


class gvx_tensor_ref(C.Structure):
    _fields_ = [("data", C.c_void_p), ("numel", C.c_int64)]


class gvx_adam_ref(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("numel", C.c_int64)]


class gvx_bptt_decoder_args(C.Structure):
    """Argument block of gvx_train_decoder_bptt (include/genvox_amd.h)."""
    _fields_ = ([(n, C.c_int32) for n in ("B", "L", "T", "A", "D", "E", "P", "a", "F", "kl")]
                + [("att_scale", C.c_float), ("dec_scale", C.c_float)]
                + [(n, C.c_void_p) for n in ("dhc_all", "pre_a", "pre_d", "c_a_all", "c_d_all", "att_keep", "dec_keep", "q_all", "ctx_all")]
                + [("ctx_ts", C.c_int64), ("ctx_bs", C.c_int64)]
                + [(n, C.c_void_p) for n in ("w_all", "memory", "pm", "w_ih_a", "w_hh_a", "w_ih_d", "w_hh_d", "wq", "v", "loc_conv", "loc_dense",
                                             "dga_all", "dgd_all", "dq_all", "dctx_all", "dpm", "dmemory", "dv", "dloc_dense", "dloc_conv")])


# name -> (restype, argtypes); every symbol include/genvox_amd.h declares
SIGNATURES = {
    "gvx_last_error": (C.c_char_p, []),
    "gvx_version": (_i, []),
    "gvx_model_create": (_i, [C.POINTER(gvx_dims), C.POINTER(_vp)]),
    "gvx_model_destroy": (None, [_vp]),
    "gvx_model_blob_bytes": (_sz, [_vp]),
    "gvx_model_pack_weights": (_i, [_vp, C.POINTER(gvx_weight_desc), _i, _vp]),
    "gvx_model_bind_blob": (_i, [_vp, _vp]),
    "gvx_model_pack_weights_device": (_i, [_vp, C.POINTER(gvx_weight_desc), _i, _vp, _vp]),
    "gvx_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "gvx_workspace_bytes_autoregressive": (_sz, [_vp, _i, _i, _i]),
    "gvx_workspace_status": (_i, [_vp, _vp, _sz, _vp, _vp]),
    "gvx_encoder_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "gvx_decoder_teacher_forced": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gvx_decoder_autoregressive": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i), _vp, _sz, _vp]),
    "gvx_postnet_workspace_bytes": (_sz, [_vp, _i, _i]),
    "gvx_postnet_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "gvx_mask_padding": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "gvx_tacotron2_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gvx_tacotron2_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "gvx_encoder_lstm_forward": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gvx_decoder_teacher_forced_train": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gvx_train_export": (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _vp, _vp]),
    "gvx_conv_train_saved_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "gvx_conv_train_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "gvx_conv_bn_act_train_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _vp, _sz, _vp, _sz, _vp]),
    "gvx_conv_bn_act_train_backward": (_i, [_vp, _vp, _sz, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gvx_tacotron2_loss_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "gvx_train_gemm_nt": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "gvx_train_gemm_tn": (_i, [_vp, _l, _vp, _l, _vp, _l, _i, _i, _l, _vp, _sz, _vp]),
    "gvx_train_transpose": (_i, [_vp, _l, _vp, _l, _i, _l, _vp]),
    "gvx_train_colsum": (_i, [_vp, _l, _i, _vp, _vp]),
    "gvx_train_axpby": (_i, [_vp, _l, _f, _vp, _l, _f, _vp, _l, _l, _i, _vp]),
    "gvx_train_relu_dropout_backward": (_i, [_vp, _vp, _vp, _f, _l, _vp, _vp]),
    "gvx_train_unblock": (_i, [_vp, _vp, _l, _i, _i, _vp]),
    "gvx_train_embedding_backward": (_i, [_vp, _vp, _l, _i, _i, _vp, _vp]),
    "gvx_train_sqnorm_scratch_bytes": (_sz, [_i]),
    "gvx_train_sqnorm_many": (_i, [_vp, _i, _vp, _vp, _vp]),
    "gvx_train_adam_step_many": (_i, [_vp, _i, _f, _f, _f, _f, _f, _f, _i, _vp]),
    "gvx_train_decoder_bptt_workspace_bytes": (_sz, [C.POINTER(gvx_bptt_decoder_args)]),
    "gvx_train_decoder_bptt": (_i, [C.POINTER(gvx_bptt_decoder_args), _vp, _sz, _vp]),
    "gvx_train_encoder_lstm_bptt_workspace_bytes": (_sz, [_i, _i]),
    "gvx_train_encoder_lstm_bptt": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "gvx_train_encoder_lstm_bptt_resident": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "gvx_train_encoder_lstm_bptt_status": (_i, [_vp, _sz, _i, _i, C.POINTER(_i), _vp]),
    "gvx_prenet_masks_generate": (_i, [_vp, _sz, C.c_uint64, _vp]),
    "gvx_stage_timing_enable": (_i, [_vp, _i]),
    "gvx_stage_times_ms": (_i, [_vp, C.POINTER(_f), C.POINTER(_i)]),
    "gvx_gl_plan_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "gvx_gl_plan_destroy": (None, [_vp]),
    "gvx_gl_workspace_bytes": (_sz, [_vp, _i, _i, _i]),
    "gvx_stft": (_i, [_vp, _vp, _vp, _i, C.c_long, _vp, _vp, _sz, _vp]),
    "gvx_istft": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _sz, _vp]),
    "gvx_mel_to_magnitude": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "gvx_griffin_lim": (_i, [_vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "gvx_wav_to_mel": (_i, [_vp, _vp, _vp, _vp, _i, C.c_long, _i, _i, _f, _vp, _vp, _sz, _vp]),
    "gvx_wav_finalize": (_i, [_vp, _i, C.c_long, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), _i, _vp, _vp, _vp]),
    "gvx_kernel_timing_enable": (_i, [_vp, _i]),
    "gvx_model_set_persistent_attention": (_i, [_vp, _i]),
    "gvx_model_set_resident_kernels": (_i, [_vp, _i]),
    "gvx_teacher_forced_rows_per_call": (_i, [_vp, _i]),
    "gvx_teacher_forced_resident": (_i, [_vp, _i, _i]),
    "gvx_teacher_forced_loop_kind": (_i, [_vp, _i, _i]),
    "gvx_autoregressive_loop_kind": (_i, [_vp, _i, _i]),
    "gvx_kernel_times_ms": (_i, [_vp, C.POINTER(_f), C.POINTER(_f), C.POINTER(_i)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library once; raise ImportError with the build recipe if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the gfx950 library first (python -m genvox_amd.build, or "
            f"__graft_entry__.build()). genvox_amd has no CPU or eager fallback.")
    # torch ships its own libamdhip64; it must be the one already loaded when this library's dependency on the HIP
    # runtime is resolved, otherwise the process ends up with two runtimes and device pointers / streams do not mix.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().gvx_last_error()
        raise GvxError(f"genvox_amd error {rc}: {msg.decode() if msg else '?'}")
