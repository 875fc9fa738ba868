"""ctypes binding of ``libripor_hip.so`` (C ABI in include/ripor_hip.h).

The product path has no CPU fallback: if the shared library is missing or no HIP device is
visible, every entry point raises (``RiporHipError``) instead of silently computing elsewhere.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# RPR_DEV_LIB=1: the development build of the same sources (-DRPR_DEV_SWITCHES: the A/B switches of kernel routes and
# generations are live, csrc/common.h dev_getenv) — tools/ and the tests that compare kernel variants bit for bit. The
# product library reads none of those switches.
LIB_NAME = "libripor_hip_dev.so" if os.environ.get("RPR_DEV_LIB", "0") not in ("", "0") else "libripor_hip.so"
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

(K_GEMM, K_DEC_SELF_ATTN, K_DEC_CROSS_ATTN, K_ENC_ATTN, K_RMSNORM, K_SELECT, K_OTHER, K_GEMM_SMALL, K_TAIL_SELF_ATTN, K_FORK,
 K_COUNT) = range(11)
KERNEL_CLASS_NAMES = ["gemm", "dec_self_attn", "dec_cross_attn", "enc_attn", "rmsnorm", "select", "other", "gemm_small",
                      "tail_self_attn", "fork"]
STATUS_SATURATED, STATUS_EMPTY_QUERY, STATUS_TAIL_LEFTOVER = 1, 2, 4
ABI_VERSION = 3

PREC_F32, PREC_F16X2, PREC_BF16 = 0, 1, 2
FLAG_LOG_SOFTMAX = 1
FLAG_NO_GRAPH = 2


class RiporHipError(RuntimeError):
    pass


GRAD_BUCKET_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)
c_f32p = C.POINTER(C.c_float)
c_f32pp = C.POINTER(c_f32p)


class ModelDesc(C.Structure):
    _fields_ = [
        ("vocab_size", C.c_int32), ("d_model", C.c_int32), ("d_kv", C.c_int32), ("d_ff", C.c_int32),
        ("num_heads", C.c_int32), ("num_layers", C.c_int32), ("num_decoder_layers", C.c_int32),
        ("rel_buckets", C.c_int32), ("rel_max_distance", C.c_int32), ("L", C.c_int32), ("V", C.c_int32),
        ("scaleup_output_hidden", C.c_int32), ("layer_norm_eps", C.c_float),
        ("shared", C.c_void_p), ("enc_rel_bias", C.c_void_p), ("dec_rel_bias", C.c_void_p),
        ("enc_final_ln", C.c_void_p), ("dec_final_ln", C.c_void_p), ("start_embed", C.c_void_p),
        ("in_embeds", C.c_void_p), ("out_embeds", C.c_void_p), ("dec_xkv", C.c_void_p),
        ("enc_ln0", C.POINTER(C.c_void_p)), ("enc_qkv", C.POINTER(C.c_void_p)), ("enc_o", C.POINTER(C.c_void_p)),
        ("enc_ln1", C.POINTER(C.c_void_p)), ("enc_wi", C.POINTER(C.c_void_p)), ("enc_wo", C.POINTER(C.c_void_p)),
        ("dec_ln0", C.POINTER(C.c_void_p)), ("dec_qkv", C.POINTER(C.c_void_p)), ("dec_o", C.POINTER(C.c_void_p)),
        ("dec_ln1", C.POINTER(C.c_void_p)), ("dec_xq", C.POINTER(C.c_void_p)), ("dec_xo", C.POINTER(C.c_void_p)),
        ("dec_ln2", C.POINTER(C.c_void_p)), ("dec_wi", C.POINTER(C.c_void_p)), ("dec_wo", C.POINTER(C.c_void_p)),
    ]


class DebugTaps(C.Structure):
    _fields_ = [("encoder_out", C.c_void_p), ("step_logits", C.c_void_p), ("step_scores", C.c_void_p),
                ("step_tokens", C.c_void_p), ("step_parent", C.c_void_p), ("step_valid", C.c_void_p)]


class KernelStats(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("launches", C.c_int64), ("flops", C.c_double), ("bytes", C.c_double)]


# every symbol include/ripor_hip.h declares: (restype, argtypes)
SIGNATURES = {
    "rpr_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rpr_free_ctx": (None, [C.c_void_p]),
    "rpr_last_error": (C.c_char_p, []),
    "rpr_abi_version": (C.c_int, []),
    "rpr_rel_bucket": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "rpr_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "rpr_get_precision": (C.c_int, [C.c_void_p]),
    "rpr_load_model": (C.c_int, [C.c_void_p, C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "rpr_free_model": (None, [C.c_void_p]),
    "rpr_build_trie": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "rpr_free_trie": (None, [C.c_void_p]),
    "rpr_trie_num_rows": (C.c_int64, [C.c_void_p]),
    "rpr_trie_perm": (C.POINTER(C.c_int64), [C.c_void_p]),
    "rpr_trie_save": (C.c_int, [C.c_void_p, C.c_char_p]),
    "rpr_trie_load": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "rpr_trie_build_file": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_char_p, C.c_int64, C.c_int64,
                                      C.c_int64, C.c_char_p]),
    "rpr_trie_file_info": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rpr_trie_file_validate": (C.c_int, [C.c_char_p]),
    "rpr_trie_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                C.POINTER(C.c_int64)]),
    "rpr_trie_keys": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rpr_trie_set_vocab": (C.c_int, [C.c_void_p, C.c_int32]),
    "rpr_d2s_open": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    "rpr_d2s_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "rpr_d2s_copy": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "rpr_d2s_close": (None, [C.c_void_p]),
    "rpr_trie_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rpr_search": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                             C.c_int32, C.c_int32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.POINTER(DebugTaps), C.c_void_p]),
    "rpr_lngknp_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                     C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "rpr_param_count": (C.c_int64, [C.c_void_p]),
    "rpr_param_total": (C.c_int64, [C.c_void_p]),
    "rpr_param_info": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rpr_lngknp_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rpr_lngknp_backward_buckets": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                              C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rpr_adamw_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "rpr_get_status": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.c_int]),
    "rpr_status_words_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "rpr_model_f32_only": (C.c_int, [C.c_void_p]),
    "rpr_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "rpr_profile_reset": (C.c_int, [C.c_void_p]),
    "rpr_profile_get": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(KernelStats)]),
    "rpr_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "rpr_set_lane_split": (C.c_int, [C.c_void_p, C.c_int32]),
    "rpr_lane_split": (C.c_int32, [C.c_void_p]),
    "rpr_set_forced_tail": (C.c_int, [C.c_void_p, C.c_int32]),
    "rpr_forced_tail": (C.c_int32, [C.c_void_p]),
    "rpr_set_fork_depths": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    "rpr_fork_depths": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32,
                                  C.POINTER(C.c_int32)]),
    "rpr_trie_single_frac": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_double)]),
    "rpr_last_fork_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rpr_op_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_void_p]),
    "rpr_op_linear_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "rpr_op_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                 C.c_void_p]),
    "rpr_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                             C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the library (once). Raises RiporHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RiporHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the search path.")
    # torch ships its own libamdhip64.so.7; the library must share that HIP runtime instance so
    # torch device pointers / streams are valid inside it. Importing torch first makes the
    # dynamic loader resolve our DT_NEEDED libamdhip64.so.7 to the copy torch already mapped.
    import torch  # noqa: F401
    hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(hip_rt):
        C.CDLL(hip_rt, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().rpr_last_error()
        raise RiporHipError(f"{what or 'libripor_hip'} failed (status {status}): "
                            f"{msg.decode('utf-8', 'replace') if msg else ''}")
