"""ctypes binding of the C-ABI in include/phonlp_heads.h.  Fails loudly when the HIP library is missing:
there is no CPU or PyTorch fallback anywhere in this package."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libphonlp_heads.so")

_i, _z, _p, _f = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float

# name -> (restype, argtypes); mirrors include/phonlp_heads.h declaration by declaration
SIGNATURES = {
    "phonlp_kp": (_i, [_i]),
    "phonlp_version": (_i, []),
    "phonlp_last_error": (ctypes.c_char_p, []),
    "phonlp_set_gemm_engine": (_i, [_i]),
    "phonlp_get_gemm_engine": (_i, []),
    "phonlp_linear": (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p]),
    "phonlp_split_bytes": (_z, [_i, _i]),
    "phonlp_split_rows": (_i, [_p, _i, _i, _i, _p, _p, _p]),
    "phonlp_linear_split": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    "phonlp_linear_split_image": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _f, _f, _i, _p]),
    "phonlp_arc_kc": (_i, [_i]),
    "phonlp_arc_scores_split": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p]),
    "phonlp_arc_scores_split_ws_bytes": (_z, [_i, _i, _i]),
    "phonlp_arc_scores_split_ws": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _z, _p]),
    "phonlp_pack_scorer_linear": (_i, [_p, _p, _p, _p, _i, _i, _p, _p, _p]),
    "phonlp_pack_bilinear": (_i, [_p, _i, _i, _p, _p]),
    "phonlp_pos_head_ws_bytes": (_z, [_i, _i, _i, _i, _i, _i]),
    "phonlp_pos_head": (_i, [_p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "phonlp_pos_head_split": (_i, [_p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "phonlp_tagger_head_ws_bytes": (_z, [_i, _i, _i, _i, _i, _i, _i]),
    "phonlp_tagger_head": (_i, [_p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "phonlp_ner_emission": (_i, [_p, _i, _p, _i, _i, _p, _p, _i, _p, _p]),
    "phonlp_dep_hidden": (_i, [_p, _i, _p, _i, _i, _p, _p, _p, _p]),
    "phonlp_biaffine_arc_ws_bytes": (_z, [_i, _i, _i, _i]),
    "phonlp_deep_biaffine_arc": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p, _p, _p]),
    "phonlp_biaffine_arc_scores_ws_bytes": (_z, [_i, _i, _i, _i]),
    "phonlp_biaffine_arc_scores": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "phonlp_biaffine_label_ws_bytes": (_z, [_i, _i, _i, _i]),
    "phonlp_deep_biaffine_label": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p, _p, _p]),
    "phonlp_crf_viterbi": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p]),
    "phonlp_crf_log_partition": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p]),
    "phonlp_crf_nll_grad_ws_bytes": (_z, [_i, _i]),
    "phonlp_crf_nll_grad": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, ctypes.c_float, _p, _p, _p, _z, _p]),
    "phonlp_crf_nll_grad_static_max_bytes": (_z, []),
    "phonlp_crf_nll_grad_static": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, ctypes.c_float, _p, _p, _p]),
    "phonlp_mst_ws_bytes": (_z, [_i, _i]),
    "phonlp_mst_one_root": (_i, [_p, _p, _i, _i, _p, _p, _p, _z, _p]),
    "phonlp_set_mst_fast_path": (_i, [_i]),
    "phonlp_get_mst_fast_path": (_i, []),
    "phonlp_mst_fallback_counter_offset": (ctypes.c_long, [_i, _i]),
    "phonlp_mst_fallback_list_offset": (ctypes.c_long, [_i, _i]),
    "phonlp_ner_entity_counts": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p]),
    "phonlp_label_project": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "phonlp_label_at_heads": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "phonlp_label_t4_bytes": (_z, [_i, _i, _i, _i]),
    "phonlp_label_stage_a_split": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "phonlp_label_at_heads_rows4": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p]),
    "phonlp_label_kpc": (_i, [_i]),
    "phonlp_label_at_heads_fused_ws_bytes": (_z, [_i, _i, _i, _i]),
    "phonlp_label_at_heads_fused": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "phonlp_gather_labels": (_i, [_p, _p, _p, _i, _i, _p, _p]),
    "phonlp_gather_first_subword": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _p, _p]),
    "phonlp_scatter_rows": (_i, [_p, _i, _i, _i, _p, _p, _i, _i, _p, _p, _p]),
    "phonlp_pack_descriptor": (_i, [_p, _i, _i, _p, _p, _i, _p]),
    "phonlp_pack_rows": (_i, [_p, _i, _i, _p, _i, _i, _p, _p]),
    "phonlp_unpack_rows": (_i, [_p, _p, _i, _i, _p, _i, _p]),
    "phonlp_biaffine_arc_scores_packed": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _z, _p]),
    "phonlp_label_at_heads_fused_packed": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "phonlp_eval_counts": (_i, [_p] * 11 + [_i, _i, _i, _p, _p]),
    "phonlp_dep_loss_ws_bytes": (_z, [_i, _i]),
    "phonlp_dep_loss": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, ctypes.c_float, _p, _p, _z, _p]),
    "phonlp_linear_backward_ws_bytes": (_z, [_i, _i, _i]),
    "phonlp_linear_backward": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _z, _p]),
    "phonlp_bilinear_backward_ws_bytes": (_z, [_i, _i, _i]),
    "phonlp_bilinear_backward": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _z, _p]),
    "phonlp_label_backward_ws_bytes": (_z, [_i, _i, _i, _i]),
    "phonlp_label_backward": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "phonlp_softmax_rows": (_i, [_p, _i, _i, _p, _p]),
    "phonlp_softmax_rows_backward": (_i, [_p, _p, _i, _i, _p, _p]),
    "phonlp_dep_loss_grad": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, ctypes.c_float, _p, _p, _p, _p, _p]),
    "phonlp_pos_loss_ws_bytes": (_z, [_i, _i]),
    "phonlp_pos_loss": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _z, _p]),
    "phonlp_pos_loss_grad": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p]),
    "phonlp_profile_begin": (_i, [_i]),
    "phonlp_profile_end": (_i, [_p, _p, _i]),
    "phonlp_profile_only": (_i, [ctypes.c_char_p]),
    "phonlp_profile_tag_name": (ctypes.c_char_p, [_i]),
}


class HeadsLibraryError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libphonlp_heads.so once.  Raises if it has not been built (python -m phonlp_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HeadsLibraryError(
                f"{LIB_PATH} is missing: build it with `python -m phonlp_amd.build` (hipcc, gfx950). "
                "phonlp_amd has no CPU fallback."
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().phonlp_last_error().decode("utf-8", "replace")
        raise HeadsLibraryError(f"{what} failed with code {rc}: {msg}")
