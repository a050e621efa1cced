"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports exactly the symbols
include/phonlp_heads.h declares; argument validation that needs no GPU returns the documented error codes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def heads_lib():
    from phonlp_amd import build as B
    from phonlp_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        B.build()
    return _lib.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "phonlp_heads.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(phonlp_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(heads_lib):
    from phonlp_amd import _lib
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(heads_lib, n), f"{n} declared in include/phonlp_heads.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and header disagree"


def test_kp_and_version(heads_lib):
    assert heads_lib.phonlp_kp(400) == 408  # 400 features + ones column, padded to 8
    assert heads_lib.phonlp_kp(16) == 24
    assert heads_lib.phonlp_version() >= 100


def test_argument_errors_without_gpu(heads_lib):
    # NULL pointers / bad shapes are rejected before any HIP call
    assert heads_lib.phonlp_crf_viterbi(None, None, None, 1, 1, 1, None, None, None) == -1
    assert b"crf_viterbi" in heads_lib.phonlp_last_error()
    assert heads_lib.phonlp_mst_one_root(None, None, 0, 0, None, None, None, 0, None) == -1
    one = ctypes.c_void_p(16)
    assert heads_lib.phonlp_crf_viterbi(one, one, one, 1, 4, 65, one, None, None) == -4  # T > 64 unsupported
    assert heads_lib.phonlp_linear(one, 6, one, 6, None, one, 4, 2, 2, 6, 0, None, 0, None) == -1  # K % 4 != 0
    # fused tagger head: NULL operands, more than 64 POS tags (one lane per tag), workspace too small
    assert heads_lib.phonlp_tagger_head(None, 768, None, 1, 1, 768, None, None, None, None, None, None, None, 400, 24, 100, 9,
                                        None, None, None, None, 0, None) == -1
    assert b"tagger_head" in heads_lib.phonlp_last_error()
    args = [one, 768, one, 2, 8, 768, one, one, one, one, one, one, one, 400, 24, 100, 9, one, one, None]
    assert heads_lib.phonlp_tagger_head(*args[:14], 65, *args[15:], one, 1 << 30, None) == -4
    assert heads_lib.phonlp_tagger_head(*args, one, 16, None) == -2
    assert heads_lib.phonlp_tagger_head_ws_bytes(2, 8, 768, 400, 24, 100, 9) > 16 * (412 + 24) * 4
    # engine-2 stage A refuses what the rows-by-4 layout cannot take before any launch
    assert heads_lib.phonlp_label_stage_a_split(None, None, None, None, None, 1, 1, 400, 33, None, None) == -1
    assert heads_lib.phonlp_profile_only(b"linear:label_stage_a") == 0 and heads_lib.phonlp_profile_only(None) == 0
    assert heads_lib.phonlp_mst_ws_bytes(256, 128) > 0
    # parser loss: NULL operands, raw scorer outputs missing, number_of_words <= 0, workspace too small
    assert heads_lib.phonlp_dep_loss(None, None, None, None, None, None, None, 1, 2, 400, 33, 1, 1, 2.0, None, None, 0, None) == -1
    assert b"dep_loss" in heads_lib.phonlp_last_error()
    assert heads_lib.phonlp_dep_loss(one, None, one, one, one, one, one, 1, 2, 400, 33, 1, 0, 2.0, one, one, 64, None) == -1
    assert heads_lib.phonlp_dep_loss(one, one, one, one, one, one, one, 1, 2, 400, 33, 1, 1, 0.0, one, one, 64, None) == -1
    assert heads_lib.phonlp_dep_loss(one, one, one, one, one, one, one, 1, 2, 400, 33, 1, 1, 2.0, one, one, 16, None) == -2
    assert heads_lib.phonlp_dep_loss_ws_bytes(256, 128) == 4 * 256 * 128 * 4
    assert heads_lib.phonlp_pos_loss(None, None, None, 1, 2, 24, 1, None, None, 0, None) == -1
    assert heads_lib.phonlp_crf_nll_grad(None, None, None, None, None, 1, 1, 9, 1.0, None, None, None, 0, None) == -1
    assert heads_lib.phonlp_crf_nll_grad(one, one, None, None, one, 1, 4, 65, 1.0, one, one, one, 1 << 20, None) == -4
    assert heads_lib.phonlp_crf_nll_grad(one, one, None, None, one, 2, 4, 9, 1.0, one, one, one, 16, None) == -2
    assert heads_lib.phonlp_pos_loss(one, one, one, 1, 2, 24, 1, one, one, 8, None) == -2
    assert heads_lib.phonlp_pos_loss_grad(one, one, one, None, 1, 2, 24, 1, one, None) == -1
    assert heads_lib.phonlp_biaffine_label_ws_bytes(256, 128, 400, 33) == 256 * 128 * (2 * 408 + 33 * 408) * 4
    # this round's entries: packed rows, device-side unsort, parser-loss gradient, the no-workspace CRF gradient
    assert heads_lib.phonlp_pack_descriptor(None, 1, 4, None, None, 4, None) == -1
    assert b"pack_descriptor" in heads_lib.phonlp_last_error()
    assert heads_lib.phonlp_pack_rows(one, 6, 8, one, 4, 6, one, None) == -1  # W % 4 != 0
    assert heads_lib.phonlp_pack_rows(one, 8, 8, one, 0, 8, one, None) == -1  # no rows
    assert heads_lib.phonlp_unpack_rows(None, one, 4, 8, one, 8, None) == -1
    assert heads_lib.phonlp_scatter_rows(one, 2, 8, 8, one, one, 8, 4, None, None, None) == -1  # col0 >= Ls
    # packed arc scores: more live rows than B*L, n_scorers not 1 or 3, workspace too small
    assert heads_lib.phonlp_biaffine_arc_scores_packed(one, 100, one, one, one, 2, 8, 400, 3, 1, 1, one, one, None, one,
                                                       1 << 30, None) == -1
    assert heads_lib.phonlp_biaffine_arc_scores_packed(one, 10, one, one, one, 2, 8, 400, 2, 0, 0, one, one, None, one,
                                                       1 << 30, None) == -1
    assert heads_lib.phonlp_biaffine_arc_scores_packed(one, 10, one, one, one, 2, 8, 400, 3, 1, 1, one, one, None, one,
                                                       16, None) == -2
    assert heads_lib.phonlp_label_at_heads_fused_packed(one, one, one, one, one, one, one, None, one, 10, 2, 8, 400, 33, one,
                                                        None, one, 1 << 30, None) == -1  # rowmap missing
    assert heads_lib.phonlp_label_at_heads_fused_packed(one, one, one, one, one, one, one, one, one, 10, 2, 8, 400, 33, one,
                                                        None, one, 16, None) == -2
    assert heads_lib.phonlp_dep_loss_grad(one, None, one, one, one, one, one, 1, 2, 400, 33, 1, 0, 2.0, one, one, one, one,
                                          None) == -1  # linearization term without the raw scorer outputs
    assert b"dep_loss_grad" in heads_lib.phonlp_last_error()
    assert heads_lib.phonlp_dep_loss_grad(one, one, one, one, one, one, one, 1, 2, 400, 33, 1, 1, 0.0, one, one, one, one,
                                          None) == -1
    # backward through the layers: NULL operands, pitches shorter than the rows, ReLU without the layer's output,
    # workspaces too small, a misaligned operand
    assert heads_lib.phonlp_linear_backward(None, 8, one, 8, None, 4, one, 4, 2, 4, 8, 0, one, 8, one, 8, one, one, 1 << 30, None) == -1
    assert b"linear_backward" in heads_lib.phonlp_last_error()
    assert heads_lib.phonlp_linear_backward(one, 4, one, 8, None, 4, one, 4, 2, 4, 8, 0, one, 8, one, 8, one, one, 1 << 30, None) == -1
    assert heads_lib.phonlp_linear_backward(one, 8, one, 8, None, 4, one, 4, 2, 4, 8, 1, one, 8, one, 8, one, one, 1 << 30, None) == -1
    assert heads_lib.phonlp_linear_backward(one, 8, one, 8, None, 4, one, 4, 2, 4, 8, 0, one, 8, one, 8, one, one, 16, None) == -2
    assert heads_lib.phonlp_linear_backward_ws_bytes(0, 4, 8) == 0 and heads_lib.phonlp_linear_backward_ws_bytes(32768, 2448, 868) > 0
    assert heads_lib.phonlp_bilinear_backward(one, one, None, 2, 8, 400, one, one, one, 1 << 30, None) == -1
    assert heads_lib.phonlp_bilinear_backward(one, one, one, 2, 8, 400, one, one, one, 16, None) == -2
    assert heads_lib.phonlp_bilinear_backward(ctypes.c_void_p(one.value + 4), one, one, 2, 8, 400, one, one, one, 1 << 30, None) == -1
    assert heads_lib.phonlp_label_backward(one, one, None, one, None, one, 2, 8, 400, 33, one, one, one, 1 << 30, None) == -1
    assert heads_lib.phonlp_label_backward(one, one, None, one, one, one, 2, 2000, 400, 33, one, one, one, 1 << 40, None) == -1  # L > 1024
    assert heads_lib.phonlp_label_backward(one, one, None, one, one, one, 2, 8, 400, 33, one, one, one, 16, None) == -2
    assert b"label_backward" in heads_lib.phonlp_last_error()
    assert heads_lib.phonlp_softmax_rows(None, 4, 24, one, None) == -1 and heads_lib.phonlp_softmax_rows(one, 4, 2000, one, None) == -1
    assert heads_lib.phonlp_softmax_rows_backward(one, None, 4, 24, one, None) == -1
    assert heads_lib.phonlp_crf_nll_grad_static_max_bytes() == 4 << 20
    assert heads_lib.phonlp_crf_nll_grad_static(one, one, None, None, one, 4096, 4, 64, 1.0, one, one, None) == -4  # B*T*T*4 > 4 MiB
    # the arc scorers' per-scorer form: a workspace only for few short sentences; bad arguments and a short workspace are refused
    assert heads_lib.phonlp_arc_scores_split_ws_bytes(32, 64, 3) == 2 * 32 * 64 * 64 * 4
    assert heads_lib.phonlp_arc_scores_split_ws_bytes(1, 32, 3) == 2 * 32 * 32 * 4
    assert heads_lib.phonlp_arc_scores_split_ws_bytes(200, 64, 3) == 0  # 3 B > 512: a workgroup per sentence
    assert heads_lib.phonlp_arc_scores_split_ws_bytes(8, 200, 3) == 0 and heads_lib.phonlp_arc_scores_split_ws_bytes(8, 64, 1) == 0
    assert heads_lib.phonlp_arc_scores_split_ws(None, one, 64, 1, 64, 400, 3, 1, 1, one, one, None, None, None, 0.0, one, None, 0, None) == -1
    assert heads_lib.phonlp_arc_scores_split_ws(one, one, 64, 1, 64, 400, 3, 1, 1, one, one, None, None, None, 0.0, one, one, 16, None) == -1
    assert b"workspace" in heads_lib.phonlp_last_error()
    assert heads_lib.phonlp_get_mst_fast_path() in (0, 1)
    assert heads_lib.phonlp_mst_fallback_counter_offset(256, 128) > 0 and heads_lib.phonlp_mst_fallback_counter_offset(64, 512) > 0
    assert heads_lib.phonlp_mst_fallback_counter_offset(8, 136) == -1  # the reference-order kernel alone decodes this length


def test_missing_library_fails_loudly(monkeypatch):
    from phonlp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libphonlp_heads.so")
    with pytest.raises(_lib.HeadsLibraryError):
        _lib.lib()


def test_fused_label_weight_image_row_order():
    """HeadWeights._pad_relations: the weight image of phonlp_label_at_heads_fused holds relation o, head feature
    32*blk + r in row (blk*O + o)*32 + r (include/phonlp_heads.h), rows past KP zero -- host logic, no GPU."""
    import torch
    from phonlp_amd import HeadWeights, ops
    H, O = 40, 5
    KP, KPC = ops.kp(H), ops.label_kpc(H)
    assert KPC % 32 == 0 and KPC >= KP
    g = torch.Generator().manual_seed(0)
    wbt = torch.randn(O * KP, KP, generator=g)
    img = HeadWeights._pad_relations(wbt, H, O)
    assert img.shape == (O * KPC, KP)
    for o in range(O):
        for d2 in range(KPC):
            blk, r = divmod(d2, 32)
            row = img[(blk * O + o) * 32 + r]
            if d2 < KP:
                assert torch.equal(row, wbt[o * KP + d2])
            else:
                assert not row.any()
