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


def test_missing_library_fails_loudly(monkeypatch):
    from phonlp_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libphonlp_heads.so")
    with pytest.raises(_lib.HeadsLibraryError):
        _lib.lib()
