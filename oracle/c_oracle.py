"""TEST INFRASTRUCTURE ONLY -- ctypes wrapper around oracle/decoders.c (the plain-C CPU oracle).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "decoders.c")
LIB = os.path.join(HERE, "_build", "liboracle.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, SRC, "-lm"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def viterbi_batch(logits, trans, sentlens):
    logits = np.ascontiguousarray(logits, np.float32)
    trans = np.ascontiguousarray(trans, np.float32)
    lens = np.ascontiguousarray(sentlens, np.int32)
    B, L, T = logits.shape
    tags = np.zeros((B, L), np.int32)
    best = np.zeros(B, np.float32)
    rc = lib().oracle_viterbi_batch(_p(logits, ctypes.c_float), _p(trans, ctypes.c_float), _p(lens, ctypes.c_int32),
                                    B, L, T, _p(tags, ctypes.c_int32), _p(best, ctypes.c_float))
    assert rc == 0
    return tags, best


def crf_log_norm(inputs, pad, trans):
    inputs = np.ascontiguousarray(inputs, np.float32)
    pad = np.ascontiguousarray(pad, np.uint8)
    trans = np.ascontiguousarray(trans, np.float32)
    B, L, T = inputs.shape
    out = np.zeros(B, np.float32)
    rc = lib().oracle_crf_log_norm(_p(inputs, ctypes.c_float), _p(pad, ctypes.c_uint8), _p(trans, ctypes.c_float),
                                   B, L, T, _p(out, ctypes.c_float))
    assert rc == 0
    return out


def mst_one_root(scores):
    scores = np.ascontiguousarray(scores, np.float32)
    l = scores.shape[0]
    heads = np.zeros(l, np.int32)
    rc = lib().oracle_mst_one_root(_p(scores, ctypes.c_float), l, l, _p(heads, ctypes.c_int32))
    return heads, rc


def mst_batch(arc, sentlens):
    arc = np.ascontiguousarray(arc, np.float32)
    lens = np.ascontiguousarray(sentlens, np.int32)
    B, L, _ = arc.shape
    heads = np.zeros((B, L), np.int32)
    status = np.zeros(B, np.int32)
    rc = lib().oracle_mst_batch(_p(arc, ctypes.c_float), _p(lens, ctypes.c_int32), B, L, _p(heads, ctypes.c_int32),
                                _p(status, ctypes.c_int32))
    assert rc >= 0
    return heads, status
