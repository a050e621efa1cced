"""PyTorch-ROCm custom ops (`torch.ops.phonlp_amd.*`) over the C-ABI of libphonlp_heads.so.

Each op allocates its outputs (and scratch) with torch's ROCm caching allocator, then enqueues the
hand-written HIP kernels on torch's current stream through ctypes.  PyTorch is plumbing here: device
memory and streams only -- none of the arithmetic of the task heads is done by torch operators.
"""
import ctypes
from typing import Optional, Tuple

import torch

from ._lib import check, lib

Tensor = torch.Tensor


def _p(t: Optional[Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current HIP stream on the current device, as the C-ABI's void*.  (torch.cuda.current_stream() builds a Stream
    object through three layers of Python: 9 us a call, a fifth of a small step's host time.)"""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t: Tensor, dtype, name: str):
    if not t.is_cuda:
        raise ValueError(f"{name}: expected a ROCm device tensor (phonlp_amd has no CPU path)")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous tensor")


def _ws(nbytes: int, device) -> Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def set_gemm_engine(engine: int) -> None:
    """0 = f32 MFMA (exact fp32 chain), 1 = fp32-equivalent 3-way bf16 split on the bf16 MFMA pipe.  Process-wide."""
    check(lib().phonlp_set_gemm_engine(int(engine)), "set_gemm_engine")


def get_gemm_engine() -> int:
    return int(lib().phonlp_get_gemm_engine())


def kp(H: int) -> int:
    return lib().phonlp_kp(H)


def label_kpc(H: int) -> int:
    return lib().phonlp_label_kpc(H)


@torch.library.custom_op("phonlp_amd::linear", mutates_args=())
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], relu: bool, sentlens: Optional[Tensor], L: int) -> Tensor:
    """y = act(x @ weight.T + bias); rows (b,i) with i >= sentlens[b] are zeroed when sentlens is given."""
    _need(x, torch.float32, "x")
    _need(weight, torch.float32, "weight")
    M, K = x.shape
    N = weight.shape[0]
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    check(lib().phonlp_linear(_p(x), K, _p(weight), K, _p(bias), _p(y), N, M, N, K, int(relu), _p(sentlens), L,
                              _stream()), "phonlp_linear")
    return y


@torch.library.custom_op("phonlp_amd::pack_scorer", mutates_args=())
def pack_scorer(W1: Tensor, b1: Tensor, W2: Tensor, b2: Tensor, Wbil: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """DeepBiaffineScorer parameters -> (W12 [2KP,D], b12 [2KP], Wbt [O*KP,KP]) in the kernels' layout."""
    for n, t in (("W1", W1), ("b1", b1), ("W2", W2), ("b2", b2), ("Wbil", Wbil)):
        _need(t, torch.float32, n)
    H, D = W1.shape
    O = Wbil.shape[2]
    if Wbil.shape[0] != H + 1 or Wbil.shape[1] != H + 1:
        raise ValueError("W_bilin.weight must be (H+1, H+1, O)")
    KP = kp(H)
    W12 = torch.empty((2 * KP, D), dtype=torch.float32, device=W1.device)
    b12 = torch.empty((2 * KP,), dtype=torch.float32, device=W1.device)
    Wbt = torch.empty((O * KP, KP), dtype=torch.float32, device=W1.device)
    check(lib().phonlp_pack_scorer_linear(_p(W1), _p(b1), _p(W2), _p(b2), H, D, _p(W12), _p(b12), _stream()),
          "phonlp_pack_scorer_linear")
    check(lib().phonlp_pack_bilinear(_p(Wbil), H, O, _p(Wbt), _stream()), "phonlp_pack_bilinear")
    return W12, b12, Wbt


@torch.library.custom_op("phonlp_amd::pos_head", mutates_args=())
def pos_head(enc: Tensor, sentlens: Tensor, Whid: Tensor, bhid: Tensor, Wclf: Tensor, bclf: Tensor,
             emb: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """enc [B,L,E] -> (feat [B,L,E+TE], pos_pred int32 [B,L], upos_logits [B,L,U]).  emb=None (U <= 32): the embedding
    columns hold the tag probabilities themselves, TE = U rounded up to 4."""
    _need(enc, torch.float32, "enc")
    _need(sentlens, torch.int32, "sentlens")
    B, L, E = enc.shape
    H, U = Whid.shape[0], Wclf.shape[0]
    TE = emb.shape[1] if emb is not None else (U + 3) // 4 * 4
    dev = enc.device
    feat = torch.empty((B, L, E + TE), dtype=torch.float32, device=dev)
    pos = torch.empty((B, L), dtype=torch.int32, device=dev)
    logits = torch.empty((B, L, U), dtype=torch.float32, device=dev)
    nb = lib().phonlp_pos_head_ws_bytes(B, L, E, H, U, TE)
    ws = _ws(nb, dev)
    check(lib().phonlp_pos_head(_p(enc), E, _p(sentlens), B, L, E, _p(Whid), _p(bhid), _p(Wclf), _p(bclf), _p(emb),
                                H, U, TE, _p(feat), _p(pos), _p(logits), _p(ws), nb, _stream()), "phonlp_pos_head")
    return feat, pos, logits


@torch.library.custom_op("phonlp_amd::tagger_head", mutates_args=())
def tagger_head(enc: Tensor, sentlens: Tensor, Whn: Tensor, bhn: Tensor, Wclf: Tensor, bclf: Tensor,
                emb: Optional[Tensor], Wner_emb: Tensor, bner: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Fused tagger head (ref annotate_model.py:132-150): enc [B,L,E] -> (pos_pred int32 [B,L], ner_logits [B,L,T],
    upos_logits [B,L,U]); Whn / bhn / Wner_emb as HeadWeights packs them (include/phonlp_heads.h).  emb=None (U <= 32):
    Wner_emb is ner_tag_clf.weight[:, E:] @ emb^T [T, U] (the embedding folded at load), upos_clf runs inside the tail."""
    _need(enc, torch.float32, "enc")
    _need(sentlens, torch.int32, "sentlens")
    B, L, E = enc.shape
    U, T = Wclf.shape[0], Wner_emb.shape[0]
    TE = emb.shape[1] if emb is not None else U
    if emb is None and Wner_emb.shape[1] != U:
        raise ValueError("tagger_head: emb=None needs Wner_emb [T, U]")
    H = Whn.shape[0] - T
    dev = enc.device
    pos = torch.empty((B, L), dtype=torch.int32, device=dev)
    ner = torch.empty((B, L, T), dtype=torch.float32, device=dev)
    logits = torch.empty((B, L, U), dtype=torch.float32, device=dev)
    nb = lib().phonlp_tagger_head_ws_bytes(B, L, E, H, U, TE, T)
    ws = _ws(nb, dev)
    check(lib().phonlp_tagger_head(_p(enc), E, _p(sentlens), B, L, E, _p(Whn), _p(bhn), _p(Wclf), _p(bclf), _p(emb),
                                   _p(Wner_emb), _p(bner), H, U, TE, T, _p(pos), _p(ner), _p(logits), _p(ws), nb,
                                   _stream()), "phonlp_tagger_head")
    return pos, ner, logits


@torch.library.custom_op("phonlp_amd::ner_emission", mutates_args=())
def ner_emission(feat: Tensor, sentlens: Tensor, Wner: Tensor, bner: Tensor) -> Tensor:
    _need(feat, torch.float32, "feat")
    _need(sentlens, torch.int32, "sentlens")
    B, L, D = feat.shape
    T = Wner.shape[0]
    out = torch.empty((B, L, T), dtype=torch.float32, device=feat.device)
    check(lib().phonlp_ner_emission(_p(feat), D, _p(sentlens), B, L, _p(Wner), _p(bner), T, _p(out), _stream()),
          "phonlp_ner_emission")
    return out


@torch.library.custom_op("phonlp_amd::dep_hidden", mutates_args=())
def dep_hidden(feat: Tensor, sentlens: Tensor, Wdep: Tensor, bdep: Tensor) -> Tensor:
    _need(feat, torch.float32, "feat")
    _need(sentlens, torch.int32, "sentlens")
    B, L, D = feat.shape
    out = torch.empty((B, L, D), dtype=torch.float32, device=feat.device)
    check(lib().phonlp_dep_hidden(_p(feat), D, _p(sentlens), B, L, _p(Wdep), _p(bdep), _p(out), _stream()),
          "phonlp_dep_hidden")
    return out


@torch.library.custom_op("phonlp_amd::deep_biaffine_arc", mutates_args=())
def deep_biaffine_arc(hidden: Tensor, W12: Tensor, b12: Tensor, Wbt: Tensor, H: int, n_scorers: int, use_lin: bool,
                      use_dist: bool, want_raw: bool,
                      pad_sentlens: Optional[Tensor] = None, b12_pad: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """hidden [B,L,D] -> (arc log-probs [B,L,L], raw scorer outputs [3,B,L,L] or an empty tensor)."""
    _need(hidden, torch.float32, "hidden")
    B, L, D = hidden.shape
    dev = hidden.device
    out = torch.empty((B, L, L), dtype=torch.float32, device=dev)
    raw = torch.zeros((3, B, L, L), dtype=torch.float32, device=dev) if want_raw else torch.empty(0, device=dev)
    nb = lib().phonlp_biaffine_arc_ws_bytes(B, L, H, n_scorers)
    ws = _ws(nb, dev)
    check(lib().phonlp_deep_biaffine_arc(_p(hidden), D, B, L, H, n_scorers, int(use_lin), int(use_dist), _p(W12),
                                         _p(b12), _p(Wbt), _p(out), _p(raw) if want_raw else None, _p(ws), nb,
                                         _stream(), _p(pad_sentlens), _p(b12_pad)), "phonlp_deep_biaffine_arc")
    return out, raw


@torch.library.custom_op("phonlp_amd::deep_biaffine_label", mutates_args=())
def deep_biaffine_label(hidden: Tensor, W12: Tensor, b12: Tensor, Wbt: Tensor, H: int, O: int,
                        want_scores: bool, pad_sentlens: Optional[Tensor] = None,
                        b12_pad: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """hidden [B,L,D] -> (label arg-max int32 [B,L,L], scores [B,L,O,L] or an empty tensor)."""
    _need(hidden, torch.float32, "hidden")
    B, L, D = hidden.shape
    dev = hidden.device
    out = torch.empty((B, L, L), dtype=torch.int32, device=dev)
    scores = torch.empty((B, L, O, L), dtype=torch.float32, device=dev) if want_scores else torch.empty(0, device=dev)
    nb = lib().phonlp_biaffine_label_ws_bytes(B, L, H, O)
    ws = _ws(nb, dev)
    check(lib().phonlp_deep_biaffine_label(_p(hidden), D, B, L, H, O, _p(W12), _p(b12), _p(Wbt), _p(out),
                                           _p(scores) if want_scores else None, _p(ws), nb, _stream(),
                                           _p(pad_sentlens), _p(b12_pad)), "phonlp_deep_biaffine_label")
    return out, scores


@torch.library.custom_op("phonlp_amd::crf_viterbi", mutates_args=())
def crf_viterbi(logits: Tensor, trans: Tensor, sentlens: Tensor) -> Tuple[Tensor, Tensor]:
    """logits [B,L,T], trans [T,T] (prev, cur), sentlens int32 [B] -> (tags int32 [B,L], best score [B])."""
    _need(logits, torch.float32, "logits")
    _need(trans, torch.float32, "trans")
    _need(sentlens, torch.int32, "sentlens")
    B, L, T = logits.shape
    tags = torch.empty((B, L), dtype=torch.int32, device=logits.device)
    best = torch.empty((B,), dtype=torch.float32, device=logits.device)
    check(lib().phonlp_crf_viterbi(_p(logits), _p(trans), _p(sentlens), B, L, T, _p(tags), _p(best), _stream()),
          "phonlp_crf_viterbi")
    return tags, best


@torch.library.custom_op("phonlp_amd::crf_log_partition", mutates_args=())
def crf_log_partition(logits: Tensor, trans: Tensor, pad_mask: Optional[Tensor], sentlens: Optional[Tensor],
                      gold: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """-> (log_norm [B], unary [B], binary [B]); unary/binary are zeros unless gold tags (int32) are given."""
    _need(logits, torch.float32, "logits")
    _need(trans, torch.float32, "trans")
    B, L, T = logits.shape
    dev = logits.device
    if pad_mask is not None:
        _need(pad_mask, torch.uint8, "pad_mask")
    if sentlens is not None:
        _need(sentlens, torch.int32, "sentlens")
    if gold is not None:
        _need(gold, torch.int32, "gold")
    log_norm = torch.empty((B,), dtype=torch.float32, device=dev)
    unary = torch.zeros((B,), dtype=torch.float32, device=dev)
    binary = torch.zeros((B,), dtype=torch.float32, device=dev)
    check(lib().phonlp_crf_log_partition(_p(logits), _p(trans), _p(pad_mask), _p(sentlens), _p(gold), B, L, T,
                                         _p(log_norm), _p(unary), _p(binary), _stream()), "phonlp_crf_log_partition")
    return log_norm, unary, binary


@torch.library.custom_op("phonlp_amd::mst_one_root", mutates_args=())
def mst_one_root(arc: Tensor, sentlens: Tensor) -> Tuple[Tensor, Tensor]:
    """arc [B,L,L] log-probs (rows = dependents), sentlens int32 [B] -> (heads int32 [B,L], status int32 [B])."""
    _need(arc, torch.float32, "arc")
    _need(sentlens, torch.int32, "sentlens")
    B, L, _ = arc.shape
    dev = arc.device
    heads = torch.empty((B, L), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    nb = lib().phonlp_mst_ws_bytes(B, L)
    ws = _ws(nb, dev)
    check(lib().phonlp_mst_one_root(_p(arc), _p(sentlens), B, L, _p(heads), _p(status), _p(ws), nb, _stream()),
          "phonlp_mst_one_root")
    return heads, status


def set_mst_fast_path(on: bool) -> None:
    """A/B and test knob: False decodes every sentence with the reference-order kernel (include/phonlp_heads.h)."""
    check(lib().phonlp_set_mst_fast_path(int(bool(on))), "set_mst_fast_path")


def get_mst_fast_path() -> bool:
    return bool(lib().phonlp_get_mst_fast_path())


def mst_one_root_counted(arc: Tensor, sentlens: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """mst_one_root plus the number of sentences the round-based decode handed to the reference-order kernel
    (int32 scalar tensor on the device; -1 when the round-based path does not apply to this L)."""
    _need(arc, torch.float32, "arc")
    _need(sentlens, torch.int32, "sentlens")
    B, L, _ = arc.shape
    dev = arc.device
    heads = torch.empty((B, L), dtype=torch.int32, device=dev)
    status = torch.empty((B,), dtype=torch.int32, device=dev)
    nb = lib().phonlp_mst_ws_bytes(B, L)
    ws = _ws(nb, dev)
    check(lib().phonlp_mst_one_root(_p(arc), _p(sentlens), B, L, _p(heads), _p(status), _p(ws), nb, _stream()),
          "phonlp_mst_one_root")
    off = lib().phonlp_mst_fallback_counter_offset(B, L)
    if off < 0 or not get_mst_fast_path():
        return heads, status, torch.full((), -1, dtype=torch.int32, device=dev)
    mst_one_root_counted.last_list = ws[lib().phonlp_mst_fallback_list_offset(B, L):][:4 * B].view(torch.int32)
    return heads, status, ws[off:off + 4].view(torch.int32)[0].clone()


@torch.library.custom_op("phonlp_amd::label_project", mutates_args=())
def label_project(hidden: Tensor, W12: Tensor, b12: Tensor, Wbt: Tensor, H: int, O: int,
                  pad_sentlens: Optional[Tensor] = None, b12_pad: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """hidden [B,L,D] -> (XF [B*L, 2*KP], T [B*L, O*KP]): the label scorer's two GEMMs (include/phonlp_heads.h).
    pad_sentlens / b12_pad select the folded form (rows before dep_hid, folded weights)."""
    _need(hidden, torch.float32, "hidden")
    B, L, D = hidden.shape
    dev = hidden.device
    KP = kp(H)
    XF = torch.empty((B * L, 2 * KP), dtype=torch.float32, device=dev)
    T = torch.empty((B * L, O * KP), dtype=torch.float32, device=dev)
    check(lib().phonlp_label_project(_p(hidden), D, B, L, H, O, _p(W12), _p(b12), _p(Wbt), _p(XF), _p(T), _stream(),
                                     _p(pad_sentlens), _p(b12_pad)), "phonlp_label_project")
    return XF, T


@torch.library.custom_op("phonlp_amd::label_at_heads", mutates_args=())
def label_at_heads(T: Tensor, XF: Tensor, heads: Tensor, sentlens: Tensor, H: int, O: int) -> Tuple[Tensor, Tensor]:
    """-> (rels int32 [B,L], winning score fp32 [B,L]) at j = heads[b,i]; 0 outside 1 <= i < len."""
    _need(T, torch.float32, "T")
    _need(XF, torch.float32, "XF")
    _need(heads, torch.int32, "heads")
    _need(sentlens, torch.int32, "sentlens")
    B, L = heads.shape
    rels = torch.empty((B, L), dtype=torch.int32, device=T.device)
    best = torch.empty((B, L), dtype=torch.float32, device=T.device)
    check(lib().phonlp_label_at_heads(_p(T), _p(XF), _p(heads), _p(sentlens), B, L, H, O, _p(rels), _p(best),
                                      _stream()), "phonlp_label_at_heads")
    return rels, best


@torch.library.custom_op("phonlp_amd::dep_loss", mutates_args=())
def dep_loss(arc_logp: Tensor, raw: Optional[Tensor], T: Tensor, XF: Tensor, gold_head: Tensor, gold_rel: Tensor,
             sentlens: Tensor, H: int, O: int, use_lin: bool, use_dist: bool, number_of_words: float) -> Tensor:
    """Parser training loss, forward value (ref models/jointmodel/model.py:209-236, dropout off) -> fp32 [5] on the
    device: arc CE, label CE, linearization CE, distance term (sums over words), total / number_of_words.
    gold_head / gold_rel: int32 [B,L] with column 0 = ROOT unused (include/phonlp_heads.h)."""
    _need(arc_logp, torch.float32, "arc_logp")
    _need(T, torch.float32, "T")
    _need(XF, torch.float32, "XF")
    _need(gold_head, torch.int32, "gold_head")
    _need(gold_rel, torch.int32, "gold_rel")
    _need(sentlens, torch.int32, "sentlens")
    if raw is not None:
        _need(raw, torch.float32, "raw")
    B, L = gold_head.shape
    if tuple(arc_logp.shape) != (B, L, L) or tuple(gold_rel.shape) != (B, L) or sentlens.numel() != B:
        raise ValueError("dep_loss: arc_logp [B,L,L], gold_head/gold_rel [B,L], sentlens [B] expected")
    if raw is not None and tuple(raw.shape) != (3, B, L, L):
        raise ValueError("dep_loss: raw must be [3,B,L,L]")
    KP = kp(H)
    if T.numel() != B * L * O * KP or XF.numel() != B * L * 2 * KP:
        raise ValueError("dep_loss: T [B*L, O*KP] and XF [B*L, 2*KP] of label_project expected")
    dev = arc_logp.device
    nb = lib().phonlp_dep_loss_ws_bytes(B, L)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    loss = torch.empty((5,), dtype=torch.float32, device=dev)
    check(lib().phonlp_dep_loss(_p(arc_logp), _p(raw), _p(T), _p(XF), _p(gold_head), _p(gold_rel), _p(sentlens), B, L,
                                H, O, int(use_lin), int(use_dist), float(number_of_words), _p(loss), _p(ws), nb,
                                _stream()), "phonlp_dep_loss")
    return loss


def dep_loss_grad(arc_logp: Tensor, raw: Optional[Tensor], T: Tensor, XF: Tensor, gold_head: Tensor, gold_rel: Tensor,
                  sentlens: Tensor, H: int, O: int, use_lin: bool, use_dist: bool,
                  number_of_words: float) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """Backward of dep_loss w.r.t. the four scorers' outputs (autograd of ref models/jointmodel/model.py:209-236):
    -> (d_arc [B,L,L], d_label [B,L,O] at the gold head, d_lin [B,L], d_dist [B,L] at the gold head); include/phonlp_heads.h."""
    _need(arc_logp, torch.float32, "arc_logp")
    _need(T, torch.float32, "T")
    _need(XF, torch.float32, "XF")
    _need(gold_head, torch.int32, "gold_head")
    _need(gold_rel, torch.int32, "gold_rel")
    _need(sentlens, torch.int32, "sentlens")
    if raw is not None:
        _need(raw, torch.float32, "raw")
    B, L = gold_head.shape
    if tuple(arc_logp.shape) != (B, L, L) or tuple(gold_rel.shape) != (B, L) or sentlens.numel() != B:
        raise ValueError("dep_loss_grad: arc_logp [B,L,L], gold_head/gold_rel [B,L], sentlens [B] expected")
    if raw is not None and tuple(raw.shape) != (3, B, L, L):
        raise ValueError("dep_loss_grad: raw must be [3,B,L,L]")
    KP = kp(H)
    if T.numel() != B * L * O * KP or XF.numel() != B * L * 2 * KP:
        raise ValueError("dep_loss_grad: T [B*L, O*KP] and XF [B*L, 2*KP] of label_project expected")
    dev = arc_logp.device
    d_arc = torch.empty((B, L, L), dtype=torch.float32, device=dev)
    d_label = torch.empty((B, L, O), dtype=torch.float32, device=dev)
    d_lin = torch.empty((B, L), dtype=torch.float32, device=dev)
    d_dist = torch.empty((B, L), dtype=torch.float32, device=dev)
    check(lib().phonlp_dep_loss_grad(_p(arc_logp), _p(raw), _p(T), _p(XF), _p(gold_head), _p(gold_rel), _p(sentlens), B,
                                     L, H, O, int(use_lin), int(use_dist), float(number_of_words), _p(d_arc), _p(d_label),
                                     _p(d_lin), _p(d_dist), _stream()), "phonlp_dep_loss_grad")
    return d_arc, d_label, d_lin, d_dist


def linear_backward(X: Tensor, W: Tensor, Yact: Optional[Tensor], dY: Tensor, relu: bool, need_dX: bool = True,
                    need_dW: bool = True, need_db: bool = True):
    """Backward of Y = act(X W^T + b) (an nn.Linear, optionally followed by ReLU): X [M,K], W [N,K], Yact = the layer's
    output [M,N] (only read when relu), dY [M,N] -> (dX [M,K], dW [N,K], db [N]), None where not asked for."""
    _need(X, torch.float32, "X")
    _need(W, torch.float32, "W")
    _need(dY, torch.float32, "dY")
    if X.dim() != 2 or W.dim() != 2 or dY.dim() != 2 or X.shape[1] != W.shape[1] or tuple(dY.shape) != (X.shape[0], W.shape[0]):
        raise ValueError("linear_backward: X [M,K], W [N,K], dY [M,N] expected")
    if relu:
        if Yact is None or tuple(Yact.shape) != tuple(dY.shape):
            raise ValueError("linear_backward: relu needs the layer's output Yact [M,N]")
        _need(Yact, torch.float32, "Yact")
    M, K = X.shape
    N = W.shape[0]
    dev = X.device
    dX = torch.empty((M, K), dtype=torch.float32, device=dev) if need_dX else None
    dW = torch.empty((N, K), dtype=torch.float32, device=dev) if need_dW else None
    db = torch.empty((N,), dtype=torch.float32, device=dev) if need_db else None
    nb = lib().phonlp_linear_backward_ws_bytes(M, N, K)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    check(lib().phonlp_linear_backward(_p(X), K, _p(W), K, _p(Yact) if relu else None, N, _p(dY), N, M, N, K, int(relu),
                                       _p(dX), K, _p(dW), K, _p(db), _p(ws), nb, _stream()), "phonlp_linear_backward")
    return dX, dW, db


def bilinear_backward(XF: Tensor, Wbt: Tensor, G: Tensor, H: int) -> Tuple[Tensor, Tensor]:
    """Backward of a one-output bilinear scorer S[b,i,j] = X1'[b,i] Wb X2'[b,j] given G = dL/dS [B,L,L]: XF [B*L, 2*KP]
    the scorer's post-ReLU features, Wbt [KP, KP] the packed weight -> (dXF [B*L, 2*KP], dWbt [KP, KP])."""
    _need(XF, torch.float32, "XF")
    _need(Wbt, torch.float32, "Wbt")
    _need(G, torch.float32, "G")
    KP = kp(H)
    if G.dim() != 3 or G.shape[1] != G.shape[2]:
        raise ValueError("bilinear_backward: G must be [B, L, L]")
    B, L = G.shape[0], G.shape[1]
    if tuple(XF.shape) != (B * L, 2 * KP) or tuple(Wbt.shape) != (KP, KP):
        raise ValueError("bilinear_backward: XF [B*L, 2*KP] and Wbt [KP, KP] expected")
    dev = XF.device
    dXF = torch.empty((B * L, 2 * KP), dtype=torch.float32, device=dev)
    dWbt = torch.empty((KP, KP), dtype=torch.float32, device=dev)
    nb = lib().phonlp_bilinear_backward_ws_bytes(B, L, H)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    check(lib().phonlp_bilinear_backward(_p(XF), _p(Wbt), _p(G), B, L, H, _p(dXF), _p(dWbt), _p(ws), nb, _stream()),
          "phonlp_bilinear_backward")
    return dXF, dWbt


def softmax_rows(Z: Tensor) -> Tensor:
    """softmax over the last dimension of a [M, U] device tensor (the soft POS distribution)."""
    _need(Z, torch.float32, "Z")
    M, U = Z.shape
    P = torch.empty_like(Z)
    check(lib().phonlp_softmax_rows(_p(Z), M, U, _p(P), _stream()), "phonlp_softmax_rows")
    return P


def softmax_rows_backward(P: Tensor, dP: Tensor) -> Tensor:
    """dZ = P * (dP - sum_u P dP) for P = softmax_rows(Z)."""
    _need(P, torch.float32, "P")
    _need(dP, torch.float32, "dP")
    if P.shape != dP.shape or P.dim() != 2:
        raise ValueError("softmax_rows_backward: P, dP [M, U] expected")
    dZ = torch.empty_like(P)
    check(lib().phonlp_softmax_rows_backward(_p(P), _p(dP), P.shape[0], P.shape[1], _p(dZ), _stream()),
          "phonlp_softmax_rows_backward")
    return dZ


def label_backward(XF: Tensor, Wbt: Tensor, g: Tensor, gold_head: Tensor, sentlens: Tensor, H: int,
                   O: int, T: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """Backward of the label scorer's scores at the gold heads: g [B,L,O] (dep_loss_grad's d_label), XF [B*L, 2*KP], Wbt
    [O*KP, KP], T = label_project's second output when kept (else recomputed) -> (dXF [B*L, 2*KP], dWbt [O*KP, KP])."""
    _need(XF, torch.float32, "XF")
    _need(Wbt, torch.float32, "Wbt")
    _need(g, torch.float32, "g")
    _need(gold_head, torch.int32, "gold_head")
    _need(sentlens, torch.int32, "sentlens")
    KP = kp(H)
    B, L = gold_head.shape
    if tuple(g.shape) != (B, L, O) or tuple(XF.shape) != (B * L, 2 * KP) or tuple(Wbt.shape) != (O * KP, KP) or sentlens.numel() != B:
        raise ValueError("label_backward: g [B,L,O], gold_head [B,L], XF [B*L, 2*KP], Wbt [O*KP, KP], sentlens [B] expected")
    dev = XF.device
    dXF = torch.empty((B * L, 2 * KP), dtype=torch.float32, device=dev)
    dWbt = torch.empty((O * KP, KP), dtype=torch.float32, device=dev)
    nb = lib().phonlp_label_backward_ws_bytes(B, L, H, O)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    if T is not None:
        _need(T, torch.float32, "T")
        if T.numel() != B * L * O * KP:
            raise ValueError("label_backward: T must be [B*L, O*KP]")
    check(lib().phonlp_label_backward(_p(XF), _p(Wbt), _p(T), _p(g), _p(gold_head), _p(sentlens), B, L, H, O, _p(dXF),
                                      _p(dWbt), _p(ws), nb, _stream()), "phonlp_label_backward")
    return dXF, dWbt


@dep_loss.register_fake
def _(arc_logp, raw, T, XF, gold_head, gold_rel, sentlens, H, O, use_lin, use_dist, number_of_words):
    return arc_logp.new_empty((5,))


@torch.library.custom_op("phonlp_amd::pos_loss", mutates_args=())
def pos_loss(upos_logits: Tensor, gold_upos: Tensor, sentlens: Tensor, ignore_index: int) -> Tensor:
    """POS training loss, forward value (ref models/jointmodel/model.py:128-135) -> fp32 [3] on the device: mean
    cross-entropy over the counted words, their sum, their number."""
    _need(upos_logits, torch.float32, "upos_logits")
    _need(gold_upos, torch.int32, "gold_upos")
    _need(sentlens, torch.int32, "sentlens")
    B, L, U = upos_logits.shape
    if tuple(gold_upos.shape) != (B, L) or sentlens.numel() != B:
        raise ValueError("pos_loss: gold_upos [B,L] and sentlens [B] expected")
    dev = upos_logits.device
    nb = lib().phonlp_pos_loss_ws_bytes(B, L)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    loss = torch.empty((3,), dtype=torch.float32, device=dev)
    check(lib().phonlp_pos_loss(_p(upos_logits), _p(gold_upos), _p(sentlens), B, L, U, int(ignore_index), _p(loss),
                                _p(ws), nb, _stream()), "phonlp_pos_loss")
    return loss


@pos_loss.register_fake
def _(upos_logits, gold_upos, sentlens, ignore_index):
    return upos_logits.new_empty((3,))


@torch.library.custom_op("phonlp_amd::crf_nll_grad", mutates_args=())
def crf_nll_grad(logits: Tensor, trans: Tensor, pad_mask: Optional[Tensor], sentlens: Optional[Tensor], gold: Tensor,
                 scale: float) -> Tuple[Tensor, Tensor]:
    """Gradient of the CRF NLL (ref crf.py:24-100 under autograd) -> (d_logits [B,L,T], d_trans [T,T]); scale = 1/B
    for the reference's batch average."""
    _need(logits, torch.float32, "logits")
    _need(trans, torch.float32, "trans")
    _need(gold, torch.int32, "gold")
    if pad_mask is not None:
        _need(pad_mask, torch.uint8, "pad_mask")
    if sentlens is not None:
        _need(sentlens, torch.int32, "sentlens")
    B, L, T = logits.shape
    if tuple(trans.shape) != (T, T) or tuple(gold.shape) != (B, L):
        raise ValueError("crf_nll_grad: trans [T,T] and gold [B,L] expected")
    dev = logits.device
    nb = lib().phonlp_crf_nll_grad_ws_bytes(B, T)
    ws = torch.empty((nb,), dtype=torch.uint8, device=dev)
    d_logits = torch.empty_like(logits)
    d_trans = torch.empty_like(trans)
    check(lib().phonlp_crf_nll_grad(_p(logits), _p(trans), _p(pad_mask), _p(sentlens), _p(gold), B, L, T, float(scale),
                                    _p(d_logits), _p(d_trans), _p(ws), nb, _stream()), "phonlp_crf_nll_grad")
    return d_logits, d_trans


@crf_nll_grad.register_fake
def _(logits, trans, pad_mask, sentlens, gold, scale):
    return torch.empty_like(logits), torch.empty_like(trans)


@torch.library.custom_op("phonlp_amd::pos_loss_grad", mutates_args=())
def pos_loss_grad(upos_logits: Tensor, gold_upos: Tensor, sentlens: Tensor, loss: Tensor, ignore_index: int) -> Tensor:
    """d(mean cross-entropy)/d(logits); `loss` is the fp32 [3] result of pos_loss (its count stays on the device)."""
    _need(upos_logits, torch.float32, "upos_logits")
    _need(gold_upos, torch.int32, "gold_upos")
    _need(sentlens, torch.int32, "sentlens")
    _need(loss, torch.float32, "loss")
    B, L, U = upos_logits.shape
    if tuple(gold_upos.shape) != (B, L) or sentlens.numel() != B or loss.numel() != 3:
        raise ValueError("pos_loss_grad: gold_upos [B,L], sentlens [B], loss [3] expected")
    d = torch.empty_like(upos_logits)
    count = ctypes.c_void_p(loss.data_ptr() + 8)
    check(lib().phonlp_pos_loss_grad(_p(upos_logits), _p(gold_upos), _p(sentlens), count, B, L, U, int(ignore_index),
                                     _p(d), _stream()), "phonlp_pos_loss_grad")
    return d


@pos_loss_grad.register_fake
def _(upos_logits, gold_upos, sentlens, loss, ignore_index):
    return torch.empty_like(upos_logits)


@torch.library.custom_op("phonlp_amd::gather_labels", mutates_args=())
def gather_labels(label_argmax: Tensor, heads: Tensor, sentlens: Tensor) -> Tensor:
    _need(label_argmax, torch.int32, "label_argmax")
    _need(heads, torch.int32, "heads")
    _need(sentlens, torch.int32, "sentlens")
    B, L = heads.shape
    rels = torch.empty((B, L), dtype=torch.int32, device=heads.device)
    check(lib().phonlp_gather_labels(_p(label_argmax), _p(heads), _p(sentlens), B, L, _p(rels), _stream()),
          "phonlp_gather_labels")
    return rels


@torch.library.custom_op("phonlp_amd::gather_first_subword", mutates_args=())
def gather_first_subword(enc: Tensor, first_subword: Tensor, sentlens: Optional[Tensor]) -> Tensor:
    """enc [B,S,E] (encoder last hidden state), first_subword [B,L] int32/int64 -> [B,L,E] word rows."""
    _need(enc, torch.float32, "enc")
    if first_subword.dtype not in (torch.int32, torch.int64) or not first_subword.is_cuda or not first_subword.is_contiguous():
        raise ValueError("first_subword: expected a contiguous int32/int64 device tensor")
    B, S, E = enc.shape
    L = first_subword.shape[1]
    out = torch.empty((B, L, E), dtype=torch.float32, device=enc.device)
    check(lib().phonlp_gather_first_subword(_p(enc), _p(first_subword), int(first_subword.dtype == torch.int64),
                                            _p(sentlens), B, S, L, E, _p(out), _stream()), "phonlp_gather_first_subword")
    return out


@torch.library.custom_op("phonlp_amd::eval_counts", mutates_args=())
def eval_counts(pos: Tensor, ner: Tensor, heads: Tensor, rels: Tensor, gold_pos: Tensor, gold_ner: Tensor,
                gold_head: Tensor, gold_rel: Tensor, lens_tag: Tensor, lens_dep: Tensor,
                mst_status: Optional[Tensor]) -> Tensor:
    """-> int64[8]: sentences, words, pos_correct, ner_correct, uas_correct, las_correct, mst_failures, 0."""
    for n, t in (("pos", pos), ("ner", ner), ("heads", heads), ("rels", rels), ("gold_pos", gold_pos),
                 ("gold_ner", gold_ner), ("gold_head", gold_head), ("gold_rel", gold_rel), ("lens_tag", lens_tag),
                 ("lens_dep", lens_dep)):
        _need(t, torch.int32, n)
    B, Lt = pos.shape
    Ld = heads.shape[1]
    counts = torch.zeros(8, dtype=torch.int64, device=pos.device)
    check(lib().phonlp_eval_counts(_p(pos), _p(ner), _p(heads), _p(rels), _p(gold_pos), _p(gold_ner), _p(gold_head),
                                   _p(gold_rel), _p(lens_tag), _p(lens_dep), _p(mst_status), B, Lt, Ld, _p(counts),
                                   _stream()), "phonlp_eval_counts")
    return counts


@torch.library.custom_op("phonlp_amd::ner_entity_counts", mutates_args=())
def ner_entity_counts(pred: Tensor, gold: Tensor, sentlens: Tensor, tag_kind: Tensor, tag_type: Tensor) -> Tensor:
    """Entity-level NER counts int64[3] = (correct, predicted, gold) -- ref models/ner/scorer.py score_by_entity.
    pred/gold int32 [B,L] tag ids; tag_kind/tag_type int32 [T] from sharding.tag_tables()."""
    for t, n in ((pred, "pred"), (gold, "gold"), (sentlens, "sentlens"), (tag_kind, "tag_kind"), (tag_type, "tag_type")):
        _need(t, torch.int32, n)
    B, L = pred.shape
    counts = torch.zeros(3, dtype=torch.int64, device=pred.device)
    check(lib().phonlp_ner_entity_counts(_p(pred), _p(gold), _p(sentlens), _p(tag_kind), _p(tag_type), B, L,
                                         tag_kind.numel(), _p(counts), _stream()), "phonlp_ner_entity_counts")
    return counts


@ner_entity_counts.register_fake
def _(pred, gold, sentlens, tag_kind, tag_type):
    return torch.empty(3, dtype=torch.int64, device=pred.device)


@torch.library.custom_op("phonlp_amd::split_rows", mutates_args=())
def split_rows(x: Tensor) -> Tuple[Tensor, Tensor]:
    """x [rows, K] fp32 (row stride free, unit column stride) -> engine-2 split image (planes uint8 [rows, 4*K16],
    inv_scale fp32 [rows]); include/phonlp_heads.h phonlp_split_rows."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("split_rows: expected a 2-d fp32 ROCm tensor with unit column stride")
    rows, K = x.shape
    nb = lib().phonlp_split_bytes(rows, K)
    planes = torch.empty((rows, nb // rows), dtype=torch.uint8, device=x.device)
    inv = torch.empty((rows,), dtype=torch.float32, device=x.device)
    check(lib().phonlp_split_rows(_p(x), x.stride(0), rows, K, _p(planes), _p(inv), _stream()), "phonlp_split_rows")
    return planes, inv


@torch.library.custom_op("phonlp_amd::linear_split", mutates_args=())
def linear_split(xp: Tensor, x_inv: Tensor, wp: Tensor, w_inv: Tensor, bias: Optional[Tensor], K: int, relu: bool,
                 sentlens: Optional[Tensor], L: int, pad_bias: Optional[Tensor], kind: int) -> Tensor:
    """Engine-2 dense layer y = act(x @ w.T + bias) on split images (split_rows); K = the unsplit inner size."""
    _need(xp, torch.uint8, "xp")
    _need(wp, torch.uint8, "wp")
    _need(x_inv, torch.float32, "x_inv")
    _need(w_inv, torch.float32, "w_inv")
    M, N = xp.shape[0], wp.shape[0]
    if xp.shape[1] != wp.shape[1] or xp.shape[1] != lib().phonlp_split_bytes(1, K):
        raise ValueError("linear_split: split images do not match K")
    y = torch.empty((M, N), dtype=torch.float32, device=xp.device)
    check(lib().phonlp_linear_split(_p(xp), _p(x_inv), _p(wp), _p(w_inv), _p(bias), _p(y), N, M, N, K, int(relu),
                                    _p(sentlens), L, _p(pad_bias), kind, _stream()), "phonlp_linear_split")
    return y


def arc_kc(H: int) -> int:
    """Columns per segment of the first-layer image the fused arc kernel reads (0: H + 1 > 416, not supported)."""
    return int(lib().phonlp_arc_kc(int(H)))


@torch.library.custom_op("phonlp_amd::linear_split_image", mutates_args=())
def linear_split_image(xp: Tensor, x_inv: Tensor, wp: Tensor, w_inv: Tensor, bias: Optional[Tensor], K: int, relu: bool,
                       sentlens: Optional[Tensor], L: int, pad_bias: Optional[Tensor], bound_c: float, bound_b: float,
                       kind: int) -> Tuple[Tensor, Tensor]:
    """linear_split with the result written as a split image (uint8 [M, 4*N], N % 16 == 0) + its inverse row scales [M];
    bound_c / bound_b: include/phonlp_heads.h phonlp_linear_split_image."""
    _need(xp, torch.uint8, "xp")
    _need(wp, torch.uint8, "wp")
    _need(x_inv, torch.float32, "x_inv")
    _need(w_inv, torch.float32, "w_inv")
    M, N = xp.shape[0], wp.shape[0]
    if xp.shape[1] != wp.shape[1] or xp.shape[1] != lib().phonlp_split_bytes(1, K) or N % 16:
        raise ValueError("linear_split_image: split images do not match K, or N % 16 != 0")
    y = torch.empty((M, 4 * N), dtype=torch.uint8, device=xp.device)
    y_inv = torch.empty((M,), dtype=torch.float32, device=xp.device)
    check(lib().phonlp_linear_split_image(_p(xp), _p(x_inv), _p(wp), _p(w_inv), _p(bias), _p(y), _p(y_inv), M, N, K,
                                          int(relu), _p(sentlens), L, _p(pad_bias), float(bound_c), float(bound_b), kind,
                                          _stream()), "phonlp_linear_split_image")
    return y, y_inv


@linear_split_image.register_fake
def _(xp, x_inv, wp, w_inv, bias, K, relu, sentlens, L, pad_bias, bound_c, bound_b, kind):
    return (torch.empty((xp.shape[0], 4 * wp.shape[0]), dtype=torch.uint8, device=xp.device),
            torch.empty((xp.shape[0],), dtype=torch.float32, device=xp.device))


@torch.library.custom_op("phonlp_amd::arc_scores_split", mutates_args=())
def arc_scores_split(xfp: Tensor, xf_inv: Tensor, B: int, L: int, H: int, n_scorers: int, use_lin: bool, use_dist: bool,
                     wbp: Tensor, wb_inv: Tensor, off: Optional[Tensor], sentlens: Optional[Tensor],
                     pad_row: Optional[Tensor], pad_inv: float) -> Tensor:
    """Engine 2: stages A and B of the arc scorers fused, from the first-layer IMAGE (linear_split_image) -> arc
    log-probs [B,L,L].  Packed rows: off / sentlens / pad_row (image row) / pad_inv; padded dependents' rows are NOT
    written.  include/phonlp_heads.h phonlp_arc_scores_split_ws (the workspace form: per-scorer workgroups at small B)."""
    _need(xfp, torch.uint8, "xfp")
    _need(wbp, torch.uint8, "wbp")
    _need(xf_inv, torch.float32, "xf_inv")
    _need(wb_inv, torch.float32, "wb_inv")
    KC = arc_kc(H)
    if KC == 0 or xfp.shape[1] != n_scorers * 2 * KC * 4 or tuple(wbp.shape) != (n_scorers * KC, KC * 4):
        raise ValueError("arc_scores_split: images do not match H / n_scorers")
    out = torch.empty((B, L, L), dtype=torch.float32, device=xfp.device)
    nb = int(lib().phonlp_arc_scores_split_ws_bytes(B, L, n_scorers))  # > 0: few short sentences, a workgroup per scorer
    ws = _ws(nb, xfp.device) if nb else None
    check(lib().phonlp_arc_scores_split_ws(_p(xfp), _p(xf_inv), xfp.shape[0], B, L, H, n_scorers, int(use_lin),
                                           int(use_dist), _p(wbp), _p(wb_inv), _p(off), _p(sentlens), _p(pad_row),
                                           float(pad_inv), _p(out), _p(ws), nb, _stream()), "phonlp_arc_scores_split_ws")
    return out


@arc_scores_split.register_fake
def _(xfp, xf_inv, B, L, H, n_scorers, use_lin, use_dist, wbp, wb_inv, off, sentlens, pad_row, pad_inv):
    return torch.empty((B, L, L), dtype=torch.float32, device=xfp.device)


@torch.library.custom_op("phonlp_amd::biaffine_arc_scores", mutates_args=())
def biaffine_arc_scores(XF: Tensor, B: int, L: int, Wbt: Tensor, H: int, n_scorers: int, use_lin: bool,
                        use_dist: bool) -> Tensor:
    """XF [B*L, n_scorers*2*KP] (the arc scorers' first-layer features) -> arc log-probs [B,L,L]."""
    _need(XF, torch.float32, "XF")
    dev = XF.device
    out = torch.empty((B, L, L), dtype=torch.float32, device=dev)
    nb = lib().phonlp_biaffine_arc_scores_ws_bytes(B, L, H, n_scorers)
    ws = _ws(nb, dev)
    check(lib().phonlp_biaffine_arc_scores(_p(XF), B, L, H, n_scorers, int(use_lin), int(use_dist), _p(Wbt), _p(out),
                                           None, _p(ws), nb, _stream()), "phonlp_biaffine_arc_scores")
    return out


def biaffine_arc_scores_raw(XF: Tensor, B: int, L: int, Wbt: Tensor, H: int, n_scorers: int, use_lin: bool,
                            use_dist: bool) -> Tuple[Tensor, Tensor]:
    """The same, also returning the raw scorer outputs [3, B, L, L] (what the training loss reads, ref jointmodel/model.py:209-236)."""
    _need(XF, torch.float32, "XF")
    dev = XF.device
    out = torch.empty((B, L, L), dtype=torch.float32, device=dev)
    raw = torch.zeros((3, B, L, L), dtype=torch.float32, device=dev)
    nb = lib().phonlp_biaffine_arc_scores_ws_bytes(B, L, H, n_scorers)
    ws = _ws(nb, dev)
    check(lib().phonlp_biaffine_arc_scores(_p(XF), B, L, H, n_scorers, int(use_lin), int(use_dist), _p(Wbt), _p(out),
                                           _p(raw), _p(ws), nb, _stream()), "phonlp_biaffine_arc_scores")
    return out, raw


@torch.library.custom_op("phonlp_amd::label_stage_a_split", mutates_args=())
def label_stage_a_split(x1p: Tensor, x1_inv: Tensor, wbtp: Tensor, wbt_inv: Tensor, B: int, L: int, H: int,
                        O: int, ones_col: Optional[Tensor] = None) -> Tensor:
    """Engine-2 stage A of the label scorer -> T4 (fp32, 'rows interleaved by 4': include/phonlp_heads.h).
    ones_col: column H of the packed bilinear weight ([O*KP], HeadWeights.lab_ones_col) or None."""
    _need(x1p, torch.uint8, "x1p")
    _need(wbtp, torch.uint8, "wbtp")
    if x1p.shape[0] != B * L or wbtp.shape[0] != O * kp(H) or x1p.shape[1] != wbtp.shape[1]:
        raise ValueError("label_stage_a_split: split images do not match B, L, H, O")
    T4 = torch.empty(lib().phonlp_label_t4_bytes(B, L, H, O) // 4, dtype=torch.float32, device=x1p.device)
    if ones_col is not None:
        _need(ones_col, torch.float32, "ones_col")
        if ones_col.numel() != O * kp(H):
            raise ValueError("label_stage_a_split: ones_col must hold O*KP floats")
    check(lib().phonlp_label_stage_a_split(_p(x1p), _p(x1_inv), _p(wbtp), _p(wbt_inv),
                                           _p(ones_col) if ones_col is not None else None, B, L, H, O, _p(T4), _stream()),
          "phonlp_label_stage_a_split")
    return T4


@torch.library.custom_op("phonlp_amd::label_at_heads_rows4", mutates_args=())
def label_at_heads_rows4(T4: Tensor, XF: Tensor, heads: Tensor, sentlens: Tensor, H: int, O: int) -> Tuple[Tensor, Tensor]:
    """label_at_heads on the T4 image of label_stage_a_split."""
    _need(T4, torch.float32, "T4")
    _need(XF, torch.float32, "XF")
    _need(heads, torch.int32, "heads")
    _need(sentlens, torch.int32, "sentlens")
    B, L = heads.shape
    rels = torch.empty((B, L), dtype=torch.int32, device=T4.device)
    best = torch.empty((B, L), dtype=torch.float32, device=T4.device)
    check(lib().phonlp_label_at_heads_rows4(_p(T4), _p(XF), _p(heads), _p(sentlens), B, L, H, O, _p(rels), _p(best),
                                            _stream()), "phonlp_label_at_heads_rows4")
    return rels, best


@torch.library.custom_op("phonlp_amd::label_at_heads_fused", mutates_args=())
def label_at_heads_fused(x1p: Tensor, x1_inv: Tensor, wbtcp: Tensor, wbtc_inv: Tensor, XF: Tensor, heads: Tensor,
                         sentlens: Tensor, H: int, O: int) -> Tuple[Tensor, Tensor]:
    """Stage A of the label scorer with the labels-at-heads contraction fused into its epilogue (engine 2):
    -> (rels int32 [B,L], winning score [B,L]).  wbtcp: HeadWeights.lab_Wbtc_s (relation rows padded to 32)."""
    _need(x1p, torch.uint8, "x1p")
    _need(wbtcp, torch.uint8, "wbtcp")
    _need(XF, torch.float32, "XF")
    _need(heads, torch.int32, "heads")
    _need(sentlens, torch.int32, "sentlens")
    B, L = heads.shape
    if x1p.shape[0] != B * L or wbtcp.shape[0] != O * lib().phonlp_label_kpc(H) or x1p.shape[1] != wbtcp.shape[1]:
        raise ValueError("label_at_heads_fused: split images do not match B, L, H, O")
    rels = torch.empty((B, L), dtype=torch.int32, device=XF.device)
    best = torch.empty((B, L), dtype=torch.float32, device=XF.device)
    nb = lib().phonlp_label_at_heads_fused_ws_bytes(B, L, H, O)
    ws = _ws(nb, XF.device)
    check(lib().phonlp_label_at_heads_fused(_p(x1p), _p(x1_inv), _p(wbtcp), _p(wbtc_inv), _p(XF), _p(heads),
                                            _p(sentlens), B, L, H, O, _p(rels), _p(best), _p(ws), nb, _stream()),
          "phonlp_label_at_heads_fused")
    return rels, best


# ---- ragged batches with their live words packed (include/phonlp_heads.h, "packed rows")
def pack_descriptor(sentlens: Tensor, L: int, m_live: int) -> Tuple[Tensor, Tensor]:
    """sentlens int32 [B] (device) -> (off int32 [B+1]: sentence b's first packed row, off[B] = live rows;
    rowmap int32 [m_live]: packed row p is word rowmap[p] = b*L + i).  m_live = sum(min(sentlens, L)), host-known."""
    _need(sentlens, torch.int32, "sentlens")
    B = sentlens.numel()
    off = torch.empty(B + 1, dtype=torch.int32, device=sentlens.device)
    rowmap = torch.empty(max(int(m_live), 1), dtype=torch.int32, device=sentlens.device)
    check(lib().phonlp_pack_descriptor(_p(sentlens), B, int(L), _p(off), _p(rowmap), rowmap.numel(), _stream()),
          "phonlp_pack_descriptor")
    return off, rowmap


def pack_rows(x: Tensor, rowmap: Tensor) -> Tensor:
    """x fp32 [rows, W] (unit column stride) -> [len(rowmap), W]: the rows rowmap names, in order."""
    _need(rowmap, torch.int32, "rowmap")
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("pack_rows: expected a 2-d fp32 ROCm tensor with unit column stride")
    out = torch.empty((rowmap.numel(), x.shape[1]), dtype=torch.float32, device=x.device)
    check(lib().phonlp_pack_rows(_p(x), x.stride(0), x.shape[0], _p(rowmap), rowmap.numel(), x.shape[1], _p(out), _stream()),
          "phonlp_pack_rows")
    return out


def unpack_rows(x: Tensor, rowmap: Tensor, n_rows: int) -> Tensor:
    """x [len(rowmap), W] or [len(rowmap)] (fp32 / int32) -> [n_rows, W] / [n_rows]: row p at rowmap[p], zeros elsewhere."""
    _need(rowmap, torch.int32, "rowmap")
    if not x.is_cuda or x.dtype not in (torch.float32, torch.int32) or not x.is_contiguous():
        raise ValueError("unpack_rows: expected a contiguous fp32 / int32 ROCm tensor")
    m = rowmap.numel()
    W = x.numel() // m
    out = torch.empty((n_rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    check(lib().phonlp_unpack_rows(_p(x), _p(rowmap), m, W, _p(out), int(n_rows), _stream()), "phonlp_unpack_rows")
    return out


def biaffine_arc_scores_packed(XF: Tensor, off: Tensor, sentlens: Tensor, pad_row: Tensor, B: int, L: int, Wbt: Tensor,
                               H: int, n_scorers: int, use_lin: bool, use_dist: bool) -> Tensor:
    """biaffine_arc_scores on packed rows: XF [m_live, n_scorers*2*KP]; pad_row [n_scorers*2*KP] = the first layer's
    output for a padded (zero) hidden row.  -> arc log-probs [B,L,L]; the rows of padded dependents are NOT written."""
    _need(XF, torch.float32, "XF")
    _need(off, torch.int32, "off")
    _need(sentlens, torch.int32, "sentlens")
    _need(pad_row, torch.float32, "pad_row")
    if pad_row.numel() != XF.shape[1]:
        raise ValueError("biaffine_arc_scores_packed: pad_row does not match XF")
    dev = XF.device
    out = torch.empty((B, L, L), dtype=torch.float32, device=dev)
    nb = lib().phonlp_biaffine_arc_scores_ws_bytes(1, XF.shape[0], H, n_scorers)
    ws = _ws(nb, dev)
    check(lib().phonlp_biaffine_arc_scores_packed(_p(XF), XF.shape[0], _p(off), _p(sentlens), _p(pad_row), B, L, H,
                                                  n_scorers, int(use_lin), int(use_dist), _p(Wbt), _p(out), None, _p(ws),
                                                  nb, _stream()), "phonlp_biaffine_arc_scores_packed")
    return out


def label_at_heads_fused_packed(x1p: Tensor, x1_inv: Tensor, wbtcp: Tensor, wbtc_inv: Tensor, XF: Tensor, heads: Tensor,
                                sentlens: Tensor, rowmap: Tensor, off: Tensor, H: int, O: int) -> Tuple[Tensor, Tensor]:
    """label_at_heads_fused on packed rows (x1p, XF: m_live rows) -> (rels int32 [B,L], winning score [B,L])."""
    _need(x1p, torch.uint8, "x1p")
    _need(wbtcp, torch.uint8, "wbtcp")
    _need(XF, torch.float32, "XF")
    _need(heads, torch.int32, "heads")
    _need(sentlens, torch.int32, "sentlens")
    _need(rowmap, torch.int32, "rowmap")
    _need(off, torch.int32, "off")
    B, L = heads.shape
    m = rowmap.numel()
    if x1p.shape[0] != m or XF.shape[0] != m or wbtcp.shape[0] != O * lib().phonlp_label_kpc(H) or \
            x1p.shape[1] != wbtcp.shape[1]:
        raise ValueError("label_at_heads_fused_packed: split images do not match rowmap, H, O")
    rels = torch.empty((B, L), dtype=torch.int32, device=XF.device)
    best = torch.empty((B, L), dtype=torch.float32, device=XF.device)
    nb = lib().phonlp_label_at_heads_fused_ws_bytes(1, m, H, O)
    ws = _ws(nb, XF.device)
    check(lib().phonlp_label_at_heads_fused_packed(_p(x1p), _p(x1_inv), _p(wbtcp), _p(wbtc_inv), _p(XF), _p(heads),
                                                   _p(sentlens), _p(rowmap), _p(off), m, B, L, H, O, _p(rels), _p(best),
                                                   _p(ws), nb, _stream()), "phonlp_label_at_heads_fused_packed")
    return rels, best


def scatter_rows(src: Tensor, col0: int, dest_row: Tensor, dst: Tensor, status: Optional[Tensor] = None,
                 fail_count: Optional[Tensor] = None) -> None:
    """dst[dest_row[b], :n] = src[b, col0:col0+n] (device-side unsort, include/phonlp_heads.h); in place on dst."""
    _need(src, torch.int32, "src")
    _need(dest_row, torch.int32, "dest_row")
    _need(dst, torch.int32, "dst")
    B, Ls = src.shape
    check(lib().phonlp_scatter_rows(_p(src), B, Ls, int(col0), _p(dest_row), _p(dst), dst.shape[1], dst.shape[0],
                                    _p(status), _p(fail_count), _stream()), "phonlp_scatter_rows")


@torch.library.custom_op("phonlp_amd::eval_counts_into", mutates_args=("counts",))
def eval_counts_into(counts: Tensor, pos: Tensor, ner: Tensor, heads: Tensor, rels: Tensor, gold_pos: Tensor,
                     gold_ner: Tensor, gold_head: Tensor, gold_rel: Tensor, lens_tag: Tensor, lens_dep: Tensor,
                     mst_status: Optional[Tensor]) -> None:
    """eval_counts ADDED to an existing int64[8] device vector (what the C entry point does: no zero-fill, no add)."""
    for n, t in (("pos", pos), ("ner", ner), ("heads", heads), ("rels", rels), ("gold_pos", gold_pos),
                 ("gold_ner", gold_ner), ("gold_head", gold_head), ("gold_rel", gold_rel), ("lens_tag", lens_tag),
                 ("lens_dep", lens_dep)):
        _need(t, torch.int32, n)
    _need(counts, torch.int64, "counts")
    if counts.numel() < 8:
        raise ValueError("eval_counts_into: counts must hold 8 int64")
    B, Lt = pos.shape
    Ld = heads.shape[1]
    check(lib().phonlp_eval_counts(_p(pos), _p(ner), _p(heads), _p(rels), _p(gold_pos), _p(gold_ner), _p(gold_head),
                                   _p(gold_rel), _p(lens_tag), _p(lens_dep), _p(mst_status), B, Lt, Ld, _p(counts),
                                   _stream()), "phonlp_eval_counts")


@torch.library.custom_op("phonlp_amd::ner_entity_counts_into", mutates_args=("counts",))
def ner_entity_counts_into(counts: Tensor, pred: Tensor, gold: Tensor, sentlens: Tensor, tag_kind: Tensor,
                           tag_type: Tensor) -> None:
    """ner_entity_counts ADDED to an existing int64[3] device vector."""
    for t, n in ((pred, "pred"), (gold, "gold"), (sentlens, "sentlens"), (tag_kind, "tag_kind"), (tag_type, "tag_type")):
        _need(t, torch.int32, n)
    _need(counts, torch.int64, "counts")
    if counts.numel() < 3:
        raise ValueError("ner_entity_counts_into: counts must hold 3 int64")
    B, L = pred.shape
    check(lib().phonlp_ner_entity_counts(_p(pred), _p(gold), _p(sentlens), _p(tag_kind), _p(tag_type), B, L,
                                         tag_kind.numel(), _p(counts), _stream()), "phonlp_ner_entity_counts")


def probs_width(U: int) -> int:
    """Columns the tag probabilities take in a feat row when they stand in for the soft POS embedding (emb=None)."""
    return (U + 3) // 4 * 4


@torch.library.custom_op("phonlp_amd::pos_head_split", mutates_args=())
def pos_head_split(enc: Tensor, sentlens: Tensor, Whid: Tensor, bhid: Tensor, Wclf: Tensor, bclf: Tensor,
                   emb: Optional[Tensor], want_feat: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """pos_head whose feat rows come out as an engine-2 split image: -> (planes uint8 [B*L, 4*K16], inv_scale [B*L],
    pos_pred int32 [B,L], upos_logits [B,L,U], feat [B,L,E+TE] or an empty tensor).  emb=None (U <= 32): the embedding
    columns hold the tag probabilities themselves, TE = U rounded up to 4 (upos_clf computed inside the tail kernel)."""
    _need(enc, torch.float32, "enc")
    _need(sentlens, torch.int32, "sentlens")
    B, L, E = enc.shape
    H, U = Whid.shape[0], Wclf.shape[0]
    TE = emb.shape[1] if emb is not None else probs_width(U)
    dev = enc.device
    D = E + TE
    planes = torch.empty((B * L, lib().phonlp_split_bytes(1, D)), dtype=torch.uint8, device=dev)
    inv = torch.empty((B * L,), dtype=torch.float32, device=dev)
    feat = torch.empty((B, L, D), dtype=torch.float32, device=dev) if want_feat else torch.empty(0, device=dev)
    pos = torch.empty((B, L), dtype=torch.int32, device=dev)
    logits = torch.empty((B, L, U), dtype=torch.float32, device=dev)
    nb = lib().phonlp_pos_head_ws_bytes(B, L, E, H, U, TE)
    ws = _ws(nb, dev)
    check(lib().phonlp_pos_head_split(_p(enc), E, _p(sentlens), B, L, E, _p(Whid), _p(bhid), _p(Wclf), _p(bclf), _p(emb),
                                      H, U, TE, _p(feat) if want_feat else None, _p(planes), _p(inv), _p(pos), _p(logits),
                                      _p(ws), nb, _stream()), "phonlp_pos_head_split")
    return planes, inv, pos, logits, feat


# ---- shape functions (torch.library `register_fake`) for the ops above that had none: with them every custom op of
# the head path traces under FakeTensorMode / torch.compile without running a kernel (VERDICT r1, row b).  Each returns
# what the op itself allocates -- same shapes and dtypes; tests/test_fake_ops.py checks them against the real outputs.
_I32, _I64, _U8 = torch.int32, torch.int64, torch.uint8


@linear.register_fake
def _(x, weight, bias, relu, sentlens, L):
    return x.new_empty((x.shape[0], weight.shape[0]))


@pack_scorer.register_fake
def _(W1, b1, W2, b2, Wbil):
    H, D = W1.shape
    KP = kp(H)
    return W1.new_empty((2 * KP, D)), W1.new_empty((2 * KP,)), W1.new_empty((Wbil.shape[2] * KP, KP))


@pos_head.register_fake
def _(enc, sentlens, Whid, bhid, Wclf, bclf, emb):
    B, L, E = enc.shape
    TE = emb.shape[1] if emb is not None else (Wclf.shape[0] + 3) // 4 * 4
    return (enc.new_empty((B, L, E + TE)), enc.new_empty((B, L), dtype=_I32),
            enc.new_empty((B, L, Wclf.shape[0])))


@tagger_head.register_fake
def _(enc, sentlens, Whn, bhn, Wclf, bclf, emb, Wner_emb, bner):
    B, L, _ = enc.shape
    return (enc.new_empty((B, L), dtype=_I32), enc.new_empty((B, L, Wner_emb.shape[0])),
            enc.new_empty((B, L, Wclf.shape[0])))


@ner_emission.register_fake
def _(feat, sentlens, Wner, bner):
    return feat.new_empty((feat.shape[0], feat.shape[1], Wner.shape[0]))


@dep_hidden.register_fake
def _(feat, sentlens, Wdep, bdep):
    return torch.empty_like(feat)


@deep_biaffine_arc.register_fake
def _(hidden, W12, b12, Wbt, H, n_scorers, use_lin, use_dist, want_raw, pad_sentlens=None, b12_pad=None):
    B, L, _ = hidden.shape
    return hidden.new_empty((B, L, L)), hidden.new_empty((3, B, L, L) if want_raw else (0,))


@deep_biaffine_label.register_fake
def _(hidden, W12, b12, Wbt, H, O, want_scores, pad_sentlens=None, b12_pad=None):
    B, L, _ = hidden.shape
    return hidden.new_empty((B, L, L), dtype=_I32), hidden.new_empty((B, L, O, L) if want_scores else (0,))


@crf_viterbi.register_fake
def _(logits, trans, sentlens):
    B, L, _ = logits.shape
    return logits.new_empty((B, L), dtype=_I32), logits.new_empty((B,))


@crf_log_partition.register_fake
def _(logits, trans, pad_mask, sentlens, gold):
    B = logits.shape[0]
    return logits.new_empty((B,)), logits.new_empty((B,)), logits.new_empty((B,))


@mst_one_root.register_fake
def _(arc, sentlens):
    B, L, _ = arc.shape
    return arc.new_empty((B, L), dtype=_I32), arc.new_empty((B,), dtype=_I32)


@label_project.register_fake
def _(hidden, W12, b12, Wbt, H, O, pad_sentlens=None, b12_pad=None):
    B, L, _ = hidden.shape
    KP = kp(H)
    return hidden.new_empty((B * L, 2 * KP)), hidden.new_empty((B * L, O * KP))


def _fake_rels(like, heads):
    B, L = heads.shape
    return like.new_empty((B, L), dtype=_I32), like.new_empty((B, L), dtype=torch.float32)


@label_at_heads.register_fake
def _(T, XF, heads, sentlens, H, O):
    return _fake_rels(T, heads)


@label_at_heads_rows4.register_fake
def _(T4, XF, heads, sentlens, H, O):
    return _fake_rels(T4, heads)


@label_at_heads_fused.register_fake
def _(x1p, x1_inv, wbtcp, wbtc_inv, XF, heads, sentlens, H, O):
    return _fake_rels(XF, heads)


@gather_labels.register_fake
def _(label_argmax, heads, sentlens):
    return torch.empty_like(heads)


@gather_first_subword.register_fake
def _(enc, first_subword, sentlens):
    return enc.new_empty((enc.shape[0], first_subword.shape[1], enc.shape[2]))


@eval_counts.register_fake
def _(pos, ner, heads, rels, gold_pos, gold_ner, gold_head, gold_rel, lens_tag, lens_dep, mst_status):
    return pos.new_empty((8,), dtype=_I64)


@eval_counts_into.register_fake
def _(counts, pos, ner, heads, rels, gold_pos, gold_ner, gold_head, gold_rel, lens_tag, lens_dep, mst_status):
    return None


@ner_entity_counts_into.register_fake
def _(counts, pred, gold, sentlens, tag_kind, tag_type):
    return None


@split_rows.register_fake
def _(x):
    rows, K = x.shape
    return x.new_empty((rows, lib().phonlp_split_bytes(1, K)), dtype=_U8), x.new_empty((rows,))


@linear_split.register_fake
def _(xp, x_inv, wp, w_inv, bias, K, relu, sentlens, L, pad_bias, kind):
    return x_inv.new_empty((xp.shape[0], wp.shape[0]))


@biaffine_arc_scores.register_fake
def _(XF, B, L, Wbt, H, n_scorers, use_lin, use_dist):
    return XF.new_empty((B, L, L))


@label_stage_a_split.register_fake
def _(x1p, x1_inv, wbtp, wbt_inv, B, L, H, O, ones_col=None):
    return x1_inv.new_empty((lib().phonlp_label_t4_bytes(B, L, H, O) // 4,))


@pos_head_split.register_fake
def _(enc, sentlens, Whid, bhid, Wclf, bclf, emb, want_feat):
    B, L, E = enc.shape
    D = E + (emb.shape[1] if emb is not None else probs_width(Wclf.shape[0]))
    return (enc.new_empty((B * L, lib().phonlp_split_bytes(1, D)), dtype=_U8), enc.new_empty((B * L,)),
            enc.new_empty((B, L), dtype=_I32), enc.new_empty((B, L, Wclf.shape[0])),
            enc.new_empty((B, L, D) if want_feat else (0,)))


class _Direct:
    """`ops.direct.<name>`: the Python function behind the custom op `<name>`, called without the torch.library dispatch
    (autograd key, redispatch, dynamo guard: ~8 us of host time per call, two thirds of what an eager step over a few
    sentences costs the host).  Same arguments, same checks, same C-ABI call; for eager loops -- under torch.compile
    the registered ops are the ones to call."""

    def __getattr__(self, name):
        f = globals()[name]
        return getattr(f, "_init_fn", f)


direct = _Direct()
