"""phonlp_amd -- MI355X-native task heads for PhoNLP (POS / NER-CRF / biaffine dependency parsing).

The arithmetic lives in hand-written HIP (phonlp_amd/csrc, built into phonlp_amd/lib/libphonlp_heads.so)
behind the C-ABI of include/phonlp_heads.h.  Importing this package does not load the library; the first
op call does, and raises if it has not been built -- there is no CPU fallback.
"""
from .heads import CapturedStep, HeadWeights, TaskHeads, as_sentlens  # noqa: F401

__all__ = ["HeadWeights", "TaskHeads", "as_sentlens"]
