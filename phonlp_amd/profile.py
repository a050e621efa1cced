"""Per-kernel timing through the library's hipEvent hooks (phonlp_profile_begin/_end)."""
import ctypes
from collections import defaultdict

import numpy as np

from ._lib import check, lib


class KernelTimer:
    """with KernelTimer(max_samples) as kt: ...launch work...;  kt.stats -> {tag: (count, mean_ms, min_ms)}

    `only`: sample the launches with this tag only (every sampled launch adds two event records to its queue)."""

    def __init__(self, max_samples=8192, only=None):
        self.max_samples = max_samples
        self.only = only
        self.stats = {}

    def __enter__(self):
        check(lib().phonlp_profile_only(self.only.encode() if self.only else None), "phonlp_profile_only")
        check(lib().phonlp_profile_begin(self.max_samples), "phonlp_profile_begin")
        return self

    def __exit__(self, *exc):
        ms = np.zeros(self.max_samples, np.float32)
        tags = np.zeros(self.max_samples, np.int32)
        n = lib().phonlp_profile_end(ms.ctypes.data_as(ctypes.c_void_p), tags.ctypes.data_as(ctypes.c_void_p),
                                     self.max_samples)
        if n < 0:
            check(n, "phonlp_profile_end")
        by = defaultdict(list)
        for i in range(n):
            by[lib().phonlp_profile_tag_name(int(tags[i])).decode()].append(float(ms[i]))
        self.stats = {k: (len(v), float(np.mean(v)), float(np.min(v))) for k, v in by.items()}
        return False
