"""Writes the hand-scheduled steady-state loop of linear_split256_kernel (phonlp_amd/csrc/linear.hip) between
the `// >>> steady-state loop` and `// <<< steady-state loop` markers.  The loop interleaves the 48 MFMAs of a
stage with the VALU / LDS / VMEM work of the next stages at a fixed ratio; this script is where the ratio and the
order are edited.  usage: python tools/gen_split_loop.py"""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "phonlp_amd", "csrc", "linear.hip")
PERM = "0x07060302u"
out = []
w = out.append
w('''    // Steady state: multiply stage s, split stage s+1 into the other buffer, fetch stage s+2.  The issue order is
    // written out by hand and pinned with sched_barrier: MFMAs execute asynchronously, so the 4-6 VALU / LDS /
    // VMEM instructions placed after each one run in its 32-cycle shadow (a run of back-to-back MFMAs would
    // leave the wave's other work for after the last one).  ABL != 0 are timing-only ablations of this loop
    // (tools/engine_probe.py): 1 = MFMAs only, 2 = + fragment reads and the barrier, 3 = + split and LDS stores.
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(n) mf(std::integral_constant<int, n>{}, f)
    Frags f;
    if constexpr (ABL == 1) read_frags(lds, f);
    for (; s + 2 < nst; ++s) {
        unsigned char *cur = lds + (s & 1) * BUF, *nxt = lds + ((s + 1) & 1) * BUF;
        if constexpr (ABL != 1) read_frags(cur, f);
        SB();
        constexpr bool SPLIT = ABL == 0 || ABL == 3;
        float r1[4], r2[4];
        unsigned t[4];
        uint2 ph, pm, pl;
''')
n = 0
def mfl(k):
    return f"            MF({k});\n            SB();\n"
for q in range(6):
    first = q == 0  # the first split chunks run before any MFMA: they cover the fragment reads' latency
    w(f"        // ---- split of R[{q}]\n        {{\n")
    w(f"            float x[4] = {{R[{q}].x, R[{q}].y, R[{q}].z, R[{q}].w}};\n")
    w("            if constexpr (SPLIT) {\n")
    w("                if constexpr (ABL == 3) x[0] += (float)s;  // keeps the split inside the loop\n")
    w("#pragma unroll\n                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(x[e]) & 0xFFFF0000u;\n")
    w("                r1[0] = x[0] - __uint_as_float(t[0]);\n                r1[1] = x[1] - __uint_as_float(t[1]);\n            }\n")
    if not first:
        w(mfl(n)); n += 1
    w("            if constexpr (SPLIT) {\n")
    w("                r1[2] = x[2] - __uint_as_float(t[2]);\n                r1[3] = x[3] - __uint_as_float(t[3]);\n")
    w("#pragma unroll\n                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(r1[e]) & 0xFFFF0000u;\n            }\n")
    if not first:
        w(mfl(n)); n += 1
    w("            if constexpr (SPLIT) {\n")
    w("#pragma unroll\n                for (int e = 0; e < 4; ++e) r2[e] = r1[e] - __uint_as_float(t[e]);\n")
    w(f"                ph.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), {PERM});\n")
    w(f"                ph.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), {PERM});\n            }}\n")
    w(mfl(n)); n += 1
    w("            if constexpr (SPLIT) {\n")
    w(f"                pm.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), {PERM});\n")
    w(f"                pm.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), {PERM});\n")
    w(f"                pl.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), {PERM});\n")
    w(f"                pl.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), {PERM});\n            }}\n")
    w(mfl(n)); n += 1
    w("            if constexpr (SPLIT) {\n")
    w(f"                unsigned char *p = nxt + wofs + {q} * 64 * 32;\n")
    w("                *reinterpret_cast<uint2 *>(p) = ph;\n                *reinterpret_cast<uint2 *>(p + PLANE) = pm;\n")
    w("                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = pl;\n            }\n")
    w(mfl(n)); n += 1
    w("        }\n")
w("        int col = (s + 2) * SK + c4;\n        col = col < a.K - 4 ? col : a.K - 4;\n")
for q in range(6):
    w(f"        if constexpr (ABL == 0) R[{q}] = *reinterpret_cast<const float4 *>(src[{q}] + col);\n")
    w(f"        MF({n});\n        SB();\n"); n += 1
w("        // the remaining MFMAs, bare\n")
for k in range(n, 48):
    w(f"        MF({k});\n")
w("        SB();\n        if constexpr (ABL != 1) lds_barrier();\n    }\n#undef MF\n#undef SB\n")
src = open(PATH).read()
a = src.index("    // >>> steady-state loop\n") + len("    // >>> steady-state loop\n")
b = src.index("    // <<< steady-state loop\n")
open(PATH, "w").write(src[:a] + "".join(out) + src[b:])
print("MFMAs interleaved:", n, "bare:", 48 - n)
