// Chu-Liu-Edmonds single-root MST, round-based decode for sentences too long for LDS (128 < L <= 1024).
//
// Same algorithm and the same condition as mst_fast.hip -- (M): every selection beats every other option by more than
// the margin tau = mst_margin(l, max|entry|), under which the maximum arborescence is unique by a margin the
// reference's float64 rounding cannot bridge, so the reference returns it too (models/common/chuliu_edmonds.py:52-169;
// the argument is in mst_fast.hip's header) -- in the form that suits a matrix that does not fit a CU's LDS:
//
//  * the matrix is never rewritten and never copied: rows are read from the arc log-probabilities in global memory
//    (fp32, L2 / Infinity-Cache resident: 1 MB per sentence at L = 512) whenever a row needs its best head;
//  * per word i: comp[i] (its contracted node), off[i] (what the reference has added to the row's entries so far, :84),
//    and the row's best column OUTSIDE its component (best_j, best_v).  A round contracts every cycle of the components'
//    selected in-edges at once; afterwards only the rows of the merged components whose best column fell inside the new
//    component are rescanned (a wavefront per row, four rows' loads in flight), and a new component's in-edge is the
//    best of its rows' (best_v + off);
//  * the whole workgroup (one thread per word) runs a round: pointer doubling on packed (smallest index, pointer) words
//    finds the cycles and their labels, LDS atomics the cycle totals (their order of addition moves a total by a few
//    ulps, far inside the margin every comparison is held to) and the components' arg-max;
//  * the expansion record of a round (component and in-edge of every word before the merge, 3 x u16 per word) goes to
//    the workgroup's region of the caller's workspace (the expansion stack of mst.hip's kernel, which only runs
//    afterwards, for the sentences this decode could not prove itself on).
//
// ~60 rounds of a few microseconds per 512-word sentence instead of ~400 contraction levels of ~17 us each in the
// reference-order kernel (mst.hip).  Single-root logic as in mst_fast.hip: phase 1 decodes the unconstrained trees,
// phase 2 the candidates of the sentences with several children of ROOT on a B x cslots grid (set_root's edits :143-149
// applied while scanning), the pick kernel takes the greatest score, lowest candidate among equals (:159).
#include "mst.h"

namespace phonlp {

enum { RS_LO = 0, RS_HI, RS_BAD, RS_FAIL, RS_ANY, RS_NLIST, RS_NROOT, RS_COUNT = 16 };
constexpr int kRoundInts = 16;

size_t mst_round_lds(int L) {
    return (size_t)L * (kRoundInts * sizeof(int) + sizeof(float) + 3 * sizeof(double) + sizeof(unsigned long long)) +
           (RS_COUNT + 16) * sizeof(int) + 2 * sizeof(double);
}
bool mst_round_fits(int L) { return L > 128 && L <= 1024 && mst_round_lds(L) <= 160 * 1024; }

__device__ __forceinline__ unsigned long long order_key(double v) {  // monotone in v, never 0 for a finite v
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Best column of up to four rows among the columns outside each row's component (and not column 0 when `skip0`), by one
// wavefront: lanes over columns, all of the four rows' loads in flight together.  Lane 0 stores best_j / best_v; a
// tied or empty row raises sc[RS_FAIL].
template <int CTRL>
__device__ __forceinline__ int rdpp_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
// (value, index, near-duplicate) arg-max step: greatest value; `dup` set when the runner-up is within tau of it (the
// bookkeeping of mst_fast.hip: a -inf operand gives a +inf or NaN difference, which compares false)
__device__ __forceinline__ void argmax_merge(float &bv, int &bj, int &dup, float v2, int j2, int d2, float tau) {
    const bool gt = v2 > bv;
    const int near = ((gt ? v2 - bv : bv - v2) <= tau) ? 1 : 0;
    dup = (gt ? d2 : dup) | near;
    bj = gt ? j2 : bj;
    bv = gt ? v2 : bv;
}
template <int CTRL>
__device__ __forceinline__ void argmax_dpp(float &bv, int &bj, int &dup, float tau) {
    argmax_merge(bv, bj, dup, __int_as_float(rdpp_i<CTRL>(__float_as_int(bv))), rdpp_i<CTRL>(bj), rdpp_i<CTRL>(dup), tau);
}

// Best column of up to four rows among the columns outside each row's component (and not column 0 when `skip0`), by one
// wavefront: lanes over columns, the loads of CG 64-column chunks of the four rows issued before their first
// comparison (a rolled loop over the groups keeps the kernel inside its 128 registers).  Columns past the row's end are
// read at the clamped column L - 1 (always inside the matrix) and masked.  Lane 0 stores best_j / best_v and the row's
// flag rowdup[i] = "its maximum does not lead by more than tau".  The flag fails the decode only if the row's edge is
// SELECTED (the row of a singleton component after the first scan; later the winning row of a merged component): the
// argument in mst_fast.hip's header needs the margin of a component's best in-edge against its other in-edges -- for
// the other rows of the component that is the margin of their best values against the winner's, checked where the
// winner is picked; their own runner-up columns never enter.  Most rescanned rows are not winners.
// (`dup` is an int on purpose: as a bool the four-row unrolled form sends the compiler into a minutes-long search.)
template <int NCH, int CG>
__device__ __forceinline__ void scan_rows4(const float *arc, int L, int l, const int *comp, const int (&rows)[4], int nrows,
                                           bool skip0, int root, int lane, int *bestJ, float *bestV, int *rowdup, float tau) {
    const float NEGF = -__builtin_inff();
    float bv[4];
    int bj[4], ci[4], ri[4], dup[4];
    const float *rp[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        ri[r] = __builtin_amdgcn_readfirstlane(r < nrows ? rows[r] : rows[0]);  // wave-uniform: scalar row bases
        ci[r] = __builtin_amdgcn_readfirstlane(comp[ri[r]]);
        rp[r] = arc + (long)ri[r] * L;
        bv[r] = NEGF; bj[r] = INT_MAX; dup[r] = 0;
    }
#pragma unroll 1
    for (int c0 = 0; c0 < NCH; c0 += CG) {
        float val[CG][4];
        int cj[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const int j = (c0 + c) * 64 + lane;
            const int jc = j < L ? j : L - 1;
#pragma unroll
            for (int r = 0; r < 4; ++r) val[c][r] = rp[r][jc];
            cj[c] = comp[j < l ? j : 0];
        }
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            const int j = (c0 + c) * 64 + lane;
            const bool colok = j < l && !(skip0 && j == 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool ok = colok && cj[c] != ci[r];
                const float v = ok ? val[c][r] : NEGF;
                const bool gt = v > bv[r];
                const int near = (ok && (gt ? v - bv[r] : bv[r] - v) <= tau) ? 1 : 0;
                dup[r] = gt ? near : (dup[r] | near);
                bj[r] = gt ? j : bj[r];
                bv[r] = gt ? v : bv[r];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int d = dup[r];
        argmax_dpp<0xB1>(bv[r], bj[r], d, tau);    // quad_perm xor 1
        argmax_dpp<0x4E>(bv[r], bj[r], d, tau);    // quad_perm xor 2
        argmax_dpp<0x141>(bv[r], bj[r], d, tau);   // row_half_mirror
        argmax_dpp<0x140>(bv[r], bj[r], d, tau);   // row_mirror: every lane of a 16-lane row holds the row's result
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1)
            argmax_merge(bv[r], bj[r], d, __shfl_xor(bv[r], o), __shfl_xor(bj[r], o), __shfl_xor(d, o), tau);
        if (r < nrows && lane == 0) {
            const int i = ri[r];
            if (i == root) { bv[r] = 0.f; bj[r] = 0; d = 0; }  // set_root: the row's only edge, weight 0 (:147-148)
            bestJ[i] = bj[r];
            bestV[i] = bv[r];
            rowdup[i] = (d || !(bv[r] > NEGF)) ? 1 : 0;
        }
    }
}

#ifdef MST_STAMPS
#define RCOUNT(k, v) do { if (tid == 0) stamp_acc[k] += (v); } while (0)
#else
#define RCOUNT(k, v) do {} while (0)
#endif

// NCH = ceil(L / 64) rounded up to 4, 8 or 16: columns per lane of a row scan, CG of them in flight at a time
template <int NCH, int CG>
__global__ __launch_bounds__(1024) void mst_round_kernel(MstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = NT >> 6;
#ifdef MST_STAMPS
    long long stamp_acc[16] = {0};
    long long stamp_t = clock64();
#endif
    const int L = a.L, mode = a.mode, CS = a.cslots;
    const int b = mode == 2 ? blockIdx.x / CS : blockIdx.x;
    const int slot = mode == 2 ? blockIdx.x % CS : 0;
    int l = a.sentlens[b];
    if (l > L) l = L;
    int32_t *hout = a.heads + (long)b * L;
    int ncand = 1;
    if (mode == 2) {
        if (blockIdx.x == 0 && tid < CT_COUNT) a.ctr[tid] = 0;  // the fallback counter: first used by the pick kernel
        if (a.status[b] != PHONLP_MST_PENDING) return;
        ncand = a.cand_n[b];
        if (slot >= ncand) return;
    } else {
        for (int t = tid; t < L; t += NT) hout[t] = 0;
        if (l <= 1) {  // l == 1: ROOT only -- the reference hits its broken assert (no candidate root)
            if (tid == 0) a.status[b] = l == 1 ? PHONLP_MST_NO_FINITE_TREE : PHONLP_MST_OK;
            return;
        }
    }
    double *off = reinterpret_cast<double *>(smem_raw);
    double *selV = off + L, *tot = selV + L;
    unsigned long long *ckey = reinterpret_cast<unsigned long long *>(tot + L);
    double *dsum = reinterpret_cast<double *>(ckey + L);  // 2 entries
    int *comp = reinterpret_cast<int *>(dsum + 2);
    int *bestJ = comp + L, *selI = bestJ + L, *selJ = selI + L, *wA = selJ + L, *wB = wA + L, *lab = wB + L;
    int *mark = lab + L, *cnt = mark + L, *list = cnt + L, *rI1 = list + L, *rJ1 = rI1 + L, *cS1 = rJ1 + L;
    int *rowdup = cS1 + L, *sp1 = rowdup + L, *sp2 = sp1 + L;  // (two spare: kRoundInts = 16)
    (void)sp1; (void)sp2;
    float *bestV = reinterpret_cast<float *>(sp2 + L);
    int *sc = reinterpret_cast<int *>(bestV + L);
    int *wcnt = sc + RS_COUNT;  // 16 entries
    const float *arc = a.arc + (long)b * L * L;
    uint16_t *hist = a.stack + (long)blockIdx.x * a.stack_stride;  // (L + 1) records of 3 L u16 fit: 3 L (L + 1) <= L (4 L + 8)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int v = tid;
    const bool node = v < l;

    // ---- the margin's scale (phase 1): the largest finite |entry| of the rows >= 1 off the diagonal, in one pass over
    // the matrix (a few microseconds from L2); NaN or +inf anywhere: fallback
    double tau;
    if (mode == 1) {
        if (tid == 0) { sc[RS_LO] = 0; sc[RS_BAD] = 0; }
        __syncthreads();
        unsigned amax = 0;
        int bad = 0;
        for (int i = 1 + wave; i < l; i += nw) {
            const float *rp = arc + (long)i * L;
            for (int j0 = 0; j0 < l; j0 += 256) {
                float x[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int j = j0 + 64 * c + lane; x[c] = rp[j < L ? j : L - 1]; }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int j = j0 + 64 * c + lane;
                    const unsigned bits = __float_as_uint(x[c]), mag = bits & 0x7FFFFFFFu;
                    const bool live = j < l && j != i;
                    if (live && mag >= 0x7F800000u) { if (mag != 0x7F800000u || !(bits >> 31)) bad = 1; }  // -inf = no edge
                    else if (live) amax = mag > amax ? mag : amax;
                }
            }
        }
        if (amax) atomicMax(reinterpret_cast<unsigned *>(&sc[RS_LO]), amax);
        if (bad) sc[RS_BAD] = 1;
        __syncthreads();
        if (sc[RS_BAD]) {  // block-uniform
            if (tid == 0) {
                a.status[b] = PHONLP_MST_FALLBACK;
                a.cand_n[b] = 1;  // why (diagnostics; the pick kernel lists the sentence): 1 = NaN / +inf entries
            }
            return;
        }
        tau = mst_margin(l, __uint_as_float((unsigned)sc[RS_LO]));
        if (tid == 0) a.tau[b] = tau;
    } else {
        tau = a.tau[b];
    }
    const float tauf = __double2float_ru(tau);

    double best_score = NEG_INF_D, second_score = NEG_INF_D;
    int best_ci = -1, failed = 0;
    for (int ci = slot; ci < ncand; ci += CS) {  // phase 1: one trip; phase 2: this slot's candidates
        const int root = mode == 2 ? a.cand_root[(long)b * L + ci] : -1;
        const bool skip0 = root >= 1;
        __syncthreads();
        if (tid == 0) sc[RS_FAIL] = 0;
        if (node) { comp[v] = v; off[v] = 0.0; selI[v] = v; mark[v] = 0; bestJ[v] = 0; bestV[v] = 0.f; rowdup[v] = 0; }
        __syncthreads();
        // ---- first arg-max of every row (ref :58)
        for (int r0 = 1 + 4 * wave; r0 < l; r0 += 4 * nw) {
            const int rows[4] = {r0, r0 + 1, r0 + 2, r0 + 3};
            const int n = l - r0 < 4 ? l - r0 : 4;
            scan_rows4<NCH, CG>(arc, L, l, comp, rows, n, skip0, root, lane, bestJ, bestV, rowdup, tauf);
        }
        __syncthreads();
        if (node && v >= 1 && rowdup[v]) sc[RS_FAIL] = 1;  // every row is its own component here: its edge is selected
        __syncthreads();
        if (node && v >= 1) { selJ[v] = bestJ[v]; selV[v] = (double)bestV[v]; }
        STAMP(0);
        RCOUNT(11, 1);
        int fail_ = sc[RS_FAIL];  // block-uniform (read after the barrier above)
        int rounds = 0;
        while (!fail_) {
            // ---- heads of the components; cycles and their smallest members by pointer doubling
            const bool rep = node && v >= 1 && comp[v] == v;
            const int hcv = rep ? comp[selJ[v]] : 0;
            if (node) { wA[v] = (v << 16) | hcv; tot[v] = 0.0; ckey[v] = 0ull; cnt[v] = 0; }
            if (tid == 0) { sc[RS_ANY] = 0; sc[RS_NLIST] = 0; }
            __syncthreads();
            int *src = wA, *dst = wB;
            for (int span = 1; span < l; span <<= 1) {
                if (node) {
                    const int w = src[v], w2 = src[w & 0xFFFF];
                    const int m1 = w >> 16, m2 = w2 >> 16;
                    dst[v] = ((m1 < m2 ? m1 : m2) << 16) | (w2 & 0xFFFF);
                }
                __syncthreads();
                int *t = src; src = dst; dst = t;
            }
            const int stamp = rounds + 1;
            const int wf = node ? src[v] : 0;
            if (rep && (wf & 0xFFFF) != 0) { mark[wf & 0xFFFF] = stamp; sc[RS_ANY] = 1; }  // where the pointer rests is on a cycle
            __syncthreads();
            STAMP(2);
            if (!sc[RS_ANY]) break;  // no cycle left: the components' in-edges form the tree
            if (rounds >= L) { fail_ = 1; break; }  // cannot happen (a round removes a component); keeps the record in bounds
            const bool oncyc = rep && mark[v] == stamp;
            if (oncyc) {
                lab[v] = wf >> 16;
                atomicAdd(&tot[wf >> 16], selV[v]);  // (:75) the order moves the total by ulps: far inside the margin
            }
            if (node) {  // this round's record: components and in-edges BEFORE the merge
                uint16_t *hr = hist + (long)rounds * 3 * L;
                hr[v] = (uint16_t)comp[v];
                hr[L + v] = (uint16_t)selI[v];
                hr[2 * L + v] = (uint16_t)selJ[v];
            }
            __syncthreads();
            STAMP(3);
            // ---- every word of a merged component: new label, row offset total - e_m (:84)
            bool merged = false;
            int nc = 0;
            if (node && v >= 1) {
                const int c = comp[v];
                if (mark[c] == stamp) {
                    merged = true;
                    nc = lab[c];
                    off[v] += tot[nc] - selV[c];
                }
            }
            __syncthreads();
            if (merged) comp[v] = nc;
            __syncthreads();
            // ---- rows whose best column is now inside their own component: rescan
            if (merged && comp[bestJ[v]] == nc) list[atomicAdd(&sc[RS_NLIST], 1)] = v;
            __syncthreads();
            STAMP(4);
            {
                const int n = sc[RS_NLIST];
                RCOUNT(9, n);
                for (int k0 = 4 * wave; k0 < n; k0 += 4 * nw) {
                    const int m = n - k0 < 4 ? n - k0 : 4;
                    const int rows[4] = {list[k0], list[k0 + (m > 1 ? 1 : 0)], list[k0 + (m > 2 ? 2 : 0)],
                                         list[k0 + (m > 3 ? 3 : 0)]};
                    scan_rows4<NCH, CG>(arc, L, l, comp, rows, m, skip0, root, lane, bestJ, bestV, rowdup, tauf);
                }
            }
            __syncthreads();
            STAMP(5);
            // ---- the new components' in-edges: the best row (:88-97 in one step), which must lead every other row of the
            // component by more than tau
            unsigned long long key = 0ull;
            double val = 0.0;
            if (merged) {
                val = (double)bestV[v] + off[v];
                key = order_key(val);
                atomicMax(&ckey[nc], key);
            }
            __syncthreads();
            if (merged && key == ckey[nc]) {
                if (atomicAdd(&cnt[nc], 1) > 0 || rowdup[v]) sc[RS_FAIL] = 1;  // two rows attain it / the winner's own runner-up
                selI[nc] = v;
                selJ[nc] = bestJ[v];
                selV[nc] = val;
            } else if (merged) {
                const unsigned long long kb = ckey[nc];  // the winner's value back from its key
                const double top = __longlong_as_double((long long)((kb >> 63) ? (kb & 0x7FFFFFFFFFFFFFFFull) : ~kb));
                if (top - val <= tau) sc[RS_FAIL] = 1;  // a runner-up inside the margin
            }
            __syncthreads();
            fail_ = sc[RS_FAIL];
            ++rounds;
            STAMP(6);
            RCOUNT(8, 1);
        }
        __syncthreads();
        if (fail_) { failed = 1; break; }
        // ---- expansion: the real in-edge of every component, the rounds undone in reverse
        int *cSc = comp, *cSn = cS1, *rIc = selI, *rIn = rI1, *rJc = selJ, *rJn = rJ1;
        int c_s = 0, s_i = 0, s_j = 0;
        if (rounds > 0 && node) {
            const uint16_t *hs = hist + (long)(rounds - 1) * 3 * L;
            c_s = hs[v]; s_i = hs[L + v]; s_j = hs[2 * L + v];
        }
        for (int s = rounds - 1; s >= 0; --s) {
            const int c0 = c_s, si = s_i, sj = s_j;
            if (node) cSn[v] = c0;
            if (s > 0 && node) {  // next record's loads in flight over this round
                const uint16_t *hs = hist + (long)(s - 1) * 3 * L;
                c_s = hs[v]; s_i = hs[L + v]; s_j = hs[2 * L + v];
            }
            __syncthreads();
            if (node && v >= 1 && c0 == v) {  // a component of this state
                const int Y = cSc[v];
                const int ri = rIc[Y], rj = rJc[Y];
                const bool inside = cSn[ri] == v;
                rIn[v] = inside ? ri : si;
                rJn[v] = inside ? rj : sj;
            }
            __syncthreads();
            int *t;
            t = cSc; cSc = cSn; cSn = t;
            t = rIc; rIc = rIn; rIn = t;
            t = rJc; rJc = rJn; rJn = t;
        }
        const int head = (node && v >= 1) ? rJc[v] : 0;
        STAMP(7);
        if (mode == 1) {
            // ---- single-root logic (ref :137-141): one child of ROOT -> done, else its children are candidates
            const int isr = (node && v >= 1 && head == 0) ? 1 : 0;
            const unsigned long long bal = __ballot(isr);
            if (lane == 0) wcnt[wave] = __popcll(bal);
            __syncthreads();
            int base = 0, nroot = 0;
            for (int w = 0; w < nw; ++w) {
                base += w < wave ? wcnt[w] : 0;
                nroot += wcnt[w];
            }
            if (nroot == 1) {
                if (node) hout[v] = head;
                if (tid == 0) a.status[b] = PHONLP_MST_OK;
            } else if (nroot == 0) {
                failed = 1;  // cannot happen for a tree; left to the reference-order kernel
            } else {
                if (isr) a.cand_root[(long)b * L + base + __popcll(bal & lt_mask)] = v;  // ascending
                if (tid == 0) { a.cand_n[b] = nroot; a.status[b] = PHONLP_MST_PENDING; }
            }
        } else {
            // ---- a candidate's score (ref :155-158); the order of the additions moves it by ulps, the pick allows for tau
            if (tid == 0) dsum[0] = (double)arc[(long)root * L];
            __syncthreads();
            double s = (node && v >= 1 && v != root) ? (double)arc[(long)v * L + head] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
            if (lane == 0) atomicAdd(&dsum[0], s);
            __syncthreads();
            const double total = dsum[0];
            second_score = total > best_score ? best_score : (total > second_score ? total : second_score);
            if (total > best_score) {  // the slot's best candidate; a runner-up within tau sends the sentence to the fallback
                best_score = total;
                best_ci = ci;
                uint16_t *tr = a.cs_tree + ((long)b * CS + slot) * L;
                if (node) tr[v] = (uint16_t)head;
            }
        }
    }
#ifdef MST_STAMPS
    if (tid == 0 && a.dbg)
        for (int k = 0; k < 16; ++k) a.dbg[((long)(mode - 1) * a.B + blockIdx.x) * 16 + k] = stamp_acc[k];
#endif
    if (tid == 0) {
        if (mode == 1) {
            if (failed) {
                a.status[b] = PHONLP_MST_FALLBACK;
                a.cand_n[b] = 2;  // 2 = a selection inside the margin, unconstrained decode
            }
        } else {
            const long qx = (long)b * CS + slot;
            if (best_score - second_score <= tau) failed = 1;
            a.cs_score[qx] = failed ? NEG_INF_D : best_score;
            a.cs_idx[qx] = failed ? -2 : best_ci;
        }
    }
}

// ctr [CT_COUNT] (zeroed per call), fb_list [B], tau [B], cand_root [B][L]
size_t mst_round_bytes(int B, int L) {
    return align_up((size_t)CT_COUNT * sizeof(int32_t), 256) + align_up((size_t)B * sizeof(int32_t), 256) +
           align_up((size_t)B * sizeof(double), 256) + align_up((size_t)B * L * sizeof(int32_t), 256);
}
void mst_round_carve(MstArgs &a, void *base) {
    char *p = (char *)base;
    a.ctr = (int32_t *)p;        p += align_up((size_t)CT_COUNT * sizeof(int32_t), 256);
    a.fb_list = (int32_t *)p;    p += align_up((size_t)a.B * sizeof(int32_t), 256);
    a.tau = (double *)p;         p += align_up((size_t)a.B * sizeof(double), 256);
    a.cand_root = (int32_t *)p;
}

int launch_mst_fast_pick(MstArgs a, hipStream_t stream);

// phases 1-3 (a.cand_n, a.cs_* carved by the caller, a.cslots set); the caller runs the reference-order kernels over the
// sentences left with status PHONLP_MST_FALLBACK afterwards
int launch_mst_round(MstArgs a, hipStream_t stream) {
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.pending(&attr_dev)) {
        const void *fns[] = {(const void *)mst_round_kernel<4, 2>, (const void *)mst_round_kernel<8, 4>, (const void *)mst_round_kernel<16, 4>};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(PHONLP_ERR_LAUNCH, "mst_one_root: cannot raise the dynamic LDS limit");
        attr_once.mark(attr_dev);
    }
    const int threads = a.L <= 256 ? 512 : 1024;  // one thread per word; the waves beyond that speed up the row rescans
    const size_t lds = mst_round_lds(a.L);
    for (int mode = 1; mode <= 2; ++mode) {
        a.mode = mode;
        const dim3 grid(mode == 1 ? a.B : a.B * a.cslots);
        if (a.L <= 256) hipLaunchKernelGGL((mst_round_kernel<4, 2>), grid, dim3(threads), lds, stream, a);
        else if (a.L <= 512) hipLaunchKernelGGL((mst_round_kernel<8, 4>), grid, dim3(threads), lds, stream, a);
        else hipLaunchKernelGGL((mst_round_kernel<16, 4>), grid, dim3(threads), lds, stream, a);
        const int rc = check_launch("mst_round_kernel");
        if (rc) return rc;
    }
    return launch_mst_fast_pick(a, stream);
}

}  // namespace phonlp
