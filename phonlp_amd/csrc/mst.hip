// Chu-Liu-Edmonds single-root MST decoding (ref models/common/chuliu_edmonds.py), one workgroup
// per sentence, float64 arithmetic, bit-identical trees to the reference.
//
// The reference is recursive: arg-max every row, list the cycles with Tarjan, contract the LAST
// listed cycle into a fresh (n-c+1)^2 matrix, recurse, expand.  This kernel runs the same
// sequence of contractions iteratively on ONE l x l float64 matrix held in LDS:
//   * a contracted level is described by `node[]`: logical index (the reference's row/column
//     order at that level) -> physical slot of the matrix.  Contracting a cycle keeps the
//     non-cycle nodes in order and appends the metanode last (ref :93-99); its row/column reuse
//     the slot of the cycle's first member, so nothing is ever compacted or copied;
//   * "the last cycle Tarjan lists" is the cycle whose smallest member is largest: a cycle of a
//     head-pointer graph can only be entered from its own members, so ref tarjan (:46-48) closes
//     cycles in ascending order of their smallest node.  Cycle membership and each cycle's
//     minimum come from log2(n) rounds of pointer doubling -- no serial DFS;
//   * rows whose best head was outside the contracted cycle keep it (their other entries are
//     unchanged, the new metanode column holds a value <= the old maximum and is ordered last,
//     so np.argmax's first-maximum rule picks the same column); only rows that pointed into
//     the cycle, and the metanode's row, are re-scanned;
//   * per level the O(n) index arrays needed for the expansion (ref :105-128) are pushed on a
//     per-sentence stack in the caller's workspace.
// Every float64 operation matches the reference's: (s - edge) + total (:84), numpy's pairwise
// summation order for cycle_scores.sum() (:75) and tree_probs.sum() (:158), strict '>' for
// first-maximum ties.
#include "mst.h"

namespace phonlp {

// scalars kept in LDS
enum { SC_M = 0, SC_NEXT, SC_NROOT, SC_FINITE, SC_NNEED, SC_TAKE, SC_COUNT = 16 };

template <bool IN_LDS>
__global__ __launch_bounds__(1024) void mst_kernel(MstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, NT = blockDim.x;
    const int mode = a.mode, R = a.cslots;
    const int wg = blockIdx.x;                      // workspace region of this workgroup
    const int b = mode == 2 ? wg / R : wg;          // sentence
    const int slot = mode == 2 ? wg % R : 0;        // mode 2: first candidate this workgroup decodes
    const int lane = tid & 63, wave = tid >> 6, nwaves = NT >> 6;
    const int L = a.L, pitch = a.pitch;
    int l = a.sentlens[b];
    if (l > L) l = L;
    int32_t *hout = a.heads + (long)b * L;
    if (mode == 2) {  // candidates of a sentence the first phase left pending (block-uniform tests)
        if (a.status[b] != PHONLP_MST_PENDING || slot >= a.cand_n[b]) return;
    } else {
        if (a.only_fb && a.status[b] != PHONLP_MST_FALLBACK) return;  // decoded by the round-based kernel already
        for (int t = tid; t < L; t += NT) hout[t] = 0;
        if (l <= 1) {  // l == 1: ROOT only -- the reference hits its broken assert (no candidate root)
            if (tid == 0) a.status[b] = l == 1 ? PHONLP_MST_NO_FINITE_TREE : PHONLP_MST_OK;
            return;
        }
    }
    // ---- carve LDS
    size_t off = 0;
    double *S;
    if (IN_LDS) {
        S = reinterpret_cast<double *>(smem_raw);
        off = (size_t)L * pitch * sizeof(double);
    } else {
        S = a.gmat + (long)wg * a.gmat_stride;
    }
    double *ev = reinterpret_cast<double *>(smem_raw + off); off += (size_t)L * 8;
    double *lv = reinterpret_cast<double *>(smem_raw + off); off += (size_t)L * 8;
    double *edge = reinterpret_cast<double *>(smem_raw + off); off += (size_t)L * 8;
    double *dsc = reinterpret_cast<double *>(smem_raw + off); off += 4 * 8;
    int *ibase = reinterpret_cast<int *>(smem_raw + off);
    int *node = ibase, *hdA = node + L, *hdB = hdA + L, *P0 = hdB + L, *P1 = P0 + L, *mn0 = P1 + L, *mn1 = mn0 + L;
    int *flag = mn1 + L, *incyc = flag + L, *rankr = incyc + L, *cyc = rankr + L, *rest = cyc + L;
    int *epick = rest + L, *lpick = epick + L, *need = lpick + L, *tree0 = need + L, *besttree = tree0 + L;
    int *sc = besttree + L;
    int *wtot = sc + SC_COUNT;  // 16 entries

    const float *arc = a.arc + (long)b * L * L;
    uint16_t *stack = a.stack + (long)wg * a.stack_stride;
#define SM(r, c) S[(long)(r) * pitch + (c)]
#define LSYNC()                     \
    do {                            \
        if (IN_LDS) lds_barrier();  \
        else __syncthreads();       \
    } while (0)

#ifdef MST_STAMPS
    long long stamp_acc[16] = {0};
    long long stamp_t = clock64();
#endif
    int root = -1;  // -1: unconstrained first run (ref :137); >= 1: set_root(root) (ref :143-149)
    double best_score = NEG_INF_D;
    int have_best = 0;
    int err = 0;
    int ci = -1, best_ci = -1;  // mode 2: number of the current / best candidate (ascending ROOT-child order)
    // the next child of ROOT in the unconstrained tree after `from` (INT_MAX: none); block-wide, ends on a barrier
    auto next_candidate = [&](int from) {
        if (tid == 0) sc[SC_NEXT] = INT_MAX;
        __syncthreads();
        if (tid < l && tid > from && tree0[tid] == 0) atomicMin(&sc[SC_NEXT], tid);
        __syncthreads();
        const int nx = sc[SC_NEXT];
        __syncthreads();
        return nx;
    };
    if (mode == 2) {
        if (tid < l) tree0[tid] = a.tree0g[(long)b * L + tid];
        __syncthreads();
        root = 0;
        for (int s2 = 0; s2 <= slot; ++s2) root = next_candidate(root);  // slot < cand_n: these exist
        ci = slot;
    }

    for (;;) {
        // -------------------------------------------------------------- load + edit the matrix
        // a wave per row, lanes over columns, eight loads in flight per lane (no index division)
        for (int i = wave; i < l; i += nwaves) {
            const float *arow = arc + (long)i * L;
            for (int j0 = lane; j0 < l; j0 += 64 * 8) {
                float vals[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) vals[t] = (j0 + 64 * t < l) ? arow[j0 + 64 * t] : 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int j = j0 + 64 * t;
                    if (j < l) {
                        double v = (double)vals[t];
                        if (root >= 1) {
                            if (j == 0 && i >= 1) v = NEG_INF_D;
                            if (i == root) v = j == 0 ? 0.0 : NEG_INF_D;
                        }
                        if (i == j) v = NEG_INF_D;                      // ref :55
                        if (i == 0) v = j == 0 ? 0.0 : NEG_INF_D;       // ref :56-57
                        SM(i, j) = v;
                    }
                }
            }
        }
        if (tid < l) { node[tid] = tid; need[tid] = tid; }  // need[]: list of rows whose arg-max must be (re)scanned
        if (tid == 0) sc[SC_NNEED] = l;
        int n = l, levels = 0;
        __syncthreads();
        STAMP(0);

        // -------------------------------------------------------------- contraction loop
        for (;;) {
            // (1) first-maximum arg-max of every row that needs it, in logical column order
            const int nneed = sc[SC_NNEED];
            for (int q = wave; q < nneed; q += nwaves) {
                const int row = need[q];
                const int rs = node[row];
                double bv = NEG_INF_D;
                int bi = INT_MAX;
                // eight columns per lane and trip: their loads (through L2 when the matrix is in global memory) are
                // in flight together; compared in ascending column order, strict '>': the first maximum
                for (int c0 = lane; c0 < n; c0 += 64 * 8) {
                    double v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = (c0 + 64 * t < n) ? SM(rs, node[c0 + 64 * t]) : NEG_INF_D;
#pragma unroll
                    for (int t = 0; t < 8; ++t) {
                        const int c = c0 + 64 * t;
                        if (c < n && (bi == INT_MAX || v[t] > bv)) { bv = v[t]; bi = c; }
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const double v2 = __shfl_xor(bv, o);
                    const int i2 = __shfl_xor(bi, o);
                    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
                }
                if (lane == 0) hdA[row] = bi;
            }
            LSYNC();
            STAMP(1);
            // (2) cycles by pointer doubling: after R rounds (2^R >= n) P[v] lies on v's cycle and
            //     mn[v] is the minimum over >= n consecutive nodes, i.e. the whole cycle for cycle nodes
            if (tid < n) { P0[tid] = hdA[tid]; mn0[tid] = tid; flag[tid] = 0; }
            if (tid == 0) { sc[SC_M] = 0; sc[SC_NNEED] = 0; }
            lds_barrier();  // only LDS arrays are exchanged here
            int *Pc = P0, *Pn = P1, *mc = mn0, *mx = mn1;
            for (int span = 1; span < n; span <<= 1) {
                if (tid < n) {
                    const int p = Pc[tid];
                    Pn[tid] = Pc[p];
                    const int m1 = mc[tid], m2 = mc[p];
                    mx[tid] = m1 < m2 ? m1 : m2;
                }
                lds_barrier();  // only LDS arrays are exchanged here
                int *t1 = Pc; Pc = Pn; Pn = t1;
                int *t2 = mc; mc = mx; mx = t2;
            }
            if (tid < n) flag[Pc[tid]] = 1;
            lds_barrier();  // only LDS arrays are exchanged here
            if (tid < n && flag[tid]) atomicMax(&sc[SC_M], mc[tid]);
            lds_barrier();  // only LDS arrays are exchanged here
            const int M = sc[SC_M];
            STAMP(2);
            if (M == 0) break;  // only ROOT's self loop is left: hdA is a tree
            if (levels >= l) { err = 1; break; }
            // (3) the chosen cycle, member / non-member lists in ascending logical order
            int f = 0;
            if (tid < n) { f = (flag[tid] && mc[tid] == M) ? 1 : 0; incyc[tid] = f; }
            const unsigned long long bal = __ballot(f);
            if (lane == 0) wtot[wave] = __popcll(bal);
            lds_barrier();  // only LDS arrays are exchanged here
            int rc = __popcll(bal & ((1ull << lane) - 1ull));
            int c = 0;
            for (int w = 0; w < nwaves; ++w) {
                const int wt = wtot[w];
                if (w < wave) rc += wt;
                c += wt;
            }
            const int k = n - c;
            if (tid < n) {
                if (f) cyc[rc] = tid;
                else rest[tid - rc] = tid;
                rankr[tid] = tid - rc;
            }
            lds_barrier();  // only LDS arrays are exchanged here
            STAMP(3);
            // (4) cycle edge scores and their numpy-ordered sum (ref :73-75)
            if (tid < c) {
                const int v = cyc[tid];
                edge[tid] = SM(node[v], node[hdA[v]]);
            }
            LSYNC();
            if (tid == 0) dsc[0] = np_sum<4>(edge, c);
            LSYNC();
            const double total = dsc[0];
            STAMP(4);
            // (5) best way into the cycle from every outside node (ref :84,:88,:97) and best head
            //     inside the cycle for every outside node (ref :86,:90,:99)
            int my_slot = 0, my_newhead = 0, my_need = 0;
            double my_ev = 0.0, my_lv = 0.0;
            const int ms = node[cyc[0]];  // slot the metanode will live in
            if (tid < k) {
                const int x = tid;
                const int old = rest[x];
                my_slot = node[old];
                double be = (SM(node[cyc[0]], my_slot) - edge[0]) + total;
                int ae = 0;
                double bl = SM(my_slot, node[cyc[0]]);
                int al = 0;
                for (int q = 1; q < c; ++q) {
                    const int cs = node[cyc[q]];
                    const double ve = (SM(cs, my_slot) - edge[q]) + total;
                    if (ve > be) { be = ve; ae = q; }
                    const double vl = SM(my_slot, cs);
                    if (vl > bl) { bl = vl; al = q; }
                }
                my_ev = be; my_lv = bl;
                epick[x] = ae; lpick[x] = al;
                const int h = hdA[old];
                if (incyc[h]) { my_need = 1; my_newhead = 0; }
                else my_newhead = rankr[h];
            }
            // (6) push this level's expansion record
            {
                uint16_t *rec = stack + (long)levels * a.lvl_stride;
                if (tid == 0) { rec[0] = (uint16_t)n; rec[1] = (uint16_t)c; }
                if (tid < n) rec[2 + tid] = (uint16_t)hdA[tid];
                if (tid < c) rec[2 + n + tid] = (uint16_t)cyc[tid];
                if (tid < k) {
                    rec[2 + n + c + tid] = (uint16_t)rest[tid];
                    rec[2 + n + c + k + tid] = (uint16_t)epick[tid];
                    rec[2 + n + c + 2 * k + tid] = (uint16_t)lpick[tid];
                }
            }
            LSYNC();
            STAMP(5);
            // (7) contract in place: metanode row/column into slot `ms`, new logical order
            if (tid < k) {
                SM(ms, my_slot) = my_ev;
                SM(my_slot, ms) = my_lv;
                node[tid] = my_slot;
                hdA[tid] = my_newhead;
                if (my_need) need[atomicAdd(&sc[SC_NNEED], 1)] = tid;
            }
            if (tid == 0) {
                SM(ms, ms) = NEG_INF_D;
                node[k] = ms;
                need[atomicAdd(&sc[SC_NNEED], 1)] = k;
            }
            n = k + 1;
            ++levels;
            LSYNC();
            STAMP(6);
#ifdef MST_STAMPS
            if (tid == 0) stamp_acc[10] += 1;
#endif
        }
        if (err) break;
        __syncthreads();  // full barrier: the level records (global stores) must be visible to the expansion
        // -------------------------------------------------------------- expansion (ref :105-128)
        int *cur = hdA, *oth = hdB;
        for (int lev = levels - 1; lev >= 0; --lev) {
            const uint16_t *rec = stack + (long)lev * a.lvl_stride;
            const int np_ = rec[0], c = rec[1], k = np_ - c;
            const uint16_t *r_hd = rec + 2, *r_cyc = r_hd + np_, *r_rest = r_cyc + c, *r_ep = r_rest + k, *r_lp = r_ep + k;
            if (tid < k) {
                const int sx = cur[tid];
                oth[r_rest[tid]] = sx < k ? (int)r_rest[sx] : (int)r_cyc[r_lp[tid]];
            }
            if (tid < c) {  // cycle members keep their head, except the one the metanode's head enters through
                const int mh = cur[k];
                oth[r_cyc[tid]] = tid == (int)r_ep[mh] ? (int)r_rest[mh] : (int)r_hd[r_cyc[tid]];
            }
            __syncthreads();
            int *t = cur; cur = oth; oth = t;
        }
        STAMP(7);
#ifdef MST_STAMPS
        if (tid == 0) stamp_acc[11] += 1;
#endif
        // `cur` now holds the tree over the original l nodes
        // -------------------------------------------------------------- single-root logic (ref :137-169)
        if (root < 0) {
            if (tid == 0) { sc[SC_NROOT] = 0; sc[SC_NEXT] = INT_MAX; }
            __syncthreads();
            if (tid < l) {
                tree0[tid] = cur[tid];
                if (tid >= 1 && cur[tid] == 0) { atomicAdd(&sc[SC_NROOT], 1); atomicMin(&sc[SC_NEXT], tid); }
            }
            __syncthreads();
            if (sc[SC_NROOT] == 1) {
#ifdef MST_STAMPS
                if (tid == 0 && a.dbg)
                    for (int k = 0; k < 16; ++k) a.dbg[(long)b * 16 + k] = stamp_acc[k];
#endif
                if (tid < l) hout[tid] = cur[tid];
                if (tid == 0) a.status[b] = PHONLP_MST_OK;
                return;
            }
            if (mode == 1 && sc[SC_NROOT] >= 2) {  // the candidates go to the second phase's grid
                if (tid < l) a.tree0g[(long)b * L + tid] = (uint16_t)cur[tid];
                if (tid == 0) { a.cand_n[b] = sc[SC_NROOT]; a.status[b] = PHONLP_MST_PENDING; }
                return;
            }
        } else {
            // tree_probs = _scores[arange, _tree] on the EDITED top-level matrix (ref :157)
            if (tid == 0) sc[SC_FINITE] = 1;
            __syncthreads();
            if (tid < l) {
                const int i = tid, j = cur[tid];
                double v;
                if (i == 0 || i == root) v = j == 0 ? 0.0 : NEG_INF_D;
                else if (j == 0 || j == i) v = NEG_INF_D;
                else v = (double)arc[(long)i * L + j];
                edge[i] = v;
                if (!(v > NEG_INF_D)) sc[SC_FINITE] = 0;
            }
            __syncthreads();
            if (tid == 0) {
                double score = NEG_INF_D;
                if (sc[SC_FINITE]) score = np_sum<4>(edge, l) + (double)arc[(long)root * L];
                dsc[1] = score;
            }
            __syncthreads();
            const double score = dsc[1];
            if (score > best_score) {  // strict: the first best candidate wins (ref :159)
                best_score = score;
                have_best = 1;
                best_ci = ci;
                if (tid < l) besttree[tid] = cur[tid];
            }
            if (mode == 2) {  // this workgroup's next candidate: R further on
                int nx = root;
                for (int s2 = 0; s2 < R && nx != INT_MAX; ++s2) nx = next_candidate(nx);
                if (tid == 0) sc[SC_NEXT] = nx;
                ci += R;
            } else {
                if (tid == 0) sc[SC_NEXT] = INT_MAX;
                __syncthreads();
                if (tid < l && tid > root && tree0[tid] == 0) atomicMin(&sc[SC_NEXT], tid);
            }
            __syncthreads();
        }
        const int next = sc[SC_NEXT];
        __syncthreads();
        STAMP(8);
        if (next == INT_MAX) break;
        root = next;
    }
    if (mode == 2) {  // this slot's best (or its failure) for mst_pick_kernel
        const long q = (long)b * R + slot;
        if (tid < l && have_best && !err) a.cs_tree[q * L + tid] = (uint16_t)besttree[tid];
        if (tid == 0) {
            a.cs_score[q] = (have_best && !err) ? best_score : NEG_INF_D;
            a.cs_idx[q] = err ? -2 : best_ci;
        }
        return;
    }
#ifdef MST_STAMPS
    if (tid == 0 && a.dbg)
        for (int k = 0; k < 16; ++k) a.dbg[(long)b * 16 + k] = stamp_acc[k];
#endif
    if (err) {
        if (tid == 0) a.status[b] = 2;
        return;
    }
    if (have_best) {
        if (tid < l) hout[tid] = besttree[tid];
        if (tid == 0) a.status[b] = PHONLP_MST_OK;
    } else if (tid == 0) {
        a.status[b] = PHONLP_MST_NO_FINITE_TREE;
    }
#undef SM
#undef LSYNC
}

// Second half of the two-phase form: the winner among a pending sentence's slots -- greatest score, lowest
// candidate number among equals, which is the reference's strict '>' over the candidates in ascending order (:159).
__global__ void mst_pick_kernel(MstArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, L = a.L, R = a.cslots;
    if (a.status[b] != PHONLP_MST_PENDING) return;
    int l = a.sentlens[b];
    if (l > L) l = L;
    const int ns = a.cand_n[b] < R ? a.cand_n[b] : R;
    double best = NEG_INF_D;
    int bs = -1, bi = INT_MAX, bad = 0;
    for (int s = 0; s < ns; ++s) {
        const double sc = a.cs_score[(long)b * R + s];
        const int ix = a.cs_idx[(long)b * R + s];
        if (ix == -2) bad = 1;
        else if (ix >= 0 && (sc > best || (sc == best && ix < bi)) && sc > NEG_INF_D) { best = sc; bs = s; bi = ix; }
    }
    if (!bad && bs >= 0)
        for (int t = tid; t < l; t += blockDim.x) a.heads[(long)b * L + t] = a.cs_tree[((long)b * R + bs) * L + t];
    if (tid == 0) a.status[b] = bad ? 2 : (bs >= 0 ? PHONLP_MST_OK : PHONLP_MST_NO_FINITE_TREE);
}

// slots per sentence of the two-phase form (candidates beyond that many share the slots round-robin)
constexpr int kCandSlots = 4;
static bool mst_two_phase() { return true; }  // long sentences' candidates side by side (one after the other: round 1's form)

static size_t mst_aux_bytes(int L) { return (size_t)L * 8 * 3 + 32 + (size_t)L * 4 * 17 + (SC_COUNT + 16) * 4; }
static bool mst_fits_lds(int L) { return (size_t)L * mst_pitch(L) * 8 + mst_aux_bytes(L) <= 160 * 1024; }

}  // namespace phonlp

using namespace phonlp;

// workgroup regions (expansion stack, global matrix) of a launch: one per sentence, or one per (sentence, slot)
static size_t mst_regions(int B, int L) { return (!mst_fits_lds(L) && !mst_wave_fits(L)) ? (size_t)B * kCandSlots : (size_t)B; }
static size_t mst_two_phase_bytes(int B, int L) {
    const size_t q = (size_t)B * kCandSlots;
    return align_up((size_t)B * L * sizeof(uint16_t), 256) + align_up((size_t)B * sizeof(int32_t), 256) +
           align_up(q * sizeof(double), 256) + align_up(q * sizeof(int32_t), 256) + align_up(q * L * sizeof(uint16_t), 256);
}

// offset of the shared region (candidate queue, counters) of the wave / round-based kernels inside the workspace
static size_t mst_share_offset(int B, int L) {
    const size_t regions = mst_regions(B, L);
    size_t off = align_up(regions * L * mst_lvl_stride(L) * sizeof(uint16_t), 256);
    if (!mst_fits_lds(L)) off += align_up(regions * L * mst_pitch(L) * sizeof(double), 256);
    return off;
}

extern "C" size_t phonlp_mst_ws_bytes(int B, int L) {
    size_t bytes = mst_share_offset(B, L);
    if (mst_wave_fits(L)) {
        bytes += mst_share_bytes(B, L);
        if (mst_fast_fits(L)) bytes += mst_fast_bytes(B, L);
    } else if (!mst_fits_lds(L)) {
        bytes += mst_two_phase_bytes(B, L);
        if (mst_round_fits(L)) bytes += mst_round_bytes(B, L);
    }
    return bytes;
}

static int g_mst_fast = -1;  // -1: not decided yet ($PHONLP_MST_FAST, default on)
static bool mst_fast_enabled() {
    if (g_mst_fast < 0) g_mst_fast = getenv("PHONLP_MST_FAST") ? (atoi(getenv("PHONLP_MST_FAST")) != 0) : 1;
    return g_mst_fast != 0;
}
extern "C" int phonlp_set_mst_fast_path(int on) {
    g_mst_fast = on ? 1 : 0;
    return PHONLP_OK;
}
extern "C" int phonlp_get_mst_fast_path(void) { return mst_fast_enabled() ? 1 : 0; }
// long sentences: the round-based decode of mst_round.hip in front of the two-phase reference-order kernels
static bool mst_round_path(int L) { return !mst_wave_fits(L) && !mst_fits_lds(L) && mst_round_fits(L) && mst_two_phase(); }
extern "C" long phonlp_mst_fallback_counter_offset(int B, int L) {
    if (B <= 0 || L <= 0) return -1;
    if (mst_round_path(L)) return (long)(mst_share_offset(B, L) + mst_two_phase_bytes(B, L) + CT_FB * sizeof(int32_t));
    if (!mst_wave_fits(L) || !mst_fast_fits(L)) return -1;
    return (long)(mst_share_offset(B, L) + CT_FB * sizeof(int32_t));
}
extern "C" long phonlp_mst_fallback_list_offset(int B, int L) {
    if (B <= 0 || L <= 0) return -1;
    if (mst_round_path(L))
        return (long)(mst_share_offset(B, L) + mst_two_phase_bytes(B, L) + align_up((size_t)CT_COUNT * sizeof(int32_t), 256));
    if (!mst_wave_fits(L) || !mst_fast_fits(L)) return -1;
    return (long)(mst_share_offset(B, L) + mst_share_bytes(B, L));
}

extern "C" int phonlp_mst_one_root(const float *arc, const int32_t *sentlens, int B, int L, int32_t *heads,
                                   int32_t *status, void *ws, size_t ws_bytes, void *stream) {
    if (!arc || !sentlens || !heads || !status || B <= 0 || L <= 0)
        return fail(PHONLP_ERR_SHAPE, "mst_one_root: bad argument");
    if (L > 1024) return fail(PHONLP_ERR_UNSUPPORTED, "mst_one_root: L > 1024 (one thread per node)");
    if (!ws || ws_bytes < phonlp_mst_ws_bytes(B, L)) return fail(PHONLP_ERR_WORKSPACE, "mst_one_root: workspace too small");
    MstArgs a{};
    a.arc = arc; a.sentlens = sentlens; a.heads = heads; a.status = status;
    a.stack = (uint16_t *)ws;
    a.lvl_stride = mst_lvl_stride(L);
    a.stack_stride = (long)L * a.lvl_stride;
    a.B = B; a.L = L; a.pitch = mst_pitch(L);
#ifdef MST_STAMPS
    a.dbg = (long long *)getenv("PHONLP_MST_DBG") ? (long long *)strtoull(getenv("PHONLP_MST_DBG"), nullptr, 0) : nullptr;
#endif
    const bool in_lds = mst_fits_lds(L);
    const int threads = L <= 256 ? 256 : (L + 63) / 64 * 64;
    size_t lds = mst_aux_bytes(L);
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.pending(&attr_dev)) {
        if (hipFuncSetAttribute((const void *)mst_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
                hipSuccess ||
            hipFuncSetAttribute((const void *)mst_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
                hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "mst_one_root: cannot raise the dynamic LDS limit");
        attr_once.mark(attr_dev);
    }
    ProfScope ps("mst", as_stream(stream));
    const size_t regions = mst_regions(B, L);
    const size_t stack_bytes = align_up(regions * L * a.lvl_stride * sizeof(uint16_t), 256);
    if (mst_wave_fits(L)) {
        const size_t off = mst_share_offset(B, L);
        mst_share_carve(a, (char *)ws + off);
        const bool fast = mst_fast_enabled() && mst_fast_fits(L);
        int rc = fast ? 0 : launch_mst_clear(a, as_stream(stream));  // the round-based path clears the region itself (phase 2)
        if (rc) return rc;
        if (fast) {
            // round-based decode where it provably returns the reference's tree; the rest lands on the fallback list,
            // which the reference-order kernel below then decodes (usually nothing)
            mst_fast_carve(a, (char *)ws + off + mst_share_bytes(B, L));
            rc = launch_mst_fast(a, as_stream(stream));
            if (rc) return rc;
            a.dbg = nullptr;  // diagnostic builds: the counters are the round-based kernel's
        }
        return launch_mst_wave(a, as_stream(stream));
    }
    if (in_lds) {
        lds += (size_t)L * a.pitch * sizeof(double);
        hipLaunchKernelGGL(mst_kernel<true>, dim3(B), dim3(threads), lds, as_stream(stream), a);
        return check_launch("mst_kernel");
    }
    a.gmat = (double *)((char *)ws + stack_bytes);
    a.gmat_stride = (long)L * a.pitch;
    if (mst_wave_fits(L) || !mst_two_phase()) {
        hipLaunchKernelGGL(mst_kernel<false>, dim3(B), dim3(threads), lds, as_stream(stream), a);
        return check_launch("mst_kernel");
    }
    // Two-phase form: the unconstrained decodes (one workgroup per sentence), then the single-root candidates of the
    // sentences that need them side by side on a B x kCandSlots grid (the reference decodes them one after the other,
    // :151-160; so did this kernel, leaving most of the chip idle when B is small), then the pick.
    {
        char *p = (char *)ws + stack_bytes + align_up(regions * L * a.pitch * sizeof(double), 256);
        const size_t q = (size_t)B * kCandSlots;
        a.tree0g = (uint16_t *)p;   p += align_up((size_t)B * L * sizeof(uint16_t), 256);
        a.cand_n = (int32_t *)p;    p += align_up((size_t)B * sizeof(int32_t), 256);
        a.cs_score = (double *)p;   p += align_up(q * sizeof(double), 256);
        a.cs_idx = (int32_t *)p;    p += align_up(q * sizeof(int32_t), 256);
        a.cs_tree = (uint16_t *)p;
    }
    a.cslots = kCandSlots;
    if (mst_fast_enabled() && mst_round_path(L)) {
        // round-based decode where it provably returns the reference's tree (mst_round.hip); what it leaves with status
        // PHONLP_MST_FALLBACK -- usually nothing -- is decoded by the reference-order kernels below
        mst_round_carve(a, (char *)ws + mst_share_offset(B, L) + mst_two_phase_bytes(B, L));
        const int rc = launch_mst_round(a, as_stream(stream));
        if (rc) return rc;
        a.only_fb = 1;
        a.dbg = nullptr;  // diagnostic builds: the counters are the round-based kernel's
    }
    a.mode = 1;
    hipLaunchKernelGGL(mst_kernel<false>, dim3(B), dim3(threads), lds, as_stream(stream), a);
    int rc = check_launch("mst_kernel");
    if (rc) return rc;
    a.mode = 2;
    hipLaunchKernelGGL(mst_kernel<false>, dim3(B * kCandSlots), dim3(threads), lds, as_stream(stream), a);
    rc = check_launch("mst_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(mst_pick_kernel, dim3(B), dim3(256), 0, as_stream(stream), a);
    return check_launch("mst_pick_kernel");
}
