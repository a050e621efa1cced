// Chu-Liu-Edmonds single-root MST, wave-synchronous variant (used when the float64 matrix fits LDS, L <= ~134).
//
// Same algorithm, same arithmetic and same tie-breaking as mst.hip (see its header for why the slot map, the
// "largest minimum index" cycle choice and the incremental row rescans reproduce ref
// models/common/chuliu_edmonds.py bit for bit).  What differs is who runs the level loop.  On the synthetic
// head distribution a sentence of 128 nodes needs ~100-130 contraction levels per run, each a dozen tiny
// dependent phases; with four waves every phase boundary is an s_barrier, and the block-synchronous kernel
// spends ~12k cycles per level mostly waiting.  Here the whole workgroup only loads the matrix and does the
// level-0 arg-max (the two O(l^2) steps); then ONE wavefront runs the contraction loop, the expansion and the
// single-root scoring with no barrier at all: LDS operations of a wave execute in program order, so a
// wavefront-scope fence (a compiler ordering point, no instruction) is all a phase boundary needs.
// Lane i owns nodes i, i+64, ... (NPL per lane); lists are built with ballot/popcount prefix ranks.
//
// Expansion records live on a byte stack in the LDS left over after the matrix (2n bytes per level: cycle
// members, their heads, and per outside node the two arg-max picks; the outside-node list is rebuilt from the
// member list).  A level that does not fit spills to the caller's global workspace; its loads are slower, the
// result is the same.
//
// Single-root retries (ref :137-169) are SHARED between the workgroups of a launch.  A sentence whose unconstrained
// tree has r >= 2 children of ROOT needs r more decodes, one per candidate root; done by the owning workgroup alone
// they made the stragglers (1 + r decodes) set the kernel's duration.  Here the owner pushes the r candidates on a
// queue in global memory and every workgroup, once its own sentence is decoded, pops candidates -- of any sentence --
// until the queue is empty.  Each candidate's score and tree go to global scratch; the workgroup that finishes a
// sentence's LAST candidate picks the winner in candidate order with strict '>' (ref :159), so the result does not
// depend on who decoded what.  A workgroup with nothing to pop and no own sentence left exits (waiting for late
// pushes was measured and dropped: the waiters' LDS keeps other streams' GEMMs off their CUs), so the grid always
// drains; whoever pushes candidates pops them itself if nobody else does.
//
// Forward progress: NO loop in this kernel waits on another workgroup without a bound.  The pop reads an entry before
// it takes it (entries are written once per call), the re-read of a reserved-but-unwritten entry and the fallback
// launch's look-out for late candidates are both counted down; when a count runs out the workgroup goes on (or
// leaves) and the work stays with the workgroup that produced it.  The result never depends on who decoded what.
#include <type_traits>

#include "mst.h"

namespace phonlp {

#define WFENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

enum { WS_NEXT = 0, WS_DONE, WS_COUNT = 8 };

size_t mst_share_zero_bytes(int B, int L) { return align_up(((size_t)CT_COUNT + B + (size_t)B * L) * sizeof(int32_t), 256); }
size_t mst_share_bytes(int B, int L) {
    const size_t Lw = (L + 3) / 4;
    return mst_share_zero_bytes(B, L) + align_up((size_t)B * sizeof(int32_t), 256) +
           align_up((size_t)B * L * sizeof(int32_t), 256) + align_up((size_t)B * L * sizeof(double), 256) +
           align_up((size_t)B * L * Lw * sizeof(uint32_t), 256);
}
void mst_share_carve(MstArgs &a, void *base) {
    char *p = (char *)base;
    const size_t B = a.B, L = a.L, Lw = (L + 3) / 4;
    a.ctr = (int32_t *)p;
    a.cand_left = a.ctr + CT_COUNT;
    a.tasks = a.cand_left + B;
    p += mst_share_zero_bytes(a.B, a.L);
    a.cand_n = (int32_t *)p;      p += align_up(B * sizeof(int32_t), 256);
    a.cand_root = (int32_t *)p;   p += align_up(B * L * sizeof(int32_t), 256);
    a.cand_score = (double *)p;   p += align_up(B * L * sizeof(double), 256);
    a.cand_tree = (uint32_t *)p;
}

// device-scope loads that bypass the (non-coherent) per-CU cache: data another workgroup published
__device__ __forceinline__ int ld_i32(const int32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned ld_u32(const uint32_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_f64(const double *p) {
    return __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT));
}

struct WaveLayout {
    size_t off_edge, off_int, off_stack, total;
    int stack_cap;
};
constexpr int kIntArrays = 16;
static WaveLayout wave_layout(int L) {
    WaveLayout w;
    const size_t mat = (size_t)L * mst_pitch(L) * sizeof(double);
    w.off_edge = mat;
    w.off_int = w.off_edge + (size_t)L * sizeof(double);
    w.off_stack = w.off_int + ((size_t)kIntArrays * L + WS_COUNT) * sizeof(int);
    const long cap = 160L * 1024 - (long)w.off_stack;
    w.stack_cap = cap > 0 ? (int)(cap & ~15L) : 0;
    w.total = w.off_stack + w.stack_cap;
    return w;
}
bool mst_wave_fits(int L) { return L <= 255 && wave_layout(L).stack_cap >= 4 * L; }

// Cross-lane exchanges inside a 16-lane row as DPP moves (quad_perm / row_half_mirror / row_mirror: VALU operand
// modifiers, a few cycles) instead of ds_bpermute round trips through the LDS crossbar (~100 cycles each, and the
// level loop is one long dependent chain).  The four patterns pair lane groups of 1, 2, 4 and 8, so applying them in
// turn leaves every lane of the row with the row's reduction; max / arg-max with the lowest-index tie-break are
// order-independent, so the results are those of the xor butterfly.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int)(b & 0xFFFFFFFFll)), hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

__device__ __forceinline__ int wave_max_i(int v) {
    int t;
    t = dpp_i<DPP_XOR1>(v); v = t > v ? t : v;
    t = dpp_i<DPP_XOR2>(v); v = t > v ? t : v;
    t = dpp_i<DPP_HALF_MIRROR>(v); v = t > v ? t : v;
    t = dpp_i<DPP_MIRROR>(v); v = t > v ? t : v;
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
        t = __shfl_xor(v, o);
        v = t > v ? t : v;
    }
    return v;
}

// (value, index) arg-max across the 16 lanes of a row: greatest value, lowest index among equals
template <int CTRL>
__device__ __forceinline__ void argmax_step(double &bv, int &bi) {
    const double v2 = dpp_d<CTRL>(bv);
    const int i2 = dpp_i<CTRL>(bi);
    if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
}
__device__ __forceinline__ int row16_argmax(double bv, int bi) {
    argmax_step<DPP_XOR1>(bv, bi);
    argmax_step<DPP_XOR2>(bv, bi);
    argmax_step<DPP_HALF_MIRROR>(bv, bi);
    argmax_step<DPP_MIRROR>(bv, bi);
    return bi;
}

// first-maximum arg-max of logical row `row` over logical columns [0, n), by the 16 lanes of a group
__device__ __forceinline__ int row_argmax16(const double *S, int pitch, const int *node, int row, int n, int sub) {
    const int rs = node[row];
    double bv = NEG_INF_D;
    int bi = INT_MAX;
    if (sub < n) { bv = S[(long)rs * pitch + node[sub]]; bi = sub; }
    for (int c = sub + 16; c < n; c += 16) {
        const double v = S[(long)rs * pitch + node[c]];
        if (v > bv) { bv = v; bi = c; }
    }
    return row16_argmax(bv, bi);
}

// The same arg-max with the lane's column slots (node[sub + 16 t]) already in registers: all of the row's loads
// are issued together instead of as a chain of dependent (slot, value) pairs.  Same comparison order.
template <int NC>
__device__ __forceinline__ int row_argmax16_cols(const double *S, int pitch, int rs, const int (&cols)[NC], int n,
                                                 int sub) {
    double vals[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) vals[t] = (sub + 16 * t < n) ? S[(long)rs * pitch + cols[t]] : NEG_INF_D;
    double bv = NEG_INF_D;
    int bi = INT_MAX;
    if (sub < n) { bv = vals[0]; bi = sub; }
#pragma unroll
    for (int t = 1; t < NC; ++t)
        if (sub + 16 * t < n && vals[t] > bv) { bv = vals[t]; bi = sub + 16 * t; }
    return row16_argmax(bv, bi);
}

template <int NPL>
__global__ __launch_bounds__(256) void mst_wave_kernel(MstArgs a, int stack_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wg = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, sub = lane & 15;
    const int L = a.L, pitch = a.pitch, Lw = (L + 3) >> 2;
    // own sentences: wg, wg + gridDim.x, ... -- entries of the fallback list when there is one (mst_fast.hip sent them
    // here), the sentences themselves otherwise
    const int nown = a.fb_list ? ld_i32(&a.ctr[CT_FB]) : a.B;
    int own_k = wg;
    if (tid == 0) atomicAdd(&a.ctr[CT_STARTED], 1);
    int b = 0, slot = -1, root = -1;  // current task: sentence, candidate slot (-1 = the unconstrained decode)
    int wait_budget = 8192;           // polls this workgroup may still spend looking for other sentences' candidates
#ifdef MST_STAMPS
    long long stamp_acc[16] = {0};
    long long stamp_t = clock64();
#endif
    double *S = reinterpret_cast<double *>(smem_raw);
    double *edge = S + (long)L * pitch;
    int *node = reinterpret_cast<int *>(edge + L);
    int *hdA = node + L, *hdB = hdA + L, *pk0 = hdB + L, *pk1 = pk0 + L, *flag = pk1 + L, *incyc = flag + L;
    int *rankr = incyc + L, *cyc = rankr + L, *rest = cyc + L, *needl = rest + L, *tree0 = needl + L;
    int *besttree = tree0 + L, *lvl_n = besttree + L, *lvl_c = lvl_n + L, *lvl_off = lvl_c + L;
    int *sc = lvl_off + L;
    unsigned char *lstack = reinterpret_cast<unsigned char *>(sc + WS_COUNT);
    unsigned char *gstack = reinterpret_cast<unsigned char *>(a.stack + (long)wg * a.stack_stride);  // per WORKER
    const int gstride = a.lvl_stride * 2;  // bytes per spilled level (>= 2n)
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#define SM(r, c) S[(long)(r) * pitch + (c)]

    for (;;) {  // one task per trip: a candidate popped from the shared queue if there is one, else the next own sentence
        if (tid == 0) {
            int code = -1;
            // Pop: a queue entry is written ONCE per call (0 -> code + 1) and read BEFORE the head is advanced past it,
            // so nobody ever owns an entry it cannot read yet.  An entry that is reserved (tail advanced) but not
            // stored yet is re-read a bounded number of times and then left to its pusher, which pops what it pushed
            // once its own sentence is done: no workgroup waits on another one without a bound.
            for (int tries = 0;;) {
                const int h = ld_i32(&a.ctr[CT_Q_HEAD]), t = ld_i32(&a.ctr[CT_Q_TAIL]);
                if (h >= t) break;
                const int v = ld_i32(&a.tasks[h]);
                if (v == 0) {
                    if (++tries > 64) break;
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                if (atomicCAS(&a.ctr[CT_Q_HEAD], h, h + 1) == h) {  // fails at most once per entry someone else took
                    code = v - 1;
                    break;
                }
            }
            if (code < 0 && own_k < nown) code = -2 - own_k;
            // Fallback list only (few sentences on a small grid: a sentence's candidates would otherwise all be decoded
            // by the workgroup that owns it): with nothing to do, look for candidates a little longer while some
            // sentence's unconstrained decode is still running -- only once every workgroup of the grid has started
            // (all of them are resident, none needs this one's CU), and only for a BOUNDED number of polls
            // (wait_budget, ~1 us each, per workgroup and launch): a workgroup that runs out leaves, and whoever pushes
            // candidates pops them itself, so the grid drains whatever the scheduler does with the other workgroups.
            if (code == -1 && a.fb_list) {
                while (wait_budget > 0 && ld_i32(&a.ctr[CT_STARTED]) >= (int)gridDim.x &&
                       ld_i32(&a.ctr[CT_FIRST_DONE]) < nown) {
                    --wait_budget;
                    __builtin_amdgcn_s_sleep(32);
                    const int h = ld_i32(&a.ctr[CT_Q_HEAD]), t = ld_i32(&a.ctr[CT_Q_TAIL]);
                    if (h < t) { code = INT_MIN; break; }  // something was pushed: go round again and pop it
                }
            }
            __threadfence();
            sc[WS_NEXT] = code;
        }
        __syncthreads();
        const int code = sc[WS_NEXT];
        __syncthreads();
        if (code == -1) break;  // nothing queued, no own sentence left
        if (code == INT_MIN) continue;
        if (code <= -2) {
            const int k = -2 - code;
            b = a.fb_list ? (a.fb_list[k] & 0xFFFFFF) : k;
            slot = -1;
            root = -1;
            own_k += gridDim.x;
        } else {
            b = code >> 8;
            slot = code & 0xFF;
            root = ld_i32(&a.cand_root[(long)b * L + slot]);
        }
        int err = 0;
        int l = a.sentlens[b];
        if (l > L) l = L;
        const float *arc = a.arc + (long)b * L * L;
        int32_t *hout = a.heads + (long)b * L;
        bool skip = false;
        if (slot < 0) {
            for (int t = tid; t < L; t += 256) hout[t] = 0;
            if (l <= 1) {
                if (tid == 0) a.status[b] = l == 1 ? PHONLP_MST_NO_FINITE_TREE : PHONLP_MST_OK;
                skip = true;  // block-uniform
            }
        }
        if (!skip) {
        // ---- whole workgroup: load + edit the matrix (ref :136, :143-149, :55-57), level-0 arg-max (ref :58)
        // a wave per row, lanes over columns (no index division); eight rows' loads are issued together so that the
        // latency of the arc matrix (HBM / L2) is paid once per batch, not once per row
        for (int i0 = wave; i0 < l; i0 += 32) {
            float vals[8][NPL];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = i0 + 4 * r;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int j = lane + 64 * q;
                    vals[r][q] = (i < l && j < l) ? arc[(long)i * L + j] : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = i0 + 4 * r;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int j = lane + 64 * q;
                    if (i < l && j < l) {
                        double v = (double)vals[r][q];
                        if (root >= 1) {
                            if (j == 0 && i >= 1) v = NEG_INF_D;
                            if (i == root) v = j == 0 ? 0.0 : NEG_INF_D;
                        }
                        if (i == j) v = NEG_INF_D;
                        if (i == 0) v = j == 0 ? 0.0 : NEG_INF_D;
                        SM(i, j) = v;
                    }
                }
            }
        }
        for (int t = tid; t < l; t += 256) node[t] = t;
        __syncthreads();
        STAMP(0);
        for (int row = tid >> 4; row < l; row += 16) {
            const int bi = row_argmax16(S, pitch, node, row, l, sub);
            if (sub == 0) hdA[row] = bi;
        }
        __syncthreads();
        STAMP(1);

        if (wave == 0) {
            int n = l, levels = 0, used = 0;
            // ------------------------------------------------------------------ contraction loop
            for (;;) {
                // cycles by pointer doubling on packed (min index << 16 | pointer) words, one word per owned node
                // in registers; the gather "word of the node I point at" is a ds_bpermute per owner register
                int word[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    word[q] = 0;
                    if (v < n) { word[q] = hdA[v] | (v << 16); flag[v] = 0; }
                }
                for (int span = 1; span < n; span <<= 1) {
                    int nw[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int x = word[q];
                        const int tn = x & 0xFFFF;
                        int y = __shfl(word[0], tn & 63);
#pragma unroll
                        for (int q2 = 1; q2 < NPL; ++q2) {
                            const int y2 = __shfl(word[q2], tn & 63);
                            if ((tn >> 6) == q2) y = y2;
                        }
                        const int mx = x >> 16, my = y >> 16;
                        nw[q] = (y & 0xFFFF) | ((mx < my ? mx : my) << 16);
                    }
#pragma unroll
                    for (int q = 0; q < NPL; ++q) word[q] = nw[q];
                }
                WFENCE();
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    if (v < n) flag[word[q] & 0xFFFF] = 1;
                }
                WFENCE();
                int M = 0;
                int oncyc[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    oncyc[q] = (v < n) ? flag[v] : 0;
                    if (oncyc[q]) { const int m = word[q] >> 16; M = m > M ? m : M; }
                }
                M = wave_max_i(M);
                STAMP(2);
                if (M == 0) break;
                if (levels >= l) { err = 1; break; }
                // the chosen cycle; member / outside lists in ascending logical order
                int c = 0;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    const int f = (oncyc[q] && (word[q] >> 16) == M) ? 1 : 0;
                    const unsigned long long bal = __ballot(f);
                    const int rc = c + __popcll(bal & lt_mask);
                    if (v < n) {
                        incyc[v] = f;
                        rankr[v] = v - rc;
                        if (f) cyc[rc] = v; else rest[v - rc] = v;
                    }
                    c += __popcll(bal);
                }
                const int k = n - c;
                WFENCE();
                STAMP(3);
                // cycle edge scores and their numpy-ordered sum (ref :73-75)
                for (int x = lane; x < c; x += 64) {
                    const int v = cyc[x];
                    edge[x] = SM(node[v], node[hdA[v]]);
                }
                WFENCE();
                const double total = np_sum<4>(edge, c);  // every lane, broadcast LDS reads
                STAMP(4);
                // expansion record of this level
                // record stores go through REC_ST: one wave-uniform branch per store site instead of a pointer that
                // may be LDS or global (which hipcc can only serve with flat_ instructions)
                const bool rec_lds = used + 2 * n <= stack_cap;
                unsigned char *rec_l = lstack + used;
                unsigned char *rec_g = gstack + (long)levels * gstride;
                if (rec_lds) { if (lane == 0) lvl_off[levels] = used; used += 2 * n; }
                else if (lane == 0) lvl_off[levels] = -1;
#define REC_ST(i, v) do { if (rec_lds) rec_l[i] = (unsigned char)(v); else rec_g[i] = (unsigned char)(v); } while (0)
                if (lane == 0) { lvl_n[levels] = n; lvl_c[levels] = c; }
                for (int x = lane; x < c; x += 64) { REC_ST(x, cyc[x]); REC_ST(c + x, hdA[cyc[x]]); }
                // best way into the cycle from every outside node / best head inside it (ref :84-99)
                const int ms = node[cyc[0]];
                int my_slot[NPL], my_newhead[NPL], my_need[NPL];
                double my_ev[NPL], my_lv[NPL];
                // CU = members handled with all loads in flight; cycles of 2 nodes dominate, then 3: two instantiations
                // so that a 2-cycle does not pay for four members' loads
                auto enter_leave = [&](auto CUc) __attribute__((always_inline)) {
                constexpr int CU = decltype(CUc)::value;
                int csm[CU];
                double em[CU];
#pragma unroll
                for (int m = 0; m < CU; ++m) {
                    csm[m] = m < c ? node[cyc[m]] : ms;
                    em[m] = m < c ? edge[m] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int x = lane + 64 * q;
                    my_slot[q] = 0; my_newhead[q] = 0; my_need[q] = 0; my_ev[q] = 0.0; my_lv[q] = 0.0;
                    if (x < k) {
                        const int old = rest[x];
                        const int slot = node[old];
                        const int h = hdA[old];
                        double se[CU], sl[CU];
#pragma unroll
                        for (int m = 0; m < CU; ++m) {
                            se[m] = SM(csm[m], slot);
                            sl[m] = SM(slot, csm[m]);
                        }
                        double be = (se[0] - em[0]) + total;
                        double bl = sl[0];
                        int ae = 0, al = 0;
#pragma unroll
                        for (int m = 1; m < CU; ++m) {
                            if (m < c) {
                                const double ve = (se[m] - em[m]) + total;
                                if (ve > be) { be = ve; ae = m; }
                                if (sl[m] > bl) { bl = sl[m]; al = m; }
                            }
                        }
                        for (int m = CU; m < c; ++m) {
                            const int cs = node[cyc[m]];
                            const double ve = (SM(cs, slot) - edge[m]) + total;
                            if (ve > be) { be = ve; ae = m; }
                            const double vl = SM(slot, cs);
                            if (vl > bl) { bl = vl; al = m; }
                        }
                        my_slot[q] = slot; my_ev[q] = be; my_lv[q] = bl;
                        REC_ST(2 * c + x, ae);
                        REC_ST(2 * c + k + x, al);
                        if (incyc[h]) my_need[q] = 1; else my_newhead[q] = rankr[h];
                    }
                }
                };
                if (c <= 2) enter_leave(std::integral_constant<int, 2>{});
                else enter_leave(std::integral_constant<int, 4>{});
                WFENCE();
                STAMP(5);
                // contract in place; rows that pointed into the cycle (and the metanode) go on the rescan list
                int nneed = 0;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int x = lane + 64 * q;
                    const int nd = (x < k && my_need[q]) ? 1 : 0;
                    if (x < k) {
                        SM(ms, my_slot[q]) = my_ev[q];
                        SM(my_slot[q], ms) = my_lv[q];
                        node[x] = my_slot[q];
                        hdA[x] = my_newhead[q];
                    }
                    const unsigned long long bal = __ballot(nd);
                    if (nd) needl[nneed + __popcll(bal & lt_mask)] = x;
                    nneed += __popcll(bal);
                }
                if (lane == 0) { SM(ms, ms) = NEG_INF_D; node[k] = ms; needl[nneed] = k; }
                ++nneed;
                n = k + 1;
                ++levels;
                WFENCE();
                STAMP(6);
#ifdef MST_STAMPS
                if (tid == 0) { stamp_acc[10] += 1; stamp_acc[13] += nneed; }
#endif
                {
                    int cols[4 * NPL];
#pragma unroll
                    for (int t = 0; t < 4 * NPL; ++t) cols[t] = (sub + 16 * t < n) ? node[sub + 16 * t] : 0;
                    for (int p = 0; p < nneed; p += 4) {
                        const int idx = p + (lane >> 4);
                        const int row = idx < nneed ? needl[idx] : 0;
                        const int bi = row_argmax16_cols<4 * NPL>(S, pitch, node[row], cols, n, sub);
                        if (idx < nneed && sub == 0) hdA[row] = bi;
                    }
                }
                WFENCE();
                STAMP(12);
            }
            // ------------------------------------------------------------------ expansion (ref :105-128)
            int *cur = hdA, *oth = hdB;
            if (!err) {
                __threadfence_block();  // spilled records (global stores of this wave) before they are re-read
                for (int lev = levels - 1; lev >= 0; --lev) {
                    const int np_ = lvl_n[lev], c = lvl_c[lev], k = np_ - c;
                    const int off = lvl_off[lev];
                    // instantiated once per address space (LDS stack / global spill), so the record reads are
                    // ds_read_u8 or global_load_ubyte rather than flat loads
                    auto expand_level = [&](const unsigned char *rec) __attribute__((always_inline)) {
                    const unsigned char *r_cyc = rec, *r_hd = rec + c, *r_ep = rec + 2 * c, *r_lp = r_ep + k;
                    constexpr int CE = 4;
                    if (c <= CE) {
                        // rest[x] = x + #{j : member_j - j <= x} (members ascending), straight from the record; the
                        // metanode's pick is folded into the member loop: one store phase, one fence per level
                        int thr[CE];
#pragma unroll
                        for (int j = 0; j < CE; ++j) thr[j] = j < c ? (int)r_cyc[j] - j : INT_MAX;
                        auto rest_of = [&](int x) {
                            int r = x;
#pragma unroll
                            for (int j = 0; j < CE; ++j) r += (x >= thr[j]) ? 1 : 0;
                            return r;
                        };
                        const int mh = cur[k];
                        const int ep = r_ep[mh];
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int x = lane + 64 * q;
                            if (x < k) {
                                const int sx = cur[x];
                                oth[rest_of(x)] = sx < k ? rest_of(sx) : (int)r_cyc[r_lp[x]];
                            }
                        }
                        if (lane < c) oth[r_cyc[lane]] = lane == ep ? rest_of(mh) : (int)r_hd[lane];
                        WFENCE();
                    } else {
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int v = lane + 64 * q;
                            if (v < np_) flag[v] = 0;
                        }
                        WFENCE();
                        for (int x = lane; x < c; x += 64) flag[r_cyc[x]] = 1;
                        WFENCE();
                        int base = 0;
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int v = lane + 64 * q;
                            const int f = (v < np_ && flag[v]) ? 1 : 0;
                            const unsigned long long bal = __ballot(f);
                            const int rc = base + __popcll(bal & lt_mask);
                            if (v < np_ && !f) rest[v - rc] = v;
                            base += __popcll(bal);
                        }
                        WFENCE();
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int x = lane + 64 * q;
                            if (x < k) {
                                const int sx = cur[x];
                                oth[rest[x]] = sx < k ? rest[sx] : (int)r_cyc[r_lp[x]];
                            }
                        }
                        for (int x = lane; x < c; x += 64) oth[r_cyc[x]] = r_hd[x];
                        WFENCE();
                        if (lane == 0) {
                            const int mh = cur[k];
                            oth[r_cyc[r_ep[mh]]] = rest[mh];
                        }
                        WFENCE();
                    }
                    };
                    if (off >= 0) expand_level(lstack + off);
                    else expand_level(gstack + (long)lev * gstride);
                    int *t = cur; cur = oth; oth = t;
                }
            }
            STAMP(7);
#ifdef MST_STAMPS
            if (tid == 0) stamp_acc[11] += 1;
#endif
            // ------------------------------------------------------------------ single-root logic (ref :137-169)
            if (slot < 0) {
                // the unconstrained tree: done if ROOT has exactly one child, else its children become candidates
                int nroot = 0;
                if (!err) {
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int v = lane + 64 * q;
                        const int isr = (v >= 1 && v < l && cur[v] == 0) ? 1 : 0;
                        const unsigned long long bal = __ballot(isr);
                        if (isr) a.cand_root[(long)b * L + nroot + __popcll(bal & lt_mask)] = v;  // ascending
                        nroot += __popcll(bal);
                    }
                }
                if (err) {
                    if (lane == 0) a.status[b] = 2;
                } else if (nroot == 1) {
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int v = lane + 64 * q;
                        if (v < l) hout[v] = cur[v];
                    }
                    if (lane == 0) a.status[b] = PHONLP_MST_OK;
                } else if (nroot == 0) {
                    if (lane == 0) a.status[b] = PHONLP_MST_NO_FINITE_TREE;
                } else {
                    if (lane == 0) {
                        a.cand_n[b] = nroot;
                        a.cand_left[b] = nroot;
                    }
                    __threadfence();  // candidate roots and counts before the queue entries that point at them
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&a.ctr[CT_Q_TAIL], nroot);
                    base = __shfl(base, 0);
                    for (int j = lane; j < nroot; j += 64)
                        __hip_atomic_store(&a.tasks[base + j], b * 256 + j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                // a candidate: its score (ref :155-158) and tree go to global scratch
                int finite = err ? 0 : 1;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int i = lane + 64 * q;
                    double v = 0.0;
                    if (i < l && !err) {
                        const int j = cur[i];
                        if (i == 0 || i == root) v = j == 0 ? 0.0 : NEG_INF_D;
                        else if (j == 0 || j == i) v = NEG_INF_D;
                        else v = (double)arc[(long)i * L + j];
                        edge[i] = v;
                    }
                    if (__ballot(i < l && !(v > NEG_INF_D))) finite = 0;
                }
                WFENCE();
                double score = NEG_INF_D;
                if (finite) score = np_sum<4>(edge, l) + (double)arc[(long)root * L];
                uint32_t *tr = a.cand_tree + ((long)b * L + slot) * Lw;
                for (int wd = lane; wd < Lw; wd += 64) {
                    unsigned pk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = 4 * wd + e;
                        if (v < l && !err) pk |= (unsigned)(cur[v] & 0xFF) << (8 * e);
                    }
                    tr[wd] = pk;
                }
                if (lane == 0) {
                    a.cand_score[(long)b * L + slot] = score;
                    if (err) atomicOr(&a.cand_n[b], 0x10000);
                }
                __threadfence();  // this candidate's result before the count that may hand the sentence to a finisher
                int left = 0;
                if (lane == 0) left = atomicSub(&a.cand_left[b], 1);
                left = __shfl(left, 0);
                if (left == 1) {  // last candidate of the sentence: pick the winner in candidate order (ref :159)
                    __threadfence();
                    const int cn = ld_i32(&a.cand_n[b]);
                    const int r = cn & 0xFFFF;
                    double best = NEG_INF_D;
                    int bj = -1;
                    for (int j = 0; j < r; ++j) {
                        const double sj = ld_f64(&a.cand_score[(long)b * L + j]);
                        if (sj > best) { best = sj; bj = j; }
                    }
                    if (cn & 0x10000) {
                        if (lane == 0) a.status[b] = 2;
                    } else if (bj < 0) {
                        if (lane == 0) a.status[b] = PHONLP_MST_NO_FINITE_TREE;
                    } else {
                        const uint32_t *bt = a.cand_tree + ((long)b * L + bj) * Lw;
                        for (int wd = lane; wd < Lw; wd += 64) {
                            const unsigned pk = ld_u32(&bt[wd]);
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int v = 4 * wd + e;
                                if (v < l) hout[v] = (pk >> (8 * e)) & 0xFF;
                            }
                        }
                        if (lane == 0) a.status[b] = PHONLP_MST_OK;
                    }
                }
            }
            STAMP(8);
        }
        __syncthreads();
        }  // !skip
        if (slot < 0 && tid == 0) {
            __threadfence();  // this sentence's pushes before the count that tells waiters no more are coming from it
            atomicAdd(&a.ctr[CT_FIRST_DONE], 1);
        }
    }
#ifdef MST_STAMPS
    if (tid == 0 && a.dbg)
        for (int k = 0; k < 16; ++k) a.dbg[(long)wg * 16 + k] = stamp_acc[k];
#endif
#undef SM
}

// Clears the shared-candidate region at the head of a launch (hipMemsetAsync's fill kernel takes ~85 us for these
// 130 KB when other streams are busy; this one is a plain grid-stride store).
__global__ void mst_clear_kernel(int32_t *p, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}

// the queue, its counters, the fallback count and the per-sentence candidate counts start at zero in every call
int launch_mst_clear(const MstArgs &a, hipStream_t stream) {
    const int nz = (int)(mst_share_zero_bytes(a.B, a.L) / sizeof(int32_t));
    hipLaunchKernelGGL(mst_clear_kernel, dim3(ceil_div(nz, 1024) < 64 ? ceil_div(nz, 1024) : 64), dim3(256), 0, stream, a.ctr, nz);
    if (hipGetLastError() != hipSuccess) return fail(PHONLP_ERR_LAUNCH, "mst_one_root: cannot clear the candidate queue");
    return PHONLP_OK;
}

// a.fb_list == NULL: one workgroup per sentence.  Otherwise the sentences of the fallback list (count in ctr[CT_FB], known
// on the device only) on a grid of at most 16 workgroups (each needs a CU's whole LDS to itself before it can even look at the list: in the step a 64-workgroup launch with nothing to do took 20-50 us), each taking entries wg, wg + grid, ...: none, normally.
int launch_mst_wave(MstArgs a, hipStream_t stream) {
    const WaveLayout w = wave_layout(a.L);
    static PerDeviceOnce attr_once;  // the attribute is per device
    int dev;
    if (attr_once.pending(&dev)) {
        if (hipFuncSetAttribute((const void *)mst_wave_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)mst_wave_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)mst_wave_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "mst_one_root: cannot raise the dynamic LDS limit");
        attr_once.mark(dev);
    }
    const dim3 grid(a.fb_list ? (a.B < 16 ? a.B : 16) : a.B);
    if (a.L <= 64) hipLaunchKernelGGL(mst_wave_kernel<1>, grid, dim3(256), w.total, stream, a, w.stack_cap);
    else if (a.L <= 128) hipLaunchKernelGGL(mst_wave_kernel<2>, grid, dim3(256), w.total, stream, a, w.stack_cap);
    else hipLaunchKernelGGL(mst_wave_kernel<4>, grid, dim3(256), w.total, stream, a, w.stack_cap);
    return check_launch("mst_wave_kernel");
}

}  // namespace phonlp
