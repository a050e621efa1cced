// Chu-Liu-Edmonds single-root MST, round-based decode ("every cycle of a round contracted together").
//
// The reference (models/common/chuliu_edmonds.py:52-129) contracts ONE cycle per recursion level; mst_wave.hip /
// mst.hip replay that order bit for bit and need ~100 dependent levels per 128-word sentence.  This kernel finds the
// same tree with ~40 much cheaper rounds, and is used only where it PROVABLY returns the reference's tree:
//
//  (M) margin.  Every selection made here -- the best head of a row among the columns outside its component, the best
//      member row of a contracted column that is chosen, the best column of a new component, the best single-root
//      candidate -- beats every other option by MORE than tau = 64 l^2 2^-53 max|entry| (mst_margin(); entries: the
//      finite ones of rows >= 1 off the diagonal; NaN or +inf anywhere: fallback).
//
//  Why that suffices (u = 2^-53, A = max|entry|, every value anyone forms is a sum of at most l entries, so |value| <= lA):
//  1. Uniqueness with a margin.  If every node of the current contracted graph has a best in-edge that beats its other
//     in-edges by g, and those edges close a cycle C, any arborescence can be re-hung to contain all of C's edges but
//     one (keep the in-edge of the first member of C on a path from ROOT, give every other member its cycle edge), and
//     every edge changed gains at least g; arborescences that contain all of C but one edge correspond one to one,
//     weight for weight, to arborescences of the graph with C contracted (:84), where the other nodes' margins are
//     unchanged (the new column is a maximum over old columns) and the new node's in-edge and the member row behind a
//     chosen column are checked here.  By induction over the rounds (disjoint cycles of a round commute): the tree T*
//     decoded here is THE maximum arborescence, and every other arborescence weighs less by at least the smallest margin
//     seen, i.e. by more than tau minus this kernel's own rounding (see 2.).
//  2. The reference returns a near-maximum arborescence.  It compares stored float64 values exactly (np.argmax); what
//     rounds are the entries it forms per contraction, (s - e_m) + sum(e) (:73-84): at most (c + 4) u lA off per entry
//     for a cycle of c members.  A tree of the contracted graph uses ONE such entry, so by induction over its recursion
//     (exact Chu-Liu-Edmonds on the entries as stored, the optimum of a level equal to that of its exact contraction)
//     the tree it returns is within sum over levels of 2 (c + 4) u lA <= 12 l^2 u A of the maximum (the cycle sizes
//     add up to at most 2l, the levels to at most l).  The values compared HERE, W[i][j] + off[i], carry the roundings of
//     off[i]'s history: every merge adds (2c + 2) u lA on top of the errors of the OTHER members' sums (a row's own
//     selected value enters the cycle total and is subtracted again: it cancels), additive over disjoint components, so
//     <= 6 l^2 u A per value.  12 + 2 * 6 = 24 l^2 u A in all, a factor 2.7 below tau.  Hence it returns T*.  The same for each single-root candidate (:143-160),
//     whose tree scores (l-term float64 sums, <= l^2 u A off) it compares with strict '>': the pick here requires the
//     best candidate to lead by tau.
//
//  No exactness is assumed: scores of any dynamic range qualify (a trained parser's near-certain arcs, log-probabilities
//  from -1e-7 to -50 in one row, never fit a common grid).  A sentence that fails (M) anywhere -- exact ties above all:
//  integer-valued or zero matrices -- gets status PHONLP_MST_FALLBACK and is decoded by the reference-order kernel
//  afterwards (launch_mst_wave over the fallback list), whose tie-breaking IS the reference's.
//
// Formulation.  The matrix is never rewritten.  comp[i] = component (contracted node) of word i; off[i] = what the
// reference has added to row i's entries so far (for every member m of a contracted cycle: total - e_m, :84), which is
// the same for every column.  The contracted row of a component X -- the reference's metanode row, max over the rows i
// of X of W[i][j] + off[i] for every column j -- is kept as ONE BYTE per column: A[X][j] = the row that attains it
// (bit 7: attained by more than one row).  Merging a cycle is then, per column, a maximum over the members' bytes
// (two or three gathers per member, all columns in parallel), and the new component's in-edge an arg-max over the
// columns outside it.  A round: heads of the components -> cycles (full pointer doubling in round 0; afterwards a
// bounded walk from the components created in the previous round, since a new cycle must pass through one of them
// and a component whose path reaches ROOT stays that way for good) -> every cycle labelled, offsets updated -> the new
// components' rows merged and their in-edges chosen, four components per pass on 16 lanes each.  One wavefront runs
// the rounds without barriers (LDS operations of a wave execute in order); the whole workgroup does the two O(l^2)
// steps (load + exactness check, first arg-max of every row).
//
// Expansion (:105-128 in one sweep per round, backwards): per round the component of every word and the in-edge of
// every component BEFORE the merge are kept (3 bytes per word and round, in the workspace; staged back into LDS for the
// sweep); a component keeps the real in-edge of the
// component it was merged into if that edge enters one of its own words, and its own recorded in-edge otherwise.
//
// Single-root logic (:137-169) as in mst.hip's two-phase form: phase 1 decodes the unconstrained tree of every
// sentence; phase 2 decodes the candidates of the sentences with several children of ROOT on a B x 4 grid
// (set_root's edits :143-149 are applied while scanning, the matrix in LDS stays the original); the pick kernel takes
// the greatest score, lowest candidate number among equals (= strict '>' in candidate order, :159).
#include "mst.h"

namespace phonlp {

#define WFENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")

constexpr int kFastSlots = 4;
constexpr int kWalkMax = 4;
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;
enum { FS_LO = 0, FS_HI, FS_BAD, FS_FAIL, FS_NNEW, FS_CMD, FS_BAD2, FS_ROUNDS, FS_COUNT = 16 };
enum { CMD_GO = 1, CMD_DONE = 2 };
constexpr int HC_SETTLED = 0x8000, HC_NEW = 0x4000, HC_MASK = 0x3FFF;

struct FastLayout {
    size_t off_d, off_a, off_i, off_b, total;  // float64 arrays | A, later the staged records | int arrays | byte arrays
    int pitch, stage_recs;
};
constexpr int kFastIntArrays = 4;   // hcAw, cycf, newlist, bvL
constexpr int kFastByteArrays = 8;  // comp, bjL, selI, selJ, lab, pool, rI, rJ (word indices < 128)
// bytes of the A region: L rows of LA, and never fewer than sixteen expansion records (3 LA bytes each)
__host__ __device__ inline size_t fast_a_bytes(int L) {
    const size_t LA = (size_t)((L + 3) & ~3);
    const size_t a = (size_t)L * LA, m = 16 * 3 * LA;
    return ((a > m ? a : m) + 15) / 16 * 16;
}
// LDS per workgroup.  At L = 128: matrix 64 KB (pitch 128: the row scan staggers its column chunks instead of padding
// the rows), A 16 KB, per-word arrays 6 KB = 86 KB -- so that ONE 72 KB dense-layer block still fits the CU beside a
// decode (the expansion records, 49 KB at three bytes per word and round, live in the workspace and are staged into
// the A region, dead by then, for the expansion).
static FastLayout fast_layout(int L) {
    FastLayout f;
    const int LA = (L + 3) & ~3;
    f.pitch = L <= 64 ? 64 : 128;
    f.off_d = (size_t)L * f.pitch * sizeof(float);
    f.off_a = f.off_d + (size_t)3 * L * sizeof(double);
    f.stage_recs = (int)(fast_a_bytes(L) / (3 * (size_t)LA));
    f.off_i = f.off_a + fast_a_bytes(L);
    f.off_b = f.off_i + ((size_t)kFastIntArrays * L + FS_COUNT) * sizeof(int);
    f.total = align_up(f.off_b + (size_t)kFastByteArrays * LA, 16);
    return f;
}
bool mst_fast_fits(int L) { return L >= 2 && L <= 128; }

template <int CTRL>
__device__ __forceinline__ int fdpp_i(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ double fdpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = fdpp_i<CTRL>((int)(b & 0xFFFFFFFFll)), hi = fdpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
// Margin bookkeeping of every arg-max here: `dup` = "the runner-up is within tau of the maximum" (tau = 0: an exact
// tie).  Folding a value v into (bv, dup): a new maximum resets dup to (v - bv <= tau), anything else ORs in
// (bv - v <= tau); merging two partial results keeps the winner's flag and ORs in the distance of the two maxima.
// -inf operands give +inf or NaN differences, which compare false: "no edge" is never a runner-up.
//
// (value, index, near-duplicate) arg-max over the 16 lanes of a row
template <int CTRL>
__device__ __forceinline__ void fast_argmax_step(float &bv, int &bi, int &dup, float tau) {
    const float v2 = __int_as_float(fdpp_i<CTRL>(__float_as_int(bv)));
    const int i2 = fdpp_i<CTRL>(bi);
    const int d2 = fdpp_i<CTRL>(dup);
    const bool gt = v2 > bv;
    const int near = ((gt ? v2 - bv : bv - v2) <= tau) ? 1 : 0;
    dup = (gt ? d2 : dup) | near;
    bi = gt ? i2 : bi;
    bv = gt ? v2 : bv;
}
// the same for the contracted rows: (value, column | row byte << 8, near-duplicate)
__device__ __forceinline__ void fast_argmax_merge_d(double &bv, int &bi, int &dup, double v2, int i2, int d2, double tau) {
    const bool gt = v2 > bv;
    const int near = ((gt ? v2 - bv : bv - v2) <= tau) ? 1 : 0;
    dup = (gt ? d2 : dup) | near;
    bi = gt ? i2 : bi;
    bv = gt ? v2 : bv;
}
template <int CTRL>
__device__ __forceinline__ void fast_argmax_step_d(double &bv, int &bi, int &dup, double tau) {
    fast_argmax_merge_d(bv, bi, dup, fdpp_d<CTRL>(bv), fdpp_i<CTRL>(bi), fdpp_i<CTRL>(dup), tau);
}

// best column of one row among the columns other than `row` itself (and not column 0 when `skip0`), by 16 lanes
// The four rows a wavefront scans at once lie a multiple of 64 words apart (pitch 64 / 128), so row group `stag` takes
// its sixteen-column chunks in the order stag, stag + 1, ...: the four groups then read four different bank groups.
// The order is free: only a maximum that beats every other entry by more than tau is ever used (`dup` sends everything
// else to the fallback).
template <int NC>
__device__ __forceinline__ void fast_scan_row16(const float *Wrow, int row, int n, int sub, int stag, bool skip0, float tau,
                                                float &bv, int &bi, int &dup) {
    float vals[NC];
#pragma unroll
    for (int t = 0; t < NC; ++t) {
        const int j = sub + 16 * ((t + stag) & (NC - 1));
        vals[t] = j < n ? Wrow[j] : -__builtin_inff();
    }
    bv = -__builtin_inff();
    bi = INT_MAX;
    dup = 0;
#pragma unroll
    for (int t = 0; t < NC; ++t) {
        const int j = sub + 16 * ((t + stag) & (NC - 1));
        const bool ok = j < n && j != row && !(skip0 && j == 0);
        const float v = ok ? vals[t] : -__builtin_inff();
        const bool gt = v > bv;
        const int near = (ok && (gt ? v - bv : bv - v) <= tau) ? 1 : 0;  // fp32 differences of near values are exact
        dup = gt ? near : (dup | near);
        bi = gt ? j : bi;
        bv = gt ? v : bv;
    }
    fast_argmax_step<DPP_XOR1>(bv, bi, dup, tau);
    fast_argmax_step<DPP_XOR2>(bv, bi, dup, tau);
    fast_argmax_step<DPP_HALF_MIRROR>(bv, bi, dup, tau);
    fast_argmax_step<DPP_MIRROR>(bv, bi, dup, tau);
}

#ifdef MST_STAMPS
#define FCOUNT(k, v) do { if (tid == 0) stamp_acc[k] += (v); } while (0)
#else
#define FCOUNT(k, v) do {} while (0)
#endif

template <int NPL>
__global__ __launch_bounds__(256) void mst_fast_kernel(MstArgs a, int pitch, int stage_recs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sub = lane & 15;
#ifdef MST_STAMPS
    long long stamp_acc[16] = {0};
    long long stamp_t = clock64();
#endif
    const int L = a.L, mode = a.mode, CS = a.cslots;
    int b = mode == 2 ? blockIdx.x / CS : blockIdx.x;
    const int slot = mode == 2 ? blockIdx.x % CS : 0;
    int ncand = 1;
    if (mode == 2) {
        // The shared region of the reference-order kernel (its counters, candidate queue and per-sentence counts) is
        // cleared here, by slot 0 of every sentence, instead of by a launch of its own in front of phase 1: nothing
        // touches the region before the pick kernel, and that launch sat on the critical path of the step, waiting
        // ~80 us for a SIMD slot behind the GEMM of the main stream.
        if (slot == 0) {
            for (int t = tid; t < L; t += 256) a.tasks[(long)b * L + t] = 0;
            if (tid == 0) a.cand_left[b] = 0;
            if (b == 0 && tid < CT_COUNT) a.ctr[tid] = 0;
        }
        // Workgroups (r, slot) decode the r-th PENDING sentence (ascending), not sentence r: the few workgroups with a
        // decode to run are then the FIRST of the grid and are placed first -- a workgroup needs 86 KB of a CU, and
        // beside another stream's dense layers those are scarce; spread over the grid, the decodes queued behind
        // hundreds of workgroups that only come to leave again.
        // (five words of the scalar block `sc` below, free until the decode starts)
        int *sel = reinterpret_cast<int *>(smem_raw + (size_t)L * pitch * sizeof(float) + (size_t)3 * L * sizeof(double) +
                                           fast_a_bytes(L)) + 4 * L + 8;
        const int rank = b;
        if (tid < 5) sel[tid] = tid == 4 ? -1 : 0;
        __syncthreads();
        int before = 0;  // pending sentences in front of this chunk of 256
        for (int c0 = 0; c0 < a.B; c0 += 256) {
            const int t = c0 + tid;
            const bool pend = t < a.B && a.status[t] == PHONLP_MST_PENDING;
            const unsigned long long bal = __ballot(pend);
            if (lane == 0) sel[wave] = __popcll(bal);
            __syncthreads();
            int pre = before;
            for (int w = 0; w < wave; ++w) pre += sel[w];
            if (pend && pre + __popcll(bal & ((1ull << lane) - 1ull)) == rank) sel[4] = t;
            before += sel[0] + sel[1] + sel[2] + sel[3];
            __syncthreads();
            if (sel[4] >= 0 || before > rank) break;  // block-uniform
        }
        b = sel[4];
        if (b < 0) return;  // fewer than rank + 1 pending sentences
        ncand = a.cand_n[b];
        if (slot >= ncand) return;
    }
    int l = a.sentlens[b];
    if (l > L) l = L;
    int32_t *hout = a.heads + (long)b * L;
    if (mode != 2) {
        for (int t = tid; t < L; t += 256) hout[t] = 0;
        if (l <= 1) {  // l == 1: ROOT only -- the reference hits its broken assert (no candidate root)
            if (tid == 0) a.status[b] = l == 1 ? PHONLP_MST_NO_FINITE_TREE : PHONLP_MST_OK;
            return;
        }
    }
    float *W = reinterpret_cast<float *>(smem_raw);
    double *selV = reinterpret_cast<double *>(smem_raw + (size_t)L * pitch * sizeof(float));
    double *dlt = selV + L, *offL = dlt + L;
    unsigned char *A = reinterpret_cast<unsigned char *>(offL + L);
    const int LA = (L + 3) & ~3;  // row pitch of A
    int *hcAw = reinterpret_cast<int *>(A + fast_a_bytes(L));
    int *cycf = hcAw + L, *newlist = cycf + L;
    float *bvL = reinterpret_cast<float *>(newlist + L);
    int *sc = reinterpret_cast<int *>(bvL + L);
    unsigned char *comp = reinterpret_cast<unsigned char *>(sc + FS_COUNT);
    unsigned char *bjL = comp + LA, *selI = bjL + LA, *selJ = selI + LA, *lab = selJ + LA, *pool = lab + LA;
    unsigned char *rI = pool + LA, *rJ = rI + LA;
    // expansion records of this workgroup (workspace): record r = planes comp / in-edge row / in-edge column of LA bytes
    const int RS = 3 * LA;
    unsigned char *hist = a.hist + (long)blockIdx.x * a.hist_stride;
    const float *arc = a.arc + (long)b * L * L;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float NEGF = -__builtin_inff();

    // ---- whole workgroup: the matrix into LDS (fp32, as given) and, in phase 1, the scale of the margin: the largest
    // finite |entry| of the rows >= 1 off the diagonal (NaN and +inf have no place in the argument: fallback)
    if (tid == 0) { sc[FS_LO] = 0; sc[FS_BAD] = 0; }
    __syncthreads();
    {
        unsigned amax = 0;  // bits of the largest finite |entry| (non-negative floats order like their bits)
        int bad = 0;
        for (int i0 = wave; i0 < l; i0 += 64) {  // sixteen rows' loads in flight per wavefront: the HBM / L2 latency is paid twice
            float vals[16][NPL];
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int i = i0 + 4 * r, j = lane + 64 * q;
                    vals[r][q] = (i < l && j < l) ? arc[(long)i * L + j] : 0.f;
                }
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int i = i0 + 4 * r, j = lane + 64 * q;
                    if (i < l && j < l) {
                        W[i * pitch + j] = vals[r][q];
                        if (i >= 1 && i != j) {
                            const unsigned bits = __float_as_uint(vals[r][q]);
                            const unsigned mag = bits & 0x7FFFFFFFu;
                            if (mag >= 0x7F800000u) { if (mag != 0x7F800000u || !(bits >> 31)) bad = 1; }  // NaN, +inf (-inf = no edge)
                            else amax = mag > amax ? mag : amax;
                        }
                    }
                }
        }
        if (mode == 1) {
            if (amax) atomicMax(reinterpret_cast<unsigned *>(&sc[FS_LO]), amax);
            if (bad) sc[FS_BAD] = 1;
        }
    }
    __syncthreads();
    double tau;
    if (mode == 1) {
        STAMP(0);
        if (sc[FS_BAD]) {  // block-uniform
            if (tid == 0) {
                a.status[b] = PHONLP_MST_FALLBACK;
                a.cand_n[b] = 1;  // why (diagnostics; the pick kernel lists the sentence): 1 = NaN / +inf entries
            }
            return;
        }
        tau = mst_margin(l, __uint_as_float((unsigned)sc[FS_LO]));
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
        // ---- whole workgroup: first arg-max of every row (ref :58), component state
        if (tid == 0) { sc[FS_FAIL] = 0; sc[FS_CMD] = CMD_GO; sc[FS_BAD2] = 0; }
        for (int t = tid; t < l; t += 256) { comp[t] = t; cycf[t] = 0; lab[t] = t; offL[t] = 0.0; }
        for (int t = tid; t < l * (LA / 4); t += 256) {  // A[X][j] = X: a word is its own component's only row
            const int X = t / (LA / 4);
            reinterpret_cast<unsigned *>(A)[t] = 0x01010101u * (unsigned)X;
        }
        __syncthreads();
        for (int row = 1 + (tid >> 4); row < l; row += 16) {
            float bv; int bi, dup;
            fast_scan_row16<4 * NPL>(W + row * pitch, row, l, sub, (tid >> 4) & 3, skip0, tauf, bv, bi, dup);
            if (row == root) { bv = 0.f; bi = 0; dup = 0; }  // set_root: the row's only edge, weight 0 (:147-148)
            if (sub == 0) {
                bjL[row] = bi;
                bvL[row] = bv;
                selI[row] = row;
                selJ[row] = bi;
                selV[row] = (double)bv;
                if (dup || !(bv > NEGF)) sc[FS_FAIL] = 1;
            }
        }
        __syncthreads();
        STAMP(1);
        const int fail0 = sc[FS_FAIL];  // block-uniform: a row without a strictly best finite head
        int fail_ = fail0;
        int head[NPL];  // wavefront 0: the decoded heads of the lane's words
        // Contracted row and in-edge of the new components of a round, one component per wavefront (lane = column):
        // per column the best member row (:84,:88,:97; A holds the row, bit 7 = more than one), then the strictly best
        // column outside the component.  Run by all four wavefronts between two workgroup barriers.
        auto merge_pass = [&](int nnew) {
            int ccol[NPL];
#pragma unroll
            for (int q = 0; q < NPL; ++q) ccol[q] = (lane + 64 * q < l) ? comp[lane + 64 * q] : -1;
#pragma unroll 1
            for (int p = 0; p + wave < nnew; p += 4) {
                const int d = newlist[p + wave];
                const int R = d & 0xFF, po = (d >> 8) & 0xFF, c = (d >> 16) & 0xFF;
                double cv[NPL];
                int ca[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) { cv[q] = NEG_INF_D; ca[q] = 0; }
#pragma unroll 1
                for (int k = 0; k < c; k += 2) {  // two members per trip: their loads are in flight together
                    const bool two = k + 1 < c;
                    const int m0 = pool[po + k], m1 = pool[po + (two ? k + 1 : k)];
                    int a0[NPL], a1[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int j = (lane + 64 * q < l) ? lane + 64 * q : 0;
                        a0[q] = A[m0 * LA + j];
                        a1[q] = A[m1 * LA + j];
                    }
                    float w0[NPL], w1[NPL];
                    double o0[NPL], o1[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int j = (lane + 64 * q < l) ? lane + 64 * q : 0;
                        w0[q] = W[(a0[q] & 0x7F) * pitch + j];
                        o0[q] = offL[a0[q] & 0x7F];
                        w1[q] = W[(a1[q] & 0x7F) * pitch + j];
                        o1[q] = offL[a1[q] & 0x7F];
                    }
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {  // selects, not control flow
                        // bit 7 of the row byte: another member row comes within tau of the best one (checked if the
                        // column is chosen); a member's own byte brings its bit along
                        const double v0 = (double)w0[q] + o0[q];
                        const bool gt0 = v0 > cv[q];
                        const int n0 = ((gt0 ? v0 - cv[q] : cv[q] - v0) <= tau) ? 0x80 : 0;
                        ca[q] = (gt0 ? a0[q] : ca[q]) | n0;
                        cv[q] = gt0 ? v0 : cv[q];
                        const double v1 = two ? (double)w1[q] + o1[q] : NEG_INF_D;
                        const bool gt1 = v1 > cv[q];
                        const int n1 = ((gt1 ? v1 - cv[q] : cv[q] - v1) <= tau) ? 0x80 : 0;
                        ca[q] = (gt1 ? a1[q] : ca[q]) | n1;
                        cv[q] = gt1 ? v1 : cv[q];
                    }
                }
                double bv = NEG_INF_D;
                int bi = 0, dup = 0;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int j = lane + 64 * q;
                    if (j < l) A[R * LA + j] = (unsigned char)ca[q];
                    const bool ok = j < l && ccol[q] != R && !(skip0 && j == 0);
                    const double v = ok ? cv[q] : NEG_INF_D;
                    const bool gt = v > bv;
                    const int near = (ok && (gt ? v - bv : bv - v) <= tau) ? 1 : 0;
                    dup = gt ? near : (dup | near);
                    bi = gt ? (j | (ca[q] << 8)) : bi;
                    bv = gt ? v : bv;
                }
                // The wave's arg-max with its margin flag WITHOUT moving (value, index, flag) triples through six exchange
                // steps: the maximum alone is reduced (four DPP steps inside the 16-lane rows, the four rows' results read
                // as scalars), the winner is the first lane that holds it, and "the runner-up is within tau" = the winner's
                // own in-lane flag, or ANY other lane's best within tau of the maximum (a ballot).  The same predicate as the
                // pairwise merges' (a merge keeps the winner's flag and ORs in the distance of the two maxima: along the
                // winner's path that is every other lane's best against the maximum), a third of the cycles.
                double m = bv;
                m = fmax(m, fdpp_d<DPP_XOR1>(m));
                m = fmax(m, fdpp_d<DPP_XOR2>(m));
                m = fmax(m, fdpp_d<DPP_HALF_MIRROR>(m));
                m = fmax(m, fdpp_d<DPP_MIRROR>(m));
                {
                    const long long mb = __double_as_longlong(m);
                    const int lo = (int)(mb & 0xFFFFFFFFll), hi = (int)(mb >> 32);
                    double r4[4];
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) {
                        const int l4 = __builtin_amdgcn_readlane(lo, 16 * w4), h4 = __builtin_amdgcn_readlane(hi, 16 * w4);
                        r4[w4] = __longlong_as_double(((long long)h4 << 32) | (unsigned)l4);
                    }
                    m = fmax(fmax(r4[0], r4[1]), fmax(r4[2], r4[3]));
                }
                const unsigned long long wmask = __ballot(bv == m && bv > NEG_INF_D);
                const int wl = wmask ? __builtin_ctzll(wmask) : 0;  // wave-uniform
                const bool others_near = __ballot(lane != wl && (m - bv) <= tau) != 0ull;  // (-inf lanes: +inf or NaN, false)
                const int wbi = __builtin_amdgcn_readlane(bi, wl), wdup = __builtin_amdgcn_readlane(dup, wl);
                if (lane == 0) {
                    selI[R] = (wbi >> 8) & 0x7F;
                    selJ[R] = wbi & 0xFF;
                    selV[R] = m;
                    if (!wmask || wdup || others_near || (wbi & 0x8000)) sc[FS_BAD2] = 1;
                }
            }
        };
        if (!fail_ && wave != 0) {  // helpers: a merge pass whenever wavefront 0 calls one
            for (;;) {
                lds_barrier();
                if (sc[FS_CMD] == CMD_DONE) break;
                merge_pass(sc[FS_NNEW] >> 16);
                lds_barrier();
            }
        }
        if (!fail_ && wave == 0) {
            double off[NPL], selVr[NPL];
            int mycomp[NPL], selIr[NPL], selJr[NPL];
            bool alive[NPL], isnew[NPL], settled[NPL];
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int v = lane + 64 * q;
                alive[q] = v >= 1 && v < l;
                mycomp[q] = v;
                off[q] = 0.0;
                selIr[q] = v;
                selJr[q] = alive[q] ? bjL[v] : 0;
                selVr[q] = alive[q] ? (double)bvL[v] : 0.0;
                isnew[q] = false; settled[q] = false;
            }
            int rounds = 0;
            for (;;) {
                const int stamp = rounds + 1;
                // heads of the components; flags: its path is known to reach ROOT / it was created in the last round
                int hc[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) hc[q] = alive[q] ? comp[selJr[q]] : 0;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    if (v < l) hcAw[v] = alive[q] ? (hc[q] | (settled[q] ? HC_SETTLED : 0) | (isnew[q] ? HC_NEW : 0)) : HC_SETTLED;
                }
                if (lane == 0) sc[FS_NNEW] = 0;
                WFENCE();
                bool found[NPL];
                int mn[NPL], clen[NPL], vis[kWalkMax][NPL];
                double tot[NPL], vsv[kWalkMax][NPL];
                bool allwalk = rounds == 0;
                if (!allwalk) {
                    // A new cycle passes through a component created in the last round: bounded walk from those.  A walk ends when it is back at its start (cycle); at a
                    // component known to reach ROOT (so does this one); at another walker with a SMALLER index (if the two
                    // share a cycle, the cycle's smallest walker goes all the way round and labels every member); or at a
                    // node it has seen before (Brent's checkpoints: it circles a cycle it is not part of: once that one is contracted, whatever
                    // cycle forms through here passes through the new component).
                    int cur[NPL], chk[NPL];
                    bool act[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        found[q] = false; act[q] = isnew[q]; cur[q] = hc[q]; chk[q] = -1;
                        mn[q] = lane + 64 * q; tot[q] = selVr[q]; clen[q] = 1;
#pragma unroll
                        for (int s = 0; s < kWalkMax; ++s) { vis[s][q] = 0; vsv[s][q] = 0.0; }
                    }
#pragma unroll
                    for (int s = 0; s <= kWalkMax; ++s) {  // unrolled: the components passed are kept in registers
                        bool any = false;
                        const bool mark = (s & (s - 1)) == 0;  // s = 0, 1, 2, 4: move the checkpoint (Brent)
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {  // selects only: no control flow inside a step
                            if (!__ballot(act[q])) continue;  // wave-uniform: no walker in this half
                            const int self = lane + 64 * q;
                            const bool hit = act[q] && cur[q] == self, atroot = act[q] && cur[q] == 0;
                            const bool loop = act[q] && cur[q] == chk[q];  // circling a cycle this walker is not part of
                            found[q] = found[q] || hit;
                            settled[q] = settled[q] || atroot;
                            act[q] = act[q] && !hit && !atroot && !loop;
                            if (s == kWalkMax) { act[q] = false; continue; }  // out of steps: forgotten
                            chk[q] = mark ? cur[q] : chk[q];
                            const int at = act[q] ? cur[q] : 0;
                            const int w = hcAw[at];
                            const double sv = selV[at];
                            const bool st = act[q] && (w & HC_SETTLED);
                            const bool go = act[q] && !st && !((w & HC_NEW) && cur[q] < self);
                            settled[q] = settled[q] || st;
                            vis[s][q] = at; vsv[s][q] = sv;
                            mn[q] = (go && cur[q] < mn[q]) ? cur[q] : mn[q];
                            tot[q] = go ? tot[q] + sv : tot[q];
                            clen[q] += go ? 1 : 0;
                            cur[q] = go ? (w & HC_MASK) : cur[q];
                            act[q] = go;
                            any = any || go;
                        }
                        if (!__ballot(any)) break;
                    }
                    // Nothing found: either no cycle is left, or one is longer than the walk (or its new components all
                    // stopped short); the full pass decides.  A walk that ran out of steps is simply forgotten: the cycle
                    // it may be on stays where it is until a full pass sees it.
                    bool anyf = false;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) anyf = anyf || found[q];
                    if (!__ballot(anyf)) allwalk = true;
                }
                if (allwalk) {
                    // pointer doubling over all components: after >= l steps a pointer rests on its cycle, or on ROOT
                    int word[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) word[q] = alive[q] ? hc[q] : 0;
                    for (int span = 1; span < l; span <<= 1) {
                        int nw[NPL];
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int tn = word[q];
                            int y = __shfl(word[0], tn & 63);
#pragma unroll
                            for (int q2 = 1; q2 < NPL; ++q2) {
                                const int y2 = __shfl(word[q2], tn & 63);
                                if ((tn >> 6) == q2) y = y2;
                            }
                            nw[q] = y;
                        }
#pragma unroll
                        for (int q = 0; q < NPL; ++q) word[q] = nw[q];
                    }
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        if (alive[q]) {
                            if (word[q] != 0) cycf[word[q]] = -stamp;  // negative: "on a cycle", not yet labelled
                            else settled[q] = true;
                        }
                    }
                    WFENCE();
                    int cur[NPL];
                    bool act[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int v = lane + 64 * q;
                        found[q] = alive[q] && cycf[v] == -stamp;
                        act[q] = found[q]; cur[q] = hc[q]; mn[q] = v; tot[q] = selVr[q]; clen[q] = 1;
                    }
#pragma unroll 1
                    for (;;) {  // every member walks its own cycle: smallest member, sum of the cycle's edges (:73-75)
                        bool any = false;
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            act[q] = act[q] && cur[q] != lane + 64 * q;
                            const int at = act[q] ? cur[q] : 0;
                            const int w = hcAw[at];
                            const double sv = selV[at];
                            mn[q] = (act[q] && cur[q] < mn[q]) ? cur[q] : mn[q];
                            tot[q] = act[q] ? tot[q] + sv : tot[q];
                            clen[q] += act[q] ? 1 : 0;
                            cur[q] = act[q] ? (w & HC_MASK) : cur[q];
                            any = any || act[q];
                        }
                        if (!__ballot(any)) break;
                    }
                }
                STAMP(2);
                FCOUNT(9, allwalk ? 1 : 0);
                bool anyf = false;
#pragma unroll
                for (int q = 0; q < NPL; ++q) anyf = anyf || found[q];
                if (!__ballot(anyf)) break;  // no cycle left: the components' in-edges form the tree
                if (rounds >= L) { fail_ = 1; break; }  // cannot happen: every round removes a component
                // this round's record: components and in-edges BEFORE the merge (to the workspace: nothing waits for it)
                {
                    unsigned char *hr = hist + (long)rounds * RS;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int v = lane + 64 * q;
                        if (v < l) {
                            hr[v] = (unsigned char)mycomp[q];
                            hr[LA + v] = (unsigned char)selIr[q];
                            hr[2 * LA + v] = (unsigned char)selJr[q];
                        }
                    }
                }
                // Labels (smallest member) and row offsets total - e_m (:84) of every member of every cycle.  One lane per
                // cycle also lists the members for the merge below: the walker that found it (it is the only one), or, after
                // a full pass (every member walked), the member that is the label.
                bool writer[NPL];
                int woff[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    writer[q] = found[q] && (!allwalk || mn[q] == v);
                    woff[q] = 0;
                    if (found[q]) {
                        lab[v] = mn[q];
                        dlt[v] = tot[q] - selVr[q];
                        cycf[v] = stamp;
                    }
                    if (writer[q]) {
                        const int x = atomicAdd(&sc[FS_NNEW], (1 << 16) | clen[q]);  // (cycles so far, members so far)
                        woff[q] = x & 0xFFFF;
                        newlist[x >> 16] = mn[q] | (woff[q] << 8) | (clen[q] << 16);
                        pool[woff[q]] = v;
                    }
                }
                if (!allwalk) {  // the finder passed every other member on its walk: labels, offsets, member list from registers
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        if (!__ballot(found[q])) continue;
#pragma unroll
                        for (int s = 0; s < kWalkMax; ++s)
                            if (found[q] && s < clen[q] - 1) {
                                const int m = vis[s][q];
                                lab[m] = mn[q];
                                dlt[m] = tot[q] - vsv[s][q];
                                cycf[m] = stamp;
                                pool[woff[q] + 1 + s] = m;
                            }
                    }
                } else {  // after a full pass every member has walked and labelled itself; the label lists the members
                    int cur[NPL], k[NPL];
                    bool act[NPL];
#pragma unroll
                    for (int q = 0; q < NPL; ++q) { act[q] = writer[q]; cur[q] = hc[q]; k[q] = 1; }
#pragma unroll 1
                    for (;;) {
                        bool any = false;
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            act[q] = act[q] && cur[q] != lane + 64 * q;
                            const int at = act[q] ? cur[q] : 0;
                            const int w = hcAw[at];
                            if (act[q]) pool[woff[q] + k[q]] = at;
                            k[q] += act[q] ? 1 : 0;
                            cur[q] = act[q] ? (w & HC_MASK) : cur[q];
                            any = any || act[q];
                        }
                        if (!__ballot(any)) break;
                    }
                }
                WFENCE();
                STAMP(3);
                // every word of a merged component: new label, offset
                bool merged[NPL];
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    const bool row = v >= 1 && v < l;
                    const int c = row ? mycomp[q] : 0;
                    const int cf = cycf[c], lb = lab[c], cfv = cycf[row ? v : 0], lbv = lab[row ? v : 0];
                    const double dl = dlt[c];
                    merged[q] = row && cf == stamp;
                    off[q] = merged[q] ? off[q] + dl : off[q];
                    mycomp[q] = merged[q] ? lb : mycomp[q];
                    alive[q] = alive[q] && !(cfv == stamp && lbv != v);
                }
#pragma unroll
                for (int q = 0; q < NPL; ++q)
                    if (merged[q]) { comp[lane + 64 * q] = mycomp[q]; offL[lane + 64 * q] = off[q]; }
                WFENCE();
                STAMP(4);
                // ---- the new components' contracted rows and in-edges: one component per wavefront, all four at work
                lds_barrier();
                merge_pass(sc[FS_NNEW] >> 16);
                lds_barrier();
                const int nnew = sc[FS_NNEW] >> 16;
                const int bad = sc[FS_BAD2];
                WFENCE();
                STAMP(5);
                FCOUNT(10, nnew);
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    isnew[q] = merged[q] && mycomp[q] == v;
                    if (isnew[q]) {
                        selIr[q] = selI[v]; selJr[q] = selJ[v]; selVr[q] = selV[v];
                        settled[q] = false;
                    }
                }
                if (bad) { fail_ = 1; break; }
                ++rounds;
                STAMP(6);
                FCOUNT(8, 1);
            }
            if (!fail_) {  // the components of the final state: record `rounds`
                unsigned char *hr = hist + (long)rounds * RS;
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    if (v < l) hr[v] = (unsigned char)mycomp[q];
                }
            }
            if (lane == 0) {
                sc[FS_CMD] = CMD_DONE;  // releases the helper wavefronts
                sc[FS_ROUNDS] = fail_ ? -1 : rounds;
            }
            lds_barrier();
        }
        if (!fail0) {
            // ---- expansion: the real in-edge of every component, the rounds undone in reverse.  The records come back
            // from the workspace into the A region (dead now), `stage_recs` at a time, fetched by the whole workgroup.
            __syncthreads();  // wavefront 0's records have reached memory
            const int R = sc[FS_ROUNDS];  // block-uniform; -1: the rounds gave up
            unsigned char *curI = selI, *curJ = selJ, *nxtI = rI, *nxtJ = rJ;
            for (int top = R; top > 0;) {
                const int lo = top - (stage_recs - 1) > 0 ? top - (stage_recs - 1) : 0;
                {
                    const unsigned *src = reinterpret_cast<const unsigned *>(hist + (long)lo * RS);
                    unsigned *dst = reinterpret_cast<unsigned *>(A);
                    const int nd = (top - lo + 1) * (RS / 4);
                    for (int t0 = tid; t0 < nd; t0 += 16 * 256) {  // sixteen loads in flight per thread
                        unsigned v[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) v[u] = t0 + 256 * u < nd ? src[t0 + 256 * u] : 0u;
#pragma unroll
                        for (int u = 0; u < 16; ++u)
                            if (t0 + 256 * u < nd) dst[t0 + 256 * u] = v[u];
                    }
                }
                __syncthreads();
                if (wave == 0) {
                    int hc0[NPL], hy[NPL], hi_[NPL], hj_[NPL];  // a round's record for the lane's words, loaded one round ahead
                    auto load_record = [&](int s_) {
                        const unsigned char *hs = A + (s_ - lo) * RS, *hn = hs + RS;
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int v = (lane + 64 * q < l) ? lane + 64 * q : 0;
                            hc0[q] = hs[v]; hy[q] = hn[v]; hi_[q] = hs[LA + v]; hj_[q] = hs[2 * LA + v];
                        }
                    };
                    load_record(top - 1);
                    for (int s = top - 1; s >= lo; --s) {
                        const unsigned char *hs = A + (s - lo) * RS;
                        int c0[NPL], y[NPL], si[NPL], sj[NPL], ri[NPL], rj[NPL];
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            c0[q] = hc0[q]; y[q] = hy[q]; si[q] = hi_[q]; sj[q] = hj_[q];
                            ri[q] = curI[y[q]]; rj[q] = curJ[y[q]];
                        }
                        if (s > lo) load_record(s - 1);
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int v = lane + 64 * q;
                            const bool inside = hs[ri[q]] == v;
                            if (v >= 1 && v < l && c0[q] == v) {  // a component of this state
                                nxtI[v] = (unsigned char)(inside ? ri[q] : si[q]);
                                nxtJ[v] = (unsigned char)(inside ? rj[q] : sj[q]);
                            }
                        }
                        WFENCE();
                        unsigned char *t1 = curI; curI = nxtI; nxtI = t1;
                        unsigned char *t2 = curJ; curJ = nxtJ; nxtJ = t2;
                    }
                }
                top = lo;
                if (top > 0) __syncthreads();  // the next group of records overwrites this one
            }
            if (wave == 0 && R >= 0) {
#pragma unroll
                for (int q = 0; q < NPL; ++q) {
                    const int v = lane + 64 * q;
                    head[q] = (v >= 1 && v < l) ? curJ[v] : 0;
                }
                STAMP(7);
                FCOUNT(11, 1);
                if (mode == 1) {
                    // ---- single-root logic (ref :137-141): one child of ROOT -> done, else its children are candidates
                    int nroot = 0;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int v = lane + 64 * q;
                        const int isr = (v >= 1 && v < l && head[q] == 0) ? 1 : 0;
                        const unsigned long long bal = __ballot(isr);
                        if (isr) a.cand_root[(long)b * L + nroot + __popcll(bal & lt_mask)] = v;  // ascending
                        nroot += __popcll(bal);
                    }
                    if (nroot == 1) {
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int v = lane + 64 * q;
                            if (v < l) hout[v] = head[q];
                        }
                        if (lane == 0) a.status[b] = PHONLP_MST_OK;
                    } else if (nroot == 0) {
                        fail_ = 1;  // cannot happen for a tree; left to the reference-order kernel
                    } else if (lane == 0) {
                        a.cand_n[b] = nroot;
                        a.status[b] = PHONLP_MST_PENDING;
                    }
                } else {
                    // ---- a candidate's score (ref :155-158); its rounding (and the reference's, whose sum runs in another
                    // order) is far inside the margin the scores are compared with
                    double s = 0.0;
#pragma unroll
                    for (int q = 0; q < NPL; ++q) {
                        const int v = lane + 64 * q;
                        if (v >= 1 && v < l && v != root) s += (double)W[v * pitch + head[q]];
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
                    s += (double)W[root * pitch];
                    second_score = s > best_score ? best_score : (s > second_score ? s : second_score);
                    if (s > best_score) {  // the slot's best candidate; a runner-up within tau sends the sentence to the fallback
                        best_score = s;
                        best_ci = ci;
                        uint16_t *tr = a.cs_tree + ((long)b * CS + slot) * L;
#pragma unroll
                        for (int q = 0; q < NPL; ++q) {
                            const int v = lane + 64 * q;
                            if (v < l) tr[v] = (uint16_t)head[q];
                        }
                    }
                }
            }
            if (tid == 0) sc[FS_FAIL] = fail_;
        }
        __syncthreads();
        fail_ = sc[FS_FAIL];
        __syncthreads();
        if (fail_) { failed = 1; break; }
    }
#ifdef MST_STAMPS
    if (tid == 0 && a.dbg)
        for (int k = 0; k < 16; ++k) a.dbg[((long)(mode - 1) * a.B + blockIdx.x) * 16 + k] = stamp_acc[k];
#endif
    if (tid == 0) {
        if (mode == 1) {
            if (failed) {
                a.status[b] = PHONLP_MST_FALLBACK;
                a.cand_n[b] = 2;  // 2 = (U) or the round limit, unconstrained decode
            }
        } else {
            const long qx = (long)b * CS + slot;
            if (best_score - second_score <= tau) failed = 1;  // (-inf scores: the difference is +inf or NaN, both false)
            a.cs_score[qx] = failed ? NEG_INF_D : best_score;
            a.cs_idx[qx] = failed ? -2 : best_ci;
        }
    }
}

// The winner among a pending sentence's slots: the greatest score, if it beats every other candidate's by more than the
// margin (then the reference's comparison of its own rounded sums, strict '>' in candidate order :159, picks the same
// one); a slot that could not decide, or two candidates within the margin, send the sentence to the fallback.
__global__ void mst_fast_pick_kernel(MstArgs a) {
    const int b = blockIdx.x, tid = threadIdx.x, L = a.L, R = a.cslots;
    if (a.status[b] == PHONLP_MST_FALLBACK) {  // phase 1 gave the sentence up: list it (reason in cand_n)
        if (tid == 0) a.fb_list[atomicAdd(&a.ctr[CT_FB], 1)] = b | (a.cand_n[b] << 24);
        return;
    }
    if (a.status[b] != PHONLP_MST_PENDING) return;
    int l = a.sentlens[b];
    if (l > L) l = L;
    const int ns = a.cand_n[b] < R ? a.cand_n[b] : R;
    const double tau = a.tau[b];
    double best = NEG_INF_D, second = NEG_INF_D;
    int bs = -1, bad = 0;
    for (int s = 0; s < ns; ++s) {
        const double sc = a.cs_score[(long)b * R + s];
        const int ix = a.cs_idx[(long)b * R + s];
        if (ix < 0) bad = 1;
        else {
            second = sc > best ? best : (sc > second ? sc : second);
            if (sc > best) { best = sc; bs = s; }
        }
    }
    if (best - second <= tau) bad = 1;  // two candidates within the margin: the reference's order decides (strict '>', :159)
    if (bad || bs < 0) {
        if (tid == 0) {
            a.status[b] = PHONLP_MST_FALLBACK;
            a.fb_list[atomicAdd(&a.ctr[CT_FB], 1)] = b | (3 << 24);  // 3 = a candidate decode
        }
        return;
    }
    for (int t = tid; t < l; t += blockDim.x) a.heads[(long)b * L + t] = a.cs_tree[((long)b * R + bs) * L + t];
    if (tid == 0) a.status[b] = PHONLP_MST_OK;
}

int launch_mst_fast_pick(MstArgs a, hipStream_t stream) {  // for mst_round.hip, which follows the same slot protocol
    hipLaunchKernelGGL(mst_fast_pick_kernel, dim3(a.B), dim3(128), 0, stream, a);
    return check_launch("mst_fast_pick_kernel");
}

// expansion records of one workgroup: up to L rounds + the final state, three planes of LA bytes each
static size_t fast_hist_stride(int L) { return (size_t)(L + 1) * 3 * ((L + 3) & ~3); }
size_t mst_fast_bytes(int B, int L) {
    const size_t q = (size_t)B * kFastSlots;
    return align_up((size_t)B * sizeof(int32_t), 256) + align_up((size_t)B * sizeof(double), 256) + align_up(q * sizeof(double), 256) +
           align_up(q * sizeof(int32_t), 256) + align_up(q * L * sizeof(uint16_t), 256) +
           align_up(q * fast_hist_stride(L), 256);
}
void mst_fast_carve(MstArgs &a, void *base) {
    char *p = (char *)base;
    const size_t q = (size_t)a.B * kFastSlots;
    a.fb_list = (int32_t *)p;  p += align_up((size_t)a.B * sizeof(int32_t), 256);
    a.tau = (double *)p;       p += align_up((size_t)a.B * sizeof(double), 256);
    a.cs_score = (double *)p;  p += align_up(q * sizeof(double), 256);
    a.cs_idx = (int32_t *)p;   p += align_up(q * sizeof(int32_t), 256);
    a.cs_tree = (uint16_t *)p; p += align_up(q * a.L * sizeof(uint16_t), 256);
    a.hist = (unsigned char *)p;  // one region per workgroup of the larger (second) launch
    a.hist_stride = (long)fast_hist_stride(a.L);
    a.cslots = kFastSlots;
}

// phases 1-3 (phase 2 clears the shared region, a.ctr included; the pick kernel then builds the fallback list); the
// caller runs the reference-order kernel over that list afterwards
int launch_mst_fast(MstArgs a, hipStream_t stream) {
    const FastLayout f = fast_layout(a.L);
    static PerDeviceOnce attr_once;
    int dev;
    if (attr_once.pending(&dev)) {
        if (hipFuncSetAttribute((const void *)mst_fast_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)mst_fast_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "mst_one_root: cannot raise the dynamic LDS limit");
        attr_once.mark(dev);
    }
    for (int mode = 1; mode <= 2; ++mode) {
        a.mode = mode;
        const dim3 grid(mode == 1 ? a.B : a.B * kFastSlots);
        if (a.L <= 64) hipLaunchKernelGGL(mst_fast_kernel<1>, grid, dim3(256), f.total, stream, a, f.pitch, f.stage_recs);
        else hipLaunchKernelGGL(mst_fast_kernel<2>, grid, dim3(256), f.total, stream, a, f.pitch, f.stage_recs);
        const int rc = check_launch("mst_fast_kernel");
        if (rc) return rc;
    }
    hipLaunchKernelGGL(mst_fast_pick_kernel, dim3(a.B), dim3(128), 0, stream, a);
    return check_launch("mst_fast_pick_kernel");
}

}  // namespace phonlp
