// Shared declarations of the two Chu-Liu-Edmonds kernels (mst.hip: block-synchronous, any L <= 1024;
// mst_wave.hip: wave-synchronous level loop, used when the float64 matrix fits LDS).
#pragma once
#include <limits.h>
#include <stdlib.h>

#include "common.h"

namespace phonlp {

struct MstArgs {
    const float *arc;
    const int32_t *sentlens;
    int32_t *heads;
    int32_t *status;
    uint16_t *stack;
    double *gmat;
    long stack_stride;  // u16 per sentence
    long gmat_stride;   // doubles per sentence
    int B, L, pitch, lvl_stride;
    long long *dbg;  // diagnostic builds only (-DMST_STAMPS): 16 counters per sentence
    // mst_wave.hip: single-root candidates shared between the workgroups of one launch (mst_share_bytes())
    int32_t *ctr;          // [8]: started, first_done, q_tail, q_head  (zeroed per launch, with tasks and cand_left)
    int32_t *tasks;        // [B*L] queue of (sentence*256 + candidate slot) + 1; 0 = not written yet
    int32_t *cand_left;    // [B] candidates of the sentence still running
    int32_t *cand_n;       // [B] number of candidates (bit 16: a decode hit the level limit)
    int32_t *cand_root;    // [B][L] candidate roots, ascending
    double *cand_score;    // [B][L]
    uint32_t *cand_tree;   // [B][L][Lw] candidate trees, one byte per node, Lw = ceil(L/4) words
    // mst_fast.hip: sentences the round-based decode could not prove itself on (ties, inexact sums); count in ctr[CT_FB]
    int32_t *fb_list;      // [B] or NULL (mst_wave_kernel then decodes every sentence)
    // mst.hip, two-phase form for sentences too long for the wave kernel: mode 0 = everything in one workgroup per
    // sentence; mode 1 = the unconstrained decode only (a sentence with >= 2 children of ROOT is left PENDING with its
    // tree in tree0g); mode 2 = grid of B x cslots workgroups, workgroup (b, s) decodes candidates s, s + cslots, ...
    // of sentence b and leaves its best in cs_*; mst_pick_kernel then takes the first best over the slots.
    int mode, cslots;
    int only_fb;           // mst.hip: decode only the sentences whose status is PHONLP_MST_FALLBACK (after mst_round.hip)
    uint16_t *tree0g;      // [B][L] unconstrained tree of a PENDING sentence
    double *cs_score;      // [B][cslots] best score of the slot's candidates (-inf: none finite)
    int32_t *cs_idx;       // [B][cslots] candidate number of that best (ascending ROOT-child order), -2: decode error
    uint16_t *cs_tree;     // [B][cslots][L] its tree
    // mst_fast.hip: expansion records, one region of hist_stride bytes per workgroup
    unsigned char *hist;
    long hist_stride;
    // mst_fast.hip / mst_round.hip: the margin of the sentence (mst_margin), written by phase 1, read by phase 2 and the pick
    double *tau;           // [B]
};

#define PHONLP_MST_PENDING 3   // internal: between the phases of the two-phase form
#define PHONLP_MST_FALLBACK 4  // internal: mst_fast.hip hands the sentence to the reference-order kernel

// counters at the head of the shared region (zeroed per call): candidate queue tail / head, fallback count
enum { CT_STARTED = 0, CT_FIRST_DONE = 1, CT_Q_TAIL = 2, CT_Q_HEAD = 3, CT_FB = 5, CT_COUNT = 8 };

#ifdef MST_STAMPS
#define STAMP(k)                                              \
    do {                                                      \
        if (tid == 0) {                                       \
            const long long now_ = clock64();                 \
            stamp_acc[k] += now_ - stamp_t;                   \
            stamp_t = now_;                                   \
        }                                                     \
    } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

#define NEG_INF_D (-__builtin_inf())

// The margin of the round-based decoders (mst_fast.hip's header has the argument): a selection is used only if its
// maximum beats every other option by MORE than tau = 64 l^2 2^-53 max|entry|.  The reference's own float64 rounding
// moves the weight of the tree it returns by at most 12 l^2 2^-53 max|entry|, these kernels' own compared values are off
// by at most 6 l^2 2^-53 max|entry| each: 24 l^2 in all for a comparison, a factor 2.7 below tau.
__host__ __device__ inline double mst_margin(int l, float amax) {
    return 64.0 * (double)l * (double)l * 1.1102230246251565e-16 * (double)amax;
}

// numpy's float64 pairwise add-reduction order (verified against numpy 2.2, n = 1..1000).
__device__ inline double np_sum_block(const double *a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return res;
}
template <int DEPTH>
__device__ inline double np_sum(const double *a, int n) {
    if (n <= 128) return np_sum_block(a, n);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum<DEPTH - 1>(a, n2) + np_sum<DEPTH - 1>(a + n2, n - n2);
}
template <>
__device__ inline double np_sum<0>(const double *a, int n) { return np_sum_block(a, n); }


inline size_t mst_lvl_stride(int L) { return 4 * (size_t)L + 8; }
inline int mst_pitch(int L) { return L | 1; }

// mst_wave.hip
bool mst_wave_fits(int L);
size_t mst_share_bytes(int B, int L);
void mst_share_carve(MstArgs &a, void *base);
size_t mst_share_zero_bytes(int B, int L);  // leading part of the region that every launch clears
int launch_mst_clear(const MstArgs &a, hipStream_t stream);
int launch_mst_wave(MstArgs a, hipStream_t stream);

// mst_fast.hip
bool mst_fast_fits(int L);
size_t mst_fast_bytes(int B, int L);
void mst_fast_carve(MstArgs &a, void *base);
int launch_mst_fast(MstArgs a, hipStream_t stream);

// mst_round.hip
bool mst_round_fits(int L);
size_t mst_round_bytes(int B, int L);
void mst_round_carve(MstArgs &a, void *base);
int launch_mst_round(MstArgs a, hipStream_t stream);

}  // namespace phonlp
