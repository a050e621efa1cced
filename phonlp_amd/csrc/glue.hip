// Data movement on either side of the head path, HBM-bound:
//   * first-sub-word gather: ref annotate_model.py:119-123 / :159-163 runs one index_select per sentence in a
//     Python loop; here one launch gathers every word row of the batch;
//   * eval counts: POS accuracy (ref models/pos/scorer.py:29-36), UAS/LAS (utils/conll18_ud_eval.py:531-532) and
//     tag-level NER agreement as one int64[8] vector -- the only quantity that crosses GPUs (all-reduced).
#include "common.h"

namespace phonlp {

// out[b, w, :] = enc[b, first[b, w], :]; rows w >= sentlens[b] are written as zeros.
// One wavefront per output row, 16 bytes per lane per step (E % 4 == 0).
template <typename IdxT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *enc, const IdxT *first, const int32_t *sentlens,
                                                          int B, int S, int L, int E, float *out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)B * L) return;
    const int b = (int)(row / L), w = (int)(row % L);
    float4 *dst = reinterpret_cast<float4 *>(out + row * E);
    const bool live = !sentlens || w < sentlens[b];
    long src_row = 0;
    if (live) {
        src_row = (long)first[row];
        if (src_row < 0 || src_row >= S) src_row = 0;  // torch.index_select would raise; clamp instead of faulting
    }
    const float4 *src = reinterpret_cast<const float4 *>(enc + ((long)b * S + src_row) * E);
    for (int c = lane; c < E / 4; c += 64) dst[c] = live ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// counts[0..7] += {sentences, words, pos_correct, ner_correct, uas_correct, las_correct, mst_failures, 0}
__global__ __launch_bounds__(256) void eval_counts_kernel(const int32_t *pos, const int32_t *ner, const int32_t *heads,
                                                          const int32_t *rels, const int32_t *gpos,
                                                          const int32_t *gner, const int32_t *ghead,
                                                          const int32_t *grel, const int32_t *lens_tag,
                                                          const int32_t *lens_dep, const int32_t *mst_status, int B,
                                                          int Lt, int Ld, unsigned long long *counts) {
    unsigned long long c[7] = {0, 0, 0, 0, 0, 0, 0};
    const long stride = (long)gridDim.x * blockDim.x;
    const long tid0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (long idx = tid0; idx < (long)B * Lt; idx += stride) {
        const int b = (int)(idx / Lt), w = (int)(idx % Lt);
        if (w < lens_tag[b]) {
            c[1] += 1;
            c[2] += pos[idx] == gpos[idx];
            c[3] += ner[idx] == gner[idx];
        }
    }
    for (long idx = tid0; idx < (long)B * Ld; idx += stride) {
        const int b = (int)(idx / Ld), w = (int)(idx % Ld);
        if (w >= 1 && w < lens_dep[b]) {
            const bool u = heads[idx] == ghead[idx];
            c[4] += u;
            c[5] += u && rels[idx] == grel[idx];
        }
    }
    for (long idx = tid0; idx < B; idx += stride) {
        c[0] += 1;
        if (mst_status) c[6] += mst_status[idx] != 0;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        unsigned long long v = c[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&counts[k], v);
    }
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_gather_first_subword(const float *enc, const void *first_subword, int index_is_int64,
                                           const int32_t *sentlens, int B, int S, int L, int E, float *out,
                                           void *stream) {
    if (!enc || !first_subword || !out || B <= 0 || S <= 0 || L <= 0 || E <= 0)
        return fail(PHONLP_ERR_SHAPE, "gather_first_subword: bad argument");
    if ((E & 3) || ((uintptr_t)enc & 15) || ((uintptr_t)out & 15))
        return fail(PHONLP_ERR_SHAPE, "gather_first_subword: E must be a multiple of 4 and enc/out 16-byte aligned");
    ProfScope ps("gather_first_subword", as_stream(stream));
    const dim3 grid((unsigned)(((long)B * L + 3) / 4));
    if (index_is_int64)
        hipLaunchKernelGGL(gather_rows_kernel<long long>, grid, dim3(256), 0, as_stream(stream), enc,
                           (const long long *)first_subword, sentlens, B, S, L, E, out);
    else
        hipLaunchKernelGGL(gather_rows_kernel<int32_t>, grid, dim3(256), 0, as_stream(stream), enc,
                           (const int32_t *)first_subword, sentlens, B, S, L, E, out);
    return check_launch("gather_rows_kernel");
}

extern "C" int phonlp_eval_counts(const int32_t *pos, const int32_t *ner, const int32_t *heads, const int32_t *rels,
                                  const int32_t *gold_pos, const int32_t *gold_ner, const int32_t *gold_head,
                                  const int32_t *gold_rel, const int32_t *lens_tag, const int32_t *lens_dep,
                                  const int32_t *mst_status, int B, int Lt, int Ld, int64_t *counts, void *stream) {
    if (!pos || !ner || !heads || !rels || !gold_pos || !gold_ner || !gold_head || !gold_rel || !lens_tag ||
        !lens_dep || !counts || B <= 0 || Lt <= 0 || Ld <= 0)
        return fail(PHONLP_ERR_SHAPE, "eval_counts: bad argument");
    ProfScope ps("eval_counts", as_stream(stream));
    const long work = (long)B * (Lt > Ld ? Lt : Ld);
    const int blocks = (int)((work + 255) / 256 < 1024 ? (work + 255) / 256 : 1024);
    hipLaunchKernelGGL(eval_counts_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), pos, ner, heads, rels,
                       gold_pos, gold_ner, gold_head, gold_rel, lens_tag, lens_dep, mst_status, B, Lt, Ld,
                       (unsigned long long *)counts);
    return check_launch("eval_counts_kernel");
}
