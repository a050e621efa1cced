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

// Entity-level NER counts (ref models/ner/scorer.py:13-57 score_by_entity over ner/utils.py:85-128
// decode_from_bioes): counts[0..2] += {predicted entities also in gold, predicted entities, gold entities}.
// tag_kind[t]: 0 'O', 1 'B-', 2 'I-', 3 'E-', 4 'S-', anything else = a tag the decoder ignores (<PAD>, <UNK>, ...);
// tag_type[t]: id of the type string after the prefix.  One wavefront per sentence, 64 positions at a time, a lane per
// position.  The reference's state machine (an entity = first index, last index, type of its LAST tag) has only local
// dependencies once the ignored tags are skipped: with p = the last decoded position before i,
//   "an entity is open before i"  <=>  kind[p] is B or I;
//   the open entity's start       =   the last position <= p that is a B, or an I / E with nothing open before it;
//   i closes the open entity (start, p, type[p]) when it is an O, B or S (and the sentence's end closes one);
//   an E emits (start or i itself, i, type[i]), an S emits (i, i, type[i]).
// So a chunk needs two ballots (decoded positions, run starts), a few cross-lane reads, and the carry of the last decoded
// position's kind / type / index / run start from the chunk before.  Gold entities are recorded at their start index --
// unique per entity -- and the predicted ones looked up there.  (Round 2's form, lane 0 walking the sentence twice: 55 us
// at B=256, L=127.)
__global__ __launch_bounds__(64) void ner_entity_counts_kernel(const int32_t *pred, const int32_t *gold,
                                                               const int32_t *sentlens, const int32_t *tag_kind,
                                                               const int32_t *tag_type, int L, int T,
                                                               unsigned long long *counts) {
    extern __shared__ int sm[];
    int *ge = sm, *gt = sm + L;  // gold entity that starts at a position: its last index and type (-1: none)
    const int b = blockIdx.x, lane = threadIdx.x;
    int len = sentlens[b];
    if (len > L) len = L;
    for (int i = lane; i < len; i += 64) { ge[i] = -1; gt[i] = -1; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned long long below = (1ull << lane) - 1ull, upto = below | (1ull << lane);
    unsigned n_correct = 0, n_pred = 0, n_gold = 0;
    for (int pass = 0; pass < 2; ++pass) {  // gold first (fills ge / gt), then the prediction
        const int32_t *tags = (pass ? pred : gold) + (long)b * L;
        int c_kind = 0, c_type = -1, c_pos = -1, c_start = -1;  // carry: the last decoded position so far (none: like an O)
        for (int base = 0; base < len; base += 64) {
            const int i = base + lane;
            int kind = 5, ty = -1;
            if (i < len) {
                const int t = tags[i];
                if (t >= 0 && t < T) { kind = tag_kind[t]; ty = tag_type[t] & 0xFFFFFF; }
                if (kind < 0 || kind > 4) kind = 5;
            }
            const bool dec = kind <= 4;
            const unsigned long long mdec = __ballot(dec);
            // the decoded position before this one: in the chunk, or the carry
            const unsigned long long pm = mdec & below;
            const int pl = pm ? 63 - __builtin_clzll(pm) : 0;
            const int pk_in = __shfl(kind, pl), pt_in = __shfl(ty, pl);
            const int pkind = pm ? pk_in : c_kind, ptype = pm ? pt_in : c_type, ppos = pm ? base + pl : c_pos;
            const bool open_before = pkind == 1 || pkind == 2;
            const bool rs = dec && (kind == 1 || ((kind == 2 || kind == 3) && !open_before));  // a run starts here
            const unsigned long long mrs = __ballot(rs);
            const unsigned long long sm_ = mrs & upto;
            const int start = sm_ ? base + 63 - __builtin_clzll(sm_) : c_start;  // of the run this position is in (B / I / E)
            const int pstart_in = __shfl(start, pl);
            const int pstart = pm ? pstart_in : c_start;  // of the entity open before this position
            // emissions of this position: (a) it closes the open entity, (b) its own (E, S)
            const bool ea = dec && open_before && (kind == 0 || kind == 1 || kind == 4);
            const bool eb = kind == 3 || kind == 4;
            const int b_start = kind == 4 ? i : start;
            if (pass == 0) {
                if (ea) { ge[pstart] = ppos; gt[pstart] = ptype; }
                if (eb) { ge[b_start] = i; gt[b_start] = ty; }
                n_gold += __popcll(__ballot(ea)) + __popcll(__ballot(eb));
            } else {
                const int ia = ea ? pstart : 0, ib = eb ? b_start : 0;  // (no read outside the tables where nothing is emitted)
                const bool ca = ea && ge[ia] == ppos && gt[ia] == ptype;
                const bool cb = eb && ge[ib] == i && gt[ib] == ty;
                n_pred += __popcll(__ballot(ea)) + __popcll(__ballot(eb));
                n_correct += __popcll(__ballot(ca)) + __popcll(__ballot(cb));
            }
            if (mdec) {  // carry: the chunk's last decoded position
                const int ll = 63 - __builtin_clzll(mdec);
                c_kind = __shfl(kind, ll); c_type = __shfl(ty, ll); c_pos = base + ll; c_start = __shfl(start, ll);
            }
        }
        if (c_kind == 1 || c_kind == 2) {  // the sentence's end closes the open entity (wave-uniform)
            if (pass == 0) {
                if (lane == 0) { ge[c_start] = c_pos; gt[c_start] = c_type; }
                ++n_gold;
            } else {
                ++n_pred;
                n_correct += (ge[c_start] == c_pos && gt[c_start] == c_type) ? 1u : 0u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (lane == 0) {
        if (n_correct) atomicAdd(&counts[0], (unsigned long long)n_correct);
        if (n_pred) atomicAdd(&counts[1], (unsigned long long)n_pred);
        if (n_gold) atomicAdd(&counts[2], (unsigned long long)n_gold);
    }
}

// Device-side unsort (ref annotate_model.py:270,277,287,291-293 / models/common/utils.py:189-195): the batch's rows go
// straight to their place in a corpus-level table: dst[dest_row[b]][0 .. n) = src[b][col0 .. col0 + n), n = min(Ls - col0,
// Ld); the rest of the destination row is left as it is.  fail_count (optional) += number of non-zero status entries.
__global__ __launch_bounds__(256) void scatter_rows_kernel(const int32_t *src, int B, int Ls, int col0,
                                                           const int32_t *dest_row, int32_t *dst, int Ld, int n_dst,
                                                           const int32_t *status, int32_t *fail_count) {
    const int n = (Ls - col0) < Ld ? (Ls - col0) : Ld;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)B * n; idx += (long)gridDim.x * blockDim.x) {
        const int b = (int)(idx / n), c = (int)(idx % n);
        const int r = dest_row[b];
        if (r >= 0 && r < n_dst) dst[(long)r * Ld + c] = src[(long)b * Ls + col0 + c];
    }
    if (status && fail_count && blockIdx.x == 0 && threadIdx.x == 0) {
        int f = 0;
        for (int b = 0; b < B; ++b) f += status[b] != 0;
        if (f) atomicAdd(fail_count, f);
    }
}

// ---- ragged batches: the live rows (b, i), i < sentlens[b], of a [B, L, .] tensor packed one sentence after the other.
// off[b] = sum of sentlens[0..b) (off[B] = number of live rows), rowmap[off[b] + i] = b*L + i.  One block per sentence;
// every block sums the lengths before its own (B is a few thousand at most).
__global__ __launch_bounds__(256) void pack_descriptor_kernel(const int32_t *sentlens, int B, int L, int32_t *off,
                                                              int32_t *rowmap, int cap) {
    __shared__ int part[256];
    const int b = blockIdx.x;
    int s = 0;
    for (int k = threadIdx.x; k < b; k += 256) {
        const int l = sentlens[k];
        s += l < 0 ? 0 : (l > L ? L : l);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    const int o0 = part[0];
    if (b == B) {  // the extra block only writes the total
        if (threadIdx.x == 0) off[B] = o0;
        return;
    }
    int len = sentlens[b];
    len = len < 0 ? 0 : (len > L ? L : len);
    if (threadIdx.x == 0) off[b] = o0;
    for (int i = threadIdx.x; i < len; i += 256)
        if (o0 + i < cap) rowmap[o0 + i] = b * L + i;
}

// dst[p][0..W) = src[rowmap[p]][0..W) (fp32 rows, W % 4 == 0): one wavefront per row
__global__ __launch_bounds__(256) void pack_rows_kernel(const float *src, long ld_src, const int32_t *rowmap, int m, int W,
                                                        int n_src, float *dst) {
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (p >= m) return;
    const int r = rowmap[p];
    if (r < 0 || r >= n_src) return;
    const float *s = src + (long)r * ld_src;
    float *d = dst + (long)p * W;
    for (int c = lane * 4; c < W; c += 256) *reinterpret_cast<float4 *>(d + c) = *reinterpret_cast<const float4 *>(s + c);
}

// dst[rowmap[p]][0..W) = src[p][0..W) for 4-byte elements (dst zeroed by the caller beforehand)
__global__ __launch_bounds__(256) void unpack_rows_kernel(const uint32_t *src, const int32_t *rowmap, int m, int W, int n_dst,
                                                          uint32_t *dst) {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)m * W; idx += (long)gridDim.x * blockDim.x) {
        const int p = (int)(idx / W), c = (int)(idx - (long)p * W);
        const int r = rowmap[p];
        if (r >= 0 && r < n_dst) dst[(long)r * W + c] = src[idx];
    }
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_pack_descriptor(const int32_t *sentlens, int B, int L, int32_t *off, int32_t *rowmap, int cap,
                                      void *stream) {
    if (!sentlens || !off || !rowmap || B <= 0 || L <= 0 || cap <= 0)
        return fail(PHONLP_ERR_SHAPE, "pack_descriptor: bad argument");
    ProfScope ps("pack_descriptor", as_stream(stream));
    hipLaunchKernelGGL(pack_descriptor_kernel, dim3(B + 1), dim3(256), 0, as_stream(stream), sentlens, B, L, off, rowmap, cap);
    return check_launch("pack_descriptor_kernel");
}

extern "C" int phonlp_pack_rows(const float *src, int ld_src, int n_src, const int32_t *rowmap, int m, int W, float *dst,
                                void *stream) {
    if (!src || !rowmap || !dst || m <= 0 || W <= 0 || n_src <= 0) return fail(PHONLP_ERR_SHAPE, "pack_rows: bad argument");
    if ((W & 3) || (ld_src & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
        return fail(PHONLP_ERR_SHAPE, "pack_rows: W, ld_src must be multiples of 4 and src, dst 16-byte aligned");
    ProfScope ps("pack_rows", as_stream(stream));
    hipLaunchKernelGGL(pack_rows_kernel, dim3((m + 3) / 4), dim3(256), 0, as_stream(stream), src, (long)ld_src, rowmap, m, W,
                       n_src, dst);
    return check_launch("pack_rows_kernel");
}

extern "C" int phonlp_unpack_rows(const void *src, const int32_t *rowmap, int m, int W, void *dst, int n_dst, void *stream) {
    if (!src || !rowmap || !dst || m <= 0 || W <= 0 || n_dst <= 0) return fail(PHONLP_ERR_SHAPE, "unpack_rows: bad argument");
    hipStream_t st = as_stream(stream);
    ProfScope ps("unpack_rows", st);
    if (hipMemsetAsync(dst, 0, (size_t)n_dst * W * 4, st) != hipSuccess)
        return fail(PHONLP_ERR_LAUNCH, "unpack_rows: hipMemsetAsync failed");
    const long work = (long)m * W;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((unsigned)((work + 255) / 256 < 2048 ? (work + 255) / 256 : 2048)), dim3(256), 0,
                       st, (const uint32_t *)src, rowmap, m, W, n_dst, (uint32_t *)dst);
    return check_launch("unpack_rows_kernel");
}

extern "C" int phonlp_scatter_rows(const int32_t *src, int B, int Ls, int col0, const int32_t *dest_row, int32_t *dst,
                                   int Ld, int n_dst, const int32_t *status_opt, int32_t *fail_count_opt, void *stream) {
    if (!src || !dest_row || !dst || B <= 0 || Ls <= 0 || Ld <= 0 || col0 < 0 || col0 >= Ls || n_dst <= 0)
        return fail(PHONLP_ERR_SHAPE, "scatter_rows: bad argument");
    ProfScope ps("scatter_rows", as_stream(stream));
    const long work = (long)B * Ls;
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)((work + 255) / 256 < 512 ? (work + 255) / 256 : 512)), dim3(256),
                       0, as_stream(stream), src, B, Ls, col0, dest_row, dst, Ld, n_dst, status_opt, fail_count_opt);
    return check_launch("scatter_rows_kernel");
}

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

extern "C" int phonlp_ner_entity_counts(const int32_t *pred, const int32_t *gold, const int32_t *sentlens,
                                        const int32_t *tag_kind, const int32_t *tag_type, int B, int L, int T,
                                        int64_t *counts, void *stream) {
    if (!pred || !gold || !sentlens || !tag_kind || !tag_type || !counts || B <= 0 || L <= 0 || T <= 0)
        return fail(PHONLP_ERR_SHAPE, "ner_entity_counts: bad argument");
    if ((size_t)L * 4 * sizeof(int) > 64 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "ner_entity_counts: L > 4096");
    ProfScope ps("ner_entity_counts", as_stream(stream));
    hipLaunchKernelGGL(ner_entity_counts_kernel, dim3(B), dim3(64), (size_t)L * 2 * sizeof(int), as_stream(stream), pred,
                       gold, sentlens, tag_kind, tag_type, L, T, (unsigned long long *)counts);
    return check_launch("ner_entity_counts_kernel");
}
