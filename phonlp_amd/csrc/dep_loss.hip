// Training-side losses of the heads as forward values: the parser loss and, at the end of
// the file, the POS loss (the NER loss is the CRF NLL of crf.hip).
//
// Parser training loss, forward value: ref models/jointmodel/model.py:209-236, the
// eval=False branch of dep_forward, with every dropout the identity (model.eval(): dev-loss monitoring; no autograd).
//
//   loss = ( CE_arc + CE_label + CE_lin + D ) / number_of_words, each a SUM over words (crit_dep :93: reduction "sum",
//   ignore_index -1):
//     CE_arc   (:211-216) rows 1 <= i < len only; soft-max over the sentence's own columns j < len (word_mask, :212);
//     CE_label (:218-223) rows 1 <= i < len only; soft-max over the O relations of the scores at the GOLD head;
//     CE_lin   (:225-230) EVERY row i >= 1, padded ones included (the reference masks neither this nor D): two-class
//              cross-entropy of (-s/2, s/2), s = lin[i, head_i], target = (head_i > i);
//     D        (:232-234) EVERY row i >= 1: -dist_kld[i, head_i] = log((|head_i - i| - 1 - softplus(dist))^2 / 2 + 1).
//
// The kernels work from what the eval path already produces: the arc log-probs (a log-soft-max over ALL L columns,
// annotate_model.py:214) differ from the masked training scores by a per-row constant, so
//   CE_arc[i] = LSE_{j < len} arc_logp[i, j] - arc_logp[i, head_i];
// lin / dist come from the scorers' raw outputs (raw_opt of phonlp_deep_biaffine_arc), the label scores at the gold
// head from phonlp_label_project's T and XF exactly as phonlp_label_at_heads forms them.
// HBM-bound like labels-at-heads: 4*O*KP bytes of T per word.
#include "common.h"

namespace phonlp {

__device__ __forceinline__ float dl_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float dl_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// One block per word (b, i).  terms[k*M + word], k = 0 arc, 1 label, 2 lin, 3 dist (M = B*L); row 0 and, for the two
// masked terms, padded rows are written as 0.
__global__ __launch_bounds__(256) void dep_loss_terms_kernel(const float *arc_logp, const float *raw, const float *T,
                                                              const float *XF, const int32_t *gold_head,
                                                              const int32_t *gold_rel, const int32_t *sentlens, int B,
                                                              int L, int O, int KP, int use_lin, int use_dist,
                                                              float *terms) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xs = sm, *sc = sm + KP;  // the gold head's h2 row; the O label scores
    const int word = blockIdx.x;
    const long M = (long)B * L;
    const int b = word / L, i = word - b * L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (i == 0) {  // ROOT row: excluded from every term (:211, :218, :227, :233)
        if (tid < 4) terms[tid * M + word] = 0.f;
        return;
    }
    int len = sentlens[b];
    if (len > L) len = L;
    const bool live = i < len;
    int h = gold_head[word];
    if (h < 0 || h >= L) h = 0;  // torch.gather would raise; read column 0 instead of faulting
    // the two unmasked terms: one lane; on live rows it is a lane of wave 2, after the row of T has been streamed
    auto lin_dist = [&]() {
        float t_lin = 0.f, t_dist = 0.f;
        const long e = ((long)b * L + i) * L + h;
        if (use_lin) {
            const float s = raw[M * L + e];
            const float a0 = -s / 2, a1 = s / 2, m = fmaxf(a0, a1);
            t_lin = m + logf(expf(a0 - m) + expf(a1 - m)) - (h > i ? a1 : a0);
        }
        if (use_dist) {
            const float d = raw[2 * M * L + e];
            const float pred = 1.f + (d > 20.f ? d : log1pf(expf(d)));  // F.softplus, threshold 20
            const float diff = fabsf((float)(h - i)) - pred;
            t_dist = logf(diff * diff / 2 + 1);
        }
        terms[2 * M + word] = t_lin;
        terms[3 * M + word] = t_dist;
    };
    if (!live) {  // block-uniform
        if (tid < 2) terms[tid * M + word] = 0.f;
        if (tid == 64) lin_dist();
        return;
    }
    const int n4 = KP >> 2;
    const float4 *x2 = reinterpret_cast<const float4 *>(XF + ((long)b * L + h) * 2 * KP + KP);
    float4 *xs4 = reinterpret_cast<float4 *>(xs);
    for (int c = tid; c < n4; c += 256) xs4[c] = x2[c];
    __syncthreads();
    const float4 *Trow = reinterpret_cast<const float4 *>(T + (long)word * O * KP);
    for (int o = wave; o < O; o += 12) {  // three relations per trip, as label_at_heads_kernel
        float s[3] = {0.f, 0.f, 0.f};
        for (int c = lane; c < n4; c += 64) {
            const float4 xv = xs4[c];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int ou = o + 4 * u;
                if (ou < O) {
                    const float4 tv = Trow[(long)ou * n4 + c];
                    s[u] = fmaf(tv.x, xv.x, s[u]);
                    s[u] = fmaf(tv.y, xv.y, s[u]);
                    s[u] = fmaf(tv.z, xv.z, s[u]);
                    s[u] = fmaf(tv.w, xv.w, s[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ou = o + 4 * u;
            const float v = dl_wave_sum(s[u]);
            if (ou < O && lane == 0) sc[ou] = v;
        }
    }
    __syncthreads();
    if (wave == 0) {  // arc: LSE over the sentence's own columns of the eval log-probs, minus the gold entry
        const float *row = arc_logp + ((long)b * L + i) * L;
        float m = -INFINITY;
        for (int j = lane; j < len; j += 64) m = fmaxf(m, row[j]);
        m = dl_wave_max(m);
        float s = 0.f;
        for (int j = lane; j < len; j += 64) s += expf(row[j] - m);
        s = dl_wave_sum(s);
        if (lane == 0) terms[word] = m + logf(s) - (h < len ? row[h] : -INFINITY);
    } else if (wave == 1) {  // label: LSE over the relations at the gold head, minus the gold relation's score
        const int r = gold_rel[word];
        float m = -INFINITY;
        for (int o = lane; o < O; o += 64) m = fmaxf(m, sc[o]);
        m = dl_wave_max(m);
        float s = 0.f;
        for (int o = lane; o < O; o += 64) s += expf(sc[o] - m);
        s = dl_wave_sum(s);
        if (lane == 0) terms[M + word] = (r >= 0 && r < O) ? m + logf(s) - sc[r] : 0.f;  // -1 = ignore_index
    } else if (tid == 128) {
        lin_dist();
    }
}


// Backward of the parser loss w.r.t. the OUTPUTS of the four scorers: what autograd derives from
// models/jointmodel/model.py:209-236 (the terms added to the unlabeled scores are .detach()ed, :197 / :204, so the arc
// cross-entropy reaches the unlabeled scorer only).  One block per word (b, i), the structure of dep_loss_terms_kernel:
//   d_arc[b,i,j]   = (softmax_{j' < len}(arc[b,i,:])[j] - [j == head_i]) / nw    rows 1 <= i < len, columns j < len
//   d_label[b,i,o] = (softmax_o(label scores at the gold head)[o] - [o == rel_i]) / nw   rows 1 <= i < len
//   d_lin[b,i]     = (sigmoid(s) - [head_i > i]) / nw, s = lin[b,i,head_i]       EVERY row i >= 1 (unmasked, :225-230)
//   d_dist[b,i]    = -(|head_i - i| - pred) sigmoid(x) / ((|head_i - i| - pred)^2 / 2 + 1) / nw,
//                    pred = 1 + softplus(x), x = dist[b,i,head_i]                EVERY row i >= 1 (:232-234)
// and zero everywhere else (every entry of the four outputs is written).  The eval log-probs differ from the masked
// training scores by a per-row constant, which the soft-max over the sentence's own columns cancels.
__global__ __launch_bounds__(256) void dep_loss_grad_kernel(const float *arc_logp, const float *raw, const float *T,
                                                             const float *XF, const int32_t *gold_head,
                                                             const int32_t *gold_rel, const int32_t *sentlens, int B,
                                                             int L, int O, int KP, int use_lin, int use_dist, float inv_nw,
                                                             float *d_arc, float *d_label, float *d_lin, float *d_dist) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *xs = sm, *sc = sm + KP;  // the gold head's h2 row; the O label scores
    const int word = blockIdx.x;
    const long M = (long)B * L;
    const int b = word / L, i = word - b * L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *darow = d_arc + (long)word * L;
    float *dlrow = d_label + (long)word * O;
    int len = sentlens[b];
    if (len > L) len = L;
    const bool live = i >= 1 && i < len;
    int h = gold_head[word];
    if (h < 0 || h >= L) h = 0;
    if (tid == 64) {  // the two unmasked terms
        float g_lin = 0.f, g_dist = 0.f;
        if (i >= 1) {
            const long e = ((long)b * L + i) * L + h;
            if (use_lin) {
                const float s = raw[M * L + e];
                g_lin = (1.f / (1.f + expf(-s)) - (h > i ? 1.f : 0.f)) * inv_nw;
            }
            if (use_dist) {
                const float x = raw[2 * M * L + e];
                const float pred = 1.f + (x > 20.f ? x : log1pf(expf(x)));  // F.softplus, threshold 20
                const float diff = fabsf((float)(h - i)) - pred;
                const float sg = x > 20.f ? 1.f : 1.f / (1.f + expf(-x));   // d softplus / dx (1 beyond the threshold)
                g_dist = -diff * sg / (diff * diff / 2 + 1) * inv_nw;
            }
        }
        d_lin[word] = g_lin;
        d_dist[word] = g_dist;
    }
    if (!live) {  // block-uniform: ROOT row or padding
        for (int j = tid; j < L; j += 256) darow[j] = 0.f;
        for (int o = tid; o < O; o += 256) dlrow[o] = 0.f;
        return;
    }
    const int n4 = KP >> 2;
    const float4 *x2 = reinterpret_cast<const float4 *>(XF + ((long)b * L + h) * 2 * KP + KP);
    float4 *xs4 = reinterpret_cast<float4 *>(xs);
    for (int c = tid; c < n4; c += 256) xs4[c] = x2[c];
    __syncthreads();
    const float4 *Trow = reinterpret_cast<const float4 *>(T + (long)word * O * KP);
    for (int o = wave; o < O; o += 12) {  // three relations per trip, as dep_loss_terms_kernel
        float s[3] = {0.f, 0.f, 0.f};
        for (int c = lane; c < n4; c += 64) {
            const float4 xv = xs4[c];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int ou = o + 4 * u;
                if (ou < O) {
                    const float4 tv = Trow[(long)ou * n4 + c];
                    s[u] = fmaf(tv.x, xv.x, s[u]);
                    s[u] = fmaf(tv.y, xv.y, s[u]);
                    s[u] = fmaf(tv.z, xv.z, s[u]);
                    s[u] = fmaf(tv.w, xv.w, s[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ou = o + 4 * u;
            const float v = dl_wave_sum(s[u]);
            if (ou < O && lane == 0) sc[ou] = v;
        }
    }
    __syncthreads();
    if (wave == 0) {  // arc: soft-max over the sentence's own columns, minus the one-hot of the gold head
        const float *row = arc_logp + ((long)b * L + i) * L;
        float m = -INFINITY;
        for (int j = lane; j < len; j += 64) m = fmaxf(m, row[j]);
        m = dl_wave_max(m);
        float s = 0.f;
        for (int j = lane; j < len; j += 64) s += expf(row[j] - m);
        s = dl_wave_sum(s);
        const float is = 1.f / s;
        for (int j = lane; j < L; j += 64)
            darow[j] = j < len ? (expf(row[j] - m) * is - (j == h ? 1.f : 0.f)) * inv_nw : 0.f;
    } else if (wave == 1) {  // label: soft-max over the relations at the gold head, minus the one-hot of the gold relation
        const int r = gold_rel[word];
        const bool counted = r >= 0 && r < O;  // -1 = ignore_index
        float m = -INFINITY;
        for (int o = lane; o < O; o += 64) m = fmaxf(m, sc[o]);
        m = dl_wave_max(m);
        float s = 0.f;
        for (int o = lane; o < O; o += 64) s += expf(sc[o] - m);
        s = dl_wave_sum(s);
        const float is = 1.f / s;
        for (int o = lane; o < O; o += 64)
            dlrow[o] = counted ? (expf(sc[o] - m) * is - (o == r ? 1.f : 0.f)) * inv_nw : 0.f;
    }
}

// loss[0..3] = the four sums, loss[4] = their total / number_of_words.  One block of 1024 threads; double accumulators
// in a fixed order (thread t takes entries t, t+1024, ..., eight loads in flight; then a tree over the block), so the
// value does not depend on timing.
__global__ __launch_bounds__(1024) void dep_loss_sum_kernel(const float *terms, long M, float number_of_words,
                                                            float *loss) {
    __shared__ double part[4][1024];
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double a = 0.0;
        const float *t = terms + k * M;
#pragma unroll 8
        for (long e = tid; e < M; e += 1024) a += (double)t[e];
        part[k][tid] = a;
    }
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w)
#pragma unroll
            for (int k = 0; k < 4; ++k) part[k][tid] += part[k][tid + w];
        __syncthreads();
    }
    if (tid == 0) {
        double tot = 0.0;
        for (int k = 0; k < 4; ++k) {
            loss[k] = (float)part[k][0];
            tot += part[k][0];
        }
        loss[4] = (float)(tot / (double)number_of_words);
    }
}

// POS loss (ref models/jointmodel/model.py:128-135, crit_pos :94 = CrossEntropyLoss(ignore_index=1), MEAN over the
// words that count): one thread per word writes its cross-entropy and a 0/1 flag; rows at or beyond the sentence's
// length (pack() drops them) and words whose gold tag is ignore_index do not count.
__global__ __launch_bounds__(256) void pos_loss_terms_kernel(const float *logits, const int32_t *gold,
                                                              const int32_t *sentlens, int B, int L, int U,
                                                              int ignore_index, float *terms) {
    const long M = (long)B * L;
    const long word = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (word >= M) return;
    const int b = (int)(word / L), i = (int)(word - (long)b * L);
    const int g = gold[word];
    float ce = 0.f, flag = 0.f;
    if (i < sentlens[b] && g != ignore_index && g >= 0 && g < U) {
        const float *row = logits + word * U;
        float m = -INFINITY;
        for (int u = 0; u < U; ++u) m = fmaxf(m, row[u]);
        float s = 0.f;
        for (int u = 0; u < U; ++u) s += expf(row[u] - m);
        ce = m + logf(s) - row[g];
        flag = 1.f;
    }
    terms[word] = ce;
    terms[M + word] = flag;
}

// loss[0] = sum / count (NaN when nothing counts, as torch), loss[1] = sum, loss[2] = count; fixed summation order.
__global__ __launch_bounds__(1024) void pos_loss_sum_kernel(const float *terms, long M, float *loss) {
    __shared__ double part[2][1024];
    const int tid = threadIdx.x;
    double a = 0.0, n = 0.0;
#pragma unroll 8
    for (long e = tid; e < M; e += 1024) {
        a += (double)terms[e];
        n += (double)terms[M + e];
    }
    part[0][tid] = a;
    part[1][tid] = n;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) {
            part[0][tid] += part[0][tid + w];
            part[1][tid] += part[1][tid + w];
        }
        __syncthreads();
    }
    if (tid == 0) {
        loss[0] = (float)(part[0][0] / part[1][0]);
        loss[1] = (float)part[0][0];
        loss[2] = (float)part[1][0];
    }
}

// Gradient of the POS loss w.r.t. the logits (what autograd derives from crit_pos): (softmax - onehot) / count on the
// words that count, 0 elsewhere; count is read from the device (loss[2] of phonlp_pos_loss), so nothing synchronises.
__global__ __launch_bounds__(256) void pos_loss_grad_kernel(const float *logits, const int32_t *gold,
                                                             const int32_t *sentlens, const float *count, int B, int L,
                                                             int U, int ignore_index, float *d_logits) {
    const long M = (long)B * L;
    const long word = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (word >= M) return;
    const int b = (int)(word / L), i = (int)(word - (long)b * L);
    const int g = gold[word];
    const float *row = logits + word * U;
    float *out = d_logits + word * U;
    if (i < sentlens[b] && g != ignore_index && g >= 0 && g < U) {
        const float inv = 1.f / count[0];
        float m = -INFINITY;
        for (int u = 0; u < U; ++u) m = fmaxf(m, row[u]);
        float s = 0.f;
        for (int u = 0; u < U; ++u) s += expf(row[u] - m);
        const float is = 1.f / s;
        for (int u = 0; u < U; ++u) out[u] = (expf(row[u] - m) * is - (u == g ? 1.f : 0.f)) * inv;
    } else {
        for (int u = 0; u < U; ++u) out[u] = 0.f;
    }
}

}  // namespace phonlp

using namespace phonlp;

extern "C" size_t phonlp_dep_loss_ws_bytes(int B, int L) {
    if (B <= 0 || L <= 0) return 0;
    return (size_t)4 * B * L * sizeof(float);
}

extern "C" int phonlp_dep_loss(const float *arc_logp, const float *raw, const float *T, const float *XF,
                               const int32_t *gold_head, const int32_t *gold_rel, const int32_t *sentlens, int B,
                               int L, int H, int O, int use_lin, int use_dist, float number_of_words, float *loss,
                               void *ws, size_t ws_bytes, void *stream) {
    if (!arc_logp || !T || !XF || !gold_head || !gold_rel || !sentlens || !loss || !ws || B <= 0 || L <= 0 || H <= 0 ||
        O <= 0)
        return fail(PHONLP_ERR_SHAPE, "dep_loss: bad argument");
    if ((use_lin || use_dist) && !raw) return fail(PHONLP_ERR_SHAPE, "dep_loss: raw scorer outputs needed");
    if (!(number_of_words > 0.f)) return fail(PHONLP_ERR_SHAPE, "dep_loss: number_of_words must be positive");
    if (ws_bytes < phonlp_dep_loss_ws_bytes(B, L)) return fail(PHONLP_ERR_WORKSPACE, "dep_loss: workspace too small");
    if (((uintptr_t)T & 15) || ((uintptr_t)XF & 15) || ((uintptr_t)ws & 15))
        return fail(PHONLP_ERR_SHAPE, "dep_loss: T, XF and ws must be 16-byte aligned");
    const int KP = kp_of(H);
    const size_t lds = (size_t)(KP + O) * sizeof(float);
    if (lds > 64 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "dep_loss: H + O too large for one LDS row");
    hipStream_t st = as_stream(stream);
    float *terms = (float *)ws;
    {
        ProfScope ps("dep_loss_terms", st);
        hipLaunchKernelGGL(dep_loss_terms_kernel, dim3(B * L), dim3(256), lds, st, arc_logp, raw, T, XF, gold_head,
                           gold_rel, sentlens, B, L, O, KP, use_lin, use_dist, terms);
        if (int rc = check_launch("dep_loss_terms_kernel")) return rc;
    }
    ProfScope ps("dep_loss_sum", st);
    hipLaunchKernelGGL(dep_loss_sum_kernel, dim3(1), dim3(1024), 0, st, terms, (long)B * L, number_of_words, loss);
    return check_launch("dep_loss_sum_kernel");
}

extern "C" int phonlp_dep_loss_grad(const float *arc_logp, const float *raw, const float *T, const float *XF,
                                    const int32_t *gold_head, const int32_t *gold_rel, const int32_t *sentlens, int B,
                                    int L, int H, int O, int use_lin, int use_dist, float number_of_words, float *d_arc,
                                    float *d_label, float *d_lin, float *d_dist, void *stream) {
    if (!arc_logp || !T || !XF || !gold_head || !gold_rel || !sentlens || !d_arc || !d_label || !d_lin || !d_dist ||
        B <= 0 || L <= 0 || H <= 0 || O <= 0)
        return fail(PHONLP_ERR_SHAPE, "dep_loss_grad: bad argument");
    if ((use_lin || use_dist) && !raw) return fail(PHONLP_ERR_SHAPE, "dep_loss_grad: raw scorer outputs needed");
    if (!(number_of_words > 0.f)) return fail(PHONLP_ERR_SHAPE, "dep_loss_grad: number_of_words must be positive");
    if (((uintptr_t)T & 15) || ((uintptr_t)XF & 15)) return fail(PHONLP_ERR_SHAPE, "dep_loss_grad: T, XF must be 16-byte aligned");
    const int KP = kp_of(H);
    const size_t lds = (size_t)(KP + O) * sizeof(float);
    if (lds > 64 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "dep_loss_grad: H + O too large for one LDS row");
    hipStream_t st = as_stream(stream);
    ProfScope ps("dep_loss_grad", st);
    hipLaunchKernelGGL(dep_loss_grad_kernel, dim3(B * L), dim3(256), lds, st, arc_logp, raw, T, XF, gold_head, gold_rel,
                       sentlens, B, L, O, KP, use_lin, use_dist, 1.f / number_of_words, d_arc, d_label, d_lin, d_dist);
    return check_launch("dep_loss_grad_kernel");
}

extern "C" size_t phonlp_pos_loss_ws_bytes(int B, int L) {
    if (B <= 0 || L <= 0) return 0;
    return (size_t)2 * B * L * sizeof(float);
}

extern "C" int phonlp_pos_loss(const float *upos_logits, const int32_t *gold_upos, const int32_t *sentlens, int B, int L,
                               int U, int ignore_index, float *loss, void *ws, size_t ws_bytes, void *stream) {
    if (!upos_logits || !gold_upos || !sentlens || !loss || !ws || B <= 0 || L <= 0 || U <= 0)
        return fail(PHONLP_ERR_SHAPE, "pos_loss: bad argument");
    if (ws_bytes < phonlp_pos_loss_ws_bytes(B, L)) return fail(PHONLP_ERR_WORKSPACE, "pos_loss: workspace too small");
    hipStream_t st = as_stream(stream);
    float *terms = (float *)ws;
    const long M = (long)B * L;
    {
        ProfScope ps("pos_loss_terms", st);
        hipLaunchKernelGGL(pos_loss_terms_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, upos_logits,
                           gold_upos, sentlens, B, L, U, ignore_index, terms);
        if (int rc = check_launch("pos_loss_terms_kernel")) return rc;
    }
    ProfScope ps("pos_loss_sum", st);
    hipLaunchKernelGGL(pos_loss_sum_kernel, dim3(1), dim3(1024), 0, st, terms, M, loss);
    return check_launch("pos_loss_sum_kernel");
}

extern "C" int phonlp_pos_loss_grad(const float *upos_logits, const int32_t *gold_upos, const int32_t *sentlens,
                                    const float *count, int B, int L, int U, int ignore_index, float *d_logits,
                                    void *stream) {
    if (!upos_logits || !gold_upos || !sentlens || !count || !d_logits || B <= 0 || L <= 0 || U <= 0)
        return fail(PHONLP_ERR_SHAPE, "pos_loss_grad: bad argument");
    const long M = (long)B * L;
    ProfScope ps("pos_loss_grad", as_stream(stream));
    hipLaunchKernelGGL(pos_loss_grad_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, as_stream(stream),
                       upos_logits, gold_upos, sentlens, count, B, L, U, ignore_index, d_logits);
    return check_launch("pos_loss_grad_kernel");
}
