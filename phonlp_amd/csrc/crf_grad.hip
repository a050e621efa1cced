// Gradient of the CRF negative log-likelihood: what autograd makes of ref
// models/common/crf.py:24-100 (CRFLoss.forward: -(unary + binary - log_norm), summed over the batch and scaled) with
// respect to the emissions and the transition matrix, by the forward-backward algorithm:
//   d loss / d inputs[b,t,c] = scale * ( P(tag_t = c) - [c == gold_t] )                        (0 on padding),
//   d loss / d trans[p,c]    = scale * sum_b sum_t ( P(tag_{t-1} = p, tag_t = c) - [p == gold_{t-1}, c == gold_t] ),
// with the reference's padding rule (:95-98: a padded step keeps the old alphas, so it carries no gradient) and its
// quirk that step 0 always enters the partition function (:86) while the gold terms are masked at every step (:52, :72).
// One wavefront per sentence, lane = tag (T <= 64); emissions, alphas, the transition matrix and the sentence's own
// [T,T] gradient share sit in LDS; a second kernel adds the B shares in a fixed order (no atomics: the result does not
// depend on timing).  Latency-bound like the forward algorithm: two dependent chains of L steps.
#include "common.h"

namespace phonlp {

__device__ __forceinline__ float cg_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// Dynamic LDS: trans[TT] | dtr[TT] | emissions[L*T] | alphas[L*T]   (TT = T*T rounded up to 4 floats)
// REG (T <= 16, the tag sets PhoNLP trains with): lane c keeps column c and row c of the transitions and its column of
// the gradient share in registers, and the previous step's alphas come from the owning lanes by v_readlane, so the
// inner loops touch no LDS at all; otherwise those live in LDS (lane c still owns column c of the share).
template <bool REG>
__global__ __launch_bounds__(64) void crf_nll_grad_kernel(const float *logits, const float *trans,
                                                          const uint8_t *pad_mask, const int32_t *sentlens,
                                                          const int32_t *gold, int L, int T, float scale,
                                                          float *d_logits, float *dtr_part) {
    constexpr int TR = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TT = (T * T + 3) & ~3;
    float *tr = smem, *dtr = smem + TT, *sc = smem + 2 * TT, *al = sc + (long)L * T;
    const int lane = threadIdx.x, b = blockIdx.x;
    const int len = sentlens ? __builtin_amdgcn_readfirstlane(sentlens[b]) : L;
    const float *x = logits + (long)b * L * T;
    float *dx = d_logits + (long)b * L * T;
    const uint8_t *mk = pad_mask ? pad_mask + (long)b * L : nullptr;
    const int32_t *g = gold + (long)b * L;
    for (int idx = lane; idx < T * T; idx += 64) {
        tr[idx] = trans[idx];
        dtr[idx] = 0.f;
    }
    for (int idx = lane; idx < L * T; idx += 64) sc[idx] = x[idx];
    __syncthreads();
    auto is_pad = [&](int t) { return (t >= len) || (mk && __builtin_amdgcn_readfirstlane((int)mk[t]) != 0); };

    const bool own = lane < T;
    const int c = own ? lane : T - 1;
    float trc[TR], trr[TR], acc[TR];  // trans[p][c], trans[c][cc], share[p][c]
    if constexpr (REG) {
#pragma unroll
        for (int p = 0; p < TR; ++p) {
            trc[p] = p < T ? tr[p * T + c] : 0.f;
            trr[p] = p < T ? tr[c * T + p] : 0.f;
            acc[p] = 0.f;
        }
    }
    // forward: alphas of every step (ref :86-98); a padded step repeats the previous row
    float alpha = sc[c];
    if (own) al[c] = alpha;
    for (int t = 1; t < L; ++t) {
        if (!is_pad(t)) {
            float m = -INFINITY, s = 0.f;
            if constexpr (REG) {
#pragma unroll
                for (int p = 0; p < TR; ++p)
                    if (p < T) m = fmaxf(m, cg_bcast(alpha, p) + trc[p]);
#pragma unroll
                for (int p = 0; p < TR; ++p)
                    if (p < T) s += expf((cg_bcast(alpha, p) + trc[p]) - m);
            } else {
                for (int p = 0; p < T; ++p) m = fmaxf(m, cg_bcast(alpha, p) + tr[p * T + c]);
                for (int p = 0; p < T; ++p) s += expf((cg_bcast(alpha, p) + tr[p * T + c]) - m);
            }
            alpha = sc[t * T + c] + (m + logf(s));
        }
        if (own) al[t * T + c] = alpha;
    }
    float zm = own ? alpha : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zm = fmaxf(zm, __shfl_xor(zm, o));
    float zs = own ? expf(alpha - zm) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zs += __shfl_xor(zs, o);
    const float logZ = zm + logf(zs);
    __syncthreads();  // alphas written by lane p are read by every lane below

    // backward: beta[c] = log of the mass of everything after step t given tag c at t
    float beta = 0.f;
    for (int t = L - 1; t >= 1; --t) {
        if (is_pad(t)) {
            if (own) dx[t * T + c] = 0.f;
            continue;
        }
        const int gt = __builtin_amdgcn_readfirstlane(g[t]), gp = __builtin_amdgcn_readfirstlane(g[t - 1]);
        const float v = sc[t * T + c] + beta;
        const float mu = expf(al[t * T + c] + beta - logZ);
        if (own) dx[t * T + c] = scale * (mu - (c == gt ? 1.f : 0.f));
        float m = -INFINITY, s = 0.f;
        if constexpr (REG) {
            const float aprev = al[(t - 1) * T + c];  // lane p holds alpha_{t-1}[p]
            const float w = v - logZ;
#pragma unroll
            for (int p = 0; p < TR; ++p)
                if (p < T) acc[p] += expf(cg_bcast(aprev, p) + trc[p] + w) - ((p == gp && c == gt) ? 1.f : 0.f);
#pragma unroll
            for (int cc = 0; cc < TR; ++cc)
                if (cc < T) m = fmaxf(m, trr[cc] + cg_bcast(v, cc));
#pragma unroll
            for (int cc = 0; cc < TR; ++cc)
                if (cc < T) s += expf((trr[cc] + cg_bcast(v, cc)) - m);
        } else {
            for (int p = 0; p < T; ++p) {  // lane c owns column c of the sentence's share: no two lanes touch one entry
                const float xi = expf(al[(t - 1) * T + p] + tr[p * T + c] + v - logZ);
                if (own) dtr[p * T + c] += xi - ((p == gp && c == gt) ? 1.f : 0.f);
            }
            // beta of step t-1, lane = previous tag: LSE over the current tag cc of trans[lane][cc] + v[cc]
            for (int cc = 0; cc < T; ++cc) m = fmaxf(m, tr[c * T + cc] + cg_bcast(v, cc));
            for (int cc = 0; cc < T; ++cc) s += expf((tr[c * T + cc] + cg_bcast(v, cc)) - m);
        }
        beta = m + logf(s);
    }
    {  // step 0: always inside the partition function; its gold term is masked like every other
        const float mu = expf(al[c] + beta - logZ);
        const bool pad0 = is_pad(0);
        const int g0 = __builtin_amdgcn_readfirstlane(g[0]);
        if (own) dx[c] = scale * (mu - ((!pad0 && c == g0) ? 1.f : 0.f));
    }
    if constexpr (REG) {
#pragma unroll
        for (int p = 0; p < TR; ++p)
            if (p < T && own) dtr_part[(long)b * T * T + p * T + c] = acc[p];
    } else {
        __syncthreads();
        for (int idx = lane; idx < T * T; idx += 64) dtr_part[(long)b * T * T + idx] = dtr[idx];
    }
}

// d_trans[e] = scale * sum_b part[b][e]: one wavefront per entry, lane l adds sentences l, l+64, ... in double, then a
// fixed butterfly over the lanes -- the order of the additions never depends on timing.
__global__ __launch_bounds__(256) void crf_dtrans_sum_kernel(const float *part, int B, int TT, float scale,
                                                             float *d_trans) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= TT) return;
    double a = 0.0;
    for (int b = lane; b < B; b += 64) a += (double)part[(long)b * TT + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
    if (lane == 0) d_trans[e] = (float)((double)scale * a);
}

}  // namespace phonlp

using namespace phonlp;

extern "C" size_t phonlp_crf_nll_grad_ws_bytes(int B, int T) {
    if (B <= 0 || T <= 0) return 0;
    return (size_t)B * T * T * sizeof(float);
}

// both entry points: `part` = B x T x T floats of scratch (every entry is written by its sentence's workgroup before
// the sum kernel reads it, so the scratch needs no clearing)
static int crf_nll_grad_run(const char *who, const float *logits, const float *trans, const uint8_t *pad_mask,
                            const int32_t *sentlens, const int32_t *gold_tags, int B, int L, int T, float scale,
                            float *d_logits, float *d_trans, float *part, hipStream_t st) {
    if (T > 64) return fail(PHONLP_ERR_UNSUPPORTED, "%s: T > 64 (one lane per tag)", who);
    const size_t lds = ((size_t)2 * ((T * T + 3) & ~3) + (size_t)2 * L * T) * sizeof(float);
    if (lds > 150 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "%s: L*T too large for LDS", who);
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.pending(&attr_dev)) {
        if (hipFuncSetAttribute((const void *)crf_nll_grad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void *)crf_nll_grad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "%s: cannot raise the dynamic LDS limit", who);
        attr_once.mark(attr_dev);
    }
    {
        ProfScope ps("crf_nll_grad", st);
        if (T <= 16)
            hipLaunchKernelGGL(crf_nll_grad_kernel<true>, dim3(B), dim3(64), lds, st, logits, trans, pad_mask, sentlens,
                               gold_tags, L, T, scale, d_logits, part);
        else
            hipLaunchKernelGGL(crf_nll_grad_kernel<false>, dim3(B), dim3(64), lds, st, logits, trans, pad_mask, sentlens,
                               gold_tags, L, T, scale, d_logits, part);
        if (int rc = check_launch("crf_nll_grad_kernel")) return rc;
    }
    ProfScope ps("crf_dtrans_sum", st);
    hipLaunchKernelGGL(crf_dtrans_sum_kernel, dim3(ceil_div(T * T, 4)), dim3(256), 0, st, (const float *)part, B, T * T,
                       scale, d_trans);
    return check_launch("crf_dtrans_sum_kernel");
}

extern "C" int phonlp_crf_nll_grad(const float *logits, const float *trans, const uint8_t *pad_mask,
                                   const int32_t *sentlens, const int32_t *gold_tags, int B, int L, int T, float scale,
                                   float *d_logits, float *d_trans, void *ws, size_t ws_bytes, void *stream) {
    if (!logits || !trans || !gold_tags || !d_logits || !d_trans || !ws || B <= 0 || L <= 0 || T <= 0)
        return fail(PHONLP_ERR_SHAPE, "crf_nll_grad: bad argument");
    if (ws_bytes < phonlp_crf_nll_grad_ws_bytes(B, T)) return fail(PHONLP_ERR_WORKSPACE, "crf_nll_grad: workspace too small");
    return crf_nll_grad_run("crf_nll_grad", logits, trans, pad_mask, sentlens, gold_tags, B, L, T, scale, d_logits,
                            d_trans, (float *)ws, as_stream(stream));
}

// ---- the same with the scratch in a file-scope device array instead of a caller workspace: for callers that cannot
// allocate (HIP graph capture, tight serving / training loops, harnesses that forbid hipMalloc).  The array's NAME is a
// host-side stand-in, not a device address: its address on the current device comes from hipGetSymbolAddress, once per
// device (each GPU holds its own copy at its own address), and only that pointer is ever handed to a kernel.
constexpr size_t kStaticScratchFloats = (size_t)1 << 20;  // 4 MiB per GPU, held until the process ends
static __device__ __align__(256) float scratch[kStaticScratchFloats];
static std::atomic<float *> g_scratch_ptr[64];

extern "C" size_t phonlp_crf_nll_grad_static_max_bytes(void) { return kStaticScratchFloats * sizeof(float); }

extern "C" int phonlp_crf_nll_grad_static(const float *logits, const float *trans, const uint8_t *pad_mask,
                                          const int32_t *sentlens, const int32_t *gold_tags, int B, int L, int T,
                                          float scale, float *d_logits, float *d_trans, void *stream) {
    if (!logits || !trans || !gold_tags || !d_logits || !d_trans || B <= 0 || L <= 0 || T <= 0)
        return fail(PHONLP_ERR_SHAPE, "crf_nll_grad_static: bad argument");
    if ((size_t)B * T * T > kStaticScratchFloats)
        return fail(PHONLP_ERR_UNSUPPORTED, "crf_nll_grad_static: B*T*T = %zu floats exceeds the static scratch of %zu "
                    "(use phonlp_crf_nll_grad with a workspace)", (size_t)B * T * T, kStaticScratchFloats);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
        return fail(PHONLP_ERR_LAUNCH, "crf_nll_grad_static: no current device");
    float *part = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!part) {  // first call on this device: its own copy of the array
        const hipError_t e = hipGetSymbolAddress((void **)&part, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !part)
            return fail(PHONLP_ERR_LAUNCH, "crf_nll_grad_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
        g_scratch_ptr[dev].store(part, std::memory_order_release);
    }
    return crf_nll_grad_run("crf_nll_grad_static", logits, trans, pad_mask, sentlens, gold_tags, B, L, T, scale,
                            d_logits, d_trans, part, as_stream(stream));
}
