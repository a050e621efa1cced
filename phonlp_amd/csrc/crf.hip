// NER CRF decoders: Viterbi (ref models/common/crf.py:103-126) and the forward algorithm
// (crf.py:79-100, log_sum_exp :129-138), one wavefront per sentence.
//
// Lane c owns tag c.  The previous step's trellis / alpha value of tag p is broadcast with
// v_readlane (p is wave-uniform); the T x T transition matrix and the sentence's emissions sit
// in LDS, so the serial chain over time steps touches no global memory.  Both recursions are
// dependent chains of length L: the bound is latency, not bandwidth.
#include "common.h"

namespace phonlp {

__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// The trellis steps with the lane's column of the transition matrix in registers and the loop over the previous tag
// unrolled (TP = T rounded up to 4; the columns beyond T hold -inf, which never wins a strict '>'): a step is TP readlane /
// add / compare triples, and the next step's emission is read while this one is computed (the loop over p with an LDS read
// per iteration cost ~1200 cycles a step at T = 9).  Same operations on the real tags in the same order: same bits.
template <int TP>
__device__ __forceinline__ float viterbi_steps(const float *sc, const float *tr, uint8_t *bp, float prev, int len, int T, int c,
                                               int lane) {
    float trc[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) trc[p] = p < T ? tr[p * T + c] : -INFINITY;
    float em = len > 1 ? sc[T + c] : 0.f;
    for (int t = 1; t < len; ++t) {
        const float em_next = t + 1 < len ? sc[(t + 1) * T + c] : 0.f;
        float bestv = lane_bcast(prev, 0) + trc[0];
        int arg = 0;
#pragma unroll
        for (int p = 1; p < TP; ++p) {
            const float v = lane_bcast(prev, p) + trc[p];
            if (v > bestv) { bestv = v; arg = p; }  // strict: first maximum wins (np.argmax)
        }
        prev = em + bestv;
        em = em_next;
        if (lane < T) bp[t * T + lane] = (uint8_t)arg;
    }
    return prev;
}

// The forward recursion's steps in the same form (columns beyond T: -inf, i.e. exp(.) = 0 terms at the end of the sum).
template <int TP>
__device__ __forceinline__ float crf_forward_steps(const float *sc, const float *tr, const uint8_t *mk, float alpha, int len, int L,
                                                   int T, int c) {
    float trc[TP];
#pragma unroll
    for (int p = 0; p < TP; ++p) trc[p] = p < T ? tr[p * T + c] : -INFINITY;
    for (int t = 1; t < L; ++t) {
        const bool pad = (t >= len) || (mk && __builtin_amdgcn_readfirstlane((int)mk[t]) != 0);
        if (pad) continue;  // ref :95-98 keeps the old alphas on padding
        float v[TP];
        float m = -INFINITY;
#pragma unroll
        for (int p = 0; p < TP; ++p) { v[p] = lane_bcast(alpha, p) + trc[p]; m = fmaxf(m, v[p]); }
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < TP; ++p) s += expf(v[p] - m);
        alpha = sc[t * T + c] + (m + logf(s));
    }
    return alpha;
}

// Dynamic LDS: trans[T*T] | per wave { emissions[L*T] f32 | backptr[L*T] u8 (Viterbi only) }
__global__ __launch_bounds__(256) void viterbi_kernel(const float *logits, const float *trans, const int32_t *sentlens,
                                                      int B, int L, int T, int wave_floats, int32_t *tags,
                                                      float *best_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tr = smem;
    const int nw = blockDim.x >> 6;
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) tr[idx] = trans[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * nw + wave;
    if (b >= B) return;
    float *sc = smem + ((T * T + 3) & ~3) + (long)wave * wave_floats;
    uint8_t *bp = reinterpret_cast<uint8_t *>(sc + L * T);
    int len = __builtin_amdgcn_readfirstlane(sentlens[b]);
    if (len > L) len = L;
    int32_t *trow = tags + (long)b * L;
    for (int t = (len > 0 ? len : 0) + lane; t < L; t += 64) trow[t] = 0;
    if (len <= 0) {
        if (best_out && lane == 0) best_out[b] = 0.f;
        return;
    }
    const float *x = logits + (long)b * L * T;
    for (int idx = lane; idx < len * T; idx += 64) sc[idx] = x[idx];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const int c = lane < T ? lane : T - 1;  // idle lanes shadow the last tag, results unused
    float prev = sc[c];
    if (T <= 16) {
        if (T <= 4) prev = viterbi_steps<4>(sc, tr, bp, prev, len, T, c, lane);
        else if (T <= 8) prev = viterbi_steps<8>(sc, tr, bp, prev, len, T, c, lane);
        else if (T <= 12) prev = viterbi_steps<12>(sc, tr, bp, prev, len, T, c, lane);
        else prev = viterbi_steps<16>(sc, tr, bp, prev, len, T, c, lane);
    } else {
        for (int t = 1; t < len; ++t) {
            float bestv = lane_bcast(prev, 0) + tr[c];
            int arg = 0;
            for (int p = 1; p < T; ++p) {
                const float v = lane_bcast(prev, p) + tr[p * T + c];
                if (v > bestv) { bestv = v; arg = p; }  // strict: first maximum wins (np.argmax)
            }
            prev = sc[t * T + c] + bestv;
            if (lane < T) bp[t * T + lane] = (uint8_t)arg;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // arg-max over the last trellis row, lowest index on ties
    float bv = lane < T ? prev : -INFINITY;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(bv, o);
        const int i2 = __shfl_xor(bi, o);
        if (v2 > bv || (v2 == bv && i2 < bi)) { bv = v2; bi = i2; }
    }
    if (bi >= T) bi = 0;  // only reachable when every score is -inf/NaN: np.argmax returns 0 for all -inf
    if (best_out && lane == 0) best_out[b] = bv;
    int tag = bi;
    if (lane == 0) trow[len - 1] = tag;
    for (int t = len - 1; t >= 1; --t) {
        tag = bp[t * T + tag];
        if (lane == 0) trow[t - 1] = tag;
    }
}

__global__ __launch_bounds__(256) void crf_forward_kernel(const float *logits, const float *trans,
                                                          const uint8_t *pad_mask, const int32_t *sentlens,
                                                          const int32_t *gold, int B, int L, int T, int wave_floats,
                                                          float *log_norm, float *unary, float *binary) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *tr = smem;
    const int nw = blockDim.x >> 6;
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) tr[idx] = trans[idx];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x * nw + wave;
    if (b >= B) return;
    float *sc = smem + ((T * T + 3) & ~3) + (long)wave * wave_floats;
    const int len = sentlens ? __builtin_amdgcn_readfirstlane(sentlens[b]) : L;
    const float *x = logits + (long)b * L * T;
    const uint8_t *mk = pad_mask ? pad_mask + (long)b * L : nullptr;
    for (int idx = lane; idx < L * T; idx += 64) sc[idx] = x[idx];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const int c = lane < T ? lane : T - 1;
    float alpha = sc[c];
    if (T <= 16) {
        if (T <= 4) alpha = crf_forward_steps<4>(sc, tr, mk, alpha, len, L, T, c);
        else if (T <= 8) alpha = crf_forward_steps<8>(sc, tr, mk, alpha, len, L, T, c);
        else if (T <= 12) alpha = crf_forward_steps<12>(sc, tr, mk, alpha, len, L, T, c);
        else alpha = crf_forward_steps<16>(sc, tr, mk, alpha, len, L, T, c);
    } else {
        for (int t = 1; t < L; ++t) {
            const bool pad = (t >= len) || (mk && __builtin_amdgcn_readfirstlane((int)mk[t]) != 0);
            if (pad) continue;  // ref :95-98 keeps the old alphas on padding
            float m = -INFINITY;
            for (int p = 0; p < T; ++p) m = fmaxf(m, lane_bcast(alpha, p) + tr[p * T + c]);
            float s = 0.f;
            for (int p = 0; p < T; ++p) s += expf((lane_bcast(alpha, p) + tr[p * T + c]) - m);
            alpha = sc[t * T + c] + (m + logf(s));
        }
    }
    float m = lane < T ? alpha : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = lane < T ? expf(alpha - m) : 0.f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) log_norm[b] = m + logf(s);

    if (gold && unary && binary) {  // ref :48-77, gold-path scores with the same padding mask
        const int32_t *g = gold + (long)b * L;
        float u = 0.f, bi = 0.f;
        for (int t = lane; t < L; t += 64) {
            const bool pad = (t >= len) || (mk && mk[t] != 0);
            if (pad) continue;
            const int gt = g[t];
            u += sc[t * T + gt];
            if (t >= 1) bi += tr[g[t - 1] * T + gt];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            u += __shfl_xor(u, o);
            bi += __shfl_xor(bi, o);
        }
        if (lane == 0) { unary[b] = u; binary[b] = bi; }
    }
}

static int plan_waves(int L, int T, bool backptr, int &wpb, int &wave_floats, size_t &lds) {
    const size_t per_wave = align_up((size_t)L * T * sizeof(float) + (backptr ? (size_t)L * T : 0), 16);
    const size_t head = (size_t)((T * T + 3) & ~3) * sizeof(float);
    wave_floats = (int)(per_wave / sizeof(float));
    wpb = 4;
    while (wpb > 1 && head + wpb * per_wave > 64 * 1024) wpb >>= 1;
    lds = head + wpb * per_wave;
    return lds <= 150 * 1024 ? 0 : -1;
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_crf_viterbi(const float *logits, const float *trans, const int32_t *sentlens, int B, int L,
                                  int T, int32_t *tags, float *best, void *stream) {
    if (!logits || !trans || !sentlens || !tags || B <= 0 || L <= 0 || T <= 0)
        return fail(PHONLP_ERR_SHAPE, "crf_viterbi: bad argument");
    if (T > 64) return fail(PHONLP_ERR_UNSUPPORTED, "crf_viterbi: T > 64 (one lane per tag)");
    int wpb, wave_floats;
    size_t lds;
    if (plan_waves(L, T, true, wpb, wave_floats, lds))
        return fail(PHONLP_ERR_UNSUPPORTED, "crf_viterbi: L*T too large for the LDS trellis");
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.pending(&attr_dev)) {
        if (hipFuncSetAttribute((const void *)viterbi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "crf_viterbi: cannot raise the dynamic LDS limit");
        attr_once.mark(attr_dev);
    }
    ProfScope ps("viterbi", as_stream(stream));
    hipLaunchKernelGGL(viterbi_kernel, dim3(ceil_div(B, wpb)), dim3(wpb * 64), lds, as_stream(stream), logits, trans,
                       sentlens, B, L, T, wave_floats, tags, best);
    return check_launch("viterbi_kernel");
}

extern "C" int phonlp_crf_log_partition(const float *logits, const float *trans, const uint8_t *pad_mask,
                                        const int32_t *sentlens, const int32_t *gold_tags, int B, int L, int T,
                                        float *log_norm, float *unary, float *binary, void *stream) {
    if (!logits || !trans || !log_norm || B <= 0 || L <= 0 || T <= 0)
        return fail(PHONLP_ERR_SHAPE, "crf_log_partition: bad argument");
    if (T > 64) return fail(PHONLP_ERR_UNSUPPORTED, "crf_log_partition: T > 64 (one lane per tag)");
    if (gold_tags && (!unary || !binary)) return fail(PHONLP_ERR_SHAPE, "crf_log_partition: gold tags need unary+binary");
    int wpb, wave_floats;
    size_t lds;
    if (plan_waves(L, T, false, wpb, wave_floats, lds))
        return fail(PHONLP_ERR_UNSUPPORTED, "crf_log_partition: L*T too large for LDS");
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.pending(&attr_dev)) {
        if (hipFuncSetAttribute((const void *)crf_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "crf_log_partition: cannot raise the dynamic LDS limit");
        attr_once.mark(attr_dev);
    }
    ProfScope ps("crf_forward", as_stream(stream));
    hipLaunchKernelGGL(crf_forward_kernel, dim3(ceil_div(B, wpb)), dim3(wpb * 64), lds, as_stream(stream), logits,
                       trans, pad_mask, sentlens, gold_tags, B, L, T, wave_floats, log_norm, unary, binary);
    return check_launch("crf_forward_kernel");
}
