// Arc scorers on engine 2, stages A and B of the three one-output scorers fused (ref: models/common/biaffine.py:21-37 per
// scorer, annotate_model.py:189-214 for the terms and the log-softmax).
//
// One workgroup = one sentence (L <= 128 words), four wavefronts, wavefront w = dependents 32w .. 32w+31.  Per scorer:
//
//   phase A  Ta^T[n][i] = sum_k Wb[n][k] X1[i][k]      (bilinear stage A, :27-29, taken TRANSPOSED: rows = the KC output
//            features n, columns = the wave's 32 dependents).  Operands are split images (f16 pieces h, l; three
//            v_mfma_f32_32x32x16_f16 per product as everywhere on engine 2): Wb, KC rows x KC, streamed from L2 through a
//            ring of three k16 stages by LDS-DMA; X1, the sentence's 128 rows of the first-layer image the W1/W2 layer
//            wrote (phonlp_linear_split_image).  The wave's accumulators are KC/32 tiles: a lane holds ONE dependent and
//            KC/2 values of n -- the row's maximum is an in-lane maximum plus one exchange between the half-waves, so
//            Ta is scaled by a power of two and split into f16 pieces right out of the registers;
//   phase B  S[i][j] = sum_n Ta[i][n] X2[j][n]          (:33), chunk by chunk of 32 features: the wave writes the chunk's
//            pieces into its own 4 KB LDS patch (eight-byte stores: four consecutive n per register group), reads them
//            back as MFMA row fragments and multiplies them with the heads' X2 fragments (ring of three 16 KB slabs).
//            Neither Ta (160 MB at the headline size) nor the fp32 first-layer features (320 MB) exist in memory.
//   terms    unlabeled + logsigmoid(lin sign(j - i)) - log((|j - i| - 1 - softplus(dist))^2 / 2 + 1), in the
//            reference's order (:192-208), accumulated in registers over the three scorers; diagonal -inf (:210-211).
//   softmax  over ALL L head columns, padding included (:214): a lane holds four columns of sixteen rows; maximum and
//            sum across the 32 lanes of a half-wave by shuffles; 128-byte row segments stored.
//
// Before: stage A as one batched GEMM on engine 1 (0.21 ms at B=256, L=128) + stage B with both fp32 operands split
// into bf16 pieces inside the kernel (0.16 ms, matrix pipe busy 0.16: ~360 conversion instructions per 24 MFMAs).
//
// Packed ragged batches: rows of sentence b at off[b] + i; a padded head column (len <= j < L) reads the ONE image row
// `pad` (what the first layer gives a zero hidden row); padded dependents are not scored.
#include <stdlib.h>

#include "linear_args.h"

namespace phonlp {

typedef _Float16 af16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 af16x4 __attribute__((ext_vector_type(4)));

#define ARC_MFMA(A, B, C) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(af16x8, A), __builtin_bit_cast(af16x8, B), C, 0, 0, 0)

__device__ __forceinline__ void arc_dma16(const unsigned char *gptr, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gptr) : "memory", "m0");
}

struct ArcSplitArgs {
    const unsigned char *XF;  // first-layer image: row pitch pitch_x bytes, segment (2s) = X1 of scorer s, (2s+1) = X2, KC*4 bytes each
    const float *x_inv;       // [rows_total] inverse scale of every image row
    const unsigned char *Wb;  // bilinear weights' image: row s*KC + n = Wbt_s[n][:] (zero rows for n >= KP), KC*4 bytes per row
    const float *wb_inv;      // [S*KC]
    float *out;               // [B, L, L] log-probabilities
    long pitch_x;
    int B, L, S, rows_total;
    int use_lin, use_dist;
    const int32_t *off, *sentlens;  // packed rows (else NULL)
    const unsigned char *pad;       // packed: image row of a padded head
    float pad_inv;
};

// logsigmoid / softplus exactly as torch defines them (F.logsigmoid; F.softplus beta=1, threshold=20).
__device__ __forceinline__ float arcs_log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float arcs_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }

template <int NT>
struct ArcSplitLayout {
    static constexpr int KC = 32 * NT;
    static constexpr int NLA = 8 + 2 * NT;          // wave-loads (16 rows x 64 bytes) of a phase-A stage: X1, then Wb
    static constexpr int NLW = (NLA + 3) / 4;       // per wavefront (the last ones may be dummies)
    static constexpr int STAGE_A = NLW * 4 * 1024;
    static constexpr int SLAB_B = 2 * 128 * 64;     // phase B: a k32 chunk of the heads' X2, two k16 stages of 128 rows
    static constexpr int RING_A = 3 * STAGE_A, RING_B = 3 * SLAB_B + 4 * 4096;
    static constexpr int RING = RING_A > RING_B ? RING_A : RING_B;
    static constexpr int TOTAL = RING + 4 * 32 * 4 + KC * 4;  // + the waves' row factors + the scorer's weight-row scales
};

template <int NT>
__global__ __launch_bounds__(256, 1) void arc_f16x3_kernel(ArcSplitArgs a) {
    using Ly = ArcSplitLayout<NT>;
    constexpr int KC = Ly::KC, NLW = Ly::NLW, STAGE_A = Ly::STAGE_A, SLAB_B = Ly::SLAB_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x, L = a.L;
    const int len = a.off ? min(max(a.sentlens[b], 0), L) : L;  // dependents scored; heads j < len are real rows
    const long rowbase = a.off ? a.off[b] : (long)b * L;
    float *rowfac = reinterpret_cast<float *>(lds + Ly::RING) + wave * 32;
    float *wscale = reinterpret_cast<float *>(lds + Ly::RING) + 4 * 32;  // wb_inv of the current scorer's KC rows
    unsigned char *patch = lds + 3 * SLAB_B + wave * 4096;

    // fragment addressing inside a k16 stage image (rows of 64 bytes, chunk c of row R at slot c ^ ((R >> 2) & 3))
    const int sw = (r >> 2) & 3;
    const int off_h = ((2 * hh) ^ sw) << 4, off_l = ((2 * hh + 1) ^ sw) << 4;
    // DMA addressing: a wave-load is 16 rows x 64 bytes; lane = (row lr, slot lane & 3), fetching chunk slot ^ ((lr >> 2) & 3)
    const int lr = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    auto xrow = [&](int i) -> const unsigned char * {  // image row of word i of this sentence (clamped: masked later)
        long row = rowbase + i;
        row = row < a.rows_total ? row : a.rows_total - 1;
        return a.XF + row * a.pitch_x;
    };

    float cell[4][16];  // the sum of the scorers' terms: rows acc_row(e, lane) of the wave's tile, column 32 jt + r
#pragma unroll
    for (int jt = 0; jt < 4; ++jt)
#pragma unroll
        for (int e = 0; e < 16; ++e) cell[jt][e] = 0.f;
    float gj[4];  // inverse scale of the heads' image rows (the X2 side)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int j = 32 * jt + r;
        long row = rowbase + j;
        row = row < a.rows_total ? row : a.rows_total - 1;
        gj[jt] = (a.off && j >= len) ? a.pad_inv : a.x_inv[row];
    }
    const float xi = a.x_inv[min(rowbase + 32 * wave + r, (long)a.rows_total - 1)];  // the lane's dependent

#pragma unroll 1
    for (int s = 0; s < a.S; ++s) {
        // ------------------------------------------------------------------------------------------ phase A
        const unsigned char *srcA[NLW];
#pragma unroll
        for (int q = 0; q < NLW; ++q) {
            const int u = 4 * q + wave;
            if (u < 8) {
                srcA[q] = xrow(16 * u + lr) + (long)(2 * s) * KC * 4 + chunk * 16;
            } else {
                int n = 16 * (u - 8) + lr;
                n = n < KC ? n : KC - 1;  // dummy wave-loads (they only keep the waves' load counts equal)
                srcA[q] = a.Wb + ((long)s * KC + n) * (KC * 4) + chunk * 16;
            }
        }
        auto issueA = [&](int st, int buf) {
#pragma unroll
            for (int q = 0; q < NLW; ++q) arc_dma16(srcA[q] + (long)st * 64, lds0 + (unsigned)(buf * STAGE_A + q * 4096));
        };
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        constexpr int nstA = KC / 16;
        lds_barrier();  // everyone is done with the previous scorer's phase-B buffers (they alias this ring) and scales
        for (int t = tid; t < KC; t += 256) wscale[t] = a.wb_inv[s * KC + t];  // published by phase A's barriers
        issueA(0, 0);
        issueA(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLW) : "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;
#pragma unroll 1
        for (int st = 0; st < nstA; ++st) {
            const int nxt2 = cur == 0 ? 2 : cur - 1;  // (cur + 2) % 3: last read in iteration st - 1, behind that barrier
            if (st + 2 < nstA) issueA(st + 2, nxt2);
            const unsigned char *buf = lds + cur * STAGE_A;
            const u32x4 xh = *reinterpret_cast<const u32x4 *>(buf + (32 * wave + r) * 64 + off_h);
            const u32x4 xl = *reinterpret_cast<const u32x4 *>(buf + (32 * wave + r) * 64 + off_l);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const u32x4 wh = *reinterpret_cast<const u32x4 *>(buf + 8192 + (32 * t + r) * 64 + off_h);
                const u32x4 wl = *reinterpret_cast<const u32x4 *>(buf + 8192 + (32 * t + r) * 64 + off_l);
                acc[t] = ARC_MFMA(wl, xh, acc[t]);  // smallest terms first
                acc[t] = ARC_MFMA(wh, xl, acc[t]);
                acc[t] = ARC_MFMA(wh, xh, acc[t]);
            }
            if (st + 2 < nstA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
        // ------------------------------------------------------------------------------------------ phase B
        // the heads' X2 slabs: chunk c = two k16 stages of 128 rows, four wave-loads per wavefront
        const unsigned char *srcB[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 4 * q + wave;         // 0..15: stage u >> 3, rows 16 (u & 7) ..
            const int j = 16 * (u & 7) + lr;
            const unsigned char *rowp = (a.off && j >= len && j < L) ? a.pad : xrow(j);
            srcB[q] = rowp + (long)(2 * s + 1) * KC * 4 + (u >> 3) * 64 + chunk * 16;
        }
        auto issueB = [&](int c, int buf) {
#pragma unroll
            for (int q = 0; q < 4; ++q) arc_dma16(srcB[q] + (long)c * 128, lds0 + (unsigned)(buf * SLAB_B + q * 4096));
        };
        issueB(0, 0);
        if (NT > 1) issueB(1, 1);
        // Ta' = acc * wb_inv[n] (exact: powers of two); the row's maximum -> its power-of-two scale.  The weight rows'
        // scales come from LDS, tile by tile, here and again at the conversion (hoisted global loads of all KC of them
        // cost 200 registers)
        float amax = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 wi = *reinterpret_cast<const float4 *>(wscale + 32 * t + 8 * g + 4 * hh);
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc[t][4 * g + 0] * wi.x), fabsf(acc[t][4 * g + 1] * wi.y)),
                                         fmaxf(fabsf(acc[t][4 * g + 2] * wi.z), fabsf(acc[t][4 * g + 3] * wi.w))));
            }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        int E = (int)(__float_as_uint(amax) >> 23);
        E = E < 16 ? 16 : (E > 254 ? 254 : E);
        const float tscale = __uint_as_float((unsigned)(268 - E) << 23);  // maximum -> [2^14, 2^15)
        const float tinv = __uint_as_float((unsigned)(E - 14) << 23);
        if (hh == 0) rowfac[r] = tinv * xi;
        f32x16 acc2[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[jt][e] = 0.f;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            // the chunk's pieces into the wave's patch: row r, stage g >> 1, eight bytes of chunk 2 (g & 1) (+1: low piece)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                af16x4 h4, l4;
                const float4 wi4 = *reinterpret_cast<const float4 *>(wscale + 32 * c + 8 * g + 4 * hh);
                const float wi[4] = {wi4.x * tscale, wi4.y * tscale, wi4.z * tscale, wi4.w * tscale};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[c][4 * g + e] * wi[e];
                    h4[e] = (_Float16)v;
                    l4[e] = (_Float16)(v - (float)h4[e]);
                }
                unsigned char *p = patch + (g >> 1) * 2048 + r * 64 + 8 * hh;
                *reinterpret_cast<af16x4 *>(p + (((2 * (g & 1)) ^ sw) << 4)) = h4;
                *reinterpret_cast<af16x4 *>(p + (((2 * (g & 1) + 1) ^ sw) << 4)) = l4;
            }
            if (c + 1 < NT) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // slab c landed (c + 1 may be in flight)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            if (c + 2 < NT) issueB(c + 2, (c + 2) % 3);
            const unsigned char *slab = lds + (c % 3) * SLAB_B;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const u32x4 th = *reinterpret_cast<const u32x4 *>(patch + st * 2048 + r * 64 + off_h);
                const u32x4 tl = *reinterpret_cast<const u32x4 *>(patch + st * 2048 + r * 64 + off_l);
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {  // all four column tiles whatever L is: branch-free (columns >= L are masked)
                    const u32x4 qh = *reinterpret_cast<const u32x4 *>(slab + st * 8192 + (32 * jt + r) * 64 + off_h);
                    const u32x4 ql = *reinterpret_cast<const u32x4 *>(slab + st * 8192 + (32 * jt + r) * 64 + off_l);
                    acc2[jt] = ARC_MFMA(tl, qh, acc2[jt]);
                    acc2[jt] = ARC_MFMA(th, ql, acc2[jt]);
                    acc2[jt] = ARC_MFMA(th, qh, acc2[jt]);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the patch is rewritten for the next chunk
        }
        // ------------------------------------------------------------------------------------------ the scorer's term
        float fi[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) fi[e] = rowfac[acc_row(e, lane)];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int j = 32 * jt + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = 32 * wave + acc_row(e, lane);
                const float x = acc2[jt][e] * fi[e] * gj[jt];
                const int d = j - i;
                float v = cell[jt][e];
                if (s == 0) {
                    v = x;
                } else if (s == 1) {
                    if (a.use_lin) {
                        const float sg = d > 0 ? 1.f : (d < 0 ? -1.f : 0.f);
                        v += arcs_log_sigmoid(x * sg);
                    }
                } else if (a.use_dist) {
                    const float pred = 1.f + arcs_softplus(x);
                    const float dd = fabsf((float)d) - pred;
                    v += -logf(dd * dd / 2.f + 1.f);
                }
                cell[jt][e] = v;
            }
            __builtin_amdgcn_sched_barrier(0);  // one column tile's transcendentals at a time (register pressure)
        }
    }
    // ---------------------------------------------------------------------------------------------- log-softmax, store
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int i = 32 * wave + acc_row(e, lane);
        float v[4], m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int j = 32 * jt + r;
            v[jt] = (j < L) ? (j == i ? -INFINITY : cell[jt][e]) : -INFINITY;
            m = fmaxf(m, v[jt]);
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) sum += (32 * jt + r < L) ? expf(v[jt] - m) : 0.f;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) sum += __shfl_xor(sum, o);
        const float ls = logf(sum);
        if (i < len) {
            float *orow = a.out + ((long)b * L + i) * L;
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int j = 32 * jt + r;
                if (j < L) orow[j] = (v[jt] - m) - ls;
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // one row at a time
    }
}

int arc_split_kc(int H) {  // columns per image segment: the smallest supported multiple of 32 that holds H + 1 features
    const int need = H + 1;
    const int opts[] = {64, 128, 256, 416};
    for (int o : opts)
        if (need <= o) return o;
    return 0;
}

template <int NT>
static int launch_arc_split(const ArcSplitArgs &a, hipStream_t stream) {
    static PerDeviceOnce once;
    int dev;
    if (once.pending(&dev)) {
        if (hipFuncSetAttribute((const void *)arc_f16x3_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ArcSplitLayout<NT>::TOTAL) != hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "arc_scores_split: cannot raise the dynamic LDS limit");
        once.mark(dev);
    }
    ProfScope ps("arc_fused", stream);
    hipLaunchKernelGGL(arc_f16x3_kernel<NT>, dim3(a.B), dim3(256), ArcSplitLayout<NT>::TOTAL, stream, a);
    return check_launch("arc_f16x3_kernel");
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_arc_kc(int H) { return H > 0 ? arc_split_kc(H) : 0; }

extern "C" int phonlp_arc_scores_split(const void *XFp, const float *x_inv, int rows_total, int B, int L, int H,
                                       int n_scorers, int use_lin, int use_dist, const void *Wbp, const float *wb_inv,
                                       const int32_t *off, const int32_t *sentlens, const void *pad_row, float pad_inv,
                                       float *arc_logp, void *stream) {
    const int KC = H > 0 ? arc_split_kc(H) : 0;
    if (!XFp || !x_inv || !Wbp || !wb_inv || !arc_logp || B <= 0 || L <= 0 || rows_total <= 0 ||
        (n_scorers != 1 && n_scorers != 3))
        return fail(PHONLP_ERR_SHAPE, "arc_scores_split: bad argument");
    if (KC == 0 || L > 128) return fail(PHONLP_ERR_UNSUPPORTED, "arc_scores_split: needs H + 1 <= 416 and L <= 128");
    if (((uintptr_t)XFp & 15) || ((uintptr_t)Wbp & 15)) return fail(PHONLP_ERR_SHAPE, "arc_scores_split: images must be 16-byte aligned");
    if ((off != nullptr) != (sentlens != nullptr) || (off && !pad_row))
        return fail(PHONLP_ERR_SHAPE, "arc_scores_split: packed rows need off, sentlens and the pad row");
    if (!off && rows_total < B * L) return fail(PHONLP_ERR_SHAPE, "arc_scores_split: rows_total < B * L");
    ArcSplitArgs a{};
    a.XF = (const unsigned char *)XFp; a.x_inv = x_inv; a.Wb = (const unsigned char *)Wbp; a.wb_inv = wb_inv;
    a.out = arc_logp; a.pitch_x = (long)n_scorers * 2 * KC * 4;
    a.B = B; a.L = L; a.S = n_scorers; a.rows_total = rows_total;
    a.use_lin = use_lin; a.use_dist = use_dist;
    a.off = off; a.sentlens = sentlens; a.pad = (const unsigned char *)pad_row; a.pad_inv = pad_inv;
    switch (KC) {
        case 64: return launch_arc_split<2>(a, as_stream(stream));
        case 128: return launch_arc_split<4>(a, as_stream(stream));
        case 256: return launch_arc_split<8>(a, as_stream(stream));
        default: return launch_arc_split<13>(a, as_stream(stream));
    }
}
