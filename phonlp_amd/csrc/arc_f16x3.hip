// Arc scorers on engine 2, stages A and B of the three one-output scorers fused (ref: models/common/biaffine.py:21-37 per
// scorer, annotate_model.py:189-214 for the terms and the log-softmax).
//
// One workgroup = 128 dependents of one sentence (L <= 512 words) against all its heads, eight wavefronts = (row tile of 32
// dependents) x (half: of the output features in phase A, of a group's head columns in phase B).  Per scorer:
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
//            back as MFMA row fragments and multiplies them with the heads' X2 fragments (ring of five 16 KB slabs, four in flight).
//            Neither Ta (160 MB at the headline size) nor the fp32 first-layer features (320 MB) exist in memory.
//   terms    unlabeled + logsigmoid(lin sign(j - i)) - log((|j - i| - 1 - softplus(dist))^2 / 2 + 1), in the
//            reference's order (:192-208), added scorer by scorer into the output rows (an L2-resident running sum: the
//            registers cannot hold three scorers' accumulators); diagonal -inf (:210-211).
//   softmax  over ALL L head columns, padding included (:214), on the finished rows: four lanes per row (L <= 128) or a
//            wavefront per row (longer sentences: head columns in groups of 128, NJG trips of phase B per scorer).
//
// Before: stage A as one batched GEMM on engine 1 (0.21 ms at B=256, L=128) + stage B with both fp32 operands split
// into bf16 pieces inside the kernel (0.16 ms, matrix pipe busy 0.16: ~360 conversion instructions per 24 MFMAs).
//
// Packed ragged batches: rows of sentence b at off[b] + i; a padded head column (len <= j < L) reads the ONE image row
// `pad` (what the first layer gives a zero hidden row); padded dependents are not scored.
#include <stdlib.h>

#include <type_traits>

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
    long long *dbg;                 // diagnostic builds only (-DARC_STAMPS): cycle stamps of block 0
    float *raw;                     // PER-SCORER form: [S - 1][B, L, L], the raw scores of scorers 1.. (scorer 0's go to `out`)
};

#ifdef ARC_STAMPS
#define ARC_STAMP(k) do { if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[k] = clock64(); } while (0)
#else
#define ARC_STAMP(k) do {} while (0)
#endif

// logsigmoid / softplus as torch defines them (F.logsigmoid; F.softplus beta=1, threshold=20), on the hardware's exp2 / log2
// (v_exp_f32 / v_log_f32, ~1 ulp each): one wavefront per SIMD has nothing to hide libm's expf / log1pf sequences
// behind -- with them the three scorers' terms and the softmax were 40 % of this kernel.  log1p(u) = log(y) - ((y - 1) - u) / y
// with y = fl(1 + u) keeps the accuracy of log1p for small u; absolute error of a term <= ~2e-7 (scores are O(1..10),
// the parity bound on them is 1e-3 relative).
__device__ __forceinline__ float arcs_log1p(float u) {
    const float y = 1.f + u;
    return __logf(y) - ((y - 1.f) - u) * __builtin_amdgcn_rcpf(y);
}
__device__ __forceinline__ float arcs_log_sigmoid(float x) { return fminf(x, 0.f) - arcs_log1p(__expf(-fabsf(x))); }
__device__ __forceinline__ float arcs_softplus(float x) { return x > 20.f ? x : arcs_log1p(__expf(x)); }

// the second and third scorers' terms (annotate_model.py:196-208) of a cell with j - i = d, shared by the fused kernel and
// the per-scorer form's combine kernel: the same operations in the same order, so both give the same bits
__device__ __forceinline__ float arc_term_lin(float x, int d) {
    const float sg = d > 0 ? 1.f : (d < 0 ? -1.f : 0.f);
    return arcs_log_sigmoid(__fmul_rn(x, sg));
}
__device__ __forceinline__ float arc_term_dist(float x, int d) {
    const float dd = __fsub_rn(fabsf((float)d), __fadd_rn(1.f, arcs_softplus(x)));
    return -__logf(__fmaf_rn(__fmul_rn(dd, dd), 0.5f, 1.f));
}

template <int NT>
struct ArcSplitLayout {
    static constexpr int KC = 32 * NT;
    static constexpr int NTW = (NT + 1) / 2;        // phase A: tiles of 32 output features per wavefront (two halves)
    static constexpr int NA = 2 * NTW;              // tiles computed (NT odd: the last one is a pad tile nobody uses)
    static constexpr int NLA = 8 + 2 * NA;          // wave-loads (16 rows x 64 bytes) of a phase-A stage: X1, then Wb
    static constexpr int NLW = (NLA + 7) / 8;       // per wavefront (the last ones may be dummies)
    static constexpr int STAGE_A = NLW * 8 * 1024;
    static constexpr int SLAB_B = 2 * 128 * 64;     // phase B: a k32 chunk of the heads' X2, two k16 stages of 128 rows
    static constexpr int NSLAB = 5;                 // phase B: slabs in the ring (NSLAB - 1 in flight: they come from HBM / MALL)
    static constexpr int RING_A = 3 * STAGE_A, RING_B = NSLAB * SLAB_B + 4 * 2 * 4096;  // B: the slabs + two patches per row tile
    static constexpr int RING = RING_A > RING_B ? RING_A : RING_B;
    static constexpr int TOTAL = RING + (4 * 32 + 2 * 4 * 32 + NA * 32) * 4;  // + row factors, row maxima, weight-row scales
};

// LONG = false: sentences of up to 128 words (one block per sentence, one group of head columns: the loops over them fold
// away, which the register allocation of the headline shape needs); true: up to 512.
// SKIP (short sentences, L <= 96: a variant of its own, so that the headline shape keeps its register allocation): the waves of
// row tiles and head columns beyond the sentence do not compute.
// PER_SCORER (few short sentences: fewer workgroups than a third of the CUs would otherwise run three scorers in a row): a workgroup
// per (sentence, scorer), blockIdx.y = the scorer; the raw scores go to `out` (scorer 0) / `raw` (the others) and
// arc_combine_kernel adds the terms in the fused form's order and takes the log-softmax.
template <int NT, bool LONG, bool SKIP = false, bool PER_SCORER = false>
__global__ __launch_bounds__(512, 1) void arc_f16x3_kernel(ArcSplitArgs a) {
    static_assert(!(LONG && PER_SCORER), "the per-scorer form is for sentences of up to 128 words");
    using Ly = ArcSplitLayout<NT>;
    constexpr int KC = Ly::KC, NTW = Ly::NTW, NLW = Ly::NLW, STAGE_A = Ly::STAGE_A, SLAB_B = Ly::SLAB_B;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // row tile (dependents 32 mt ..), half (phase A: output features; B: head columns).  Waves 2 mt and 2 mt + 1 share a row tile and
    // sit on different SIMDs (wave % 4), so a short sentence's live row tiles spread over the SIMDs: the waves of a row tile
    // beyond the sentence (`dead`) only take part in the loads and barriers, and with up to 64 words every SIMD runs ONE
    // computing wave instead of two (B = 1, L = 32: 0.14 -> 0.09 ms)
    const int mt = LONG ? (wave & 3) : (wave >> 1), nh = LONG ? (wave >> 2) : (wave & 1);  // (LONG: the mapping it was tuned with)
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.x, L = a.L;
    const int i0 = LONG ? 128 * blockIdx.y : 0;  // the block's dependents: i0 .. i0 + 127 (sentences longer than 128 words: several blocks)
    const int len = a.off ? min(max(a.sentlens[b], 0), L) : L;  // dependents scored; heads j < len are real rows
    if (i0 >= len) return;  // packed: nothing but padded dependents here (block-uniform)
    const bool dead = SKIP && i0 + 32 * mt >= len;  // wave-uniform
    const int NJG = LONG ? (L + 127) >> 7 : 1;   // groups of 128 head columns
    const long rowbase = a.off ? a.off[b] : (long)b * L;
    float *rowfac = reinterpret_cast<float *>(lds + Ly::RING);  // [4][32]
    float *rmax = rowfac + 4 * 32;                              // [2][4][32]
    float *wscale = rmax + 2 * 4 * 32;                          // wb_inv of the current scorer's rows
    unsigned char *patch = lds + Ly::NSLAB * SLAB_B + mt * 8192;  // the row tile's two Ta patches

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

    const float xi = a.x_inv[min(rowbase + i0 + 32 * mt + r, (long)a.rows_total - 1)];  // the lane's dependent
    float *obase = a.out + ((long)b * L + i0 + 32 * mt) * L;

    ARC_STAMP(0);
    const int s_first = PER_SCORER ? blockIdx.y : 0, s_end = PER_SCORER ? blockIdx.y + 1 : a.S;
#pragma unroll 1
    for (int s = s_first; s < s_end; ++s) {
        // ------------------------------------------------------------------------------------------ phase A
        const unsigned char *srcA[NLW];
#pragma unroll
        for (int q = 0; q < NLW; ++q) {
            const int u = 8 * q + wave;
            if (u < 8) {
                srcA[q] = xrow(i0 + 16 * u + lr) + (long)(2 * s) * KC * 4 + chunk * 16;
            } else {
                int n = 16 * (u - 8) + lr;
                n = n < KC ? n : KC - 1;  // the pad tile's rows and dummy wave-loads: any valid row
                srcA[q] = a.Wb + ((long)s * KC + n) * (KC * 4) + chunk * 16;
            }
        }
        auto issueA = [&](int st, int buf) {
#pragma unroll
            for (int q = 0; q < NLW; ++q) arc_dma16(srcA[q] + (long)st * 64, lds0 + (unsigned)(buf * STAGE_A + q * 8192));
        };
        f32x16 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        constexpr int nstA = KC / 16;
        __syncthreads();  // everyone is done with the previous scorer's phase-B buffers (they alias this ring) and scales
        for (int t = tid; t < KC; t += 512) wscale[t] = a.wb_inv[s * KC + t];  // published by phase A's barriers
        issueA(0, 0);
        issueA(1, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLW) : "memory");  // stage 0 landed
        __builtin_amdgcn_s_barrier();
        // A stage as pairs of tiles: the pair's six MFMAs (two accumulators alternating), the NEXT pair's fragment reads
        // in front of them, the stage's LDS-DMA issues spread behind the pairs (an issue costs the wave 60-180 cycles; the
        // SIMD's other wavefront computes meanwhile).
        constexpr int NP = (NTW + 1) / 2;
        auto stageA = [&](int st, int cur, auto issue) {
            const int nxt2 = cur == 0 ? 2 : cur - 1;  // (cur + 2) % 3: last read in iteration st - 1, behind that barrier
            const unsigned char *buf = lds + cur * STAGE_A;
            const unsigned char *wbuf = buf + 8192 + nh * (NTW * 2048);
            const unsigned ld = lds0 + (unsigned)(nxt2 * STAGE_A);
            const long so = (long)(st + 2) * 64;
            if (dead) {  // no dependents in this wave's row tile: its share of the loads only
                if constexpr (decltype(issue)::value) {
#pragma unroll
                    for (int q = 0; q < NLW; ++q) arc_dma16(srcA[q] + so, ld + q * 8192);
                }
                return;
            }
            const u32x4 xh = *reinterpret_cast<const u32x4 *>(buf + (32 * mt + r) * 64 + off_h);
            const u32x4 xl = *reinterpret_cast<const u32x4 *>(buf + (32 * mt + r) * 64 + off_l);
            u32x4 wh[2][2], wl[2][2];
            auto read_pair = [&](int p, int slot) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int t = 2 * p + k < NTW ? 2 * p + k : NTW - 1;
                    wh[slot][k] = *reinterpret_cast<const u32x4 *>(wbuf + (32 * t + r) * 64 + off_h);
                    wl[slot][k] = *reinterpret_cast<const u32x4 *>(wbuf + (32 * t + r) * 64 + off_l);
                }
            };
            read_pair(0, 0);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const int sl = p & 1, t0 = 2 * p, t1 = 2 * p + 1;
                if (p + 1 < NP) read_pair(p + 1, sl ^ 1);
                acc[t0] = ARC_MFMA(wl[sl][0], xh, acc[t0]);  // smallest terms first
                if (t1 < NTW) acc[t1] = ARC_MFMA(wl[sl][1], xh, acc[t1]);
                acc[t0] = ARC_MFMA(wh[sl][0], xl, acc[t0]);
                if (t1 < NTW) acc[t1] = ARC_MFMA(wh[sl][1], xl, acc[t1]);
                acc[t0] = ARC_MFMA(wh[sl][0], xh, acc[t0]);
                if (t1 < NTW) acc[t1] = ARC_MFMA(wh[sl][1], xh, acc[t1]);
                if constexpr (decltype(issue)::value) {  // the stage's NLW issues spread evenly over its NP pairs
#pragma unroll
                    for (int q = p * NLW / NP; q < (p + 1) * NLW / NP; ++q) arc_dma16(srcA[q] + so, ld + q * 8192);
                }
            }
        };
        int cur = 0, st = 0;
#pragma unroll 1
        for (; st + 2 < nstA; ++st) {
            stageA(st, cur, std::true_type{});
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLW) : "memory");  // stage st + 1 has landed (st + 2 may be in flight)
            lds_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
#pragma unroll 1
        for (; st < nstA; ++st) {
            stageA(st, cur, std::false_type{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
        ARC_STAMP(1 + 4 * s);
        // ------------------------------------------------------------------------------------------ phase B
        constexpr int NSLAB = Ly::NSLAB, DB = NSLAB - 1;  // DB slabs ahead
        // Ta' = acc * wb_inv[n] (exact: powers of two), once, in place; the row's maximum over BOTH halves -> its
        // power-of-two scale.  (The weight rows' scales come from LDS through a base the compiler cannot see through:
        // it hoists the fifty lane-dependent LDS addresses of an unrolled form out of the scorer loop, spills them, and
        // every reload waits for vmcnt(0), i.e. for the LDS-DMA transfers in flight.)
        float amax = 0.f;
        {
            const float *wsc = wscale + 4 * hh;
            asm volatile("" : "+v"(wsc));
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                if (nh * NTW + t < NT) {  // wave-uniform: not the pad tile
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 wi = *reinterpret_cast<const float4 *>(wsc + 32 * (nh * NTW + t) + 8 * g);
                        acc[t][4 * g + 0] *= wi.x; acc[t][4 * g + 1] *= wi.y; acc[t][4 * g + 2] *= wi.z; acc[t][4 * g + 3] *= wi.w;
                        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(acc[t][4 * g + 0]), fabsf(acc[t][4 * g + 1])),
                                                 fmaxf(fabsf(acc[t][4 * g + 2]), fabsf(acc[t][4 * g + 3]))));
                    }
                }
            }
        }
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        if (hh == 0) rmax[(nh * 4 + mt) * 32 + r] = amax;
        lds_barrier();
        amax = fmaxf(rmax[mt * 32 + r], rmax[(4 + mt) * 32 + r]);
        int E = (int)(__float_as_uint(amax) >> 23);
        E = E < 16 ? 16 : (E > 254 ? 254 : E);
        const float tscale = __uint_as_float((unsigned)(268 - E) << 23);  // maximum -> [2^14, 2^15)
        const float tinv = __uint_as_float((unsigned)(E - 14) << 23);
        if (nh == 0 && hh == 0) rowfac[mt * 32 + r] = tinv * xi;
        // a chunk's pieces into one of the row tile's two patches: row r, stage g >> 1, eight bytes of chunk 2 (g & 1) (+1: low
        // piece); written by the wavefront that holds the chunk (tile t of its half)
        auto convert = [&](int c, int t) {
            (void)c;
            unsigned char *pt = patch + (c & 1) * 4096;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                af16x4 h4, l4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = acc[t][4 * g + e] * tscale;
                    h4[e] = (_Float16)v;
                    l4[e] = (_Float16)(v - (float)h4[e]);
                }
                unsigned char *p = pt + (g >> 1) * 2048 + r * 64 + 8 * hh;
                *reinterpret_cast<af16x4 *>(p + (((2 * (g & 1)) ^ sw) << 4)) = h4;
                *reinterpret_cast<af16x4 *>(p + (((2 * (g & 1) + 1) ^ sw) << 4)) = l4;
            }
        };
        ARC_STAMP(2 + 4 * s);
        float fi[16];
#pragma unroll 1
        for (int jg = 0; jg < NJG; ++jg) {  // 128 head columns at a time (one trip for sentences of up to 128 words)
        const int jbase = 128 * jg;
        // the heads' X2 slabs: chunk c = two k16 stages of 128 rows, two wave-loads per wavefront
        const unsigned char *srcB[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = 8 * q + wave;         // 0..15: stage u >> 3, rows 16 (u & 7) ..
            const int j = jbase + 16 * (u & 7) + lr;
            const unsigned char *rowp = (a.off && j >= len && j < L) ? a.pad : xrow(j);
            srcB[q] = rowp + (long)(2 * s + 1) * KC * 4 + (u >> 3) * 64 + chunk * 16;
        }
        auto issueB = [&](int c, int buf) {
#pragma unroll
            for (int q = 0; q < 2; ++q) arc_dma16(srcB[q] + (long)c * 128, lds0 + (unsigned)(buf * SLAB_B + q * 8192));
        };
        float gj[2];  // inverse scale of the heads' image rows (the X2 side) of the wave's two column tiles
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = jbase + 32 * (2 * nh + k) + r;
            long row = rowbase + j;
            row = row < a.rows_total ? row : a.rows_total - 1;
            gj[k] = (a.off && j >= len) ? a.pad_inv : a.x_inv[row];
        }
        if (jg > 0) lds_barrier();  // the previous group's slabs and patches have been read
#pragma unroll
        for (int c = 0; c < DB && c < NT; ++c) issueB(c, c);
        f32x16 acc2[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[k][e] = 0.f;
        if (nh == 0 && !dead) convert(0, 0);
        const bool idle = SKIP && (dead || jbase + 64 * nh >= L);  // no dependents, or none of the wave's head columns: loads and barriers only
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            {  // slab c landed; the slabs issued behind it (up to DB - 1 of them) may stay in flight
                const int behind = (NT - 1 - c) < (DB - 1) ? (NT - 1 - c) : (DB - 1);  // a constant once the loop is unrolled
                if (behind >= 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if (behind == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if (behind == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            lds_barrier();  // ... and the chunk's patch is written
            const unsigned char *slab = lds + (c % NSLAB) * SLAB_B;
            const unsigned char *pt = patch + (c & 1) * 4096;
            const unsigned ld = lds0 + (unsigned)(((c + DB) % NSLAB) * SLAB_B);
            if (!idle) {
                u32x4 th[2], tl[2];
#pragma unroll
                for (int st2 = 0; st2 < 2; ++st2) {
                    th[st2] = *reinterpret_cast<const u32x4 *>(pt + st2 * 2048 + r * 64 + off_h);
                    tl[st2] = *reinterpret_cast<const u32x4 *>(pt + st2 * 2048 + r * 64 + off_l);
                }
                // four groups (stage, column tile) of three MFMAs: the next group's fragments are read in front of a group, one
                // of the two LDS-DMA issues of slab c + 2 follows every other group
                u32x4 qh[2], ql[2];
                auto read_q = [&](int gi, int slot) {
                    const int st2 = gi >> 1, jt = 2 * nh + (gi & 1);
                    qh[slot] = *reinterpret_cast<const u32x4 *>(slab + st2 * 8192 + (32 * jt + r) * 64 + off_h);
                    ql[slot] = *reinterpret_cast<const u32x4 *>(slab + st2 * 8192 + (32 * jt + r) * 64 + off_l);
                };
                read_q(0, 0);
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {  // both column tiles whatever L is: branch-free (columns >= L are masked)
                    const int st2 = gi >> 1, k = gi & 1, sl = gi & 1;
                    if (gi + 1 < 4) read_q(gi + 1, sl ^ 1);
                    acc2[k] = ARC_MFMA(tl[st2], qh[sl], acc2[k]);
                    acc2[k] = ARC_MFMA(th[st2], ql[sl], acc2[k]);
                    acc2[k] = ARC_MFMA(th[st2], qh[sl], acc2[k]);
                    if ((gi & 1) && c + DB < NT) arc_dma16(srcB[gi >> 1] + (long)(c + DB) * 128, ld + (gi >> 1) * 8192);
                }
            } else if (c + DB < NT) {
#pragma unroll
                for (int q = 0; q < 2; ++q) arc_dma16(srcB[q] + (long)(c + DB) * 128, ld + q * 8192);
            }
            // the next chunk into the other patch, beside this chunk's MFMAs, by the half that holds it
            if (c + 1 < NT && nh == (c + 1) / NTW && !dead) convert(c + 1, (c + 1) % NTW);
        }
        ARC_STAMP(3 + 4 * s);
        // ------------------------------------------------------------------------------------------ the scorer's term
        // The running sum of the scorers' terms lives in the output rows themselves (every lane re-reads what it wrote:
        // 128-byte row segments, L2-resident); the softmax below reads the finished rows back.
#pragma unroll
        for (int e = 0; e < 16; ++e) fi[e] = rowfac[mt * 32 + acc_row(e, lane)];
        float *ob = obase + (lane >> 5) * 4 * (long)L + jbase + 64 * nh + r;  // row 4 (lane >> 5), the wave's first column tile
        int dj = jbase + 64 * nh + r - (i0 + 32 * mt + 4 * hh);                   // j - i of that cell
        asm volatile("" : "+v"(ob), "+v"(dj));  // (everything derived from them is derived HERE: hoisted out of the scorer loop,
                                                // the 32 cells' addresses and offsets were spilled and reloaded one by one)
        // cell (k, e): row 4 hh + (e & 3) + 8 (e >> 2) of the row tile, column 32 k of the wave's pair; one loop per scorer, the
        // scorer switch outside it
        auto cell_ok = [&](int k, int e) {
            return jbase + 32 * (2 * nh + k) + r < L && i0 + 32 * mt + acc_row(e, lane) < len;  // (padded dependents' rows stay untouched)
        };
        if (idle) {
            // nothing of this wave's cells is inside the sentence
        } else if (PER_SCORER) {
            float *rb = s == 0 ? ob : a.raw + (long)(s - 1) * a.B * L * L + (ob - a.out);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (cell_ok(k, e)) rb[(long)((e & 3) + 8 * (e >> 2)) * L + 32 * k] = acc2[k][e] * fi[e] * gj[k];
        } else if (s == 0) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (cell_ok(k, e)) ob[(long)((e & 3) + 8 * (e >> 2)) * L + 32 * k] = acc2[k][e] * fi[e] * gj[k];
        } else if ((s == 1 && a.use_lin) || (s == 2 && a.use_dist)) {
            float prev[2][16];  // both column tiles' loads in flight together
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 16; ++e) prev[k][e] = cell_ok(k, e) ? ob[(long)((e & 3) + 8 * (e >> 2)) * L + 32 * k] : 0.f;
            if (s == 1) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int d = dj + 32 * k - ((e & 3) + 8 * (e >> 2));
                        const float x = acc2[k][e] * fi[e] * gj[k];
                        prev[k][e] = __fadd_rn(prev[k][e], arc_term_lin(x, d));
                    }
            } else {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int d = dj + 32 * k - ((e & 3) + 8 * (e >> 2));
                        const float x = acc2[k][e] * fi[e] * gj[k];
                        prev[k][e] = __fadd_rn(prev[k][e], arc_term_dist(x, d));
                    }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (cell_ok(k, e)) ob[(long)((e & 3) + 8 * (e >> 2)) * L + 32 * k] = prev[k][e];
        }
        }  // column groups
        ARC_STAMP(4 + 4 * s);
    }
    // ---------------------------------------------------------------------------------------------- log-softmax, store
    // over ALL L head columns, padding included (:214), diagonal -inf (:210-211): a wavefront per row, sixteen rows each,
    // on the sums the block's wavefronts left in the output rows
    if constexpr (PER_SCORER) return;  // arc_combine_kernel finishes the rows
    __syncthreads();
    if constexpr (!LONG) {
        // sixteen rows per wavefront, FOUR LANES PER ROW (lane = 4 row + quarter): a lane reduces its 32 columns in its own
        // registers and two exchanges finish the row (a lane per column pair took twelve dependent cross-lane steps per row)
        const int i = 16 * wave + (lane >> 2), q4 = lane & 3;
        float *orow = a.out + ((long)b * L + (i < L ? i : 0)) * L + 32 * q4;
        const bool vec = (L & 3) == 0;  // 16-byte aligned rows
        float v[32];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float4 t = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            const int j = 32 * q4 + 4 * c;
            if (i < len) {
                if (vec && j + 3 < L) t = *reinterpret_cast<const float4 *>(orow + 4 * c);
                else {
                    if (j < L) t.x = orow[4 * c];
                    if (j + 1 < L) t.y = orow[4 * c + 1];
                    if (j + 2 < L) t.z = orow[4 * c + 2];
                    if (j + 3 < L) t.w = orow[4 * c + 3];
                }
            }
            v[4 * c] = j == i ? -INFINITY : t.x;
            v[4 * c + 1] = j + 1 == i ? -INFINITY : t.y;
            v[4 * c + 2] = j + 2 == i ? -INFINITY : t.z;
            v[4 * c + 3] = j + 3 == i ? -INFINITY : t.w;
        }
        float m = -INFINITY;
#pragma unroll
        for (int c = 0; c < 32; ++c) m = fmaxf(m, v[c]);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        float sum = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) sum += __expf(v[c] - m);  // (exp(-inf - m) = 0: masked columns, the diagonal)
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        const float ls = __logf(sum);
        if (i < len) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int j = 32 * q4 + 4 * c;
                const float4 t = make_float4((v[4 * c] - m) - ls, (v[4 * c + 1] - m) - ls, (v[4 * c + 2] - m) - ls,
                                             (v[4 * c + 3] - m) - ls);
                if (vec && j + 3 < L) *reinterpret_cast<float4 *>(orow + 4 * c) = t;
                else {
                    if (j < L) orow[4 * c] = t.x;
                    if (j + 1 < L) orow[4 * c + 1] = t.y;
                    if (j + 2 < L) orow[4 * c + 2] = t.z;
                    if (j + 3 < L) orow[4 * c + 3] = t.w;
                }
            }
        }
    } else {  // up to 512 columns: eight values per lane, two rows' loads in flight
#pragma unroll 1
        for (int k0 = 0; k0 < 16; k0 += 2) {
            float v[2][8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = i0 + 16 * wave + k0 + k;
                const float *orow = a.out + ((long)b * L + (i < L ? i : 0)) * L;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int j = lane + 64 * q;
                    v[k][q] = (i < len && j < L && j != i) ? orow[j] : -INFINITY;
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = i0 + 16 * wave + k0 + k;
                float m = -INFINITY;
#pragma unroll
                for (int q = 0; q < 8; ++q) m = fmaxf(m, v[k][q]);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
                float sum = 0.f;
#pragma unroll
                for (int q = 0; q < 8; ++q) sum += __expf(v[k][q] - m);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                const float ls = __logf(sum);
                if (i < len) {
                    float *orow = a.out + ((long)b * L + i) * L;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int j = lane + 64 * q;
                        if (j < L) orow[j] = (v[k][q] - m) - ls;
                    }
                }
            }
        }
    }
    ARC_STAMP(13);
}

// Per-scorer form, second launch: out[b][i][j] = log-softmax_j(raw0 + term_lin(raw1) + term_dist(raw2)) -- the fused
// kernel's sums, operation by operation -- sixteen rows per wavefront, four lanes per row as above.  L <= 128.
__global__ __launch_bounds__(512) void arc_combine_kernel(ArcSplitArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, L = a.L;
    const int len = a.off ? min(max(a.sentlens[b], 0), L) : L;
    const int i = 16 * wave + (lane >> 2), q4 = lane & 3;
    if (i >= len) return;
    float *orow = a.out + ((long)b * L + i) * L + 32 * q4;
    const float *r1 = a.raw + (orow - a.out), *r2 = r1 + (long)a.B * L * L;
    const bool lin = a.S == 3 && a.use_lin, dist = a.S == 3 && a.use_dist;
    float v[32];
    if ((L & 3) == 0) {  // 16-byte aligned rows: all of a lane's loads in flight together
        float4 t0[8], t1[8], t2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const bool in = 32 * q4 + 4 * c < L;
            t0[c] = in ? *reinterpret_cast<const float4 *>(orow + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            t1[c] = in && lin ? *reinterpret_cast<const float4 *>(r1 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            t2[c] = in && dist ? *reinterpret_cast<const float4 *>(r2 + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int j = 32 * q4 + c;
            const float x0 = reinterpret_cast<const float *>(&t0[c >> 2])[c & 3];
            const float x1 = reinterpret_cast<const float *>(&t1[c >> 2])[c & 3];
            const float x2 = reinterpret_cast<const float *>(&t2[c >> 2])[c & 3];
            float t = x0;
            if (lin) t = __fadd_rn(t, arc_term_lin(x1, j - i));
            if (dist) t = __fadd_rn(t, arc_term_dist(x2, j - i));
            v[c] = (j < L && j != i) ? t : -INFINITY;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const int j = 32 * q4 + c;
            float t = -INFINITY;
            if (j < L) {
                t = orow[c];
                if (lin) t = __fadd_rn(t, arc_term_lin(r1[c], j - i));
                if (dist) t = __fadd_rn(t, arc_term_dist(r2[c], j - i));
            }
            v[c] = j == i ? -INFINITY : t;
        }
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < 32; ++c) m = fmaxf(m, v[c]);
    m = fmaxf(m, __shfl_xor(m, 1));
    m = fmaxf(m, __shfl_xor(m, 2));
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) sum += __expf(v[c] - m);
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    const float ls = __logf(sum);
    if ((L & 3) == 0) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (32 * q4 + 4 * c < L)
                *reinterpret_cast<float4 *>(orow + 4 * c) = make_float4((v[4 * c] - m) - ls, (v[4 * c + 1] - m) - ls,
                                                                         (v[4 * c + 2] - m) - ls, (v[4 * c + 3] - m) - ls);
    } else {
#pragma unroll
        for (int c = 0; c < 32; ++c)
            if (32 * q4 + c < L) orow[c] = (v[c] - m) - ls;
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
        if (hipFuncSetAttribute((const void *)arc_f16x3_kernel<NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ArcSplitLayout<NT>::TOTAL) != hipSuccess ||
            hipFuncSetAttribute((const void *)arc_f16x3_kernel<NT, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ArcSplitLayout<NT>::TOTAL) != hipSuccess ||
            hipFuncSetAttribute((const void *)arc_f16x3_kernel<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ArcSplitLayout<NT>::TOTAL) != hipSuccess ||
            hipFuncSetAttribute((const void *)arc_f16x3_kernel<NT, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ArcSplitLayout<NT>::TOTAL) != hipSuccess ||
            hipFuncSetAttribute((const void *)arc_f16x3_kernel<NT, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                ArcSplitLayout<NT>::TOTAL) != hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "arc_scores_split: cannot raise the dynamic LDS limit");
        once.mark(dev);
    }
    ProfScope ps("arc_fused", stream);
    if (a.raw) {  // the per-scorer form (arc_per_scorer() said so and the caller brought the buffer)
        if (a.L <= 96)
            hipLaunchKernelGGL((arc_f16x3_kernel<NT, false, true, true>), dim3(a.B, a.S), dim3(512), ArcSplitLayout<NT>::TOTAL, stream, a);
        else
            hipLaunchKernelGGL((arc_f16x3_kernel<NT, false, false, true>), dim3(a.B, a.S), dim3(512), ArcSplitLayout<NT>::TOTAL, stream, a);
        const int rc = check_launch("arc_f16x3_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(arc_combine_kernel, dim3(a.B), dim3(512), 0, stream, a);
        return check_launch("arc_combine_kernel");
    }
    if (a.L <= 96) hipLaunchKernelGGL((arc_f16x3_kernel<NT, false, true>), dim3(a.B), dim3(512), ArcSplitLayout<NT>::TOTAL, stream, a);
    else if (a.L <= 128) hipLaunchKernelGGL((arc_f16x3_kernel<NT, false>), dim3(a.B), dim3(512), ArcSplitLayout<NT>::TOTAL, stream, a);
    else hipLaunchKernelGGL((arc_f16x3_kernel<NT, true>), dim3(a.B, (a.L + 127) / 128), dim3(512), ArcSplitLayout<NT>::TOTAL, stream, a);
    return check_launch("arc_f16x3_kernel");
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_arc_kc(int H) { return H > 0 ? arc_split_kc(H) : 0; }

// the per-scorer form pays when three scorers in a row would run on fewer than a third of the CUs' worth of workgroups twice over
static bool arc_per_scorer(int B, int L, int n_scorers) { return n_scorers == 3 && L <= 128 && 3L * B <= 512; }

extern "C" size_t phonlp_arc_scores_split_ws_bytes(int B, int L, int n_scorers) {
    if (B <= 0 || L <= 0 || !arc_per_scorer(B, L, n_scorers)) return 0;
    return (size_t)(n_scorers - 1) * B * L * L * sizeof(float);
}

extern "C" int phonlp_arc_scores_split(const void *XFp, const float *x_inv, int rows_total, int B, int L, int H,
                                       int n_scorers, int use_lin, int use_dist, const void *Wbp, const float *wb_inv,
                                       const int32_t *off, const int32_t *sentlens, const void *pad_row, float pad_inv,
                                       float *arc_logp, void *stream) {
    return phonlp_arc_scores_split_ws(XFp, x_inv, rows_total, B, L, H, n_scorers, use_lin, use_dist, Wbp, wb_inv, off, sentlens,
                                      pad_row, pad_inv, arc_logp, nullptr, 0, stream);
}

extern "C" int phonlp_arc_scores_split_ws(const void *XFp, const float *x_inv, int rows_total, int B, int L, int H,
                                          int n_scorers, int use_lin, int use_dist, const void *Wbp, const float *wb_inv,
                                          const int32_t *off, const int32_t *sentlens, const void *pad_row, float pad_inv,
                                          float *arc_logp, void *ws, size_t ws_bytes, void *stream) {
    const int KC = H > 0 ? arc_split_kc(H) : 0;
    if (!XFp || !x_inv || !Wbp || !wb_inv || !arc_logp || B <= 0 || L <= 0 || rows_total <= 0 ||
        (n_scorers != 1 && n_scorers != 3))
        return fail(PHONLP_ERR_SHAPE, "arc_scores_split: bad argument");
    if (KC == 0 || L > 512) return fail(PHONLP_ERR_UNSUPPORTED, "arc_scores_split: needs H + 1 <= 416 and L <= 512");
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
    if (ws) {
        const size_t need = phonlp_arc_scores_split_ws_bytes(B, L, n_scorers);
        if (need && (ws_bytes < need || ((uintptr_t)ws & 15)))
            return fail(PHONLP_ERR_SHAPE, "arc_scores_split_ws: workspace too small (phonlp_arc_scores_split_ws_bytes) or unaligned");
        if (need) a.raw = (float *)ws;
    }
#ifdef ARC_STAMPS
    a.dbg = getenv("PHONLP_ARC_DBG") ? (long long *)strtoull(getenv("PHONLP_ARC_DBG"), nullptr, 0) : nullptr;
#endif
    switch (KC) {
        case 64: return launch_arc_split<2>(a, as_stream(stream));
        case 128: return launch_arc_split<4>(a, as_stream(stream));
        case 256: return launch_arc_split<8>(a, as_stream(stream));
        default: return launch_arc_split<13>(a, as_stream(stream));
    }
}
