// Engine 2, "f16x3": fp32-equivalent dense layers on the f16 matrix pipe with HALF the MFMA work of engine 1.
//
// An fp32 operand row x is scaled by a power of two s (exact) so that its largest magnitude lies in
// [2^14, 2^15), and split into two f16 pieces  h = rn16(s x),  l = rn16(s x - h)  (the subtraction is exact).
// h + l carries 22-23 significand bits plus the sign trick of round-to-nearest: |s x - h - l| <= 2^-23 |s x| for
// every element within 2^16 of the row maximum; smaller elements lose relative precision against f16's
// subnormal grid (2^-24), i.e. at most 2^-39 of the row maximum in absolute terms -- far below one fp32 rounding
// of the row's large elements.  The product is  a.b ~ al.bh + ah.bl + ah.bh  (the dropped al.bl <= 2^-22 |a||b|
// worst case, 2^-24 typical), three v_mfma_f32_32x32x16_f16 accumulated in fp32 smallest terms first, and the
// epilogue multiplies by the two inverse scales (exact).  Measured against an fp64 product the error is that of
// engine 1 (tests/test_gpu_parity.py::test_engines_against_fp64).
//
// The split is done ONCE PER ELEMENT by split_rows_kernel (weights: at load; activations: one HBM-bound pass,
// 8 bytes per element), not once per tile as engine 1 does while staging, so the GEMM's loop is nothing but
// global_load_dwordx4 -> ds_write_b128 -> ds_read_b128 -> MFMA.
//
// Split image of a [rows, K] matrix: row pitch K16*4 bytes (K16 = K rounded up to 16, zero filled); every group of
// 8 consecutive k is 32 bytes: h[8] then l[8].  A k16 stage of a row is therefore one 64-byte segment of four
// 16-byte chunks (h0-7, l0-7, h8-15, l8-15), and an MFMA fragment (lane r, half hh: k = 8hh..8hh+7 of row r) is
// chunk 2hh (h) / 2hh+1 (l).  inv_scale[row] = 1/s.
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "linear_args.h"

namespace phonlp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define PHONLP_F16_MFMA(A, B, C) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)

// ------------------------------------------------------------------------------------------ split
// One wavefront per row.  K % 4 == 0, X 16-byte aligned, ldx % 4 == 0.
// `pitch` = bytes between image rows (>= 4*K16: the image may be the leading columns of a wider one).
// A launch splits one matrix or two of the same K (b.rows > 0: an activation and the weight it meets, one launch
// instead of two on the step's critical chain).
struct SplitSet {
    const float *X;
    long ldx;
    int rows;
    unsigned char *planes;
    long pitch;
    float *inv_scale;
};
__global__ __launch_bounds__(256) void split_rows_kernel(SplitSet a, SplitSet b, int K, int K16) {
    const int lane = threadIdx.x & 63;
    const int nb1 = (a.rows + 3) >> 2;
    const bool second = (int)blockIdx.x >= nb1;  // block-uniform
    const float *X = second ? b.X : a.X;
    const long ldx = second ? b.ldx : a.ldx, pitch = second ? b.pitch : a.pitch;
    const int rows = second ? b.rows : a.rows;
    unsigned char *planes = second ? b.planes : a.planes;
    float *inv_scale = second ? b.inv_scale : a.inv_scale;
    const int row = ((int)blockIdx.x - (second ? nb1 : 0)) * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *x = X + (long)row * ldx;
    unsigned char *out = planes + (long)row * pitch;
    auto amax4 = [](unsigned mx, const float4 &v) {
        mx = max(mx, __float_as_uint(v.x) & 0x7FFFFFFFu);
        mx = max(mx, __float_as_uint(v.y) & 0x7FFFFFFFu);
        mx = max(mx, __float_as_uint(v.z) & 0x7FFFFFFFu);
        return max(mx, __float_as_uint(v.w) & 0x7FFFFFFFu);
    };
    auto put = [&](int g, const float4 &t0, const float4 &t1, float scale) {
        const float v[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        f16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float s = v[e] * scale;            // exact (power of two, no overflow: |s| < 2^15)
            h[e] = (_Float16)s;                      // round to nearest even
            l[e] = (_Float16)(s - (float)h[e]);      // the difference is exact in fp32
        }
        *reinterpret_cast<f16x8 *>(out + g * 32) = h;
        *reinterpret_cast<f16x8 *>(out + g * 32 + 16) = l;
    };
    auto scale_of = [&](unsigned mx) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        int E = (int)(mx >> 23);  // biased exponent of the row maximum
        E = E < 16 ? 16 : (E > 254 ? 254 : E);
        if (lane == 0) inv_scale[row] = __uint_as_float((unsigned)(E - 14) << 23);
        return __uint_as_float((unsigned)(268 - E) << 23);  // 2^(14 - (E - 127)): maximum -> [2^14, 2^15)
    };
    constexpr int NG = 4;  // rows of up to 64 * NG groups of eight (K16 <= 2048) are read ONCE and kept in registers
    if (K16 <= 512 * NG) {
        float4 t[NG][2];
        unsigned mx = 0;
#pragma unroll
        for (int j = 0; j < NG; ++j) {
            const int c = (lane + 64 * j) * 8;
            t[j][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            t[j][1] = t[j][0];
            if (c < K) t[j][0] = *reinterpret_cast<const float4 *>(x + c);
            if (c + 4 < K) t[j][1] = *reinterpret_cast<const float4 *>(x + c + 4);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) mx = amax4(amax4(mx, t[j][0]), t[j][1]);
        const float scale = scale_of(mx);
#pragma unroll
        for (int j = 0; j < NG; ++j)
            if ((lane + 64 * j) * 8 < K16) put(lane + 64 * j, t[j][0], t[j][1], scale);
        return;
    }
    unsigned mx = 0;
    for (int c = lane * 4; c < K; c += 256) mx = amax4(mx, *reinterpret_cast<const float4 *>(x + c));
    const float scale = scale_of(mx);
    for (int g = lane; g * 8 < K16; g += 64) {
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        if (g * 8 < K) t0 = *reinterpret_cast<const float4 *>(x + g * 8);
        if (g * 8 + 4 < K) t1 = *reinterpret_cast<const float4 *>(x + g * 8 + 4);
        put(g, t0, t1, scale);
    }
}

// ------------------------------------------------------------------------------------------ GEMM
// Epilogue in the "rows interleaved by 4" layout: Y4[(m/4)*N + n][m%4] = C[m][n].  In the MFMA C/D layout four
// consecutive accumulator registers of a lane are four consecutive ROWS of one column, so a lane's float4 is stored
// as it stands (no LDS transpose): lanes 0-31 of a wave-store write 512 contiguous bytes, lanes 32-63 the next row
// group's 512.  Scaling (exact powers of two) and an optional per-column bias; no activation / padding in this form.  Y4 holds
// ceil(M/4) row groups, so a group that straddles M is stored whole.
template <int TM, int TN>
__device__ __forceinline__ void rows4_epilogue(f32x16 (&acc)[TM][TN], const LinearArgs &a, int m0, int n0, int wm, int wn,
                                               int lane) {
    float cs[TN], cb[TN];
    int col[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        col[j] = n0 + (wn * TN + j) * 32 + (lane & 31);
        cs[j] = (a.col_scale && col[j] < a.N) ? a.col_scale[col[j]] : 1.f;
        cb[j] = (a.bias && col[j] < a.N) ? a.bias[col[j]] : 0.f;  // per-column term added to all four rows
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = m0 + (wm * TM + i) * 32 + 8 * g + 4 * (lane >> 5);  // multiple of 4
            if (row >= a.M) continue;
            float4 rs = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.row_scale) {
                rs.x = a.row_scale[row];
                rs.y = a.row_scale[row + 1 < a.M ? row + 1 : a.M - 1];
                rs.z = a.row_scale[row + 2 < a.M ? row + 2 : a.M - 1];
                rs.w = a.row_scale[row + 3 < a.M ? row + 3 : a.M - 1];
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (col[j] >= a.N) continue;
                float4 v;
                v.x = acc[i][j][4 * g + 0] * rs.x * cs[j] + cb[j];
                v.y = acc[i][j][4 * g + 1] * rs.y * cs[j] + cb[j];
                v.z = acc[i][j][4 * g + 2] * rs.z * cs[j] + cb[j];
                v.w = acc[i][j][4 * g + 3] * rs.w * cs[j] + cb[j];
#ifdef PHONLP_ABLATION
                if (a.dbg == 1 && v.x != 1234.5f) continue;
#endif
                *reinterpret_cast<float4 *>(a.Y + ((long)(row >> 2) * a.N + col[j]) * 4) = v;
            }
        }
}

// Labels-at-heads epilogue (mode 2): the tile of T never leaves the chip.  The product is taken TRANSPOSED -- the
// kernel's row operand is the bilinear weight image (row (blk * O + o) * 32 + d2 % 32 = relation o, head feature
// d2 = 32 blk + d2 % 32; kpc / 32 blocks, so a 32-row accumulator tile lies inside one relation) and its column operand
// the words' X1 image --
// because of where the MFMA leaves the results: a lane holds ONE column (one word) and 16 rows (16 values of d2), so the
// contraction over d2 with the head's features is a per-lane sum over its own registers, plus one exchange between
// the two half-waves.  (With words as rows every sum ran across 32 lanes: eleven cross-lane exchanges per tile and
// 0.40 of the kernel's 1.42 ms.)  The head's features come from x2t[d2/4][word][d2%4] -- gathered through the decoded
// heads and transposed once per batch (gather_heads_t_kernel) -- so a lane's four consecutive rows are one float4 and a
// half-wave's load is 512 contiguous bytes (scalar loads from x2t[d2][word]: 128 per lane and tile, 0.19 of 1.19 ms).
// P[row/32][word] = this tile's share of score[word][o]; label_reduce_kernel adds the kpc/32 shares of a relation in
// ascending feature-block order, so the sum has one fixed order: deterministic, and the same whatever batch a sentence
// is decoded in.
template <int TM, int TN>
__device__ __forceinline__ void at_heads_epilogue(f32x16 (&acc)[TM][TN], const LinearArgs &a, int m0, int n0, int wm,
                                                  int wn, int lane) {
    const int half = lane >> 5, l32 = lane & 31;
    int word[TN];
    float cs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        word[j] = n0 + (wn * TN + j) * 32 + l32;
        cs[j] = word[j] < a.N ? a.col_scale[word[j]] : 0.f;
    }
    // A 32-row accumulator tile is group grp = row / 32 of the weight image = (feature block grp / O, relation grp % O):
    // the wave's TM tiles are TM consecutive relations of (nearly always) ONE block of 32 head features, so the heads'
    // features are loaded once per wave and used by all of them -- with the rows ordered (relation, feature) every tile
    // re-read its own 32 KB of x2t and the epilogue moved a third as many bytes again as the product's operands.
    // A lane's 16 registers are four groups of four consecutive rows (d2 = 32 blk + 4 half + 8g + 0..3): one float4 of the
    // rows-by-4 image of the features and one of the row scales per group; kpx and a.M are multiples of 4 and 32, so a
    // group is valid or invalid as a whole.
    const int grp0 = __builtin_amdgcn_readfirstlane((m0 >> 5) + wm * TM);
    float4 x[TN][4];
    int cur_blk = -1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int grp = grp0 + i, tile_row = grp << 5;
        const int blk = grp / a.nrel;
        const bool tv = tile_row < a.M;
        if (blk != cur_blk) {  // wave-uniform
            cur_blk = blk;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d2 = blk * 32 + 4 * half + 8 * g;
                const long xo = (long)(tv && d2 < a.kpx ? d2 >> 2 : 0) * a.ldx2 * 4;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    x[j][g] = *reinterpret_cast<const float4 *>(a.x2 + (word[j] < a.N ? word[j] : 0) * 4L + xo);
            }
        }
        float4 rs[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const bool valid = tv && blk * 32 + 4 * half + 8 * g < a.kpx;
            rs[g] = valid ? *reinterpret_cast<const float4 *>(a.row_scale + tile_row + 4 * half + 8 * g)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // ascending register index: the order of the sum is part of the result
                sum += acc[i][j][4 * g + 0] * rs[g].x * x[j][g].x;
                sum += acc[i][j][4 * g + 1] * rs[g].y * x[j][g].y;
                sum += acc[i][j][4 * g + 2] * rs[g].z * x[j][g].z;
                sum += acc[i][j][4 * g + 3] * rs[g].w * x[j][g].w;
            }
            sum *= cs[j];
            sum += __shfl_xor(sum, 32);  // the rows of the other half-wave
#ifdef PHONLP_ABLATION
            if (a.dbg == 1 && sum != 1234.5f) continue;
#endif
            if (half == 0 && word[j] < a.N && tv) a.P[(long)grp * a.N + word[j]] = sum;
        }
    }
}

// Transposed dense-layer epilogue (persistent wide kernel, EPI 1 / 2): the product is taken with the WEIGHT image as the
// row operand and the words' image as the column operand, as for the labels-at-heads above, so an accumulator lane holds
// ONE word and four groups of four CONSECUTIVE output features.  A 32 x 32 accumulator tile is 128 contiguous bytes of
// each of 32 words' rows, both as fp32 (IMAGE = false) and as f16 pieces of the split image (IMAGE = true;
// LinearArgs::img, scale from the bound as in linear_epilogue) -- but spread over a word's two lanes and four register
// groups.  Stored from there (8 / 16 bytes per lane, 32 different lines per instruction) the stores cost a third of the
// layer: the CU's path to L2 takes a request per LINE touched, whatever it carries (halving the bytes: -3 %; an
// L2-resident target: still 60 % of the cost).  So the tile goes through the wave's own 4 KB of LDS (the 32 KB the two
// pair buffers leave free; fp32, scaled by the word's inverse scale; chunk c of word r at slot c ^ ((r >> 1) & 7):
// conflict-free both ways) and is read back EIGHT LANES A WORD, lane = (word, four consecutive features): in that layout
// the per-feature terms (inverse scale, bias, ReLU, padding value) depend on lane & 7 alone -- a dozen ds_bpermute per
// tile row instead of a dozen per register group --, the f16 pieces of two neighbouring lanes are exchanged by DPP into
// whole 16-byte chunks (h[8] | l[8]), and every store instruction writes eight FULL 128-byte lines.
// a.M = output features (a.row_scale, a.bias, a.pad_bias per feature; a multiple of 4 -- 8 for the image --, the three
// arrays 16-byte aligned: the launcher checks), a.N = words (a.col_scale per word, a.sentlens / a.L their padding mask),
// Y[word * ldy + feature].
template <int TM, int TN, bool IMAGE>
__device__ __forceinline__ void transposed_epilogue(f32x16 (&acc)[TM][TN], const LinearArgs &a, int m0, int n0, int wm,
                                                    int wn, int lane, unsigned char *lds_base, unsigned patch_ofs) {
    // (opaque: everything lane-dependent below is derived HERE, per tile -- hoisted out of the persistent kernel's tile loop
    // it would be spilled across the main loop, and a scratch reload waits for every vector-memory operation in flight.
    // The patch is addressed as an OFFSET into the kernel's LDS array: a pointer through the asm would lose its address
    // space and turn the ds_ accesses into flat ones.)
    asm volatile("" : "+v"(lane), "+v"(patch_ofs));
    static_assert(TM * 32 == 128 && TN == 2, "two registers of 64 features, two column tiles per wave");
    unsigned char *patch = lds_base + patch_ofs;
    const int half = lane >> 5, l32 = lane & 31;
    const int rw = lane >> 3, rk = lane & 7;  // read side: word 8q + rw of a column tile, features 4 rk .. 4 rk + 3 of a tile row
    // write side: the word's inverse scale
    float cs[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int w = n0 + (wn * TN + j) * 32 + l32;
        cs[j] = w < a.N ? a.col_scale[w] : 0.f;
    }
    // read side: the lane's eight words of the wave's two column tiles
    unsigned wofs[TN][4];   // byte offset of the word's output row
    float isc[TN][4];       // image: the word's scale
    unsigned okmask = 0, livemask = 0;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int wd = n0 + (wn * TN + j) * 32 + 8 * q + rw;
            const bool ok = wd < a.N;
            const int wc = ok ? wd : 0;
            const bool lv = ok && (!a.sentlens || (wc % a.L) < a.sentlens[wc / a.L]);
            okmask |= (ok ? 1u : 0u) << (4 * j + q);
            livemask |= (lv ? 1u : 0u) << (4 * j + q);
            isc[j][q] = 1.f;
            if constexpr (IMAGE) {
                float inv;
                image_scale_of_bound(a.img_c * (ok ? a.col_scale[wc] : 0.f) * 32768.f + a.img_b, isc[j][q], inv);
                if (ok && m0 == 0 && wm == 0 && rk == 0) a.img_inv[wd] = inv;
                wofs[j][q] = (unsigned)wc * (unsigned)a.img_pitch;
            } else {
                wofs[j][q] = (unsigned)wc * (unsigned)a.ldy * 4u;
            }
        }
    const int relu_n = a.relu > 0 ? INT_MAX : -a.relu;
    const bool vec_ok = IMAGE || (((a.ldy & 3) == 0) && (((uintptr_t)a.Y & 15) == 0));
    unsigned char *wr_row = patch + l32 * 128;
    const int wsw = (l32 >> 1) & 7;
    // Per-feature scales, biases and padding values of the wave's TM * 32 features: ONE load per lane and array at the top
    // (lane t holds feature t, a second register features 64 ..), handed to the lanes that need them by ds_bpermute.  Loads
    // inside the loop would sit behind this tile's stores in the in-order vector-memory queue: waiting for one would mean
    // waiting for those stores' acknowledgements.
    const int fw = m0 + wm * (TM * 32);
    const int fa = fw + lane < a.M ? fw + lane : a.M - 1, fb = fw + 64 + lane < a.M ? fw + 64 + lane : a.M - 1;
    float rsA = a.row_scale[fa], rsB = a.row_scale[fb];
    float bA = 0.f, bB = 0.f, pA = 0.f, pB = 0.f;
    if (a.bias) { bA = a.bias[fa]; bB = a.bias[fb]; }
    if (a.pad_bias) { pA = a.pad_bias[fa]; pB = a.pad_bias[fb]; }
    // (waited for HERE, and handed on as the asm's results: the compiler's own wait for a loaded register would be placed at its
    // first use inside the loop as vmcnt(n) counted over ALL vector-memory operations, this tile's stores included)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(rsA), "+v"(rsB), "+v"(bA), "+v"(bB), "+v"(pA), "+v"(pB));
    asm volatile("" : "+v"(cs[0]), "+v"(cs[1]), "+v"(okmask), "+v"(livemask));
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(wofs[j][q]), "+v"(isc[j][q]));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int f0 = __builtin_amdgcn_readfirstlane(m0 + (wm * TM + i) * 32);  // the tile row's first feature
        if (f0 >= a.M) continue;
        // the read-side lane's four features f0 + 4 rk + e
        const int t = 32 * (i & 1) + 4 * rk;
        const float rv = i < 2 ? rsA : rsB, bb = i < 2 ? bA : bB, pp = i < 2 ? pA : pB;
        const float4 rs = make_float4(__shfl(rv, t), __shfl(rv, t + 1), __shfl(rv, t + 2), __shfl(rv, t + 3));
        const float4 bv = make_float4(__shfl(bb, t), __shfl(bb, t + 1), __shfl(bb, t + 2), __shfl(bb, t + 3));
        const int f = f0 + 4 * rk;
        float4 lo;  // ReLU as a lower clamp: 0 where it applies, -inf elsewhere
        lo.x = (a.relu && f < relu_n) ? 0.f : -INFINITY;
        lo.y = (a.relu && f + 1 < relu_n) ? 0.f : -INFINITY;
        lo.z = (a.relu && f + 2 < relu_n) ? 0.f : -INFINITY;
        lo.w = (a.relu && f + 3 < relu_n) ? 0.f : -INFINITY;
        float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);  // value of a padded word: act(pad_bias), or zeros
        if (a.pad_bias) {
            pv = make_float4(fmaxf(__shfl(pp, t), lo.x), fmaxf(__shfl(pp, t + 1), lo.y), fmaxf(__shfl(pp, t + 2), lo.z),
                             fmaxf(__shfl(pp, t + 3), lo.w));
        }
        const bool piece_ok = f < a.M;  // (a.M % 4 == 0; the image's a.M % 8 == 0 makes both lanes of a chunk agree)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // chunk 2g + half = features 8g + 4 half .. + 3 of the word; (acc * 2^-ka): exact
                const float4 v = make_float4(acc[i][j][4 * g + 0] * cs[j], acc[i][j][4 * g + 1] * cs[j], acc[i][j][4 * g + 2] * cs[j],
                                             acc[i][j][4 * g + 3] * cs[j]);
                *reinterpret_cast<float4 *>(wr_row + (((2 * g + half) ^ wsw) << 4)) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float4 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 8 * q + rw;
                d[q] = *reinterpret_cast<const float4 *>(patch + r * 128 + ((rk ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v;  // (. * 2^-kb) + bias, activation: as linear_epilogue
                v.x = fmaxf(d[q].x * rs.x + bv.x, lo.x);
                v.y = fmaxf(d[q].y * rs.y + bv.y, lo.y);
                v.z = fmaxf(d[q].z * rs.z + bv.z, lo.z);
                v.w = fmaxf(d[q].w * rs.w + bv.w, lo.w);
                if (!((livemask >> (4 * j + q)) & 1u)) v = pv;
                const bool st = ((okmask >> (4 * j + q)) & 1u) && piece_ok;
                if constexpr (IMAGE) {
                    const float s4[4] = {v.x * isc[j][q], v.y * isc[j][q], v.z * isc[j][q], v.w * isc[j][q]};  // exact (powers of two)
                    img_f16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        h[e] = (_Float16)s4[e];
                        l[e] = (_Float16)(s4[e] - (float)h[e]);
                    }
                    // features 8m .. 8m+7 are the lane pair (rk = 2m, 2m+1): the even lane stores the chunk h[8], the odd one l[8]
                    const uint2 hh2 = __builtin_bit_cast(uint2, h), ll2 = __builtin_bit_cast(uint2, l);
                    const uint2 give = (rk & 1) ? hh2 : ll2;
                    uint2 got;
                    got.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)give.x, 0xB1, 0xF, 0xF, false);  // quad_perm [1,0,3,2]
                    got.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)give.y, 0xB1, 0xF, 0xF, false);
                    u32x4 out;
                    if (rk & 1) { out[0] = got.x; out[1] = got.y; out[2] = ll2.x; out[3] = ll2.y; }
                    else { out[0] = hh2.x; out[1] = hh2.y; out[2] = got.x; out[3] = got.y; }
#ifdef PHONLP_ABLATION
                    if (a.dbg == 1 && out[0] != 0x12345u) continue;
#endif
                    if (st) *reinterpret_cast<u32x4 *>(a.img + wofs[j][q] + (unsigned)f0 * 4u + rk * 16) = out;
                } else {
#ifdef PHONLP_ABLATION
                    if (a.dbg == 1 && v.x != 1234.5f) continue;
#endif
                    float *y = reinterpret_cast<float *>(reinterpret_cast<unsigned char *>(a.Y) + wofs[j][q]) + f;
                    if (st) {
                        if (vec_ok) {
                            *reinterpret_cast<float4 *>(y) = v;
                        } else {
                            y[0] = v.x; y[1] = v.y; y[2] = v.z; y[3] = v.w;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

// hrow[m] = row of the head's features for word m = (b, i), 1 <= i < sentlens[b]; -1 elsewhere (ROOT, padding); and
// x2t[d2 / 4][m][d2 % 4] = x2[hrow[m]][d2] (0 where there is no head; kpx % 4 == 0): the head-side features gathered
// through the decoded heads, transposed and interleaved by four so that the epilogue above reads a word's four
// consecutive features as one float4, 512 contiguous bytes per half-wave.  Block = 32 words x 32 features through LDS.
__global__ __launch_bounds__(1024) void gather_heads_t_kernel(const int32_t *heads, const int32_t *sentlens,
                                                              const float *x2, long ldx2, int M, int L, int kpx,
                                                              int32_t *hrow, float *x2t, const int32_t *rowmap,
                                                              const int32_t *off, int *tile_ctr) {
    __shared__ float tile[32][33];
    __shared__ int hr[32];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int m0 = blockIdx.x * 32, d0 = blockIdx.y * 32;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8) tile_ctr[threadIdx.x] = 0;  // the persistent stage A's counters
    if (ty == 0) {
        const int m = m0 + tx;
        int v = -1;
        if (m < M) {
            // packed (rowmap != NULL): word m is the live word rowmap[m] = b*L + i, its head's features are row off[b] + h
            const int dm = rowmap ? rowmap[m] : m;
            const int b = dm / L, i = dm - b * L;
            if (i >= 1 && i < sentlens[b]) {
                const int h = heads[dm];
                if (h >= 0 && h < (rowmap ? sentlens[b] : L)) v = (rowmap ? off[b] : b * L) + h;
            }
            if (blockIdx.y == 0) hrow[m] = v;
        }
        hr[tx] = v;
    }
    __syncthreads();
    const int h = hr[ty];
    tile[ty][tx] = (h >= 0 && d0 + tx < kpx) ? x2[(long)h * ldx2 + d0 + tx] : 0.f;
    __syncthreads();
    // rows-by-4 image: x2t[(d2 >> 2) * M + m][d2 & 3]; 128 consecutive threads write 512 contiguous bytes
    const int f = (threadIdx.x & 3) + 4 * (threadIdx.x >> 7), w = (threadIdx.x >> 2) & 31;
    if (d0 + f < kpx && m0 + w < M) x2t[((long)((d0 + f) >> 2) * M + m0 + w) * 4 + (f & 3)] = tile[w][f];
}

// The same for whole rows (the form used whenever 32 rows fit 64 KB of LDS): a block takes 32 words, reads each head's
// kpx features as ONE contiguous row (float4 per thread) into LDS rows of kpx/4 + 1 float4 (the odd pitch makes the
// transposed read conflict-free), and writes, per group of four features, the 32 words' float4 as 512 contiguous bytes.
// The 32 x 32 form above reads 128-byte row segments: 45-75 us against 30 for this one at B=256, L=128.
__global__ __launch_bounds__(256) void gather_heads_rows_kernel(const int32_t *heads, const int32_t *sentlens, const float *x2,
                                                                long ldx2, int M, int L, int kpx, int32_t *hrow, float *x2t,
                                                                const int32_t *rowmap, const int32_t *off, int *tile_ctr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm[];
    float4 *tile = reinterpret_cast<float4 *>(gsm);
    __shared__ int hr[32];
    const int tid = threadIdx.x, m0 = blockIdx.x * 32, c4n = kpx >> 2, pitch = c4n + 1;
    if (blockIdx.x == 0 && tid < 8) tile_ctr[tid] = 0;  // the persistent stage A's counters
    if (tid < 32) {
        const int m = m0 + tid;
        int v = -1;
        if (m < M) {
            const int dm = rowmap ? rowmap[m] : m;
            const int b = dm / L, i = dm - b * L;
            if (i >= 1 && i < sentlens[b]) {
                const int h = heads[dm];
                if (h >= 0 && h < (rowmap ? sentlens[b] : L)) v = (rowmap ? off[b] : b * L) + h;
            }
            hrow[m] = v;
        }
        hr[tid] = v;
    }
    __syncthreads();
    const int total = 32 * c4n;
    for (int idx0 = tid; idx0 < total; idx0 += 4 * 256) {  // four loads in flight per thread
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = idx0 + 256 * u;
            const int w = idx < total ? idx / c4n : 0, c4 = idx - w * c4n;
            const int h = idx < total ? hr[w] : -1;
            v[u] = h >= 0 ? *reinterpret_cast<const float4 *>(x2 + (long)h * ldx2 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int idx = idx0 + 256 * u;
            const int w = idx / c4n, c4 = idx - w * c4n;
            if (idx < total) tile[w * pitch + c4] = v[u];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < total; idx += 256) {
        const int w = idx & 31, c4 = idx >> 5;
        if (m0 + w < M) *reinterpret_cast<float4 *>(x2t + ((long)c4 * M + m0 + w) * 4) = tile[w * pitch + c4];
    }
}

// rels[m] = argmax_o sum_{t < gpo} P[t*O + o][m] (shares added in ascending t; strict '>' in ascending o), 0 for
// words without a label.  Block = 64 words x 8 relation classes (o mod 8), combined through LDS.
__global__ __launch_bounds__(512) void label_reduce_kernel(const float *P, const int32_t *hrow, int M, int O, int gpo,
                                                           int32_t *rels, float *best_opt, const int32_t *rowmap) {
    __shared__ float sv[8][64];
    __shared__ int so[8][64];
    const int lx = threadIdx.x & 63, oq = threadIdx.x >> 6;
    const int m = blockIdx.x * 64 + lx;
    const bool live = m < M && hrow[m] >= 0;
    float bv = -INFINITY;
    int bo = INT_MAX;
    if (live)
        for (int o = oq; o < O; o += 8) {
            float s = 0.f;
            if (gpo <= 16) {  // a relation's shares in flight together, added in the same ascending order
                float v[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = t < gpo ? P[(long)(t * O + o) * M + m] : 0.f;
#pragma unroll
                for (int t = 0; t < 16; ++t) s = t < gpo ? s + v[t] : s;
            } else {
                for (int t = 0; t < gpo; ++t) s += P[(long)(t * O + o) * M + m];
            }
            if (s > bv) { bv = s; bo = o; }
        }
    sv[oq][lx] = bv;
    so[oq][lx] = bo;
    __syncthreads();
    if (oq == 0 && m < M) {
        for (int q = 1; q < 8; ++q)
            if (sv[q][lx] > bv || (sv[q][lx] == bv && so[q][lx] < bo)) { bv = sv[q][lx]; bo = so[q][lx]; }
        const int dm = rowmap ? rowmap[m] : m;  // packed: the caller zeroed rels / best_opt, the live words are scattered
        rels[dm] = live && bo != INT_MAX ? bo : 0;
        if (best_opt) best_opt[dm] = live && bo != INT_MAX ? bv : 0.f;
    }
}

// 256 x 128 block tile, four waves of 128 x 64 (8 accumulator tiles), k16 stages.  a.X / a.W are the split images (row
// pitch a.ldx / a.ldw BYTES), a.K = K16.  Per stage and wave: 12 ds_read_b128 feed 24 MFMAs and there is no other VALU
// work.
//
// LDS image of a stage: row R at R*64 bytes (rows 0-255 A tile, 256-383 B tile), chunk c stored at slot
// c ^ ((R>>2)&3): the 16 rows of a ds_read_b128 lane group cover all 64 banks.  Rows past M / N are not zero-filled:
// their pointers are clamped to the last valid row and the garbage only reaches outputs the epilogue never stores.
//
// The operands are streamed global -> LDS by the LDS-DMA path (global_load_lds_dwordx4: 16 bytes per lane land at
// M0 + lane*16, no VGPR staging) into a ring of THREE stage buffers, so the loads of stage s+2 are in flight while stage
// s is multiplied: two whole stages (~1500-3000 cycles) cover the L2 / Infinity-Cache latency.  A wave-load covers 16
// rows x 64 bytes; lane i writes slot i&3 of row i>>2, so it FETCHES chunk (i&3) ^ ((i>>4)&3) of that row: the swizzle
// of the LDS image is applied on the global side.  The DMA and its s_waitcnt are inline asm: hipcc's own waitcnt
// insertion would make every LDS read wait for all LDS-DMA loads in flight (it cannot tell the ring buffers apart).
//
// PREFETCH chooses the loop form (launch_linear_f16x3 picks by K): true = fragments double-buffered in registers, the
// ds_read_b128 of stage s+1 issued BEFORE the MFMAs of stage s (256 VGPRs; wins for K >= 512: 0.456 vs 0.482 ms on the
// W1/W2 layer); false = fragments read at the top of their own stage (196 VGPRs; wins for short K, i.e. label stage A,
// where the leaner epilogue matters more: 1.55 vs 1.65 ms).
__device__ __forceinline__ void lds_dma16(const unsigned char *gptr, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_addr), "v"(gptr) : "memory", "m0");
}

// the same with a uniform (SGPR) base and a 32-bit byte offset per lane: half the address registers, and a uniform term
// (the k offset of a pair) is added on the scalar unit
__device__ __forceinline__ void lds_dma16_s(const unsigned char *sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(sbase) : "memory", "m0");
}

template <int TAG, bool PREFETCH, bool ILV = false>
__global__ __launch_bounds__(256, 2) void linear_f16x3_dma_kernel(LinearArgs a) {
    constexpr int BM = 256, BN = 128, WM = 2, WN = 2, TM = 4, TN = 2, NBUF = 3;
    constexpr int ROWS = BM + BN, BUF = ROWS * 64;
    static_assert(WM * WN * 32 * (TN * 32 + 4) * 4 <= NBUF * BUF, "epilogue patch must fit");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = a.num_m * a.num_n;
    const int nst = a.K >> 4;
    int m0, n0;
    {   // one tile per block; blocks b, b+8, ... (one XCD) take a contiguous chunk of a GM-tall column-major walk
        const int xg = blockIdx.x & 7, jl = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int t = (xg < r ? xg * (q + 1) : r * (q + 1) + (xg - r) * q) + jl;
        const int GM = a.gm > 0 ? a.gm : 4;
        const int per_group = GM * a.num_n;
        const int g = t / per_group;
        const int first_m = g * GM;
        const int gsz = (a.num_m - first_m) < GM ? (a.num_m - first_m) : GM;
        const int in = t - g * per_group;
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * BN;
    }
    // wave-load u = 4q + wave (q = 0..5) fills rows 16u .. 16u+15 of the stage image (A rows 0-255, B rows 256-383)
    const int lr = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned char *src[6];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = m0 + 16 * (4 * q + wave) + lr;
        row = row < a.M ? row : a.M - 1;
        src[q] = reinterpret_cast<const unsigned char *>(a.X) + (long)row * a.ldx + chunk * 16;
    }
#pragma unroll
    for (int q = 4; q < 6; ++q) {
        int row = n0 + 16 * (4 * (q - 4) + wave) + lr;
        row = row < a.N ? row : a.N - 1;
        src[q] = reinterpret_cast<const unsigned char *>(a.W) + (long)row * a.ldw + chunk * 16;
    }
#ifdef PHONLP_ABLATION
    if (a.dbg == 4)  // timing only: every load from the same 64 bytes (no L2 traffic, same issue count)
#pragma unroll
        for (int q = 0; q < 6; ++q) src[q] = reinterpret_cast<const unsigned char *>(a.X) + chunk * 16;
#endif
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    auto issue_stage = [&](int st, int buf) {
#pragma unroll
        for (int q = 0; q < 6; ++q) lds_dma16(src[q] + (long)st * 64, lds0 + (unsigned)(buf * BUF + q * 4096));
    };

    const int r = lane & 31, hh = lane >> 5;
    const int sw = (r >> 2) & 3;
    const int off_h = ((2 * hh) ^ sw) << 4, off_l = ((2 * hh + 1) ^ sw) << 4;
    const int a_ofs = (wm * 128 + r) * 64;
    const int b_ofs = (BM + wn * 64 + r) * 64;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Frags {
        u32x4 ah[TM], al[TM], bh[TN], bl[TN];
    };
    auto read_frags = [&](int bufi, Frags &f) {
        const unsigned char *buf = lds + bufi * BUF;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.bh[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 2048 + off_h);
            f.bl[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 2048 + off_l);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.al[i] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + i * 2048 + off_l);
            f.ah[i] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + i * 2048 + off_h);
        }
    };
    auto mma_frags = [&](const Frags &f) {  // smallest terms first
#define PHONLP_X3_ROUND(AA, BB)                      \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)   \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = PHONLP_F16_MFMA(f.AA[i], f.BB[j], acc[i][j]);
        PHONLP_X3_ROUND(al, bh)
        PHONLP_X3_ROUND(ah, bl)
        PHONLP_X3_ROUND(ah, bh)
#undef PHONLP_X3_ROUND
    };

    issue_stage(0, 0);
    if (nst > 1) {
        issue_stage(1, 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (PREFETCH && ILV) {
        // Register-prefetch loop with the LDS-DMA issues spread between the MFMAs.  The fragments of stage s are in
        // registers before stage s starts, so behind the barrier at its top buffer s%3 is free: the loads issued DURING
        // stage s are those of stage s+3 (into buffer s%3), one behind every fourth MFMA, and every load still has between
        // two and three stages to land.
        auto pin = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
        };
        if (nst > 2) issue_stage(2, 2);
        Frags f0, f1;
        read_frags(0, f0);
        auto full = [&](int s, int b0, const Frags &fc, Frags &fn) {  // b0 = s % 3; needs s + 3 < nst
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // stage s+1 landed (s+2 may be in flight)
            lds_barrier();
            read_frags(b0 == 2 ? 0 : b0 + 1, fn);
            asm volatile("" ::: "memory");
            pin();
            const long so = (long)(s + 3) * 64;
            const unsigned ld = lds0 + (unsigned)(b0 * BUF);
#define PHONLP_X3_HALF(AA, BB, I0, Q)                                                                          \
    _Pragma("unroll") for (int i = I0; i < I0 + TM / 2; ++i)                                                   \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = PHONLP_F16_MFMA(fc.AA[i], fc.BB[j], acc[i][j]); \
    pin();                                                                                                     \
    lds_dma16(src[Q] + so, ld + Q * 4096);
            PHONLP_X3_HALF(al, bh, 0, 0)
            PHONLP_X3_HALF(al, bh, TM / 2, 1)
            PHONLP_X3_HALF(ah, bl, 0, 2)
            PHONLP_X3_HALF(ah, bl, TM / 2, 3)
            PHONLP_X3_HALF(ah, bh, 0, 4)
            PHONLP_X3_HALF(ah, bh, TM / 2, 5)
#undef PHONLP_X3_HALF
        };
        auto tail = [&](int s, int b0, const Frags &fc, Frags &fn) {
            if (s + 1 < nst) {
                if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
                if (s + 3 < nst) issue_stage(s + 3, b0);
                read_frags(b0 == 2 ? 0 : b0 + 1, fn);
            }
            mma_frags(fc);
        };
        int s = 0, b0 = 0;
        for (; s + 4 < nst; s += 2) {
            full(s, b0, f0, f1);
            b0 = b0 == 2 ? 0 : b0 + 1;
            full(s + 1, b0, f1, f0);
            b0 = b0 == 2 ? 0 : b0 + 1;
        }
        for (; s < nst; s += 2) {
            tail(s, b0, f0, f1);
            b0 = b0 == 2 ? 0 : b0 + 1;
            if (s + 1 < nst) tail(s + 1, b0, f1, f0);
            b0 = b0 == 2 ? 0 : b0 + 1;
        }
        lds_barrier();  // the epilogue's LDS patch aliases the stage buffers
    } else if constexpr (PREFETCH) {
        // one stage: fetch stage s+2 into the buffer whose readers are all past the previous barrier, publish stage s+1
        // (own pieces landed, then the barrier), prefetch its fragments, multiply stage s.
        auto stage = [&](int s, const Frags &fc, Frags &fn) {
            if (s + 2 < nst) issue_stage(s + 2, (s + 2) % NBUF);
            if (s + 1 < nst) {
                if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_barrier();
                read_frags((s + 1) % NBUF, fn);
            }
            mma_frags(fc);
        };
        Frags f0, f1;
        read_frags(0, f0);
        // stage s+2 goes to buffer (s+2)%3, last read as stage s-1 by the prefetch of iteration s-2: every wave has been
        // through the barrier of iteration s-1 since (lds_barrier waits for the wave's outstanding LDS reads first).
        // The steady state is branch-free (a conditional around the prefetch makes hipcc wait for ALL LDS reads at the
        // join, i.e. in front of the MFMAs); the last stages run through the guarded form.
        auto stage_full = [&](int s, const Frags &fc, Frags &fn) {
            issue_stage(s + 2, (s + 2) % NBUF);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            lds_barrier();
            read_frags((s + 1) % NBUF, fn);
            // MFMAs are register-only: unpinned, hipcc hoists them above the barrier, and the next stage's MFMAs then
            // sit right behind their own reads.  An empty asm that "modifies" every accumulator pins them behind the
            // reads (sched_barrier does too, but drives the register allocator into 500 spills here).
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
            mma_frags(fc);
        };
        int s = 0;
        for (; s + 3 < nst; s += 2) {
            stage_full(s, f0, f1);
            stage_full(s + 1, f1, f0);
        }
        for (; s < nst; s += 2) {
            stage(s, f0, f1);
            if (s + 1 < nst) stage(s + 1, f1, f0);
        }
        lds_barrier();  // the epilogue's LDS patch aliases the stage buffers
    } else if constexpr (ILV) {
        // The plain loop with the six LDS-DMA issues of stage s+2 spread between the MFMAs of stage s (one behind every
        // fourth) instead of in a burst at its top: an issue costs the wave 60-180 cycles, and in a burst those add up
        // to most of a stage's MFMA time during which only the SIMD's other wave can feed the matrix pipe.  The empty
        // asm "modifying" the accumulators keeps hipcc from regrouping the MFMAs around the (volatile) issues.
        auto pin = [&]() {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
        };
        auto body = [&](int s, int cur, auto issue) {
            const int nxt2 = cur == 0 ? 2 : cur - 1;
#ifdef PHONLP_ABLATION
            const long so = a.dbg == 4 ? 0 : (long)(s + 2) * 64;
#define PHONLP_X3_SKIP(Q) (a.dbg == 3 && Q >= 3)  // timing only: half of the loads
#else
            const long so = (long)(s + 2) * 64;
#define PHONLP_X3_SKIP(Q) false
#endif
            const unsigned ld = lds0 + (unsigned)(nxt2 * BUF);
            Frags f;
            read_frags(cur, f);
#define PHONLP_X3_HALF(AA, BB, I0, Q)                                                                        \
    _Pragma("unroll") for (int i = I0; i < I0 + TM / 2; ++i)                                                 \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = PHONLP_F16_MFMA(f.AA[i], f.BB[j], acc[i][j]); \
    pin();                                                                                                   \
    if constexpr (decltype(issue)::value) if (!PHONLP_X3_SKIP(Q)) lds_dma16(src[Q] + so, ld + Q * 4096);
            PHONLP_X3_HALF(al, bh, 0, 0)
            PHONLP_X3_HALF(al, bh, TM / 2, 1)
            PHONLP_X3_HALF(ah, bl, 0, 2)
            PHONLP_X3_HALF(ah, bl, TM / 2, 3)
            PHONLP_X3_HALF(ah, bh, 0, 4)
            PHONLP_X3_HALF(ah, bh, TM / 2, 5)
#undef PHONLP_X3_HALF
#undef PHONLP_X3_SKIP
        };
        int cur = 0, s = 0;
        for (; s + 2 < nst; ++s) {
            body(s, cur, std::true_type{});
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // stage s+1 (issued one iteration ago) has landed
            lds_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
        for (; s < nst; ++s) {
            body(s, cur, std::false_type{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
    } else {
        int cur = 0;
        for (int s = 0; s < nst; ++s) {
            const int nxt2 = cur == 0 ? 2 : cur - 1;  // (cur + 2) % 3: last read in iteration s-1, behind that barrier
            if (s + 2 < nst) issue_stage(s + 2, nxt2);
            Frags f;
            read_frags(cur, f);
            mma_frags(f);
            // stage s+1 (issued one iteration ago) must have landed before anyone reads it; the 6 loads of stage s+2
            // issued above may stay in flight
            if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            cur = cur == 2 ? 0 : cur + 1;
        }
    }
#ifdef PHONLP_ABLATION
    if (a.dbg == 2) {  // timing only: keep the accumulators alive without the epilogue
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 1234.5f) a.Y[tid] = t;
        return;
    }
#endif
    if (a.rows4 == 2) {
        at_heads_epilogue<TM, TN>(acc, a, m0, n0, wm, wn, lane);
        return;
    }
    if (a.rows4) {
        rows4_epilogue<TM, TN>(acc, a, m0, n0, wm, wn, lane);
        return;
    }
    linear_epilogue<TM, TN, WM, WN>(acc, reinterpret_cast<float *>(lds), a, m0, n0, wm, wn, lane, wave);
}

// The SMALL form, for launches of a few hundred 64 x 64 tiles at most (B = 1 ... 32 sentences): such a product is a
// LATENCY chain, not a bandwidth or MFMA problem -- the 256 x 128 kernel above keeps two k16 stages in flight per
// workgroup and, with a handful of workgroups on the chip, every stage waits most of a memory round trip (B = 1, L = 32,
// K = 768: 48 stages in 36 us).  Here: 64 x 64 block tile (four waves of 32 x 32, one accumulator tile each: four to eight
// times as many workgroups pull the operands), k16 stages of 8 KB in a ring of SIXTEEN (fifteen in flight: 120 KB per
// workgroup, one workgroup per CU), the same LDS image, DMA pattern and MFMA order (al.bh, ah.bl, ah.bh per stage) as
// above, hence the same bits.  The loop is branch-free: the loads of the stages beyond K re-fetch the last stage into
// buffers nobody reads again, so one s_waitcnt immediate serves every iteration.
template <int NBUF>
__global__ __launch_bounds__(256, 1) void linear_f16x3_small_kernel(LinearArgs a) {
    constexpr int BM = 64, BN = 64, WM = 2, WN = 2, TM = 1, TN = 1;
    constexpr int ROWS = BM + BN, BUF = ROWS * 64;
    static_assert(WM * WN * 32 * (TN * 32 + 4) * 4 <= NBUF * BUF, "epilogue patch must fit");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int nst = a.K >> 4;
    const int m0 = (blockIdx.x % a.num_m) * BM, n0 = (blockIdx.x / a.num_m) * BN;
    // wave-load u = 4q + wave (q = 0, 1) fills rows 16u .. 16u+15 of the stage image (A rows 0-63, B rows 64-127)
    const int lr = lane >> 2, chunk = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned char *src[2];
    {
        int row = m0 + 16 * wave + lr;
        row = row < a.M ? row : a.M - 1;
        src[0] = reinterpret_cast<const unsigned char *>(a.X) + (long)row * a.ldx + chunk * 16;
        row = n0 + 16 * wave + lr;
        row = row < a.N ? row : a.N - 1;
        src[1] = reinterpret_cast<const unsigned char *>(a.W) + (long)row * a.ldw + chunk * 16;
    }
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;
    auto issue_stage = [&](int st, int buf) {
        const long so = (long)(st < nst ? st : nst - 1) * 64;
        lds_dma16(src[0] + so, lds0 + (unsigned)(buf * BUF));
        lds_dma16(src[1] + so, lds0 + (unsigned)(buf * BUF + 4096));
    };
    const int r = lane & 31, hh = lane >> 5;
    const int sw = (r >> 2) & 3;
    const int off_h = ((2 * hh) ^ sw) << 4, off_l = ((2 * hh + 1) ^ sw) << 4;
    const int a_ofs = (wm * 32 + r) * 64;
    const int b_ofs = (BM + wn * 32 + r) * 64;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] = 0.f;

#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s) issue_stage(s, s);
    int cur = 0;
    for (int s = 0; s < nst; ++s) {
        // stage s has landed (the 14 stages behind it may be in flight); behind the barrier every wave's pieces of it
        // have, and every wave is through reading stage s-1, whose buffer takes stage s+15
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NBUF - 2)) : "memory");
        lds_barrier();
        issue_stage(s + NBUF - 1, cur == 0 ? NBUF - 1 : cur - 1);
        const unsigned char *buf = lds + cur * BUF;
        const u32x4 bh = *reinterpret_cast<const u32x4 *>(buf + b_ofs + off_h);
        const u32x4 al = *reinterpret_cast<const u32x4 *>(buf + a_ofs + off_l);
        const u32x4 bl = *reinterpret_cast<const u32x4 *>(buf + b_ofs + off_l);
        const u32x4 ah = *reinterpret_cast<const u32x4 *>(buf + a_ofs + off_h);
        acc[0][0] = PHONLP_F16_MFMA(al, bh, acc[0][0]);  // smallest terms first, as everywhere
        acc[0][0] = PHONLP_F16_MFMA(ah, bl, acc[0][0]);
        acc[0][0] = PHONLP_F16_MFMA(ah, bh, acc[0][0]);
        cur = cur == NBUF - 1 ? 0 : cur + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus loads land before the epilogue's patch reuses the ring
    lds_barrier();
    linear_epilogue<TM, TN, WM, WN>(acc, reinterpret_cast<float *>(lds), a, m0, n0, wm, wn, lane, wave);
}

// The WIDE form: 256 x 256 block tile, EIGHT waves of 128 x 64, one workgroup per CU (two waves per SIMD as above), and
// the operands fetched a k32 PAIR of stages at a time, 128 contiguous bytes per row.
//
// Why: with every load served from L1 the fused stage A above runs in 0.83 instead of 1.08 ms (PHONLP_X3_ABLATION=4), i.e.
// a quarter of its time is the L2 -> LDS path: 8.8 GB per launch in 64-byte half-line requests, every 128-byte line of
// L2 asked for twice, a stage apart.  Here a line is asked for once (a lane group of eight covers it) and the square
// tile moves a third fewer bytes per FLOP: a third of the requests.  The price is one barrier domain of eight waves --
// both waves of a SIMD stop at it together -- but there is one barrier per 48 MFMAs of a wave instead of one per 24.
//
// LDS: two pair buffers of 512 rows x 128 bytes (rows 0-255 A tile, 256-511 B tile) = 128 KB.  Row R's eight 16-byte
// chunks (k16 stage t = chunk / 4; inside it h0-7, l0-7, h8-15, l8-15) are stored at slot chunk ^ ((R >> 1) & 7): with
// the 128-byte pitch the sixteen rows of a ds_read_b128 lane group ({0-3,12-15,20-27} / {4-11,16-19,28-31} of the 32)
// then cover all 64 banks.  A wave-load (1 KB) is 8 rows x 128 bytes, lane i writing slot i & 7 of row i >> 3, so it
// fetches chunk (i & 7) ^ ((R >> 1) & 7): the swizzle is applied on the global side as above.
//
// Per pair and wave: 24 ds_read_b128, 48 MFMAs, 8 LDS-DMA issues, one barrier (schedule: the comment in the kernel).
// K16 must be a multiple of 32 (the launcher checks).
template <int TAG>
__global__ __launch_bounds__(512, 1) void linear_f16x3_wide_kernel(LinearArgs a) {
    constexpr int BM = 256, BN = 256, WM = 2, WN = 4, TM = 4, TN = 2;
    constexpr int ROWS = BM + BN, PBUF = ROWS * 128;
    static_assert(WM * WN * 32 * (TN * 32 + 4) * 4 <= 2 * PBUF, "epilogue patch must fit");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = a.num_m * a.num_n;
    const int npair = a.K >> 5;
    int m0, n0;
    {   // as in the kernel above: one tile per block, an XCD's blocks walk GM-tall groups column-major
        const int xg = blockIdx.x & 7, jl = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int t = (xg < r ? xg * (q + 1) : r * (q + 1) + (xg - r) * q) + jl;
        const int GM = a.gm > 0 ? a.gm : 8;  // measured: 8 rows of tiles per group 1 % ahead of 4, 16 level, 1 / 54 3-5 % behind
        const int per_group = GM * a.num_n;
        const int g = t / per_group;
        const int first_m = g * GM;
        const int gsz = (a.num_m - first_m) < GM ? (a.num_m - first_m) : GM;
        const int in = t - g * per_group;
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * BN;
    }
    // wave-load u = 8q + wave (q = 0..7) fills rows 8u .. 8u+7 of the pair image (A: q < 4, B: q >= 4)
    const int lr = lane >> 3;
    const int chunk = (lane & 7) ^ (((wave & 1) << 2) + (lane >> 4));  // (R >> 1) & 7 with R = 64q + 8 wave + lr
    const unsigned char *src[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = m0 + 64 * q + 8 * wave + lr;
        row = row < a.M ? row : a.M - 1;
        src[q] = reinterpret_cast<const unsigned char *>(a.X) + (long)row * a.ldx + chunk * 16;
    }
#pragma unroll
    for (int q = 4; q < 8; ++q) {
        int row = n0 + 64 * (q - 4) + 8 * wave + lr;
        row = row < a.N ? row : a.N - 1;
        src[q] = reinterpret_cast<const unsigned char *>(a.W) + (long)row * a.ldw + chunk * 16;
    }
#ifdef PHONLP_ABLATION
    if (a.dbg == 4)  // timing only: every load from the same 128 bytes
#pragma unroll
        for (int q = 0; q < 8; ++q) src[q] = reinterpret_cast<const unsigned char *>(a.X) + chunk * 16;
#endif
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    const int r = lane & 31, hh = lane >> 5;
    const int sw = (r >> 1) & 7;
    // byte offsets inside a row of the four fragment kinds of k16 stage 0 (stage 1: ^ 64)
    const int off_h = ((2 * hh) ^ sw) << 4, off_l = ((2 * hh + 1) ^ sw) << 4;
    const int a_ofs = (wm * 128 + r) * 128;
    const int b_ofs = (BM + wn * 64 + r) * 128;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Frags {
        u32x4 ah[TM], al[TM], bh[TN], bl[TN];
    };
    auto read_frags = [&](const unsigned char *buf, int t, Frags &f) {
        const int oh = off_h ^ (t << 6), ol = off_l ^ (t << 6);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.bh[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 4096 + oh);
            f.bl[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 4096 + ol);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.al[i] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + i * 4096 + ol);
            f.ah[i] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + i * 4096 + oh);
        }
    };
    auto pin = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
    };
    auto mma_frags = [&](const Frags &f) {  // smallest terms first
#define PHONLP_X3_ROUND(AA, BB)                      \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)   \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = PHONLP_F16_MFMA(f.AA[i], f.BB[j], acc[i][j]);
        PHONLP_X3_ROUND(al, bh)
        PHONLP_X3_ROUND(ah, bl)
        PHONLP_X3_ROUND(ah, bh)
#undef PHONLP_X3_ROUND
    };
    auto issue_pair = [&](int p, int pb) {
#pragma unroll
        for (int q = 0; q < 8; ++q) lds_dma16(src[q] + (long)p * 128, lds0 + (unsigned)(pb * PBUF + q * 8192));
    };
    // Software pipeline over the k16 stages (F = fragments of a pair's stage 0, G = of its stage 1; a stage = three rounds
    // of 8 MFMAs: al.bh, ah.bl, ah.bh):
    //   rounds 1-2 of F | read G | round 3 of F | rounds 1-2 of G | wait for pair p+1 + barrier | read the next F |
    //   round 3 of G with the 8 loads of pair p+2 between its MFMAs
    // so every fragment read has 8 MFMAs (256+ cycles of the matrix pipe) between its issue and its first use, and a
    // pair's loads have the 40 MFMAs up to the next barrier to land.
#define PHONLP_X3_ROW(FR, AA, BB, I) \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[I][j] = PHONLP_F16_MFMA(FR.AA[I], FR.BB[j], acc[I][j]);
#define PHONLP_X3_ROUND(FR, AA, BB)                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = PHONLP_F16_MFMA(FR.AA[i], FR.BB[j], acc[i][j]);
    Frags F, G;
    auto body = [&](int p, auto more, auto more2) {
        constexpr bool MORE = decltype(more)::value, MORE2 = decltype(more2)::value;  // pairs p+1 / p+2 exist
        const int pb = p & 1;
        const unsigned char *buf = lds + pb * PBUF;
        PHONLP_X3_ROUND(F, al, bh)
        PHONLP_X3_ROUND(F, ah, bl)
        pin();
        read_frags(buf, 1, G);
        pin();
        PHONLP_X3_ROUND(F, ah, bh)
        pin();
        PHONLP_X3_ROUND(G, al, bh)
        PHONLP_X3_ROUND(G, ah, bl)
        pin();
        if constexpr (MORE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // pair p+1 (issued 40+ MFMAs ago) has landed
            lds_barrier();  // ... everyone's; and everyone has read this pair's fragments: buffer pb is free
            read_frags(lds + (pb ^ 1) * PBUF, 0, F);
            pin();
        }
        if constexpr (MORE2) {  // round 3 of G with the loads of pair p+2 (into this pair's buffer), one behind every MFMA
#ifdef PHONLP_ABLATION
            const long so = a.dbg == 4 ? 0 : (long)(p + 2) * 128;
#else
            const long so = (long)(p + 2) * 128;
#endif
            const unsigned ld = lds0 + (unsigned)(pb * PBUF);
#define PHONLP_X3_ONE_DMA(I, J, Q)                                      \
    acc[I][J] = PHONLP_F16_MFMA(G.ah[I], G.bh[J], acc[I][J]);           \
    asm volatile("" : "+v"(acc[I][J]));                                 \
    lds_dma16(src[Q] + so, ld + Q * 8192);
            PHONLP_X3_ONE_DMA(0, 0, 0)
            PHONLP_X3_ONE_DMA(0, 1, 1)
            PHONLP_X3_ONE_DMA(1, 0, 2)
            PHONLP_X3_ONE_DMA(1, 1, 3)
            PHONLP_X3_ONE_DMA(2, 0, 4)
            PHONLP_X3_ONE_DMA(2, 1, 5)
            PHONLP_X3_ONE_DMA(3, 0, 6)
            PHONLP_X3_ONE_DMA(3, 1, 7)
#undef PHONLP_X3_ONE_DMA
        } else {
            PHONLP_X3_ROUND(G, ah, bh)
        }
    };
    issue_pair(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    read_frags(lds, 0, F);
    if (npair > 1) issue_pair(1, 1);
    int p = 0;
    for (; p + 2 < npair; ++p) body(p, std::true_type{}, std::true_type{});
    if (p + 1 < npair) { body(p, std::true_type{}, std::false_type{}); ++p; }
    body(p, std::false_type{}, std::false_type{});
#undef PHONLP_X3_ROW
#undef PHONLP_X3_ROUND
    lds_barrier();  // the epilogue's LDS patch aliases the pair buffers
#ifdef PHONLP_ABLATION
    if (a.dbg == 2) {  // timing only: keep the accumulators alive without the epilogue
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) t += acc[i][j][e];
        if (t == 1234.5f) a.Y[tid] = t;
        return;
    }
#endif
    if (a.rows4 == 2) {
        at_heads_epilogue<TM, TN>(acc, a, m0, n0, wm, wn, lane);
        return;
    }
    if (a.rows4) {
        rows4_epilogue<TM, TN>(acc, a, m0, n0, wm, wn, lane);
        return;
    }
    linear_epilogue<TM, TN, WM, WN>(acc, reinterpret_cast<float *>(lds), a, m0, n0, wm, wn, lane, wave);
}

// tile t of the walk (GM-tall groups of row tiles, column-major inside a group) -> its origin
__device__ __forceinline__ void wide_tile_origin(const LinearArgs &a, int t, int &m0, int &n0) {
    const int GM = a.gm > 0 ? a.gm : 8;
    const int per_group = GM * a.num_n;
    const int g = t / per_group;
    const int first_m = g * GM;
    const int gsz = (a.num_m - first_m) < GM ? (a.num_m - first_m) : GM;
    const int in = t - g * per_group;
    m0 = (first_m + in % gsz) * 256;
    n0 = (in / gsz) * 256;
}

// The wide kernel as a PERSISTENT workgroup for the plain dense layers taken transposed (EPI 1: the result as a split
// image, EPI 2: as fp32 rows; transposed_epilogue above): one workgroup per CU takes every stride-th tile of its XCD's chunk
// of the walk, and the operand loads of a tile's first two pairs are issued during the LAST two pairs of the tile before
// it, so they land behind that tile's epilogue: no workgroup launch and no exposed first-load latency per tile.  The pair
// pipeline is the one above, continued across tiles (the buffer parity runs on); the last pair of a tile only takes a
// barrier where the others wait for the next pair and read its fragments -- those registers are the epilogue's -- and the
// next tile's first fragments are read behind the epilogue.  Needs a.K / 32 >= 4.  No waiting on other workgroups.  (The
// labels-at-heads form of this kernel is linear_f16x3_wide16_heads_kernel below.)
template <int EPI>
__global__ __launch_bounds__(512, 1) void linear_f16x3_wide_persist_kernel(LinearArgs a) {
    constexpr int BM = 256, BN = 256, WN = 4, TM = 4, TN = 2;
    constexpr int ROWS = BM + BN, PBUF = ROWS * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int npair = a.K >> 5;
    const int xg = blockIdx.x & 7;
    const int ntiles = a.num_m * a.num_n, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int chunk0 = xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8;  // this XCD's tiles: chunk0 .. chunk0 + chunkn
    const int chunkn = q8 + (xg < r8 ? 1 : 0);
    // no counter to zero (these layers are launched from several streams at once): workgroup jl of the XCD takes tiles jl,
    // jl + stride, ... of the chunk
    const int stride = gridDim.x >> 3;
    const int first = blockIdx.x >> 3;
    if (first >= chunkn) return;
    int cur_tile = first;
    int m0, n0;
    wide_tile_origin(a, chunk0 + first, m0, n0);

    const int lr = lane >> 3;
    const int chunk = (lane & 7) ^ (((wave & 1) << 2) + (lane >> 4));
    const unsigned char *src[8];
    auto set_src = [&](int mm, int nn) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int row = mm + 64 * q + 8 * wave + lr;
            row = row < a.M ? row : a.M - 1;
            src[q] = reinterpret_cast<const unsigned char *>(a.X) + (long)row * a.ldx + chunk * 16;
        }
#pragma unroll
        for (int q = 4; q < 8; ++q) {
            int row = nn + 64 * (q - 4) + 8 * wave + lr;
            row = row < a.N ? row : a.N - 1;
            src[q] = reinterpret_cast<const unsigned char *>(a.W) + (long)row * a.ldw + chunk * 16;
        }
#ifdef PHONLP_ABLATION
        if (a.dbg == 4)  // timing only: every load from the same 128 bytes
#pragma unroll
            for (int q = 0; q < 8; ++q) src[q] = reinterpret_cast<const unsigned char *>(a.X) + chunk * 16;
#endif
    };
    set_src(m0, n0);
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    const int r = lane & 31, hh = lane >> 5;
    const int sw = (r >> 1) & 7;
    const int off_h = ((2 * hh) ^ sw) << 4, off_l = ((2 * hh + 1) ^ sw) << 4;
    const int a_ofs = (wm * 128 + r) * 128;
    const int b_ofs = (BM + wn * 64 + r) * 128;

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    struct Frags {
        u32x4 ah[TM], al[TM], bh[TN], bl[TN];
    };
    auto read_frags = [&](const unsigned char *buf, int t, Frags &f) {
        const int oh = off_h ^ (t << 6), ol = off_l ^ (t << 6);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.bh[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 4096 + oh);
            f.bl[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 4096 + ol);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.al[i] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + i * 4096 + ol);
            f.ah[i] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + i * 4096 + oh);
        }
    };
    auto pin = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));
    };
#define PHONLP_X3_ROUND(FR, AA, BB)                \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = PHONLP_F16_MFMA(FR.AA[i], FR.BB[j], acc[i][j]);
    Frags F, G;
    // One pair out of buffer pb.  JOIN: 2 = wait for the following pair, barrier, read its first fragments; 3 = barrier
    // only, and only if `flag`; 0 = nothing.  LOADS: 1 = the pair two ahead (byte offset `so` of the rows in src[]) goes
    // into this pair's buffer between the last eight MFMAs (the barrier is what frees the buffer); 2 = the same if `flag`
    // (wave-uniform: the branches hold nothing but the issue -- two code paths with MFMAs in them made hipcc shuffle the
    // accumulators at the merge and spill a thousand registers); 0 = none.
    // (the row operand is the weight image, so the two small terms are taken in the order ah.bl, al.bh -- the
    // words' low pieces first, as the 256 x 128 kernel does with the words as rows: products commute exactly and k runs the
    // same way, so both kernels give the same bits and a sentence's results do not depend on which of them a batch size picks)
    auto body = [&](int pb, long so, auto join, auto loads, bool flag) {
        constexpr int JOIN = decltype(join)::value, LOADS = decltype(loads)::value;
        const unsigned char *buf = lds + pb * PBUF;
        PHONLP_X3_ROUND(F, ah, bl)
        PHONLP_X3_ROUND(F, al, bh)
        pin();
        read_frags(buf, 1, G);
        pin();
        PHONLP_X3_ROUND(F, ah, bh)
        pin();
        PHONLP_X3_ROUND(G, ah, bl)
        PHONLP_X3_ROUND(G, al, bh)
        pin();
        if constexpr (JOIN == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
            read_frags(lds + (pb ^ 1) * PBUF, 0, F);
            pin();
        } else if constexpr (JOIN == 3) {
            if (flag) lds_barrier();
        }
        if constexpr (LOADS != 0) {
#ifdef PHONLP_ABLATION
            if (a.dbg == 4) so = 0;
#endif
            const unsigned ld = lds0 + (unsigned)(pb * PBUF);
#define PHONLP_X3_ONE_DMA(I, J, Q)                                      \
    acc[I][J] = PHONLP_F16_MFMA(G.ah[I], G.bh[J], acc[I][J]);           \
    asm volatile("" : "+v"(acc[I][J]));                                 \
    if (LOADS == 1 || flag) lds_dma16(src[Q] + so, ld + Q * 8192);
            PHONLP_X3_ONE_DMA(0, 0, 0)
            PHONLP_X3_ONE_DMA(0, 1, 1)
            PHONLP_X3_ONE_DMA(1, 0, 2)
            PHONLP_X3_ONE_DMA(1, 1, 3)
            PHONLP_X3_ONE_DMA(2, 0, 4)
            PHONLP_X3_ONE_DMA(2, 1, 5)
            PHONLP_X3_ONE_DMA(3, 0, 6)
            PHONLP_X3_ONE_DMA(3, 1, 7)
#undef PHONLP_X3_ONE_DMA
        } else {
            PHONLP_X3_ROUND(G, ah, bh)
        }
    };
    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using C3 = std::integral_constant<int, 3>;
    // first tile: pairs 0 and 1
#pragma unroll
    for (int q = 0; q < 8; ++q) lds_dma16(src[q], lds0 + (unsigned)(q * 8192));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    read_frags(lds, 0, F);
#pragma unroll
    for (int q = 0; q < 8; ++q) lds_dma16(src[q] + 128, lds0 + (unsigned)(PBUF + q * 8192));
    int pb = 0;
    for (;;) {
        for (int p = 0; p + 3 < npair; ++p) {
            body(pb, (long)(p + 2) * 128, C2{}, C1{}, true);
            pb ^= 1;
        }
        body(pb, (long)(npair - 1) * 128, C2{}, C1{}, true);  // the tile's last loads
        pb ^= 1;
        const int nxt = cur_tile + stride;
        cur_tile = nxt;
        const bool has_next = nxt < chunkn;
        int m1 = m0, n1 = n0;
        if (has_next) {
            wide_tile_origin(a, chunk0 + nxt, m1, n1);
            set_src(m1, n1);
        }
        body(pb, 0, C2{}, C2{}, has_next);    // pair npair-2: the next tile's pair 0 takes its buffer
        pb ^= 1;
        body(pb, 128, C3{}, C2{}, has_next);  // pair npair-1: the next tile's pair 1 takes its buffer
        pb ^= 1;
#ifdef PHONLP_ABLATION
        if (a.dbg == 2) {  // timing only: keep the accumulators alive without the epilogue
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) t += acc[i][j][e];
            if (t == 1234.5f) a.P[tid] = t;
        } else
#endif
        {
            transposed_epilogue<TM, TN, EPI == 1>(acc, a, m0, n0, wm, wn, lane, lds, 2u * PBUF + wave * 4096u);
        }
        if (!has_next) break;
        zero_acc();
        m0 = m1;
        n0 = n1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's pairs 0 and 1 (issued before the epilogue)
        lds_barrier();
        read_frags(lds + pb * PBUF, 0, F);
    }
#undef PHONLP_X3_ROUND
}

// ---------------------------------------------------------------------------------------------------------------------
// The persistent wide kernel for the labels-at-heads epilogue on v_mfma_f32_16x16x32_f16.  Same tile (256 x 256, eight
// waves of 128 x 64, k32 pairs, two pair buffers, tiles drawn per XCD), same three products per operand pair; what
// changes is the instruction shape: one MFMA takes a whole k32 pair of a 16 x 16 accumulator (four passes) instead of
// a k16 stage of a 32 x 32 one (eight).  MI355X_MICROARCH.md measures 1.12-1.15 x the FLOP/s for this shape at the clock
// the chip holds under MFMA load, at equal cycles per FLOP and equal LDS traffic (a fragment is 16 rows x 32 k instead
// of 32 rows x 16 k: 24 ds_read_b128 per pair and wave either way).
//
// Fragment of lane l: row l & 15, k-group g = l >> 4 (k = 8g .. 8g + 7 of the pair), pieces h and l: two 16-byte chunks
// of the row's 128 bytes.  LDS slot of chunk (g, piece) of row R: (piece + {0, 6, 2, 4}[g]) ^ ((R >> 1) & 7) -- the
// hardware serves a ds_read_b128 sixteen lanes at a time, lanes {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} of each
// half-wave, i.e. eight rows with k-group g and eight with g + 1: with the chunk numbers of g and g + 1 differing by 6
// every such group covers all 64 banks once (SQ_LDS_BANK_CONFLICT = 0, tools/pmc_fused.sh).  The DMA side applies
// the mapping on the global addresses as before.
//
// Schedule of a pair (ah, al: row fragments of the wave's eight 16-row tiles; bh, bl: of its four column tiles; 96
// registers as before): read bh | rows 0-3: ah.bl | read rows 4-7 | rows 0-3: al.bh, ah.bh | rows 4-7: ah.bl, al.bh |
// wait for the next pair + barrier | read the next pair's rows 0-3 and bl | rows 4-7: ah.bh with the eight LDS-DMA
// issues of the pair after next between its MFMAs.  Every accumulator takes its three products in the same order
// (ah.bl, al.bh, ah.bh); results differ from the 32 x 32 kernels' in the last bits (the order inside a k32 sum).
typedef float f32x4v __attribute__((ext_vector_type(4)));
#define PHONLP_F16_MFMA16(A, B, C) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)

// labels-at-heads epilogue in the 16 x 16 accumulator layout: a lane holds ONE word (column l & 15) and four consecutive
// rows 4 (l >> 4) .. + 3 of each 16-row tile; a relation's 32 head features are two such tiles; the sum over them is
// eight products per lane plus two exchanges between the four lane groups; lane group q then stores column tile q, so a
// (32-row group, 64 words) share leaves as ONE 256-byte store.
__device__ __forceinline__ void at_heads_epilogue16(f32x4v (&acc)[8][4], const LinearArgs &a, int m0, int n0, int wm, int wn,
                                                    int lane) {
    asm volatile("" : "+v"(lane));  // (opaque: nothing lane-dependent of the epilogue is hoisted out of the tile loop and kept -- or
                                    // spilled -- across the main loop)
    const int q = lane >> 4, c16 = lane & 15;
    int word[4];
    float cs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        word[j] = n0 + wn * 64 + 16 * j + c16;
        cs[j] = word[j] < a.N ? a.col_scale[word[j]] : 0.f;
    }
    const int grp0 = __builtin_amdgcn_readfirstlane((m0 >> 5) + wm * 4);
    float4 x[4][2];
    int cur_blk = -1;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
        const int grp = grp0 + gi, tile_row = grp << 5;
        const int blk = grp / a.nrel;
        const bool tv = tile_row < a.M;
        if (blk != cur_blk) {  // wave-uniform: the wave's four relations nearly always share one block of 32 head features
            cur_blk = blk;
#ifdef PHONLP_ABLATION
            if (a.dbg == 5) {  // timing only: no loads of the heads' features (is their latency exposed?)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[j][hf] = make_float4(cs[j], cs[j], 1.f, 2.f);
            } else
#endif
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int d2 = blk * 32 + 16 * hf + 4 * q;
                const long xo = (long)(tv && d2 < a.kpx ? d2 >> 2 : 0) * a.ldx2 * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    x[j][hf] = *reinterpret_cast<const float4 *>(a.x2 + (word[j] < a.N ? word[j] : 0) * 4L + xo);
            }
        }
        float4 rs[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const bool valid = tv && blk * 32 + 16 * hf + 4 * q < a.kpx;
            rs[hf] = valid ? *reinterpret_cast<const float4 *>(a.row_scale + tile_row + 16 * hf + 4 * q)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        float sums[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sum = 0.f;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {  // ascending feature index: the order of the sum is part of the result
                sum += acc[2 * gi + hf][j][0] * rs[hf].x * x[j][hf].x;
                sum += acc[2 * gi + hf][j][1] * rs[hf].y * x[j][hf].y;
                sum += acc[2 * gi + hf][j][2] * rs[hf].z * x[j][hf].z;
                sum += acc[2 * gi + hf][j][3] * rs[hf].w * x[j][hf].w;
            }
            sum *= cs[j];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            sums[j] = sum;
        }
        const float mine = q == 0 ? sums[0] : (q == 1 ? sums[1] : (q == 2 ? sums[2] : sums[3]));
        const int wq = n0 + wn * 64 + lane;  // = the word of column tile q at this lane's column
        if (wq < a.N && tv) a.P[(long)grp * a.N + wq] = mine;
    }
}

__global__ __launch_bounds__(512, 1) void linear_f16x3_wide16_heads_kernel(LinearArgs a) {
    constexpr int BM = 256, ROWS = 512, PBUF = ROWS * 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ int s_next;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int npair = a.K >> 5;
    const int xg = blockIdx.x & 7;
    const int ntiles = a.num_m * a.num_n, q8 = ntiles >> 3, r8 = ntiles & 7;
    const int chunk0 = xg < r8 ? xg * (q8 + 1) : r8 * (q8 + 1) + (xg - r8) * q8;
    const int chunkn = q8 + (xg < r8 ? 1 : 0);
    if (tid == 0) s_next = atomicAdd(a.tile_ctr + xg, 1);
    __syncthreads();
    const int first = __builtin_amdgcn_readfirstlane(s_next);
    if (first >= chunkn) return;
    int m0, n0;
    wide_tile_origin(a, chunk0 + first, m0, n0);

    // DMA: wave-load u = 8q + wave fills rows 8u .. 8u+7; lane i writes slot i & 7 of row i >> 3, fetching the global chunk whose
    // slot that is: global chunk (2g + piece) lives at slot (piece + {0,6,2,4}[g]) ^ ((R >> 1) & 7)
    // The lane's eight loads of a pair: rows 64 q + (8 wave + lr) of the tile's 256 A rows (q < 4) and B rows, chunk `chunk`
    // of the pair's 128 bytes.  Address = uniform base (image + tile origin + pair, SGPRs) + a 32-bit lane offset that is
    // computed per issue from ONE register per operand (the offset inside a 64-row group) by an add and a min: the min
    // against the lane's last valid offset in the tile clamps rows beyond the matrix to its last row, as the 32 x 32 kernels'
    // eight precomputed pointers do -- sixteen registers there, four here.
    const unsigned char *const baseX = reinterpret_cast<const unsigned char *>(a.X);
    const unsigned char *const baseW = reinterpret_cast<const unsigned char *>(a.W);
    unsigned offA, offB, maxA = 0, maxB = 0;
    const unsigned char *tileX = baseX, *tileW = baseW;
    auto lane_chunk16 = [&]() {
        int ln = lane;
        asm volatile("" : "+v"(ln));  // (recomputed where needed rather than kept in a register across the main loop)
        const int cphys = (ln & 7) ^ (((wave & 1) << 2) + (ln >> 4));  // (slot) ^ ((R >> 1) & 7), R = 64q + 8 wave + (ln >> 3)
        return (unsigned)((0x32765410 >> (4 * cphys)) & 7) * 16u;      // physical chunk number -> global chunk 2g + piece
    };
    {
        const unsigned r8 = 8 * wave + (lane >> 3), c16 = lane_chunk16();
        offA = r8 * (unsigned)a.ldx + c16;
        offB = r8 * (unsigned)a.ldw + c16;
    }
    auto set_src = [&](int mm, int nn) {
        const unsigned c16 = lane_chunk16();
        tileX = baseX + (long)mm * a.ldx;
        tileW = baseW + (long)nn * a.ldw;
        maxA = (unsigned)(a.M - 1 - mm) * (unsigned)a.ldx + c16;  // (the tile's origin is inside the matrix)
        maxB = (unsigned)(a.N - 1 - nn) * (unsigned)a.ldw + c16;
    };
    const unsigned strideA = 64u * (unsigned)a.ldx, strideB = 64u * (unsigned)a.ldw;
    auto dma = [&](int q, const unsigned char *sx, const unsigned char *sw, unsigned lds_addr) {  // sx / sw: tile base + pair offset
        if (q < 4) {
            const unsigned v = offA + (unsigned)q * strideA;
            lds_dma16_s(sx, v < maxA ? v : maxA, lds_addr);
        } else {
            const unsigned v = offB + (unsigned)(q - 4) * strideB;
            lds_dma16_s(sw, v < maxB ? v : maxB, lds_addr);
        }
    };
    set_src(m0, n0);
    const unsigned lds0 = (unsigned)(uintptr_t)lds + (unsigned)__builtin_amdgcn_readfirstlane(wave) * 1024u;

    const int r16 = lane & 15, g = lane >> 4;
    const int c0 = (0x4260 >> (4 * g)) & 7;  // physical chunk number of (g, h); (g, l) = + 1
    const int sw = r16 >> 1;
    const int off_h = (c0 ^ sw) << 4, off_l = ((c0 + 1) ^ sw) << 4;
    const int a_ofs = (wm * 128 + r16) * 128;
    const int b_ofs = (BM + wn * 64 + r16) * 128;

    f32x4v acc[8][4];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();
    // Row fragments in TWO sets of two 16-row tiles (the wave's eight row tiles = four groups G0..G3; G0 and G2 use set 0,
    // G1 and G3 set 1): a set is re-read as soon as its group's 24 MFMAs have been issued, and the other group's 24 MFMAs
    // (384+ cycles of the matrix pipe) cover the read.  64 fragment registers instead of 96.
    u32x4 ah[2][2], al[2][2], bh[4], bl[4];
    auto pin = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(acc[i][j]));
    };
    auto read_g = [&](const unsigned char *buf, int G, int S) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            al[S][t] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + (2 * G + t) * 2048 + off_l);
            ah[S][t] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + (2 * G + t) * 2048 + off_h);
        }
    };
    auto read_bl = [&](const unsigned char *buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bl[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 2048 + off_l);
    };
    auto read_bh = [&](const unsigned char *buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bh[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 2048 + off_h);
    };
#define PHONLP_W16_ROUND(G, S, AA, BB)                 \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)      \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)      \
        acc[2 * (G) + t][j] = PHONLP_F16_MFMA16(AA[S][t], BB[j], acc[2 * (G) + t][j]);
#define PHONLP_W16_GROUP(G, S)        \
    PHONLP_W16_ROUND(G, S, ah, bl)    \
    PHONLP_W16_ROUND(G, S, al, bh)    \
    PHONLP_W16_ROUND(G, S, ah, bh)
    // One pair out of buffer pb, the same code for every pair of every tile (no branch in it: specialised tail bodies made
    // hipcc move the accumulators between them and spill; branches around the LDS reads made it read into temporaries, wait,
    // and copy).  The eight loads at byte offset `so` of the rows set by set_src() go into this pair's buffer between the last
    // sixteen MFMAs: the pair after next of the tile, or the first two pairs of the NEXT tile during a tile's last two pairs (a
    // workgroup's last tile loads its own first pairs again: two wasted pair loads per workgroup).
    // Schedule: read bh, G1 | G0 | read G2 | G1 | read G3 | G2 | G3: ah.bl | wait for the next pair + barrier | read the next
    // pair's G0 and bl | G3: al.bh, ah.bh with the LDS-DMA issues between the MFMAs.
    auto body = [&](int pb, long so) {
        const unsigned char *buf = lds + pb * PBUF;
        const unsigned char *sx = tileX + so, *sw = tileW + so;
        read_bh(buf);
        read_g(buf, 1, 1);
        pin();
        PHONLP_W16_GROUP(0, 0)
        pin();
        read_g(buf, 2, 0);
        pin();
        PHONLP_W16_GROUP(1, 1)
        pin();
        read_g(buf, 3, 1);
        pin();
        PHONLP_W16_GROUP(2, 0)
        pin();
        PHONLP_W16_ROUND(3, 1, ah, bl)
        pin();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next pair (issued 80+ MFMAs ago) has landed
        lds_barrier();  // ... everyone's; and everyone has read this pair's fragments: buffer pb is free
        read_g(lds + (pb ^ 1) * PBUF, 0, 0);
        read_bl(lds + (pb ^ 1) * PBUF);
        pin();
        const unsigned ld = lds0 + (unsigned)(pb * PBUF);
#define PHONLP_W16_TWO_DMA(AA, T, J, Q)                                                  \
    acc[6 + T][J] = PHONLP_F16_MFMA16(AA[1][T], bh[J], acc[6 + T][J]);                   \
    acc[6 + T][J + 1] = PHONLP_F16_MFMA16(AA[1][T], bh[J + 1], acc[6 + T][J + 1]);       \
    asm volatile("" : "+v"(acc[6 + T][J]), "+v"(acc[6 + T][J + 1]));                     \
    dma(Q, sx, sw, ld + Q * 8192);
        PHONLP_W16_TWO_DMA(al, 0, 0, 0)
        PHONLP_W16_TWO_DMA(al, 0, 2, 1)
        PHONLP_W16_TWO_DMA(al, 1, 0, 2)
        PHONLP_W16_TWO_DMA(al, 1, 2, 3)
        PHONLP_W16_TWO_DMA(ah, 0, 0, 4)
        PHONLP_W16_TWO_DMA(ah, 0, 2, 5)
        PHONLP_W16_TWO_DMA(ah, 1, 0, 6)
        PHONLP_W16_TWO_DMA(ah, 1, 2, 7)
#undef PHONLP_W16_TWO_DMA
    };
    // first tile: pairs 0 and 1
#pragma unroll
    for (int q = 0; q < 8; ++q) dma(q, tileX, tileW, lds0 + (unsigned)(q * 8192));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    read_g(lds, 0, 0);
    read_bl(lds);
#pragma unroll
    for (int q = 0; q < 8; ++q) dma(q, tileX + 128, tileW + 128, lds0 + (unsigned)(PBUF + q * 8192));
    int pb = 0;
    for (;;) {
        int drawn = 0;  // the next tile's number: drawn now, needed npair - 3 pairs from here
        if (tid == 0) drawn = atomicAdd(a.tile_ctr + xg, 1);
        bool has_next = false;
        int m1 = m0, n1 = n0;
#pragma unroll 1
        for (int p = 0; p < npair; ++p) {
            if (p == npair - 3 && tid == 0) s_next = drawn;  // published by this pair's barrier
            if (p == npair - 2) {  // from here on the loads are the next tile's first two pairs
                const int nxt = __builtin_amdgcn_readfirstlane(s_next);
                has_next = nxt < chunkn;
                if (has_next) {
                    wide_tile_origin(a, chunk0 + nxt, m1, n1);
                    set_src(m1, n1);
                }
            }
            body(pb, (long)(p + 2 < npair ? p + 2 : p + 2 - npair) * 128);
            pb ^= 1;
        }
        // (the last pair's barrier stood behind the arrival of the next tile's pair 0, whose first fragments are in registers)
        at_heads_epilogue16(acc, a, m0, n0, wm, wn, lane);
        if (!has_next) break;
        zero_acc();
        m0 = m1;
        n0 = n1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last tile's surplus loads land before the workgroup's LDS is released
#undef PHONLP_W16_GROUP
#undef PHONLP_W16_ROUND
}

int launch_linear_f16x3(const char *tag, const void *Xp, long pitch_x, const float *x_inv, const void *Wp, long pitch_w,
                        const float *w_inv, const float *bias, float *Y, long ldy, int M, int N, int K16, int relu,
                        const int32_t *sentlens, int L, hipStream_t stream, const float *pad_bias, int rows4 = 0,
                        const LinearArgs *at_heads = nullptr, const LinearArgs *image = nullptr) {
    if (!Xp || !Wp || !x_inv || !w_inv || (!Y && !at_heads && !image) || M <= 0 || N <= 0 || K16 <= 0)
        return fail(PHONLP_ERR_SHAPE, "linear_split: bad argument");
    if (image && (!image->img || !image->img_inv || (N & 7) || (image->img_pitch & 15) || image->img_pitch < 4L * N ||
                  ((uintptr_t)image->img & 15) || rows4 || at_heads || !(image->img_c >= 0.f) || !(image->img_b >= 0.f)))
        return fail(PHONLP_ERR_SHAPE, "linear_split: image output needs N %% 8 == 0, a 16-byte aligned image and its bound");
    if ((K16 & 15) || (pitch_x & 15) || (pitch_w & 15) || pitch_x < 4L * K16 || pitch_w < 4L * K16 ||
        ((uintptr_t)Xp & 15) || ((uintptr_t)Wp & 15))
        return fail(PHONLP_ERR_SHAPE, "linear_split: split images must be 16-byte aligned with K16 %% 16 == 0");
    if (sentlens && L <= 0) return fail(PHONLP_ERR_SHAPE, "linear_split: sentlens given but L <= 0");
    if (pad_bias && !sentlens) return fail(PHONLP_ERR_SHAPE, "linear_split: pad_bias needs sentlens");
    LinearArgs a{reinterpret_cast<const float *>(Xp), reinterpret_cast<const float *>(Wp), bias, pad_bias, Y, sentlens,
                 pitch_x, pitch_w, ldy, M, N, K16, L, relu, ceil_div(M, 256), ceil_div(N, 128), x_inv, w_inv};
    constexpr int LDS_DMA = 3 * 384 * 64;
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.pending(&attr_dev)) {
        const void *fns[] = {(const void *)linear_f16x3_dma_kernel<0, false, true>,
                             (const void *)linear_f16x3_dma_kernel<1, false, true>,
                             (const void *)linear_f16x3_dma_kernel<0, true, true>,
                             (const void *)linear_f16x3_dma_kernel<1, true, true>};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DMA) != hipSuccess)
                return fail(PHONLP_ERR_LAUNCH, "linear_split: cannot raise the dynamic LDS limit");
        attr_once.mark(attr_dev);
    }
    const bool stage_a = tag && !strcmp(tag, "linear:label_stage_a");  // TAG only names the instantiation for the profiler
#ifdef PHONLP_ABLATION
    static const int abl = getenv("PHONLP_X3_ABLATION") ? atoi(getenv("PHONLP_X3_ABLATION")) : 0;  // timing only
    a.dbg = abl;
#endif
    a.rows4 = rows4;
    if (image) {
        a.img = image->img; a.img_pitch = image->img_pitch; a.img_inv = image->img_inv;
        a.img_c = image->img_c; a.img_b = image->img_b;
    }
    if (at_heads) {
        a.rows4 = 2;
        a.x2 = at_heads->x2; a.P = at_heads->P; a.ldx2 = at_heads->ldx2;
        a.kpc = at_heads->kpc; a.kpx = at_heads->kpx; a.nrel = at_heads->nrel;
        a.gm = at_heads->gm;  // row tiles per walk group (8 wide, 4 otherwise: flat between 4 and 16, 1 and 54 are 3-5 % behind)
    }
    if (rows4 == 1 && (sentlens || relu || ((uintptr_t)Y & 15)))
        return fail(PHONLP_ERR_SHAPE, "linear_split: the rows-by-4 output takes no activation / padding mask");
    constexpr int LDS_WIDE = 2 * 512 * 128, LDS_WIDE_T = LDS_WIDE + 8 * 4096;  // + the transposed epilogue's patch per wave
    // Persistent wide kernels (256 x 256 tile, k32 pairs, one workgroup per CU): attributes and the CU count, once per device
    static PerDeviceOnce persist_once;
    static int n_cu[64];
    auto persist_cus = [&]() -> int {
        int pdev;
        if (persist_once.pending(&pdev)) {
            int cus = 0;
            const void *pf[] = {(const void *)linear_f16x3_wide16_heads_kernel, (const void *)linear_f16x3_wide_persist_kernel<1>,
                                (const void *)linear_f16x3_wide_persist_kernel<2>};
            for (int k = 0; k < 3; ++k)
                if (hipFuncSetAttribute(pf[k], hipFuncAttributeMaxDynamicSharedMemorySize, k ? LDS_WIDE_T : LDS_WIDE) != hipSuccess)
                    return -1;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, pdev) != hipSuccess || cus < 8) return -1;
            n_cu[pdev & 63] = cus;
            persist_once.mark(pdev);
        }
        int cur_dev = 0;
        (void)hipGetDevice(&cur_dev);
        return n_cu[cur_dev & 63] & ~7;
    };
    // few tiles: the small form (64 x 64 tiles; a ring of 16 stages with one workgroup per CU while every tile has a CU of its
    // own, of 8 with two per CU beyond)
    {
        constexpr long small_max = 1280;  // measured crossover (M = 2048: 1248 tiles of the arc layer still win, M = 8192 loses by 40 %)
        const long nsmall = (long)ceil_div(M, 64) * ceil_div(N, 64);
        if (!rows4 && !at_heads && nsmall <= small_max) {
            constexpr int STAGE_SMALL = 128 * 64;
            static PerDeviceOnce small_once;
            int sdev;
            if (small_once.pending(&sdev)) {
                if (hipFuncSetAttribute((const void *)linear_f16x3_small_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        16 * STAGE_SMALL) != hipSuccess ||
                    hipFuncSetAttribute((const void *)linear_f16x3_small_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        8 * STAGE_SMALL) != hipSuccess)
                    return fail(PHONLP_ERR_LAUNCH, "linear_split: cannot raise the dynamic LDS limit");
                small_once.mark(sdev);
            }
            a.num_m = ceil_div(M, 64);
            a.num_n = ceil_div(N, 64);
            ProfScope pss(tag, stream);
            if (nsmall <= 256)
                hipLaunchKernelGGL((linear_f16x3_small_kernel<16>), dim3((unsigned)nsmall), dim3(256), 16 * STAGE_SMALL, stream, a);
            else
                hipLaunchKernelGGL((linear_f16x3_small_kernel<8>), dim3((unsigned)nsmall), dim3(256), 8 * STAGE_SMALL, stream, a);
            return check_launch("linear_f16x3_small_kernel");
        }
    }
    // The plain layers at large sizes (W1/W2 of the scorers, upos_hid) on the persistent wide kernel, taken TRANSPOSED (row
    // operand = the weight image, column operand = the words): the accumulator lane then holds four consecutive features of
    // ONE word, so the epilogue stores 16-byte (fp32 rows) or 8-byte (split image) pieces straight from the registers -- the
    // row-major epilogue's LDS transposition was what made the wide tile lose on these layers (one workgroup per CU: nothing
    // hides it).  Same bits as the 256 x 128 kernel below (see the kernel), which keeps the small products and the two
    // layers the step runs BESIDE the MST decode (label W1/W2, the tagger's hidden layer): its 72 KB workgroups share a CU
    // with a decode's 86 KB, this kernel's 128 KB would wait for the decodes to leave (label W1/W2 0.32 -> 0.53 ms).
    const bool beside_decode = tag && (!strcmp(tag, "linear:label_w12") || !strcmp(tag, "linear:tagger_hid"));
    if (!beside_decode && !rows4 && !at_heads && (K16 & 31) == 0 && (K16 >> 5) >= 4 && (N & 3) == 0 && !((uintptr_t)w_inv & 15) &&
        !((uintptr_t)bias & 15) && !((uintptr_t)pad_bias & 15) && (long)ceil_div(N, 256) * ceil_div(M, 256) >= 64) {
        const int cap = persist_cus();
        if (cap < 0) return fail(PHONLP_ERR_LAUNCH, "linear_split: cannot set up the persistent wide kernel");
        LinearArgs t = a;
        t.X = reinterpret_cast<const float *>(Wp); t.ldx = pitch_w; t.M = N; t.row_scale = w_inv;
        t.W = reinterpret_cast<const float *>(Xp); t.ldw = pitch_x; t.N = M; t.col_scale = x_inv;
        t.num_m = ceil_div(N, 256); t.num_n = ceil_div(M, 256);
        t.gm = t.num_m <= 16 ? t.num_m : 8;  // all of a word tile's feature tiles in one group of the walk
        const int ntiles_t = t.num_m * t.num_n, want = ntiles_t & ~7;
        ProfScope pst(tag, stream);
        const dim3 gt(want < cap ? want : cap);
        if (image) hipLaunchKernelGGL(linear_f16x3_wide_persist_kernel<1>, gt, dim3(512), LDS_WIDE_T, stream, t);
        else hipLaunchKernelGGL(linear_f16x3_wide_persist_kernel<2>, gt, dim3(512), LDS_WIDE_T, stream, t);
        return check_launch("linear_f16x3_wide_persist_kernel");
    }
    // the wide form for the fused stage A (its epilogue is light)
    if ((K16 & 31) == 0 && at_heads) {
        static PerDeviceOnce wide_once;
        int wdev;
        if (wide_once.pending(&wdev)) {
            const void *wf[] = {(const void *)linear_f16x3_wide_kernel<0>, (const void *)linear_f16x3_wide_kernel<1>};
            for (const void *f : wf)
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_WIDE) != hipSuccess)
                    return fail(PHONLP_ERR_LAUNCH, "linear_split: cannot raise the dynamic LDS limit");
            wide_once.mark(wdev);
        }
        a.num_n = ceil_div(N, 256);
        // persistent form for the labels-at-heads epilogue (four pairs per tile at least): a workgroup per CU, or per tile when there
        // are fewer tiles than CUs (small batches: B = 1, L = 32 is 54 tiles -- two or three tiles per workgroup doubled its time)
        const int ntiles_w = a.num_m * a.num_n;
        if (at_heads->tile_ctr && (K16 >> 5) >= 4 && ntiles_w >= 16) {
            const int cap = persist_cus(), want = (ntiles_w + 7) & ~7;
            if (cap < 0) return fail(PHONLP_ERR_LAUNCH, "linear_split: cannot set up the persistent stage A");
            a.tile_ctr = at_heads->tile_ctr;
            ProfScope psh(tag, stream);
            hipLaunchKernelGGL(linear_f16x3_wide16_heads_kernel, dim3(want < cap ? want : cap), dim3(512), LDS_WIDE, stream, a);
            return check_launch("linear_f16x3_wide16_heads_kernel");
        }
        ProfScope psw(tag, stream);
        const dim3 gw(a.num_m * a.num_n);
        if (stage_a) hipLaunchKernelGGL((linear_f16x3_wide_kernel<1>), gw, dim3(512), LDS_WIDE, stream, a);
        else hipLaunchKernelGGL((linear_f16x3_wide_kernel<0>), gw, dim3(512), LDS_WIDE, stream, a);
        return check_launch("linear_f16x3_wide_kernel");
    }
    ProfScope ps(tag, stream);
    const dim3 g(a.num_m * a.num_n);
    // the register-prefetch loop wins on long K (W1/W2 layers), the plain loop on short K (stage A): K16 picks; both with the
    // LDS-DMA issues spread between the MFMAs (3-7 % on every layer over a burst at the top of the stage)
    const bool prefetch = K16 >= 512;
    if (!prefetch) {
        if (stage_a) hipLaunchKernelGGL((linear_f16x3_dma_kernel<1, false, true>), g, dim3(256), LDS_DMA, stream, a);
        else hipLaunchKernelGGL((linear_f16x3_dma_kernel<0, false, true>), g, dim3(256), LDS_DMA, stream, a);
    } else {
        if (stage_a) hipLaunchKernelGGL((linear_f16x3_dma_kernel<1, true, true>), g, dim3(256), LDS_DMA, stream, a);
        else hipLaunchKernelGGL((linear_f16x3_dma_kernel<0, true, true>), g, dim3(256), LDS_DMA, stream, a);
    }
    return check_launch("linear_f16x3_dma_kernel");
}

}  // namespace phonlp

using namespace phonlp;

extern "C" size_t phonlp_split_bytes(int rows, int K) {
    if (rows <= 0 || K <= 0) return 0;
    return (size_t)rows * (size_t)((K + 15) / 16 * 16) * 4;
}

namespace phonlp {
// Split image of X [rows, K] written with row pitch `pitch` bytes (0 = packed, 4*K16).
int launch_split_rows(const float *X, long ldx, int rows, int K, void *planes, long pitch, float *inv_scale,
                      hipStream_t stream) {
    if (!X || !planes || !inv_scale || rows <= 0 || K <= 0) return fail(PHONLP_ERR_SHAPE, "split_rows: bad argument");
    const int K16 = (K + 15) / 16 * 16;
    if (pitch == 0) pitch = 4L * K16;
    if ((K & 3) || (ldx & 3) || ((uintptr_t)X & 15) || ((uintptr_t)planes & 15) || (pitch & 15) || pitch < 4L * K16)
        return fail(PHONLP_ERR_SHAPE, "split_rows: K, ldx must be multiples of 4 and X, planes 16-byte aligned");
    ProfScope ps("split_rows", stream);
    const SplitSet a{X, ldx, rows, reinterpret_cast<unsigned char *>(planes), pitch, inv_scale}, none{};
    hipLaunchKernelGGL(split_rows_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, a, none, K, K16);
    return check_launch("split_rows_kernel");
}
// Two matrices of the same K in one launch (packed image of the second).
int launch_split_rows2(const float *X, long ldx, int rows, void *planes, long pitch, float *inv_scale, const float *X2,
                       long ldx2, int rows2, void *planes2, float *inv_scale2, int K, hipStream_t stream) {
    if (!X || !planes || !inv_scale || !X2 || !planes2 || !inv_scale2 || rows <= 0 || rows2 <= 0 || K <= 0)
        return fail(PHONLP_ERR_SHAPE, "split_rows: bad argument");
    const int K16 = (K + 15) / 16 * 16;
    if (pitch == 0) pitch = 4L * K16;
    if ((K & 3) || (ldx & 3) || (ldx2 & 3) || ((uintptr_t)X & 15) || ((uintptr_t)X2 & 15) || ((uintptr_t)planes & 15) ||
        ((uintptr_t)planes2 & 15) || (pitch & 15) || pitch < 4L * K16)
        return fail(PHONLP_ERR_SHAPE, "split_rows: K, ldx must be multiples of 4 and X, planes 16-byte aligned");
    ProfScope ps("split_rows", stream);
    const SplitSet a{X, ldx, rows, reinterpret_cast<unsigned char *>(planes), pitch, inv_scale};
    const SplitSet b{X2, ldx2, rows2, reinterpret_cast<unsigned char *>(planes2), 4L * K16, inv_scale2};
    hipLaunchKernelGGL(split_rows_kernel, dim3(ceil_div(rows, 4) + ceil_div(rows2, 4)), dim3(256), 0, stream, a, b, K, K16);
    return check_launch("split_rows_kernel");
}
}  // namespace phonlp

extern "C" int phonlp_split_rows(const float *X, int ldx, int rows, int K, void *planes, float *inv_scale, void *stream) {
    return launch_split_rows(X, ldx, rows, K, planes, 0, inv_scale, as_stream(stream));
}

extern "C" int phonlp_linear_split(const void *Xp, const float *x_inv, const void *Wp, const float *w_inv,
                                   const float *bias, float *Y, int ldy, int M, int N, int K, int relu,
                                   const int32_t *sentlens, int L, const float *pad_bias, int kind, void *stream) {
    if (K <= 0) return fail(PHONLP_ERR_SHAPE, "linear_split: bad argument");
    const int K16 = (K + 15) / 16 * 16;
    static const char *tags[] = {"linear_split", "linear:label_stage_a", "linear:label_w12", "linear:arc_w12"};
    if (kind < 0 || kind > 3) return fail(PHONLP_ERR_SHAPE, "linear_split: kind must be 0..3");
    return launch_linear_f16x3(tags[kind], Xp, 4L * K16, x_inv, Wp, 4L * K16, w_inv, bias, Y, ldy, M, N, K16, relu, sentlens,
                               L, as_stream(stream), pad_bias);
}

// the same layer with its result written as the split image the next engine-2 product reads (LinearArgs::img)
extern "C" int phonlp_linear_split_image(const void *Xp, const float *x_inv, const void *Wp, const float *w_inv,
                                         const float *bias, void *Yp, float *y_inv, int M, int N, int K, int relu,
                                         const int32_t *sentlens, int L, const float *pad_bias, float bound_c,
                                         float bound_b, int kind, void *stream) {
    if (K <= 0 || !Yp || !y_inv) return fail(PHONLP_ERR_SHAPE, "linear_split_image: bad argument");
    if (N & 15) return fail(PHONLP_ERR_SHAPE, "linear_split_image: N must be a multiple of 16 (the image's k16 stages)");
    const int K16 = (K + 15) / 16 * 16;
    static const char *tags[] = {"linear_split", "linear:label_stage_a", "linear:label_w12", "linear:arc_w12"};
    if (kind < 0 || kind > 3) return fail(PHONLP_ERR_SHAPE, "linear_split_image: kind must be 0..3");
    LinearArgs im{};
    im.img = (unsigned char *)Yp; im.img_pitch = 4L * N; im.img_inv = y_inv; im.img_c = bound_c; im.img_b = bound_b;
    return launch_linear_f16x3(tags[kind], Xp, 4L * K16, x_inv, Wp, 4L * K16, w_inv, bias, nullptr, 0, M, N, K16, relu,
                               sentlens, L, as_stream(stream), pad_bias, 0, nullptr, &im);
}

extern "C" size_t phonlp_label_t4_bytes(int B, int L, int H, int O) {
    if (B <= 0 || L <= 0 || H <= 0 || O <= 0) return 0;
    return (size_t)(((long)B * L + 3) / 4 * 4) * (size_t)O * kp_of(H) * sizeof(float);
}

extern "C" int phonlp_label_stage_a_split(const void *X1p, const float *x1_inv, const void *Wbtp, const float *wbt_inv,
                                          const float *ones_col, int B, int L, int H, int O, float *T4, void *stream) {
    if (B <= 0 || L <= 0 || H <= 0 || O <= 0) return fail(PHONLP_ERR_SHAPE, "label_stage_a_split: bad argument");
    const int KP = kp_of(H), K16 = (KP + 15) / 16 * 16;
    // X1's ones column (index H) times row H of the bilinear weight is a per-column constant: given as ones_col it is
    // added in the epilogue and the product runs over the H real features only -- a k16 stage less when H % 16 == 0
    // (H = 400: 25 stages instead of 26)
    const bool drop = ones_col && (H % 16) == 0;
    return launch_linear_f16x3("linear:label_stage_a", X1p, 4L * K16, x1_inv, Wbtp, 4L * K16, wbt_inv, drop ? ones_col : nullptr,
                               T4, 0, B * L, O * KP, drop ? H : K16, 0, nullptr, 0, as_stream(stream), nullptr, 1);
}

/* Labels at the decoded heads with stage A fused in: T never reaches memory. */
extern "C" int phonlp_label_kpc(int H) { return (kp_of(H) + 31) / 32 * 32; }

extern "C" size_t phonlp_label_at_heads_fused_ws_bytes(int B, int L, int H, int O) {
    if (B <= 0 || L <= 0 || H <= 0 || O <= 0) return 0;
    const size_t M = (size_t)B * L;
    return align_up(M * sizeof(int32_t), 256) + align_up((size_t)O * (phonlp_label_kpc(H) / 32) * M * sizeof(float), 256) +
           align_up((size_t)kp_of(H) * M * sizeof(float), 256) + 256;  // hrow, P, x2t, tile counters
}

static int label_at_heads_fused_run(const void *X1p, const float *x1_inv, const void *Wbtcp, const float *wbtc_inv,
                                    const float *XF, const int32_t *heads, const int32_t *sentlens, int B, int L, int M,
                                    const int32_t *rowmap, const int32_t *off, int H, int O, int32_t *rels,
                                    float *best_opt, void *ws, hipStream_t st) {
    const int KP = kp_of(H), K16 = (KP + 15) / 16 * 16, KPC = phonlp_label_kpc(H);
    if (((uintptr_t)wbtc_inv & 15) || ((uintptr_t)ws & 15))  // the epilogue reads the row scales and x2t as float4
        return fail(PHONLP_ERR_SHAPE, "label_at_heads_fused: wbtc_inv and the workspace must be 16-byte aligned");
    int32_t *hrow = (int32_t *)ws;
    float *P = (float *)((char *)ws + align_up((size_t)M * sizeof(int32_t), 256));
    float *x2t = (float *)((char *)P + align_up((size_t)O * (KPC / 32) * M * sizeof(float), 256));
    int *tile_ctr = (int *)((char *)x2t + align_up((size_t)KP * M * sizeof(float), 256));  // zeroed by gather_heads_t_kernel
    if (rowmap) {  // ROOT and padding keep label 0
        if (hipMemsetAsync(rels, 0, (size_t)B * L * sizeof(int32_t), st) != hipSuccess ||
            (best_opt && hipMemsetAsync(best_opt, 0, (size_t)B * L * sizeof(float), st) != hipSuccess))
            return fail(PHONLP_ERR_LAUNCH, "label_at_heads_fused_packed: hipMemsetAsync failed");
    }
    {
        ProfScope ps("gather_heads_t", st);
        const size_t rows_lds = (size_t)32 * (KP / 4 + 1) * sizeof(float4);
        if (rows_lds <= 64 * 1024 && (((uintptr_t)XF) & 15) == 0)
            hipLaunchKernelGGL(gather_heads_rows_kernel, dim3(ceil_div(M, 32)), dim3(256), rows_lds, st, heads, sentlens, XF + KP,
                               2L * KP, M, L, KP, hrow, x2t, rowmap, off, tile_ctr);
        else
            hipLaunchKernelGGL(gather_heads_t_kernel, dim3(ceil_div(M, 32), ceil_div(KP, 32)), dim3(1024), 0, st, heads, sentlens,
                               XF + KP, 2L * KP, M, L, KP, hrow, x2t, rowmap, off, tile_ctr);
        const int rc = check_launch("gather_heads_t_kernel");
        if (rc) return rc;
    }
    // transposed product: rows = the weight image's (relation, head feature) rows, columns = words
    LinearArgs ah{};
    ah.x2 = x2t; ah.ldx2 = M; ah.P = P; ah.kpc = KPC; ah.kpx = KP; ah.nrel = O; ah.tile_ctr = tile_ctr;
    int rc = launch_linear_f16x3("linear:label_stage_a", Wbtcp, 4L * K16, wbtc_inv, X1p, 4L * K16, x1_inv, nullptr, nullptr,
                                 0, O * KPC, M, K16, 0, nullptr, 0, st, nullptr, 0, &ah);
    if (rc) return rc;
    ProfScope ps("label_reduce", st);
    hipLaunchKernelGGL(label_reduce_kernel, dim3(ceil_div(M, 64)), dim3(512), 0, st, P, hrow, M, O, KPC / 32, rels, best_opt,
                       rowmap);
    return check_launch("label_reduce_kernel");
}

extern "C" int phonlp_label_at_heads_fused(const void *X1p, const float *x1_inv, const void *Wbtcp, const float *wbtc_inv,
                                           const float *XF, const int32_t *heads, const int32_t *sentlens, int B, int L,
                                           int H, int O, int32_t *rels, float *best_opt, void *ws, size_t ws_bytes,
                                           void *stream) {
    if (!X1p || !x1_inv || !Wbtcp || !wbtc_inv || !XF || !heads || !sentlens || !rels || B <= 0 || L <= 0 || H <= 0 || O <= 0)
        return fail(PHONLP_ERR_SHAPE, "label_at_heads_fused: bad argument");
    if (!ws || ws_bytes < phonlp_label_at_heads_fused_ws_bytes(B, L, H, O))
        return fail(PHONLP_ERR_WORKSPACE, "label_at_heads_fused: workspace too small");
    return label_at_heads_fused_run(X1p, x1_inv, Wbtcp, wbtc_inv, XF, heads, sentlens, B, L, B * L, nullptr, nullptr, H, O,
                                    rels, best_opt, ws, as_stream(stream));
}

/* The same for a ragged batch whose live words are packed: X1p / XF hold m_live rows (rowmap[p] = b*L + i of packed row p,
 * off[b] = sentence b's first packed row); heads, rels, best_opt stay [B, L].  Workspace: the dense size for (1, m_live). */
extern "C" int phonlp_label_at_heads_fused_packed(const void *X1p, const float *x1_inv, const void *Wbtcp,
                                                  const float *wbtc_inv, const float *XF, const int32_t *heads,
                                                  const int32_t *sentlens, const int32_t *rowmap, const int32_t *off,
                                                  int m_live, int B, int L, int H, int O, int32_t *rels, float *best_opt,
                                                  void *ws, size_t ws_bytes, void *stream) {
    if (!X1p || !x1_inv || !Wbtcp || !wbtc_inv || !XF || !heads || !sentlens || !rowmap || !off || !rels || B <= 0 || L <= 0 ||
        H <= 0 || O <= 0 || m_live <= 0 || (long)m_live > (long)B * L)
        return fail(PHONLP_ERR_SHAPE, "label_at_heads_fused_packed: bad argument");
    if (!ws || ws_bytes < phonlp_label_at_heads_fused_ws_bytes(1, m_live, H, O))
        return fail(PHONLP_ERR_WORKSPACE, "label_at_heads_fused_packed: workspace too small");
    return label_at_heads_fused_run(X1p, x1_inv, Wbtcp, wbtc_inv, XF, heads, sentlens, B, L, m_live, rowmap, off, H, O, rels,
                                    best_opt, ws, as_stream(stream));
}
