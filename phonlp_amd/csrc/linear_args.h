// Arguments and the shared epilogue of the dense-layer kernels (linear.hip, linear_f16x3.hip).
#pragma once
#include <limits.h>

#include "mma_f32.h"

namespace phonlp {

struct LinearArgs {
    const float *X;
    const float *W;
    const float *bias;
    const float *pad_bias;  // NULL: padded rows are written as zeros; else as act(pad_bias) (see launch_linear)
    float *Y;
    const int32_t *sentlens;
    long ldx, ldw, ldy;
    int M, N, K, L;
    int relu;   // 0: none; 1: ReLU on every column; -n (internal callers): ReLU on the first n columns only
    int num_m, num_n;
    // engine 2 only (linear_f16x3.hip): per-row / per-column powers of two that undo the operand scaling
    const float *row_scale, *col_scale;
    int rows4;  // linear_f16x3.hip: 1 = write Y in the "rows interleaved by 4" layout straight from the accumulators;
                // 2 = labels-at-heads epilogue (no Y; rows = (relation, head feature) weight rows, columns = words):
                //     P[row/32][col] = sum over the tile's 32 rows of C[row][col] * x2[((d2/4) * ldx2 + col) * 4 + d2%4],
                //     d2 = 32 * (row / 32 / nrel) + row % 32
    const float *x2;       // mode 2: head-side features of every word's head, transposed by fours: [kpx/4][ldx2][4]
    float *P;              // mode 2: [M/32][N]
    long ldx2;
    int kpc, kpx, nrel;    // mode 2: rows per relation (multiple of 32), how many of them are real, relations; weight row
                           //         (blk * nrel + o) * 32 + d2 % 32 holds relation o, head feature d2 = 32 blk + d2 % 32
    long bsx, bsw, bsy;  // batched launch (gridDim.y > 1): float strides of X, W, Y between the items
    int gm;   // linear_f16x3.hip: row tiles per group of the tile walk (0 = 4)
    int *tile_ctr;  // mode 2, persistent wide kernel: eight tile counters (one per XCD chunk of the walk), zero at launch
    int dbg;  // -DPHONLP_ABLATION builds only (PHONLP_X3_ABLATION): 1 = epilogue without its global stores, 2 = no epilogue
    // linear_f16x3.hip, image output (img != NULL, Y unused): the layer's result is written as the SPLIT IMAGE the next
    // engine-2 product reads (row pitch img_pitch bytes, N % 8 == 0 columns: h[8] l[8] per group of eight), scaled per
    // row by a power of two chosen BEFORE the product from a bound on the row's entries,
    //     |y[m][n]| <= img_c * max_k|x[m][k]| + img_b,  img_c >= sqrt(K) max_n ||W[n]||_2  (Cauchy-Schwarz),  img_b >= max|bias|,
    // with max_k|x[m][k]| < 2^15 row_scale[m] read off the input image's own scale.  img_inv[m] = the inverse scale.  A
    // bound instead of the row's true maximum costs nothing in accuracy until it is ~2^10 too large (f16 pieces keep 22
    // bits of every element within 2^-(17 - log2 looseness) of the row's maximum; smaller ones are off by < 2^-39 of
    // the bound) and saves the separate split pass over the result (8 bytes per element read + written).
    unsigned char *img;
    long img_pitch;
    float *img_inv;
    float img_c, img_b;
};

typedef _Float16 img_f16x4 __attribute__((ext_vector_type(4)));

// scale (a power of two) that maps |y| <= U into f16's range with the bound just below 2^15, and its inverse
__device__ __forceinline__ void image_scale_of_bound(float U, float &scale, float &inv) {
    int e = (int)((__float_as_uint(U) >> 23) & 0xFF);  // biased exponent: U < 2^(e - 126)
    e = e < 16 ? 16 : (e > 254 ? 254 : e);
    scale = __uint_as_float((unsigned)(268 - e) << 23);
    inv = __uint_as_float((unsigned)(e - 14) << 23);
}

// Epilogue shared by the dense-layer kernels: the wave's accumulators are transposed through a per-wave LDS
// patch (32 rows x (TN*32 + 4) floats at lds + wave * 32 * EP; the caller guarantees WM*WN such patches fit and
// that no wave still reads the slab images they alias) so that every global store is a 16-byte-per-lane row
// segment; bias / ReLU / padding mask fused.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void linear_epilogue(f32x16 (&acc)[TM][TN], float *lds, const LinearArgs &a, int m0, int n0,
                                                int wm, int wn, int lane, int wave) {
            constexpr int WCOLS = TN * 32, EP = WCOLS + 4;
            float *ep = lds + wave * 32 * EP;
            constexpr int LPR = WCOLS / 4;   // lanes per row
            constexpr int RPI = 64 / LPR;    // rows per wave-instruction
            const int c4 = (lane % LPR) * 4;
            const int col = n0 + wn * WCOLS + c4;
            const int rsub = lane / LPR;
            const bool vec_ok = ((a.ldy & 3) == 0) && (((uintptr_t)a.Y & 15) == 0) && (col + 3 < a.N);
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a.bias) {
                if (col < a.N) bv.x = a.bias[col];
                if (col + 1 < a.N) bv.y = a.bias[col + 1];
                if (col + 2 < a.N) bv.z = a.bias[col + 2];
                if (col + 3 < a.N) bv.w = a.bias[col + 3];
            }
            float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
            if (a.col_scale) {
                if (col < a.N) cs.x = a.col_scale[col];
                if (col + 1 < a.N) cs.y = a.col_scale[col + 1];
                if (col + 2 < a.N) cs.z = a.col_scale[col + 2];
                if (col + 3 < a.N) cs.w = a.col_scale[col + 3];
            }
            // ReLU as a per-column lower clamp: 0 where it applies, -inf elsewhere
            const int relu_n = a.relu > 0 ? INT_MAX : -a.relu;
            float4 lo;
            lo.x = col < relu_n ? 0.f : -INFINITY;
            lo.y = col + 1 < relu_n ? 0.f : -INFINITY;
            lo.z = col + 2 < relu_n ? 0.f : -INFINITY;
            lo.w = col + 3 < relu_n ? 0.f : -INFINITY;
            float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);  // value of a padded row
            if (a.pad_bias) {
                if (col < a.N) pv.x = a.pad_bias[col];
                if (col + 1 < a.N) pv.y = a.pad_bias[col + 1];
                if (col + 2 < a.N) pv.z = a.pad_bias[col + 2];
                if (col + 3 < a.N) pv.w = a.pad_bias[col + 3];
                if (a.relu) { pv.x = fmaxf(pv.x, lo.x); pv.y = fmaxf(pv.y, lo.y); pv.z = fmaxf(pv.z, lo.z); pv.w = fmaxf(pv.w, lo.w); }
            }
            unsigned live[TM];
    #pragma unroll
            for (int i = 0; i < TM; ++i) {
                live[i] = 0xFFFFFFFFu;
                if (a.sentlens) {
                    live[i] = 0;
    #pragma unroll
                    for (int it = 0; it < 32 / RPI; ++it) {
                        const int row = m0 + (wm * TM + i) * 32 + it * RPI + rsub;
                        if (row < a.M && (row % a.L) < a.sentlens[row / a.L]) live[i] |= 1u << it;
                    }
                }
            }
    #pragma unroll
            for (int i = 0; i < TM; ++i) {
    #pragma unroll
                for (int j = 0; j < TN; ++j)
    #pragma unroll
                    for (int r = 0; r < 16; ++r) ep[acc_row(r, lane) * EP + j * 32 + (lane & 31)] = acc[i][j][r];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    #pragma unroll
                for (int it = 0; it < 32 / RPI; ++it) {
                    const int lrow = it * RPI + rsub;
                    const int row = m0 + (wm * TM + i) * 32 + lrow;
                    float4 v = *reinterpret_cast<const float4 *>(ep + lrow * EP + c4);
                    if (a.row_scale) {  // (v * 2^-ka) * 2^-kb: exact
                        const float rs = a.row_scale[row < a.M ? row : a.M - 1];
                        v.x = v.x * rs * cs.x; v.y = v.y * rs * cs.y; v.z = v.z * rs * cs.z; v.w = v.w * rs * cs.w;
                    }
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (a.relu) { v.x = fmaxf(v.x, lo.x); v.y = fmaxf(v.y, lo.y); v.z = fmaxf(v.z, lo.z); v.w = fmaxf(v.w, lo.w); }
                    if (!((live[i] >> it) & 1u)) v = pv;
                    if (row >= a.M || col >= a.N) continue;
#ifdef PHONLP_ABLATION
                    if (a.dbg == 1 && v.x != 1234.5f) continue;
#endif
                    if (a.img) {  // the result as a split image (a.N % 8 == 0: the four columns are inside the row)
                        float sc, inv;
                        image_scale_of_bound(a.img_c * a.row_scale[row] * 32768.f + a.img_b, sc, inv);
                        const float s4[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};  // exact (powers of two)
                        img_f16x4 h, l;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            h[e] = (_Float16)s4[e];
                            l[e] = (_Float16)(s4[e] - (float)h[e]);
                        }
                        unsigned char *o = a.img + (long)row * a.img_pitch + (col >> 3) * 32 + (col & 7) * 2;
                        *reinterpret_cast<img_f16x4 *>(o) = h;
                        *reinterpret_cast<img_f16x4 *>(o + 16) = l;
                        if (col == 0) a.img_inv[row] = inv;
                        continue;
                    }
                    float *y = a.Y + (long)row * a.ldy + col;
                    if (vec_ok) {
                        *reinterpret_cast<float4 *>(y) = v;
                    } else {
                        y[0] = v.x;
                        if (col + 1 < a.N) y[1] = v.y;
                        if (col + 2 < a.N) y[2] = v.z;
                        if (col + 3 < a.N) y[3] = v.w;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
}

}  // namespace phonlp
