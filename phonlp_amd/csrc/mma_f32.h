// fp32-in / fp32-accumulate MFMA tile engine for gfx950 (v_mfma_f32_32x32x2_f32).
//
// All GEMM-shaped kernels of the library (dense layers, bilinear stage A, arc stage B, label
// stage B) are "NT" products  C[m][n] = sum_k A[m][k] * B[n][k]  with both operands
// K-contiguous in global memory.  A block stages BK = 32 deep slabs of A and B through LDS
// (register staging: global_load_dwordx4 -> ds_write_b128) and every wave multiplies its
// (TM*32) x (TN*32) sub-tile with 32x32x2 f32 MFMAs.
//
// LDS slab image: [rows][PITCH] floats, PITCH = 36 (144-byte rows).  A lane reads 4 consecutive
// k values of its row with one ds_read_b128 and uses them for 4 MFMA k-steps; lanes 0-31 take
// k-chunk 0 and lanes 32-63 k-chunk 1 of an 8-deep group, so MFMA step t multiplies the k pair
// {kb + t, kb + 4 + t}: a permutation of the summation order that A and B share.  The 144-byte
// pitch makes the ds_read_b128 lane groups hit 16 distinct 16-byte slots (9 is odd mod 16).
#pragma once
#include "common.h"

namespace phonlp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
constexpr int PITCH = BK + 4;

template <int ROWS, int THREADS>
struct Slab {
    static_assert((ROWS * 8) % THREADS == 0, "slab rows must divide evenly over the block");
    static constexpr int N = ROWS * 8 / THREADS;
    float4 v[N];
};

// Loads a ROWS x BK slab (rows `0..rows_valid-1` of `base`, columns k0..k0+31, zero beyond K or
// beyond rows_valid) into registers.  `base` points at the tile's first row; ld in floats.
template <int ROWS, int THREADS>
__device__ __forceinline__ void slab_load(Slab<ROWS, THREADS> &s, const float *base, long ld, int rows_valid,
                                          int k0, int K, int tid) {
#pragma unroll
    for (int q = 0; q < Slab<ROWS, THREADS>::N; ++q) {
        const int idx = tid + q * THREADS;
        const int row = idx >> 3;
        const int c4 = (idx & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows_valid && k0 + c4 < K) v = *reinterpret_cast<const float4 *>(base + (long)row * ld + k0 + c4);
        s.v[q] = v;
    }
}

template <int ROWS, int THREADS>
__device__ __forceinline__ void slab_store(const Slab<ROWS, THREADS> &s, float *lds, int tid) {
#pragma unroll
    for (int q = 0; q < Slab<ROWS, THREADS>::N; ++q) {
        const int idx = tid + q * THREADS;
        const int row = idx >> 3;
        const int c4 = (idx & 7) * 4;
        *reinterpret_cast<float4 *>(lds + row * PITCH + c4) = s.v[q];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Engine 0: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain, 64 FLOP/clk/SIMD).
// One slab of MFMAs for this wave's sub-tile.  As/Bs point at the wave's first row of the A / B slab
// images.  acc[i][j] register r, lane l holds
//   C[row = i*32 + (r&3) + 8*(r>>2) + 4*(l>>5)][col = j*32 + (l&31)].
template <int TM, int TN, int DEPTH>
__device__ __forceinline__ void mma_slab_f32(const float *As, const float *Bs, int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31;
    const int h = lane >> 5;
#pragma unroll
    for (int kb = 0; kb < DEPTH; kb += 8) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4 *>(As + (i * 32 + r) * PITCH + kb + 4 * h);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4 *>(Bs + (j * 32 + r) * PITCH + kb + 4 * h);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][t], b[j][t], acc[i][j], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Engine 1: fp32-equivalent products on the bf16 matrix pipe ("bf16x6").  Every fp32 operand x is split IN
// REGISTERS, exactly, into three bf16 pieces by truncation: hi = top 16 bits of x, mid = top 16 bits of
// x - hi, lo = x - hi - mid (8 + 8 + 8 significand bits, so hi + mid + lo == x bit for bit).  The product
// a*b is then the sum of six v_mfma_f32_32x32x16_bf16 terms (hh, hm, mh, hl, lh, mm; the dropped ml, lm, ll
// terms are <= 3 * 2^-24 |a||b|, i.e. at the level of one fp32 rounding), accumulated in fp32 smallest first.
// Six bf16 MFMAs of 32 cycles replace sixteen f32 MFMAs of 64 cycles for the same 32x32x16 block: 5.3x fewer
// matrix-pipe cycles; the ~5.5 VALU ops per operand element of the split then set the pace (~2x overall).
// Memory, LDS image and accumulator layout are those of engine 0 (C/D layout is dtype-independent on gfx950),
// so staging and epilogues are shared.  Inputs must be finite (inf - inf in the split would give NaN).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Split3 {
    u32x4 h, m, l;  // 8 bf16 each, element j in half (j&1) of dword j>>1
};

__device__ __forceinline__ void split3(const f32x4 &x0, const f32x4 &x1, Split3 &s) {
    unsigned hb[8], mb[8], lb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = e < 4 ? x0[e] : x1[e - 4];
        const unsigned u = __float_as_uint(x);
        const float r1 = x - __uint_as_float(u & 0xFFFF0000u);  // exact
        const unsigned u1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);  // exact, <= 8 significant bits
        hb[e] = u;
        mb[e] = u1;
        lb[e] = __float_as_uint(r2);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {  // v_perm_b32: upper halves of two words -> one packed bf16 pair
        s.h[p] = __builtin_amdgcn_perm(hb[2 * p + 1], hb[2 * p], 0x07060302u);
        s.m[p] = __builtin_amdgcn_perm(mb[2 * p + 1], mb[2 * p], 0x07060302u);
        s.l[p] = __builtin_amdgcn_perm(lb[2 * p + 1], lb[2 * p], 0x07060302u);
    }
}

// The same split for the four consecutive-k floats a thread stages: 4 bf16 (two dwords) per piece.
__device__ __forceinline__ void split3_f4(const float4 &v, uint2 &h, uint2 &m, uint2 &l) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned hb[4], mb[4], lb[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned u = __float_as_uint(x[e]);
        const float r1 = x[e] - __uint_as_float(u & 0xFFFF0000u);
        const unsigned u1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
        hb[e] = u;
        mb[e] = u1;
        lb[e] = __float_as_uint(r2);
    }
    h.x = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);
    h.y = __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u);
    m.x = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
    m.y = __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u);
    l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
    l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
}

#define PHONLP_BF16_MFMA(A, B, C) \
    __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), C, 0, 0, 0)

template <int TM, int TN, int DEPTH>
__device__ __forceinline__ void mma_slab_x6(const float *As, const float *Bs, int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31;
    const int h = lane >> 5;
#pragma unroll
    for (int kb = 0; kb < DEPTH; kb += 16) {
        Split3 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float *p = As + (i * 32 + r) * PITCH + kb + 8 * h;
            split3(*reinterpret_cast<const f32x4 *>(p), *reinterpret_cast<const f32x4 *>(p + 4), a[i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float *p = Bs + (j * 32 + r) * PITCH + kb + 8 * h;
            split3(*reinterpret_cast<const f32x4 *>(p), *reinterpret_cast<const f32x4 *>(p + 4), b[j]);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x16 c = acc[i][j];
                c = PHONLP_BF16_MFMA(a[i].l, b[j].h, c);
                c = PHONLP_BF16_MFMA(a[i].h, b[j].l, c);
                c = PHONLP_BF16_MFMA(a[i].m, b[j].m, c);
                c = PHONLP_BF16_MFMA(a[i].m, b[j].h, c);
                c = PHONLP_BF16_MFMA(a[i].h, b[j].m, c);
                c = PHONLP_BF16_MFMA(a[i].h, b[j].h, c);
                acc[i][j] = c;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
template <int TM, int TN, int DEPTH, int ENG>
__device__ __forceinline__ void mma_slab_d(const float *As, const float *Bs, int lane, f32x16 (&acc)[TM][TN]) {
    if constexpr (ENG == 0) mma_slab_f32<TM, TN, DEPTH>(As, Bs, lane, acc);
    else mma_slab_x6<TM, TN, (DEPTH + 15) / 16 * 16>(As, Bs, lane, acc);
}

// `depth` = number of valid k columns in this slab rounded up to 8 (wave-uniform).  The last slab of a K
// that is not a multiple of 32 (868, 408) skips the k-groups that hold only zero fill (the slab images are
// zero-filled beyond K, so rounding the depth up is always safe).
template <int TM, int TN, int ENG>
__device__ __forceinline__ void mma_slab(const float *As, const float *Bs, int lane, f32x16 (&acc)[TM][TN],
                                         int depth = BK) {
    if constexpr (ENG == 0) {
        if (depth >= 32) mma_slab_f32<TM, TN, 32>(As, Bs, lane, acc);
        else if (depth > 16) mma_slab_f32<TM, TN, 24>(As, Bs, lane, acc);
        else if (depth > 8) mma_slab_f32<TM, TN, 16>(As, Bs, lane, acc);
        else mma_slab_f32<TM, TN, 8>(As, Bs, lane, acc);
    } else {
        if (depth > 16) mma_slab_x6<TM, TN, 32>(As, Bs, lane, acc);
        else mma_slab_x6<TM, TN, 16>(As, Bs, lane, acc);
    }
}

// valid depth (multiple of 8) of the slab starting at k0
__device__ __forceinline__ int slab_depth(int k0, int K) {
    const int d = K - k0;
    return d >= BK ? BK : ((d + 7) & ~7);
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// blockIdx -> output tile.  Blocks b and b+8 share an XCD (and its 4 MiB L2), so each XCD is
// given a contiguous run of the tile sequence (bijective remap), and the sequence itself walks
// 8-tile-tall column-major groups so that 64 consecutive tiles form an 8 x 8 patch sharing
// 8 A panels and 8 B panels.
__device__ __forceinline__ void tile_coords(int bid, int nblocks, int num_m, int num_n, int &tm, int &tn) {
    const int xcd = bid & 7;
    const int q = nblocks >> 3, r = nblocks & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GM = 8;
    const int per_group = GM * num_n;
    const int g = t / per_group;
    const int first_m = g * GM;
    const int gsz = (num_m - first_m) < GM ? (num_m - first_m) : GM;
    const int in = t - g * per_group;
    tm = first_m + in % gsz;
    tn = in / gsz;
}

}  // namespace phonlp
