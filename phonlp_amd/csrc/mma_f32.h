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

// One slab of MFMAs for this wave's sub-tile.  As/Bs point at the wave's first row of the A / B slab
// images.  acc[i][j] register r, lane l holds
//   C[row = i*32 + (r&3) + 8*(r>>2) + 4*(l>>5)][col = j*32 + (l&31)].
template <int TM, int TN, int DEPTH>
__device__ __forceinline__ void mma_slab_d(const float *As, const float *Bs, int lane, f32x16 (&acc)[TM][TN]) {
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

// `depth` = number of valid k columns in this slab rounded up to 8 (wave-uniform).  The last slab of a K
// that is not a multiple of 32 (868, 408) skips the k-groups that hold only zero fill.
template <int TM, int TN>
__device__ __forceinline__ void mma_slab(const float *As, const float *Bs, int lane, f32x16 (&acc)[TM][TN],
                                         int depth = BK) {
    if (depth >= 32) mma_slab_d<TM, TN, 32>(As, Bs, lane, acc);
    else if (depth > 16) mma_slab_d<TM, TN, 24>(As, Bs, lane, acc);
    else if (depth > 8) mma_slab_d<TM, TN, 16>(As, Bs, lane, acc);
    else mma_slab_d<TM, TN, 8>(As, Bs, lane, acc);
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
