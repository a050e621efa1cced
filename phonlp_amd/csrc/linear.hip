// Dense layer  Y = act(X . W^T + bias)  on the fp32 MFMA tile engine, plus the one-time weight
// packing kernels.  See include/phonlp_heads.h for the reference call sites each entry replaces.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "linear_args.h"

namespace phonlp {

// BM x BN block tile, WM x WN waves, each wave a (BM/WM) x (BN/WN) sub-tile.  TAG only names the
// instantiation (TAG 1 = bilinear stage A of the label scorer, so profilers list it on its own row).
//
// PERSISTENT: the grid is (at most) two blocks per CU and every block walks a strided sequence of output
// tiles.  K is short on this path (408 / 868), so a one-tile-per-block kernel spends ~17 % of its life in
// block dispatch, the first slab's load latency and the epilogue (measured: 146 TFLOP/s for the bare
// ds_read+MFMA loop at K=4096 vs 122 at K=408).  Here the first slab of the NEXT tile is fetched while the
// last slab of the current one is multiplied, and blocks are never re-dispatched.
//
// K loop: two LDS slab buffers, ONE barrier per slab.  Iteration k issues the global loads of slab k+1,
// multiplies slab k out of buffer k&1, then writes slab k+1 into buffer (k+1)&1.  All slabs but the last
// are full depth, so the loop has a single MFMA code path and the 64 accumulators stay pinned (a depth
// switch inside the loop made hipcc shuttle them through v_accvgpr_write/read every slab, -9 %).
// Epilogue: accumulators are transposed through a per-wave LDS patch so that every global store is a
// 16-byte-per-lane row segment; bias / ReLU / padding mask fused.
//
// Tile order: block b belongs to XCD group x = b % 8 (blocks b, b+8, ... share an L2), local index
// j = b / 8; in round i it takes tile (i*8 + x)*Gx + j of a sequence that walks 8-tile-tall column-major
// groups, so the Gx blocks of one XCD work on one compact patch of tiles sharing A and B panels.
template <int BM, int BN, int WM, int WN, int TAG, int ENG>
__global__ __launch_bounds__(WM *WN * 64, 2) void linear_kernel(LinearArgs a) {
    if (gridDim.y > 1) {  // batched launch: item blockIdx.y
        a.X += blockIdx.y * a.bsx;
        a.W += blockIdx.y * a.bsw;
        a.Y += blockIdx.y * a.bsy;
    }
    constexpr int THREADS = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int BUF = (BM + BN) * PITCH;
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    static_assert(WM * WN * 32 * (TN * 32 + 4) <= 2 * BUF, "epilogue patch must fit the slab buffers");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = a.num_m * a.num_n;
    const int gx = gridDim.x >> 3;  // blocks per XCD group (gridDim.x is a multiple of 8, or == ntiles < 8)
    const int xg = blockIdx.x & 7, jl = blockIdx.x >> 3;
    const int nslab = (a.K + BK - 1) / BK;

    // tile sequence index of this block in `round`.  With gridDim.x == ntiles (the default: blocks of other
    // kernels on other streams can then interleave, which a statically partitioned persistent grid cannot
    // tolerate) there is one round and the map is the bijective "contiguous chunk per XCD group" remap.
    auto tile_of = [&](int round, int &m0, int &n0) -> bool {
        long t;
        if ((int)gridDim.x == ntiles) {
            if (round > 0) return false;
            const int q = ntiles >> 3, r = ntiles & 7;
            t = (xg < r ? xg * (q + 1) : r * (q + 1) + (xg - r) * q) + jl;
        } else {
            t = ((long)round * 8 + xg) * gx + jl;
        }
        if (t >= ntiles) return false;
        constexpr int GM = 8;
        const int per_group = GM * a.num_n;
        const int g = (int)(t / per_group);
        const int first_m = g * GM;
        const int gsz = (a.num_m - first_m) < GM ? (a.num_m - first_m) : GM;
        const int in = (int)(t - (long)g * per_group);
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * BN;
        return true;
    };

    f32x16 acc[TM][TN];
    Slab<BM, THREADS> ra;
    Slab<BN, THREADS> rb;
    int m0 = 0, n0 = 0, round = 0;
    bool have = tile_of(0, m0, n0);
    if (have) {
        slab_load(ra, a.X + (long)m0 * a.ldx, a.ldx, a.M - m0, 0, a.K, tid);
        slab_load(rb, a.W + (long)n0 * a.ldw, a.ldw, a.N - n0, 0, a.K, tid);
    }
    while (have) {
        const float *Xt = a.X + (long)m0 * a.ldx;
        const float *Wt = a.W + (long)n0 * a.ldw;
        const int mv = a.M - m0, nv = a.N - n0;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        slab_store(ra, lds, tid);
        slab_store(rb, lds + BM * PITCH, tid);
        __syncthreads();
        int cur = 0;
        for (int sidx = 0; sidx < nslab - 1; ++sidx) {
            const int k1 = (sidx + 1) * BK;
            slab_load(ra, Xt, a.ldx, mv, k1, a.K, tid);
            slab_load(rb, Wt, a.ldw, nv, k1, a.K, tid);
            const float *As = lds + cur * BUF, *Bs = As + BM * PITCH;
            mma_slab_d<TM, TN, 32, ENG>(As + wm * TM * 32 * PITCH, Bs + wn * TN * 32 * PITCH, lane, acc);
            float *An = lds + (cur ^ 1) * BUF;
            slab_store(ra, An, tid);
            slab_store(rb, An + BM * PITCH, tid);
            __syncthreads();
            cur ^= 1;
        }
        // last slab of this tile; meanwhile fetch the first slab of the next one
        int m1 = 0, n1 = 0;
        const bool have1 = tile_of(round + 1, m1, n1);
        if (have1) {
            slab_load(ra, a.X + (long)m1 * a.ldx, a.ldx, a.M - m1, 0, a.K, tid);
            slab_load(rb, a.W + (long)n1 * a.ldw, a.ldw, a.N - n1, 0, a.K, tid);
        }
        {
            const float *As = lds + cur * BUF, *Bs = As + BM * PITCH;
            mma_slab<TM, TN, ENG>(As + wm * TM * 32 * PITCH, Bs + wn * TN * 32 * PITCH, lane, acc,
                             slab_depth((nslab - 1) * BK, a.K));
            lds_barrier();  // not __syncthreads(): the next tile's prefetch must stay in flight
        }

        linear_epilogue<TM, TN, WM, WN>(acc, lds, a, m0, n0, wm, wn, lane, wave);
        lds_barrier();  // the patches alias the slab buffers the next tile is about to fill; the tile's global
                        // stores are NOT waited for (a full barrier would expose their write latency per tile)
        have = have1;
        m0 = m1;
        n0 = n1;
        ++round;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Engine 1 dense layer: fp32-equivalent product on the bf16 matrix pipe, operands split ONCE PER BLOCK while
// they are staged.  The fp32 slab a thread fetched (global_load_dwordx4, as in engine 0) is split exactly into
// three bf16 pieces (mma_f32.h: hi / mid / lo by truncation) and written to three bf16 LDS planes per operand;
// the MFMA loop is then nothing but ds_read_b128 + v_mfma_f32_32x32x16_bf16 (12 reads feed 24 MFMAs per k16
// step of a 64x64 wave tile -- twice the MFMA work per LDS byte of a plain bf16 GEMM).  Splitting at staging
// costs ~5 VALU ops per element once per 128-wide tile, instead of once per wave per fragment use.
//
// LDS image (one buffer, 48 KB, so two or three blocks share a CU and cover each other's barriers):
//   plane p of A at p*BM*64 bytes, of B at (3*BM + p*BN)*64; a plane row is the slab's 32 bf16 (64 bytes) as
//   four 16-byte chunks, chunk c of row R stored at chunk slot c ^ ((R>>2)&3): the 16 lanes of a
//   ds_read_b128 group (16 rows, one chunk index) then cover all 64 banks, and the 16 lanes of a
//   ds_write_b64 group (two whole rows) cover the 32 banks of a store exactly once.
// Loop per slab: split+store slab k | barrier | issue global loads of slab k+1 | MFMAs on slab k | barrier.
template <int TAG>
__global__ __launch_bounds__(256, 2) void linear_split_kernel(LinearArgs a) {
    if (gridDim.y > 1) {  // batched launch: item blockIdx.y
        a.X += blockIdx.y * a.bsx;
        a.W += blockIdx.y * a.bsw;
        a.Y += blockIdx.y * a.bsy;
    }
    constexpr int BM = 128, BN = 128, WM = 2, WN = 2, TM = 2, TN = 2, THREADS = 256;
    constexpr int PLANE_A = BM * 64, PLANE_B = BN * 64;        // bytes
    constexpr int LDS_BYTES = 3 * PLANE_A + 3 * PLANE_B;       // 49152
    static_assert(WM * WN * 32 * (TN * 32 + 4) * 4 <= LDS_BYTES, "epilogue patch must fit the slab image");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = a.num_m * a.num_n;
    const int nslab = (a.K + BK - 1) / BK;
    int m0, n0;
    {   // one tile per block; blocks b, b+8, ... (one XCD) take a contiguous chunk of the 8-tall column-major walk
        const int xg = blockIdx.x & 7, jl = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int t = (xg < r ? xg * (q + 1) : r * (q + 1) + (xg - r) * q) + jl;
        constexpr int GM = 8;
        const int per_group = GM * a.num_n;
        const int g = t / per_group;
        const int first_m = g * GM;
        const int gsz = (a.num_m - first_m) < GM ? (a.num_m - first_m) : GM;
        const int in = t - g * per_group;
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * BN;
    }
    const float *Xt = a.X + (long)m0 * a.ldx;
    const float *Wt = a.W + (long)n0 * a.ldw;
    const int mv = a.M - m0, nv = a.N - n0;

    // staging: thread -> (row = idx>>3, float4 column (idx&7)); its 4 bf16 per plane are half of chunk (idx&7)>>1
    auto stage = [&](const float4 (&v)[4], unsigned char *plane0, int plane_bytes) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + q * THREADS;
            const int row = idx >> 3, c8 = idx & 7;
            uint2 h, m, l;
            split3_f4(v[q], h, m, l);
            unsigned char *p = plane0 + row * 64 + ((((c8 >> 1) ^ (row >> 2)) & 3) << 4) + ((c8 & 1) << 3);
            *reinterpret_cast<uint2 *>(p) = h;
            *reinterpret_cast<uint2 *>(p + plane_bytes) = m;
            *reinterpret_cast<uint2 *>(p + 2 * plane_bytes) = l;
        }
    };
    // fragment reads: lane (r = lane&31, h = lane>>5) takes k = kb + 8h .. +7 of row r: chunk (kb>>3) + h
    const int r = lane & 31, hh = lane >> 5;
    const int sw = (r >> 2) & 3;
    const unsigned char *Ab = lds + (wm * 64 + r) * 64;
    const unsigned char *Bb = lds + 3 * PLANE_A + (wn * 64 + r) * 64;
    const int off0 = ((hh ^ sw) & 3) << 4;  // k16 step 0; step 1 is chunk (2 + h): off0 ^ 32

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto k16 = [&](int off) {
        u32x4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const u32x4 *>(Bb + j * 32 * 64 + off);
#pragma unroll
        for (int i = 0; i < TM; ++i) al[i] = *reinterpret_cast<const u32x4 *>(Ab + 2 * PLANE_A + i * 32 * 64 + off);
#pragma unroll
        for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const u32x4 *>(Ab + i * 32 * 64 + off);
#pragma unroll
        for (int j = 0; j < TN; ++j) bl[j] = *reinterpret_cast<const u32x4 *>(Bb + 2 * PLANE_B + j * 32 * 64 + off);
#pragma unroll
        for (int i = 0; i < TM; ++i) am[i] = *reinterpret_cast<const u32x4 *>(Ab + PLANE_A + i * 32 * 64 + off);
#pragma unroll
        for (int j = 0; j < TN; ++j) bm[j] = *reinterpret_cast<const u32x4 *>(Bb + PLANE_B + j * 32 * 64 + off);
#define PHONLP_X6_ROUND(AA, BB)                                               \
    _Pragma("unroll") for (int i = 0; i < TM; ++i)                            \
    _Pragma("unroll") for (int j = 0; j < TN; ++j) acc[i][j] = PHONLP_BF16_MFMA(AA[i], BB[j], acc[i][j]);
        PHONLP_X6_ROUND(al, bh)
        PHONLP_X6_ROUND(ah, bl)
        PHONLP_X6_ROUND(am, bm)
        PHONLP_X6_ROUND(am, bh)
        PHONLP_X6_ROUND(ah, bm)
        PHONLP_X6_ROUND(ah, bh)
#undef PHONLP_X6_ROUND
    };

    Slab<BM, THREADS> ra;
    Slab<BN, THREADS> rb;
    slab_load(ra, Xt, a.ldx, mv, 0, a.K, tid);
    slab_load(rb, Wt, a.ldw, nv, 0, a.K, tid);
    for (int sidx = 0; sidx < nslab - 1; ++sidx) {
        stage(ra.v, lds, PLANE_A);
        stage(rb.v, lds + 3 * PLANE_A, PLANE_B);
        lds_barrier();
        const int k1 = (sidx + 1) * BK;
        slab_load(ra, Xt, a.ldx, mv, k1, a.K, tid);
        slab_load(rb, Wt, a.ldw, nv, k1, a.K, tid);
        k16(off0);
        k16(off0 ^ 32);
        lds_barrier();
    }
    stage(ra.v, lds, PLANE_A);
    stage(rb.v, lds + 3 * PLANE_A, PLANE_B);
    lds_barrier();
    k16(off0);
    if (a.K - (nslab - 1) * BK > 16) k16(off0 ^ 32);
    lds_barrier();
    linear_epilogue<TM, TN, WM, WN>(acc, reinterpret_cast<float *>(lds), a, m0, n0, wm, wn, lane, wave);
}

// ---------------------------------------------------------------------------------------------------------
// Engine 1, large layers: 256 x 128 block tile, four waves of 128 x 64, k16 stages, two LDS buffers, ONE
// barrier per stage, and the split of stage s+1 interleaved with the MFMAs of stage s inside every wave
// (the matrix pipe runs asynchronously: a wave can issue ~7 VALU/LDS instructions in the shadow of each
// 32-cycle MFMA).  Per stage and wave: 18 ds_read_b128 feed 48 MFMAs, and the thread splits its 24 staged
// floats (~130 VALU ops, 2.7 per MFMA -- the 128 x 128 kernel above needs 5.7).
//
// LDS (dynamic, 2 x 36 KB): buffer b, plane p at (b*3 + p) * 384*32 bytes; rows 0-255 = A tile, 256-383 = B
// tile; a plane row is the stage's 16 bf16 (32 bytes) as two 16-byte chunks, chunk c stored at slot
// c ^ ((row>>3)&1) so that the 16 rows of a ds_read_b128 lane group cover all 64 banks.
// Rows past M / N are not zero-filled: their pointers are clamped to the last valid row and the garbage only
// reaches output rows / columns that the epilogue never stores.  Columns past K (last stage only) are
// clamped to K-4 for the load and zeroed before the split.
template <int TAG, int ABL = 0>
__global__ __launch_bounds__(256, 2) void linear_split256_kernel(LinearArgs a) {
    if (gridDim.y > 1) {  // batched launch: item blockIdx.y
        a.X += blockIdx.y * a.bsx;
        a.W += blockIdx.y * a.bsw;
        a.Y += blockIdx.y * a.bsy;
    }
    constexpr int BM = 256, BN = 128, WM = 2, WN = 2, TM = 4, TN = 2, SK = 16;
    constexpr int ROWS = BM + BN, PLANE = ROWS * 32, BUF = 3 * PLANE;
    static_assert(WM * WN * 32 * (TN * 32 + 4) * 4 <= 2 * BUF, "epilogue patch must fit");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntiles = a.num_m * a.num_n;
    const int nst = (a.K + SK - 1) / SK;
    int m0, n0;
    {
        const int xg = blockIdx.x & 7, jl = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int t = (xg < r ? xg * (q + 1) : r * (q + 1) + (xg - r) * q) + jl;
        constexpr int GM = 4;
        const int per_group = GM * a.num_n;
        const int g = t / per_group;
        const int first_m = g * GM;
        const int gsz = (a.num_m - first_m) < GM ? (a.num_m - first_m) : GM;
        const int in = t - g * per_group;
        m0 = (first_m + in % gsz) * BM;
        n0 = (in / gsz) * BN;
    }
    // staging map: thread -> float4 column c4 = (tid&3)*4 of rows (tid>>2) + 64q; q 0-3 are A rows, 4-5 B rows
    const int r0 = tid >> 2, c4 = (tid & 3) * 4;
    const float *src[6];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row = m0 + r0 + 64 * q;
        row = row < a.M ? row : a.M - 1;
        src[q] = a.X + (long)row * a.ldx;
    }
#pragma unroll
    for (int q = 4; q < 6; ++q) {
        int row = n0 + r0 + 64 * (q - 4);
        row = row < a.N ? row : a.N - 1;
        src[q] = a.W + (long)row * a.ldw;
    }
    const int wofs = r0 * 32 + ((((c4 >> 3) ^ (r0 >> 3)) & 1) << 4) + (((c4 >> 2) & 1) << 3);
    float4 R[6];
    auto load_stage = [&](int st) {
        int col = st * SK + c4;
        col = col < a.K - 4 ? col : a.K - 4;
#pragma unroll
        for (int q = 0; q < 6; ++q) R[q] = *reinterpret_cast<const float4 *>(src[q] + col);
    };
    auto split_stage = [&](unsigned char *buf) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            uint2 h, m, l;
            split3_f4(R[q], h, m, l);
            unsigned char *p = buf + wofs + q * 64 * 32;
            *reinterpret_cast<uint2 *>(p) = h;
            *reinterpret_cast<uint2 *>(p + PLANE) = m;
            *reinterpret_cast<uint2 *>(p + 2 * PLANE) = l;
        }
    };
    auto zero_tail = [&](int st) {  // last stage: the float4 whose columns lie past K
        if (st * SK + c4 >= a.K) {
#pragma unroll
            for (int q = 0; q < 6; ++q) R[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    const int r = lane & 31, hh = lane >> 5;
    const int rofs = r * 32 + (((hh ^ (r >> 3)) & 1) << 4);
    const int a_ofs = (wm * 128) * 32 + rofs;
    const int b_ofs = (BM + wn * 64) * 32 + rofs;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragments of one stage: all 18 are read before the stage's LDS stores are issued (the compiler cannot
    // prove that reads of one buffer and stores to the other do not alias, so source order is issue order)
    struct Frags {
        u32x4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
    };
    auto read_frags = [&](const unsigned char *buf, Frags &f) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f.bh[j] = *reinterpret_cast<const u32x4 *>(buf + b_ofs + j * 1024);
            f.bl[j] = *reinterpret_cast<const u32x4 *>(buf + 2 * PLANE + b_ofs + j * 1024);
            f.bm[j] = *reinterpret_cast<const u32x4 *>(buf + PLANE + b_ofs + j * 1024);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.al[i] = *reinterpret_cast<const u32x4 *>(buf + 2 * PLANE + a_ofs + i * 1024);
            f.ah[i] = *reinterpret_cast<const u32x4 *>(buf + a_ofs + i * 1024);
            f.am[i] = *reinterpret_cast<const u32x4 *>(buf + PLANE + a_ofs + i * 1024);
        }
    };
    auto mma_frags = [&](const Frags &f) {
#pragma unroll
        for (int ip = 0; ip < TM; ip += 2) {
#define PHONLP_X6_ROUND(AA, BB)                                    \
    _Pragma("unroll") for (int i = ip; i < ip + 2; ++i)            \
    _Pragma("unroll") for (int j = 0; j < TN; ++j)                 \
        acc[i][j] = PHONLP_BF16_MFMA(f.AA[i], f.BB[j], acc[i][j]);
            PHONLP_X6_ROUND(al, bh)
            PHONLP_X6_ROUND(ah, bl)
            PHONLP_X6_ROUND(am, bm)
            PHONLP_X6_ROUND(am, bh)
            PHONLP_X6_ROUND(ah, bm)
            PHONLP_X6_ROUND(ah, bh)
#undef PHONLP_X6_ROUND
        }
    };
    // MFMA number n of a stage, in the order mma_frags issues them
    auto mf = [&](auto N, const Frags &f) {
        constexpr int n = decltype(N)::value;
        constexpr int ip = n / 24, rem = n % 24, prod = rem / 4, i = ip * 2 + (rem % 4) / 2, j = rem % 2;
        const u32x4 &A = prod == 0 ? f.al[i] : (prod == 2 || prod == 3) ? f.am[i] : f.ah[i];
        const u32x4 &B = (prod == 0 || prod == 3 || prod == 5) ? f.bh[j] : prod == 1 ? f.bl[j] : f.bm[j];
        acc[i][j] = PHONLP_BF16_MFMA(A, B, acc[i][j]);
    };
    auto mma_stage = [&](const unsigned char *buf) {
        Frags f;
        read_frags(buf, f);
        mma_frags(f);
    };

    // prologue: stage 0 in buffer 0, stage 1 in registers
    load_stage(0);
    if (nst == 1) zero_tail(0);
    split_stage(lds);
    if (nst > 1) load_stage(1);
    lds_barrier();
    int s = 0;
    // >>> steady-state loop
    // Steady state: multiply stage s, split stage s+1 into the other buffer, fetch stage s+2.  The issue order is
    // written out by hand and pinned with sched_barrier: MFMAs execute asynchronously, so the 4-6 VALU / LDS /
    // VMEM instructions placed after each one run in its 32-cycle shadow (a run of back-to-back MFMAs would
    // leave the wave's other work for after the last one).  ABL != 0 are timing-only ablations of this loop
    // (tools/engine_probe.py): 1 = MFMAs only, 2 = + fragment reads and the barrier, 3 = + split and LDS stores.
#define SB() __builtin_amdgcn_sched_barrier(0)
#define MF(n) mf(std::integral_constant<int, n>{}, f)
    Frags f;
    if constexpr (ABL == 1) read_frags(lds, f);
    for (; s + 2 < nst; ++s) {
        unsigned char *cur = lds + (s & 1) * BUF, *nxt = lds + ((s + 1) & 1) * BUF;
        if constexpr (ABL != 1) read_frags(cur, f);
        SB();
        constexpr bool SPLIT = ABL == 0 || ABL == 3;
        float r1[4], r2[4];
        unsigned t[4];
        uint2 ph, pm, pl;
        // ---- split of R[0]
        {
            float x[4] = {R[0].x, R[0].y, R[0].z, R[0].w};
            if constexpr (SPLIT) {
                if constexpr (ABL == 3) x[0] += (float)s;  // keeps the split inside the loop
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(x[e]) & 0xFFFF0000u;
                r1[0] = x[0] - __uint_as_float(t[0]);
                r1[1] = x[1] - __uint_as_float(t[1]);
            }
            if constexpr (SPLIT) {
                r1[2] = x[2] - __uint_as_float(t[2]);
                r1[3] = x[3] - __uint_as_float(t[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(r1[e]) & 0xFFFF0000u;
            }
            if constexpr (SPLIT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] = r1[e] - __uint_as_float(t[e]);
                ph.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                ph.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
            }
            MF(0);
            SB();
            if constexpr (SPLIT) {
                pm.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                pm.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                pl.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                pl.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            }
            MF(1);
            SB();
            if constexpr (SPLIT) {
                unsigned char *p = nxt + wofs + 0 * 64 * 32;
                *reinterpret_cast<uint2 *>(p) = ph;
                *reinterpret_cast<uint2 *>(p + PLANE) = pm;
                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = pl;
            }
            MF(2);
            SB();
        }
        // ---- split of R[1]
        {
            float x[4] = {R[1].x, R[1].y, R[1].z, R[1].w};
            if constexpr (SPLIT) {
                if constexpr (ABL == 3) x[0] += (float)s;  // keeps the split inside the loop
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(x[e]) & 0xFFFF0000u;
                r1[0] = x[0] - __uint_as_float(t[0]);
                r1[1] = x[1] - __uint_as_float(t[1]);
            }
            MF(3);
            SB();
            if constexpr (SPLIT) {
                r1[2] = x[2] - __uint_as_float(t[2]);
                r1[3] = x[3] - __uint_as_float(t[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(r1[e]) & 0xFFFF0000u;
            }
            MF(4);
            SB();
            if constexpr (SPLIT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] = r1[e] - __uint_as_float(t[e]);
                ph.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                ph.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
            }
            MF(5);
            SB();
            if constexpr (SPLIT) {
                pm.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                pm.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                pl.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                pl.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            }
            MF(6);
            SB();
            if constexpr (SPLIT) {
                unsigned char *p = nxt + wofs + 1 * 64 * 32;
                *reinterpret_cast<uint2 *>(p) = ph;
                *reinterpret_cast<uint2 *>(p + PLANE) = pm;
                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = pl;
            }
            MF(7);
            SB();
        }
        // ---- split of R[2]
        {
            float x[4] = {R[2].x, R[2].y, R[2].z, R[2].w};
            if constexpr (SPLIT) {
                if constexpr (ABL == 3) x[0] += (float)s;  // keeps the split inside the loop
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(x[e]) & 0xFFFF0000u;
                r1[0] = x[0] - __uint_as_float(t[0]);
                r1[1] = x[1] - __uint_as_float(t[1]);
            }
            MF(8);
            SB();
            if constexpr (SPLIT) {
                r1[2] = x[2] - __uint_as_float(t[2]);
                r1[3] = x[3] - __uint_as_float(t[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(r1[e]) & 0xFFFF0000u;
            }
            MF(9);
            SB();
            if constexpr (SPLIT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] = r1[e] - __uint_as_float(t[e]);
                ph.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                ph.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
            }
            MF(10);
            SB();
            if constexpr (SPLIT) {
                pm.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                pm.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                pl.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                pl.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            }
            MF(11);
            SB();
            if constexpr (SPLIT) {
                unsigned char *p = nxt + wofs + 2 * 64 * 32;
                *reinterpret_cast<uint2 *>(p) = ph;
                *reinterpret_cast<uint2 *>(p + PLANE) = pm;
                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = pl;
            }
            MF(12);
            SB();
        }
        // ---- split of R[3]
        {
            float x[4] = {R[3].x, R[3].y, R[3].z, R[3].w};
            if constexpr (SPLIT) {
                if constexpr (ABL == 3) x[0] += (float)s;  // keeps the split inside the loop
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(x[e]) & 0xFFFF0000u;
                r1[0] = x[0] - __uint_as_float(t[0]);
                r1[1] = x[1] - __uint_as_float(t[1]);
            }
            MF(13);
            SB();
            if constexpr (SPLIT) {
                r1[2] = x[2] - __uint_as_float(t[2]);
                r1[3] = x[3] - __uint_as_float(t[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(r1[e]) & 0xFFFF0000u;
            }
            MF(14);
            SB();
            if constexpr (SPLIT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] = r1[e] - __uint_as_float(t[e]);
                ph.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                ph.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
            }
            MF(15);
            SB();
            if constexpr (SPLIT) {
                pm.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                pm.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                pl.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                pl.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            }
            MF(16);
            SB();
            if constexpr (SPLIT) {
                unsigned char *p = nxt + wofs + 3 * 64 * 32;
                *reinterpret_cast<uint2 *>(p) = ph;
                *reinterpret_cast<uint2 *>(p + PLANE) = pm;
                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = pl;
            }
            MF(17);
            SB();
        }
        // ---- split of R[4]
        {
            float x[4] = {R[4].x, R[4].y, R[4].z, R[4].w};
            if constexpr (SPLIT) {
                if constexpr (ABL == 3) x[0] += (float)s;  // keeps the split inside the loop
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(x[e]) & 0xFFFF0000u;
                r1[0] = x[0] - __uint_as_float(t[0]);
                r1[1] = x[1] - __uint_as_float(t[1]);
            }
            MF(18);
            SB();
            if constexpr (SPLIT) {
                r1[2] = x[2] - __uint_as_float(t[2]);
                r1[3] = x[3] - __uint_as_float(t[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(r1[e]) & 0xFFFF0000u;
            }
            MF(19);
            SB();
            if constexpr (SPLIT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] = r1[e] - __uint_as_float(t[e]);
                ph.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                ph.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
            }
            MF(20);
            SB();
            if constexpr (SPLIT) {
                pm.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                pm.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                pl.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                pl.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            }
            MF(21);
            SB();
            if constexpr (SPLIT) {
                unsigned char *p = nxt + wofs + 4 * 64 * 32;
                *reinterpret_cast<uint2 *>(p) = ph;
                *reinterpret_cast<uint2 *>(p + PLANE) = pm;
                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = pl;
            }
            MF(22);
            SB();
        }
        // ---- split of R[5]
        {
            float x[4] = {R[5].x, R[5].y, R[5].z, R[5].w};
            if constexpr (SPLIT) {
                if constexpr (ABL == 3) x[0] += (float)s;  // keeps the split inside the loop
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(x[e]) & 0xFFFF0000u;
                r1[0] = x[0] - __uint_as_float(t[0]);
                r1[1] = x[1] - __uint_as_float(t[1]);
            }
            MF(23);
            SB();
            if constexpr (SPLIT) {
                r1[2] = x[2] - __uint_as_float(t[2]);
                r1[3] = x[3] - __uint_as_float(t[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) t[e] = __float_as_uint(r1[e]) & 0xFFFF0000u;
            }
            MF(24);
            SB();
            if constexpr (SPLIT) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r2[e] = r1[e] - __uint_as_float(t[e]);
                ph.x = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
                ph.y = __builtin_amdgcn_perm(__float_as_uint(x[3]), __float_as_uint(x[2]), 0x07060302u);
            }
            MF(25);
            SB();
            if constexpr (SPLIT) {
                pm.x = __builtin_amdgcn_perm(__float_as_uint(r1[1]), __float_as_uint(r1[0]), 0x07060302u);
                pm.y = __builtin_amdgcn_perm(__float_as_uint(r1[3]), __float_as_uint(r1[2]), 0x07060302u);
                pl.x = __builtin_amdgcn_perm(__float_as_uint(r2[1]), __float_as_uint(r2[0]), 0x07060302u);
                pl.y = __builtin_amdgcn_perm(__float_as_uint(r2[3]), __float_as_uint(r2[2]), 0x07060302u);
            }
            MF(26);
            SB();
            if constexpr (SPLIT) {
                unsigned char *p = nxt + wofs + 5 * 64 * 32;
                *reinterpret_cast<uint2 *>(p) = ph;
                *reinterpret_cast<uint2 *>(p + PLANE) = pm;
                *reinterpret_cast<uint2 *>(p + 2 * PLANE) = pl;
            }
            MF(27);
            SB();
        }
        int col = (s + 2) * SK + c4;
        col = col < a.K - 4 ? col : a.K - 4;
        if constexpr (ABL == 0) R[0] = *reinterpret_cast<const float4 *>(src[0] + col);
        MF(28);
        SB();
        if constexpr (ABL == 0) R[1] = *reinterpret_cast<const float4 *>(src[1] + col);
        MF(29);
        SB();
        if constexpr (ABL == 0) R[2] = *reinterpret_cast<const float4 *>(src[2] + col);
        MF(30);
        SB();
        if constexpr (ABL == 0) R[3] = *reinterpret_cast<const float4 *>(src[3] + col);
        MF(31);
        SB();
        if constexpr (ABL == 0) R[4] = *reinterpret_cast<const float4 *>(src[4] + col);
        MF(32);
        SB();
        if constexpr (ABL == 0) R[5] = *reinterpret_cast<const float4 *>(src[5] + col);
        MF(33);
        SB();
        // the remaining MFMAs, bare
        MF(34);
        MF(35);
        MF(36);
        MF(37);
        MF(38);
        MF(39);
        MF(40);
        MF(41);
        MF(42);
        MF(43);
        MF(44);
        MF(45);
        MF(46);
        MF(47);
        SB();
        if constexpr (ABL != 1) lds_barrier();
    }
#undef MF
#undef SB
    // <<< steady-state loop
    if (s + 1 < nst) {  // stage nst-2: the last stage is in registers
        unsigned char *cur = lds + (s & 1) * BUF, *nxt = lds + ((s + 1) & 1) * BUF;
        zero_tail(s + 1);
        split_stage(nxt);
        mma_stage(cur);
        lds_barrier();
        ++s;
    }
    mma_stage(lds + (s & 1) * BUF);
    lds_barrier();
    linear_epilogue<TM, TN, WM, WN>(acc, reinterpret_cast<float *>(lds), a, m0, n0, wm, wn, lane, wave);
}

// GEMM engine shared by every MFMA kernel of the library: 0 = f32 MFMA (exact fp32 chain), 1 = bf16x6, 2 = f16x3 on
// the split-image entry points (linear_f16x3.hip) and bf16x6 everywhere else (default).
static int g_engine = -1;
int gemm_engine() {
    if (g_engine < 0) {
        const char *e = getenv("PHONLP_GEMM_ENGINE");
        const int v = e ? atoi(e) : 2;
        g_engine = v < 0 ? 0 : (v > 2 ? 2 : v);
    }
    return g_engine;
}
void set_gemm_engine(int e) { g_engine = e; }

// Grid size: one block per tile.  (A persistent grid of two blocks per CU walking strided tiles was +3 % with the chip
// to itself, but its static partition stalls when kernels of other streams hold CUs, which is how
// TaskHeads.annotate_batch runs: removed.)
static int persistent_grid(int ntiles) { return ntiles; }

// `sentlens` (with L) marks the padded rows (row % L >= sentlens[row / L]).  They are written as zeros -- or, when
// `pad_bias` is given, as act(pad_bias): that is what a following layer makes of an all-zero input row, used when
// the layer that produced those zero rows has been folded into this one (heads.py: dep_hid folded into W1/W2).
int launch_linear(const char *tag, const float *X, long ldx, const float *W, long ldw, const float *bias, float *Y, long ldy, int M,
                  int N, int K, int relu, const int32_t *sentlens, int L, hipStream_t stream, const float *pad_bias, int nbatch,
                  long bsx, long bsw, long bsy) {
    if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return fail(PHONLP_ERR_SHAPE, "linear: bad argument");
    if ((K & 3) || (ldx & 3) || (ldw & 3) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15))
        return fail(PHONLP_ERR_SHAPE, "linear: K, ldx, ldw must be multiples of 4 and X, W 16-byte aligned");
    if (sentlens && L <= 0) return fail(PHONLP_ERR_SHAPE, "linear: sentlens given but L <= 0");
    if (pad_bias && !sentlens) return fail(PHONLP_ERR_SHAPE, "linear: pad_bias needs sentlens");
    LinearArgs a{X, W, bias, pad_bias, Y, sentlens, ldx, ldw, ldy, M, N, K, L, relu, 0, 0};
    if (nbatch < 1 || (nbatch > 1 && ((bsx & 3) || (bsw & 3)))) return fail(PHONLP_ERR_SHAPE, "linear: bad batch strides");
    a.bsx = bsx; a.bsw = bsw; a.bsy = bsy;
    const bool stage_a = tag && !strcmp(tag, "linear:label_stage_a");
    ProfScope ps(tag, stream);
    a.num_m = ceil_div(M, 128);
    const int eng = gemm_engine();
    if (N <= 64) {
        // Narrow layers (upos_clf, ner_tag_clf) are HBM-bound row streams: 128-row blocks give one block per CU at
        // B*L = 32768 (latency-bound); 32-row blocks of one wave give 4x the loads in flight (PHONLP_NARROW_BM=128: A/B).
        a.num_n = ceil_div(N, 32);
        if (M >= 4096) {
            a.num_m = ceil_div(M, 32);
            const dim3 g(a.num_m * a.num_n, nbatch);
            if (eng) hipLaunchKernelGGL((linear_kernel<32, 32, 1, 1, 0, 1>), g, dim3(64), 0, stream, a);
            else hipLaunchKernelGGL((linear_kernel<32, 32, 1, 1, 0, 0>), g, dim3(64), 0, stream, a);
            return check_launch("linear_kernel");
        }
        const dim3 g(persistent_grid(a.num_m * a.num_n), nbatch);
        if (eng) hipLaunchKernelGGL((linear_kernel<128, 32, 4, 1, 0, 1>), g, dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((linear_kernel<128, 32, 4, 1, 0, 0>), g, dim3(256), 0, stream, a);
    } else {
        a.num_n = ceil_div(N, 128);
        const dim3 g(persistent_grid(a.num_m * a.num_n), nbatch);
        constexpr bool in_regs = false;  // (the split-per-fragment-use form of engine 1: measured behind, kept for small M)
        const dim3 g1(a.num_m * a.num_n, nbatch);
        if (eng && M >= 1024 && K >= 8) {
            static PerDeviceOnce attr_once;
    int attr_dev;
            constexpr int LDS = 2 * 3 * 384 * 32;
            if (attr_once.pending(&attr_dev)) {
                if (hipFuncSetAttribute((const void *)linear_split256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess ||
                    hipFuncSetAttribute((const void *)linear_split256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess)
                    return fail(PHONLP_ERR_LAUNCH, "linear: cannot raise the dynamic LDS limit");
                attr_once.mark(attr_dev);
            }
            a.num_m = ceil_div(M, 256);
            const dim3 g2(a.num_m * a.num_n, nbatch);
#ifdef PHONLP_ABLATION
            static const int abl = getenv("PHONLP_X6_ABLATION") ? atoi(getenv("PHONLP_X6_ABLATION")) : 0;  // timing only
#else
            constexpr int abl = 0;  // the ablated loop variants are only instantiated in -DPHONLP_ABLATION builds
#endif
            (void)abl;
#ifdef PHONLP_ABLATION
            if (abl) {
                const void *fns[] = {(const void *)linear_split256_kernel<0, 1>, (const void *)linear_split256_kernel<0, 2>,
                                     (const void *)linear_split256_kernel<0, 3>};
                for (const void *f : fns) (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
                if (abl == 1) hipLaunchKernelGGL((linear_split256_kernel<0, 1>), g2, dim3(256), LDS, stream, a);
                else if (abl == 2) hipLaunchKernelGGL((linear_split256_kernel<0, 2>), g2, dim3(256), LDS, stream, a);
                else hipLaunchKernelGGL((linear_split256_kernel<0, 3>), g2, dim3(256), LDS, stream, a);
            } else
#endif
            if (stage_a) hipLaunchKernelGGL((linear_split256_kernel<1>), g2, dim3(256), LDS, stream, a);
            else hipLaunchKernelGGL((linear_split256_kernel<0>), g2, dim3(256), LDS, stream, a);
            return check_launch("linear_split256_kernel");
        }
        if (stage_a) {
            if (eng && !in_regs) hipLaunchKernelGGL((linear_split_kernel<1>), g1, dim3(256), 0, stream, a);
            else if (eng) hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 1, 1>), g, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 1, 0>), g, dim3(256), 0, stream, a);
        } else {
            if (eng && !in_regs) hipLaunchKernelGGL((linear_split_kernel<0>), g1, dim3(256), 0, stream, a);
            else if (eng) hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 0, 1>), g, dim3(256), 0, stream, a);
            else hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2, 0, 0>), g, dim3(256), 0, stream, a);
        }
    }
    return check_launch("linear_kernel");
}

// ---------------------------------------------------------------------------------- packing
__global__ void pack_scorer_linear_kernel(const float *W1, const float *b1, const float *W2, const float *b2, int H,
                                          int D, int KP, float *W12, float *b12) {
    const long total = 2L * KP * D;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int row = idx / D, d = idx % D;
        const int side = row / KP, r = row % KP;
        const float *W = side ? W2 : W1;
        W12[idx] = r < H ? W[(long)r * D + d] : 0.f;
        if (d == 0) {
            const float *b = side ? b2 : b1;
            b12[row] = r < H ? b[r] : (r == H ? 1.f : 0.f);
        }
    }
}

__global__ void pack_bilinear_kernel(const float *weight, int H, int O, int KP, float *Wbt) {
    const int H1 = H + 1;
    const long total = (long)O * KP * KP;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int d1 = idx % KP;
        const long row = idx / KP;
        const int d2 = row % KP, o = row / KP;
        float v = 0.f;
        if (d1 < H1 && d2 < H1) v = weight[(long)d1 * H1 * O + (long)o * H1 + d2];
        Wbt[idx] = v;
    }
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_set_gemm_engine(int engine) {
    if (engine < 0 || engine > 2) return fail(PHONLP_ERR_SHAPE, "set_gemm_engine: 0 (f32 MFMA), 1 (bf16x6) or 2 (f16x3)");
    set_gemm_engine(engine);
    return PHONLP_OK;
}
extern "C" int phonlp_get_gemm_engine(void) { return gemm_engine(); }

extern "C" int phonlp_linear(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy,
                             int M, int N, int K, int relu, const int32_t *sentlens, int L, void *stream) {
    return launch_linear("linear", X, ldx, W, ldw, bias, Y, ldy, M, N, K, relu, sentlens, L, as_stream(stream), nullptr, 1, 0, 0, 0);
}

extern "C" int phonlp_pack_scorer_linear(const float *W1, const float *b1, const float *W2, const float *b2, int H,
                                         int D, float *W12, float *b12, void *stream) {
    if (!W1 || !b1 || !W2 || !b2 || !W12 || !b12 || H <= 0 || D <= 0)
        return fail(PHONLP_ERR_SHAPE, "pack_scorer_linear: bad argument");
    const int KP = kp_of(H);
    hipLaunchKernelGGL(pack_scorer_linear_kernel, dim3(1024), dim3(256), 0, as_stream(stream), W1, b1, W2, b2, H, D,
                       KP, W12, b12);
    return check_launch("pack_scorer_linear_kernel");
}

extern "C" int phonlp_pack_bilinear(const float *weight, int H, int O, float *Wbt, void *stream) {
    if (!weight || !Wbt || H <= 0 || O <= 0) return fail(PHONLP_ERR_SHAPE, "pack_bilinear: bad argument");
    const int KP = kp_of(H);
    hipLaunchKernelGGL(pack_bilinear_kernel, dim3(2048), dim3(256), 0, as_stream(stream), weight, H, O, KP, Wbt);
    return check_launch("pack_bilinear_kernel");
}
