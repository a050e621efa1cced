// Dense layer  Y = act(X . W^T + bias)  on the fp32 MFMA tile engine, plus the one-time weight
// packing kernels.  See include/phonlp_heads.h for the reference call sites each entry replaces.
#include "mma_f32.h"

namespace phonlp {

struct LinearArgs {
    const float *X;
    const float *W;
    const float *bias;
    float *Y;
    const int32_t *sentlens;
    long ldx, ldw, ldy;
    int M, N, K, L;
    int relu;
    int num_m, num_n;
};

// BM x BN block tile, WM x WN waves, each wave a (BM/WM) x (BN/WN) sub-tile.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM *WN * 64) void linear_kernel(LinearArgs a) {
    constexpr int THREADS = WM * WN * 64;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * PITCH];
    float *As = lds, *Bs = lds + BM * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int tm, tn;
    tile_coords(blockIdx.x, gridDim.x, a.num_m, a.num_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const float *Xt = a.X + (long)m0 * a.ldx;
    const float *Wt = a.W + (long)n0 * a.ldw;
    const int mv = a.M - m0, nv = a.N - n0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Slab<BM, THREADS> ra;
    Slab<BN, THREADS> rb;
    slab_load(ra, Xt, a.ldx, mv, 0, a.K, tid);
    slab_load(rb, Wt, a.ldw, nv, 0, a.K, tid);
    slab_store(ra, As, tid);
    slab_store(rb, Bs, tid);
    __syncthreads();
    for (int k0 = 0; k0 < a.K; k0 += BK) {
        const bool more = k0 + BK < a.K;
        if (more) {
            slab_load(ra, Xt, a.ldx, mv, k0 + BK, a.K, tid);
            slab_load(rb, Wt, a.ldw, nv, k0 + BK, a.K, tid);
        }
        mma_slab<TM, TN>(As + wm * TM * 32 * PITCH, Bs + wn * TN * 32 * PITCH, lane, acc);
        __syncthreads();
        if (more) {
            slab_store(ra, As, tid);
            slab_store(rb, Bs, tid);
            __syncthreads();
        }
    }

#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + (wm * TM + i) * 32 + acc_row(r, lane);
            if (row >= a.M) continue;
            bool live = true;
            if (a.sentlens) live = (row % a.L) < a.sentlens[row / a.L];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + (wn * TN + j) * 32 + (lane & 31);
                if (col >= a.N) continue;
                float v = acc[i][j][r] + (a.bias ? a.bias[col] : 0.f);
                if (a.relu) v = fmaxf(v, 0.f);
                a.Y[(long)row * a.ldy + col] = live ? v : 0.f;
            }
        }
    }
}

int launch_linear(const char *tag, const float *X, long ldx, const float *W, long ldw, const float *bias, float *Y, long ldy, int M,
                  int N, int K, int relu, const int32_t *sentlens, int L, hipStream_t stream) {
    if (!X || !W || !Y || M <= 0 || N <= 0 || K <= 0) return fail(PHONLP_ERR_SHAPE, "linear: bad argument");
    if ((K & 3) || (ldx & 3) || (ldw & 3) || ((uintptr_t)X & 15) || ((uintptr_t)W & 15))
        return fail(PHONLP_ERR_SHAPE, "linear: K, ldx, ldw must be multiples of 4 and X, W 16-byte aligned");
    if (sentlens && L <= 0) return fail(PHONLP_ERR_SHAPE, "linear: sentlens given but L <= 0");
    LinearArgs a{X, W, bias, Y, sentlens, ldx, ldw, ldy, M, N, K, L, relu, 0, 0};
    ProfScope ps(tag, stream);
    if (N <= 64) {
        a.num_m = ceil_div(M, 128);
        a.num_n = ceil_div(N, 32);
        hipLaunchKernelGGL((linear_kernel<128, 32, 4, 1>), dim3(a.num_m * a.num_n), dim3(256), 0, stream, a);
    } else {
        a.num_m = ceil_div(M, 128);
        a.num_n = ceil_div(N, 128);
        hipLaunchKernelGGL((linear_kernel<128, 128, 2, 2>), dim3(a.num_m * a.num_n), dim3(256), 0, stream, a);
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

extern "C" int phonlp_linear(const float *X, int ldx, const float *W, int ldw, const float *bias, float *Y, int ldy,
                             int M, int N, int K, int relu, const int32_t *sentlens, int L, void *stream) {
    return launch_linear("linear", X, ldx, W, ldw, bias, Y, ldy, M, N, K, relu, sentlens, L, as_stream(stream));
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
