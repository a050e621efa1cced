// Training-side heads, backward THROUGH the layers (the gradients of the losses w.r.t. the
// scores are dep_loss.hip / crf_grad.hip): what autograd does behind the reference's nn.Linear + ReLU layers
// (annotate_model.py:43-94 / jointmodel/model.py:29-91) and behind a one-output PairwiseBilinear (biaffine.py:21-37, the
// unlabeled / linearization / distance scorers), as NT products on the library's own dense-layer kernels
// (launch_linear: C[m][n] = sum_k A[m][k] B[n][k], fp32-grade on every engine) around transposing / masking passes.
//
//   linear layer  Y = act(X W^T + b):   dZ = dY * [Y > 0] (ReLU) or dY;   dX = dZ W;   dW = dZ^T X;   db = column sums of dZ
//   bilinear      S[b,i,j] = sum_{d1,d2} X1[b,i,d1] Wb[d2][d1] X2[b,j,d2]   (Wb = the phonlp_pack_bilinear image, O = 1):
//       U[b] = G[b] X2[b]          dX1 = U Wb
//       T    = X1 Wb^T             dX2[b] = G[b]^T T[b]
//       dWb[d2][d1] = sum_m U[m][d2] X1[m][d1]
//
// Every product's reduction dimension is padded to a multiple of 4 with zeros inside the workspace (the kernels need
// K % 4 == 0), so the caller's shapes are free.  Column sums are taken in a fixed two-level order (no float atomics):
// the results do not depend on the launch.
#include "common.h"

namespace phonlp {

int launch_linear(const char *tag, const float *X, long ldx, const float *W, long ldw, const float *bias, float *Y, long ldy, int M,
                  int N, int K, int relu, const int32_t *sentlens, int L, hipStream_t stream, const float *pad_bias, int nbatch,
                  long bsx, long bsw, long bsy);

// Y[c][r] = X[r][c] for r < rows, c < cols; Y has `ldy` >= rows columns per row and its columns rows .. rows_pad-1 are
// zero-filled (the padding of a reduction dimension).  Batched over blockIdx.z.
__global__ __launch_bounds__(1024) void bw_transpose_kernel(const float *X, long ldx, int rows, int cols, float *Y, long ldy,
                                                            int rows_pad, long bsx, long bsy) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    X += (long)blockIdx.z * bsx;
    Y += (long)blockIdx.z * bsy;
    const int r = r0 + ty, c = c0 + tx;
    tile[ty][tx] = (r < rows && c < cols) ? X[(long)r * ldx + c] : 0.f;
    __syncthreads();
    const int rr = r0 + tx, cc = c0 + ty;  // output element (row cc, column rr)
    if (cc < cols && rr < rows_pad) Y[(long)cc * ldy + rr] = tile[tx][ty];
}

// dZ[m][n] = dY[m][n] * (relu ? Yact[m][n] > 0 : 1) for n < N; columns N .. N4-1 of dZ zero (ld of dZ = N4)
__global__ __launch_bounds__(256) void bw_mask_kernel(const float *dY, long lddy, const float *Yact, long ldy, int M, int N,
                                                      int N4, int relu, float *dZ) {
    const long total = (long)M * N4;
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx / N4;
        const int n = (int)(idx - m * N4);
        float v = 0.f;
        if (n < N) {
            v = dY[m * lddy + n];
            if (relu && !(Yact[m * ldy + n] > 0.f)) v = 0.f;
        }
        dZ[idx] = v;
    }
}

// part[g][n] = sum of dZ[m][n] over the rows m of group g (128 rows, ascending); then db[n] = sum_g part[g][n], ascending
__global__ __launch_bounds__(256) void bw_colsum_part_kernel(const float *dZ, long ld, int M, int N, float *part) {
    const int n = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
    if (n >= N) return;
    const int m1 = min(M, (g + 1) * 128);
    float s = 0.f;
    for (int m = g * 128; m < m1; ++m) s += dZ[(long)m * ld + n];
    part[(long)g * N + n] = s;
}
__global__ __launch_bounds__(256) void bw_colsum_final_kernel(const float *part, int G, int N, float *db) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[(long)g * N + n];
    db[n] = s;
}

// Label scorer, gradient at the gold heads.  One block per word m = (b, i): Z[m][o*KP + d2] = g[m][o] * X2[b, head_m][d2]
// (a word's gradient w.r.t. its row of stage A's output T).  Words without a label have g = 0; their head is clamped.
__global__ __launch_bounds__(256) void bw_label_z_kernel(const float *g, const int32_t *gold_head, const float *XF, int L, int KP,
                                                         int O, float *Z) {
    const long m = blockIdx.x;
    const int b = (int)(m / L);
    int h = gold_head[m];
    h = h < 0 ? 0 : (h >= L ? L - 1 : h);
    const float *x2 = XF + ((long)b * L + h) * 2 * KP + KP;
    float *z = Z + m * (long)O * KP;
    for (int t = threadIdx.x; t < O * KP; t += 256) {
        const int o = t / KP, d2 = t - o * KP;
        z[t] = g[m * O + o] * x2[d2];
    }
}
// dX2h[m][d2] = sum_o g[m][o] T[m][o*KP + d2]  (ascending o)
__global__ __launch_bounds__(256) void bw_label_dx2h_kernel(const float *g, const float *T, int KP, int O, float *dX2h) {
    const long m = blockIdx.x;
    const float *t = T + m * (long)O * KP;
    for (int d2 = threadIdx.x; d2 < KP; d2 += 256) {
        float s = 0.f;
        for (int o = 0; o < O; ++o) s += g[m * O + o] * t[(long)o * KP + d2];
        dX2h[m * KP + d2] = s;
    }
}
// dX2[b, j] = sum over the labelled dependents i (1 <= i < len_b, ascending) with gold head j of dX2h[b, i]: one block per
// (b, j), written for EVERY j (zeros where nobody points at j)
__global__ __launch_bounds__(256) void bw_label_scatter_kernel(const float *dX2h, const int32_t *gold_head, const int32_t *sentlens,
                                                               int L, int KP, float *dXF) {
    const int b = blockIdx.x / L, j = blockIdx.x - b * L;
    int len = sentlens[b];
    len = len > L ? L : len;
    __shared__ int deps[1024];
    __shared__ int nd;
    if (threadIdx.x == 0) {
        int n = 0;
        for (int i = 1; i < len; ++i)
            if (gold_head[(long)b * L + i] == j) deps[n++] = i;
        nd = n;
    }
    __syncthreads();
    float *out = dXF + ((long)b * L + j) * 2 * KP + KP;
    for (int d2 = threadIdx.x; d2 < KP; d2 += 256) {
        float s = 0.f;
        for (int k = 0; k < nd; ++k) s += dX2h[((long)b * L + deps[k]) * KP + d2];
        out[d2] = s;
    }
}

// Row soft-max (the soft POS distribution, ref jointmodel/model.py:135 / annotate_model.py:140,177) and its backward
// dZ = P * (dP - sum_u P dP): one wavefront per row, U <= 1024.
__global__ __launch_bounds__(256) void bw_softmax_fwd_kernel(const float *Z, int M, int U, float *P) {
    const int lane = threadIdx.x & 63;
    const long m = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (m >= M) return;
    const float *z = Z + m * U;
    float mx = -INFINITY;
    for (int u = lane; u < U; u += 64) mx = fmaxf(mx, z[u]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int u = lane; u < U; u += 64) sum += expf(z[u] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    for (int u = lane; u < U; u += 64) P[m * U + u] = expf(z[u] - mx) / sum;
}
__global__ __launch_bounds__(256) void bw_softmax_bwd_kernel(const float *P, const float *dP, int M, int U, float *dZ) {
    const int lane = threadIdx.x & 63;
    const long m = blockIdx.x * 4L + (threadIdx.x >> 6);
    if (m >= M) return;
    float dot = 0.f;
    for (int u = lane; u < U; u += 64) dot += P[m * U + u] * dP[m * U + u];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    for (int u = lane; u < U; u += 64) dZ[m * U + u] = P[m * U + u] * (dP[m * U + u] - dot);
}

static int transpose(const float *X, long ldx, int rows, int cols, float *Y, long ldy, int rows_pad, int nbatch, long bsx,
                     long bsy, hipStream_t st) {
    ProfScope ps("bw_transpose", st);
    hipLaunchKernelGGL(bw_transpose_kernel, dim3(ceil_div(cols, 32), ceil_div(rows_pad, 32), nbatch), dim3(1024), 0, st, X, ldx,
                       rows, cols, Y, ldy, rows_pad, bsx, bsy);
    return check_launch("bw_transpose_kernel");
}

static inline int up4(int v) { return (v + 3) & ~3; }

// A product whose output is small and whose reduction is the batch (dW = dZ^T X, dWb = U^T X1: a few dozen output tiles
// over K = B*L) would leave most CUs idle: the reduction is cut into S chunks run as ONE batched product into S partial
// outputs, which are then added in ascending order (fixed: no float atomics).  Kc: chunk length (multiple of 4); the
// operands' rows are zero-padded to S*Kc.
struct SplitK {
    int S, Kc;
    long kpad() const { return (long)S * Kc; }
};
static SplitK plan_splitk(int Mo, int No, int K) {
    const int tiles = ceil_div(Mo, 256) * ceil_div(No, 128);
    int S = 512 / tiles;
    if (S > K / 256) S = K / 256;
    if (S > 64) S = 64;
    if (S < 1) S = 1;
    SplitK p;
    p.S = S;
    p.Kc = up4(ceil_div(K, S));
    return p;
}
__global__ __launch_bounds__(256) void bw_sum_parts_kernel(const float *part, long stride, int S, long total, int No, float *Y,
                                                           long ldy) {
    for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int k = 0; k < S; ++k) v += part[k * stride + idx];
        const long m = idx / No;
        Y[m * ldy + (idx - m * No)] = v;
    }
}
static int gemm_splitk(const char *tag, const float *A, const float *Bm, float *Y, long ldy, int Mo, int No, SplitK sk,
                       float *part, hipStream_t st) {
    const long ld = sk.kpad();
    if (sk.S == 1)
        return launch_linear(tag, A, ld, Bm, ld, nullptr, Y, ldy, Mo, No, sk.Kc, 0, nullptr, 0, st, nullptr, 1, 0, 0, 0);
    int rc = launch_linear(tag, A, ld, Bm, ld, nullptr, part, No, Mo, No, sk.Kc, 0, nullptr, 0, st, nullptr, sk.S, sk.Kc, sk.Kc,
                           (long)Mo * No);
    if (rc) return rc;
    ProfScope ps("bw_sum_parts", st);
    hipLaunchKernelGGL(bw_sum_parts_kernel, dim3(1024), dim3(256), 0, st, part, (long)Mo * No, sk.S, (long)Mo * No, No, Y, ldy);
    return check_launch("bw_sum_parts_kernel");
}
static inline float *carve(char *&p, size_t floats) {
    float *r = (float *)p;
    p += align_up(floats * sizeof(float), 256);
    return r;
}

}  // namespace phonlp

using namespace phonlp;

extern "C" size_t phonlp_linear_backward_ws_bytes(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const size_t M4 = (size_t)plan_splitk(N, K, M).kpad(), N4 = up4(N);
    return align_up((size_t)M * N4 * 4, 256) + align_up((size_t)N * M4 * 4, 256) + align_up((size_t)K * M4 * 4, 256) +
           align_up((size_t)K * N4 * 4, 256) + align_up((size_t)ceil_div(M, 128) * N * 4, 256) +
           align_up((size_t)plan_splitk(N, K, M).S * N * K * 4, 256);  // + the split product's partial outputs
}

extern "C" int phonlp_linear_backward(const float *X, int ldx, const float *W, int ldw, const float *Yact, int ldy,
                                      const float *dY, int lddy, int M, int N, int K, int relu, float *dX, int lddx,
                                      float *dW, int lddw, float *db, void *ws, size_t ws_bytes, void *stream) {
    if (!X || !W || !dY || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldw < K || lddy < N || (relu && (!Yact || ldy < N)) ||
        (dX && lddx < K) || (dW && lddw < K))
        return fail(PHONLP_ERR_SHAPE, "linear_backward: bad argument");
    if (!ws || ((uintptr_t)ws & 15) || ws_bytes < phonlp_linear_backward_ws_bytes(M, N, K))
        return fail(PHONLP_ERR_WORKSPACE, "linear_backward: workspace too small or not 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int N4 = up4(N), G = ceil_div(M, 128);
    const SplitK sk = plan_splitk(N, K, M);
    const long M4 = sk.kpad();  // the padded reduction length of dW = dZ^T X
    char *p = (char *)ws;
    float *dZ = carve(p, (size_t)M * N4), *dZT = carve(p, (size_t)N * M4), *XT = carve(p, (size_t)K * M4);
    float *WT = carve(p, (size_t)K * N4), *part = carve(p, (size_t)G * N), *parts = carve(p, (size_t)sk.S * N * K);
    {
        ProfScope ps("bw_mask", st);
        hipLaunchKernelGGL(bw_mask_kernel, dim3(2048), dim3(256), 0, st, dY, (long)lddy, Yact, (long)ldy, M, N, N4, relu, dZ);
        const int rc = check_launch("bw_mask_kernel");
        if (rc) return rc;
    }
    int rc;
    if (db) {
        ProfScope ps("bw_colsum", st);
        hipLaunchKernelGGL(bw_colsum_part_kernel, dim3(ceil_div(N, 256), G), dim3(256), 0, st, dZ, (long)N4, M, N, part);
        hipLaunchKernelGGL(bw_colsum_final_kernel, dim3(ceil_div(N, 256)), dim3(256), 0, st, part, G, N, db);
        if ((rc = check_launch("bw_colsum_kernel"))) return rc;
    }
    if (dX) {  // dX[m][k] = sum_n dZ[m][n] WT[k][n]
        if ((rc = transpose(W, ldw, N, K, WT, N4, N4, 1, 0, 0, st))) return rc;
        if ((rc = launch_linear("bw:dX", dZ, N4, WT, N4, nullptr, dX, lddx, M, K, N4, 0, nullptr, 0, st, nullptr, 1, 0, 0, 0)))
            return rc;
    }
    if (dW) {  // dW[n][k] = sum_m dZT[n][m] XT[k][m]
        if ((rc = transpose(dZ, N4, M, N, dZT, M4, (int)M4, 1, 0, 0, st))) return rc;
        if ((rc = transpose(X, ldx, M, K, XT, M4, (int)M4, 1, 0, 0, st))) return rc;
        if ((rc = gemm_splitk("bw:dW", dZT, XT, dW, lddw, N, K, sk, parts, st))) return rc;
    }
    return PHONLP_OK;
}

extern "C" size_t phonlp_bilinear_backward_ws_bytes(int B, int L, int H) {
    if (B <= 0 || L <= 0 || H <= 0) return 0;
    const size_t KP = kp_of(H), M = (size_t)B * L, L4 = up4(L);
    const SplitK sk = plan_splitk((int)KP, (int)KP, (int)M);
    const size_t M4 = (size_t)sk.kpad();
    return align_up((size_t)sk.S * KP * KP * 4, 256) +   // partial outputs of dWb
           2 * align_up(KP * M4 * 4, 256) +              // X1T, UT
           2 * align_up(M * KP * 4, 256) +               // T, U
           2 * align_up((size_t)B * L * L4 * 4, 256) +   // G, G^T with the reduction dimension padded
           align_up(KP * KP * 4, 256) +                  // Wb^T
           2 * align_up((size_t)B * KP * L4 * 4, 256);   // per-sentence X2^T, T^T
}

/* One-output bilinear scorer (ref biaffine.py:21-37 with O = 1), backward of S[b,i,j] given G = dL/dS [B, L, L]:
 * dXF [M, 2KP] = (dX1' | dX2') w.r.t. the scorer's post-ReLU features XF = (X1' | X2') (the ones / padding columns get
 * whatever the algebra gives them; callers ignore columns >= H), dWbt [KP, KP] in the phonlp_pack_bilinear layout. */
extern "C" int phonlp_bilinear_backward(const float *XF, const float *Wbt, const float *G, int B, int L, int H, float *dXF,
                                        float *dWbt, void *ws, size_t ws_bytes, void *stream) {
    if (!XF || !Wbt || !G || !dXF || !dWbt || B <= 0 || L <= 0 || H <= 0)
        return fail(PHONLP_ERR_SHAPE, "bilinear_backward: bad argument");
    if (((uintptr_t)XF & 15) || ((uintptr_t)Wbt & 15) || ((uintptr_t)dXF & 15))
        return fail(PHONLP_ERR_SHAPE, "bilinear_backward: XF, Wbt, dXF must be 16-byte aligned");
    if (!ws || ((uintptr_t)ws & 15) || ws_bytes < phonlp_bilinear_backward_ws_bytes(B, L, H))
        return fail(PHONLP_ERR_WORKSPACE, "bilinear_backward: workspace too small or not 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int KP = kp_of(H), M = B * L, L4 = up4(L);
    const SplitK sk = plan_splitk(KP, KP, M);
    const long M4 = sk.kpad();
    char *p = (char *)ws;
    float *parts = carve(p, (size_t)sk.S * KP * KP);
    float *X1T = carve(p, (size_t)KP * M4), *UT = carve(p, (size_t)KP * M4);
    float *T = carve(p, (size_t)M * KP), *U = carve(p, (size_t)M * KP);
    float *Gp = carve(p, (size_t)B * L * L4), *GTp = carve(p, (size_t)B * L * L4);
    float *WbT = carve(p, (size_t)KP * KP);
    float *X2Tb = carve(p, (size_t)B * KP * L4), *TTb = carve(p, (size_t)B * KP * L4);
    int rc;
    // the per-sentence operands of the two batched products, reduction dimension (a sentence's L words) padded to L4:
    //   Gp[b][i][j] = G[b][i][j], GTp[b][j][i] = G[b][i][j] (columns L .. L4-1 zero);  X2Tb[b][d2][j] = X2[b][j][d2]
    if ((rc = transpose(G, L, L, L, GTp, L4, L4, B, (long)L * L, (long)L * L4, st))) return rc;
    if ((rc = transpose(GTp, L4, L, L, Gp, L4, L4, B, (long)L * L4, (long)L * L4, st))) return rc;
    if ((rc = transpose(XF + KP, 2L * KP, L, KP, X2Tb, L4, L4, B, (long)L * 2 * KP, (long)KP * L4, st))) return rc;
    // U[b] = G[b] X2[b]:  C[i][d2] = sum_j Gp[b][i][j] X2Tb[b][d2][j]
    if ((rc = launch_linear("bw:U", Gp, L4, X2Tb, L4, nullptr, U, KP, L, KP, L4, 0, nullptr, 0, st, nullptr, B, (long)L * L4,
                            (long)KP * L4, (long)L * KP)))
        return rc;
    // dX1 = U Wb:  C[m][d1] = sum_d2 U[m][d2] WbT[d1][d2]
    if ((rc = transpose(Wbt, KP, KP, KP, WbT, KP, KP, 1, 0, 0, st))) return rc;
    if ((rc = launch_linear("bw:dX1", U, KP, WbT, KP, nullptr, dXF, 2L * KP, M, KP, KP, 0, nullptr, 0, st, nullptr, 1, 0, 0, 0)))
        return rc;
    // T = X1 Wb^T (the forward's stage A):  C[m][d2] = sum_d1 X1[m][d1] Wb[d2][d1]
    if ((rc = launch_linear("bw:T", XF, 2L * KP, Wbt, KP, nullptr, T, KP, M, KP, KP, 0, nullptr, 0, st, nullptr, 1, 0, 0, 0)))
        return rc;
    // dX2[b] = G[b]^T T[b]:  C[j][d2] = sum_i GTp[b][j][i] TTb[b][d2][i]
    if ((rc = transpose(T, KP, L, KP, TTb, L4, L4, B, (long)L * KP, (long)KP * L4, st))) return rc;
    if ((rc = launch_linear("bw:dX2", GTp, L4, TTb, L4, nullptr, dXF + KP, 2L * KP, L, KP, L4, 0, nullptr, 0, st, nullptr, B,
                            (long)L * L4, (long)KP * L4, (long)L * 2 * KP)))
        return rc;
    // dWb[d2][d1] = sum_m UT[d2][m] X1T[d1][m]
    if ((rc = transpose(U, KP, M, KP, UT, M4, (int)M4, 1, 0, 0, st))) return rc;
    if ((rc = transpose(XF, 2L * KP, M, KP, X1T, M4, (int)M4, 1, 0, 0, st))) return rc;
    return gemm_splitk("bw:dWb", UT, X1T, dWbt, KP, KP, KP, sk, parts, st);
}

extern "C" size_t phonlp_label_backward_ws_bytes(int B, int L, int H, int O) {
    if (B <= 0 || L <= 0 || H <= 0 || O <= 0) return 0;
    const size_t KP = kp_of(H), M = (size_t)B * L, M4 = up4((int)M), OK = (size_t)O * KP;
    return 2 * align_up(M * OK * 4, 256) +     // Z, T
           align_up(OK * M4 * 4, 256) +        // Z^T
           align_up(KP * OK * 4, 256) +        // Wbt^T
           align_up(KP * M4 * 4, 256) +        // X1^T
           align_up(M * KP * 4, 256);          // dX2 at the heads, before the scatter
}

/* The label scorer (O relations; ref biaffine.py:21-37, jointmodel/model.py:218-223), backward of the scores AT THE GOLD
 * HEADS: g [B, L, O] = dL/d deprel_scores[b, i, gold_head[b,i], :] (phonlp_dep_loss_grad's d_label: zero on ROOT and
 * padding rows), XF / Wbt as phonlp_label_project takes and makes them, T_opt = that call's T [B*L, O*KP] or NULL
 * (recomputed).  dXF [B*L, 2*KP] = (dX1' | dX2'), dWbt [O*KP, KP]. */
extern "C" int phonlp_label_backward(const float *XF, const float *Wbt, const float *T_opt, const float *g,
                                     const int32_t *gold_head, const int32_t *sentlens, int B, int L, int H, int O,
                                     float *dXF, float *dWbt, void *ws, size_t ws_bytes, void *stream) {
    if (!XF || !Wbt || !g || !gold_head || !sentlens || !dXF || !dWbt || B <= 0 || L <= 0 || L > 1024 || H <= 0 || O <= 0)
        return fail(PHONLP_ERR_SHAPE, "label_backward: bad argument (L <= 1024)");
    if (((uintptr_t)XF & 15) || ((uintptr_t)Wbt & 15) || ((uintptr_t)dXF & 15))
        return fail(PHONLP_ERR_SHAPE, "label_backward: XF, Wbt, dXF must be 16-byte aligned");
    if (!ws || ((uintptr_t)ws & 15) || ws_bytes < phonlp_label_backward_ws_bytes(B, L, H, O))
        return fail(PHONLP_ERR_WORKSPACE, "label_backward: workspace too small or not 16-byte aligned");
    hipStream_t st = as_stream(stream);
    const int KP = kp_of(H), M = B * L, M4 = up4(M), OK = O * KP;
    char *p = (char *)ws;
    float *Z = carve(p, (size_t)M * OK), *T = carve(p, (size_t)M * OK), *ZT = carve(p, (size_t)OK * M4);
    float *WbtT = carve(p, (size_t)KP * OK), *X1T = carve(p, (size_t)KP * M4), *dX2h = carve(p, (size_t)M * KP);
    int rc;
    {
        ProfScope ps("bw_label_z", st);
        hipLaunchKernelGGL(bw_label_z_kernel, dim3(M), dim3(256), 0, st, g, gold_head, XF, L, KP, O, Z);
        if ((rc = check_launch("bw_label_z_kernel"))) return rc;
    }
    // dX1 = Z Wbt:  C[m][d1] = sum_(o,d2) Z[m][(o,d2)] WbtT[d1][(o,d2)]
    if ((rc = transpose(Wbt, KP, OK, KP, WbtT, OK, OK, 1, 0, 0, st))) return rc;
    if ((rc = launch_linear("bw:label_dX1", Z, OK, WbtT, OK, nullptr, dXF, 2L * KP, M, KP, OK, 0, nullptr, 0, st, nullptr, 1, 0, 0, 0)))
        return rc;
    // T = X1 Wbt^T (the forward's stage A; taken from the caller when it kept phonlp_label_project's), then dX2 at the
    // heads and its scatter to the heads' rows
    if (!T_opt && (rc = launch_linear("bw:label_T", XF, 2L * KP, Wbt, KP, nullptr, T, OK, M, OK, KP, 0, nullptr, 0, st, nullptr,
                                      1, 0, 0, 0)))
        return rc;
    {
        ProfScope ps("bw_label_dx2", st);
        hipLaunchKernelGGL(bw_label_dx2h_kernel, dim3(M), dim3(256), 0, st, g, T_opt ? T_opt : T, KP, O, dX2h);
        hipLaunchKernelGGL(bw_label_scatter_kernel, dim3(M), dim3(256), 0, st, dX2h, gold_head, sentlens, L, KP, dXF);
        if ((rc = check_launch("bw_label_dx2_kernel"))) return rc;
    }
    // dWbt[(o,d2)][d1] = sum_m ZT[(o,d2)][m] X1T[d1][m]
    if ((rc = transpose(Z, OK, M, OK, ZT, M4, M4, 1, 0, 0, st))) return rc;
    if ((rc = transpose(XF, 2L * KP, M, KP, X1T, M4, M4, 1, 0, 0, st))) return rc;
    return launch_linear("bw:label_dWbt", ZT, M4, X1T, M4, nullptr, dWbt, KP, OK, KP, M4, 0, nullptr, 0, st, nullptr, 1, 0, 0, 0);
}

/* Row soft-max P = softmax(Z) over the last dimension ([M, U], U <= 1024) and its backward dZ = P * (dP - sum_u P dP):
 * the soft POS distribution that feeds the tag embedding of the NER and parser heads (jointmodel/model.py:135,150,171). */
extern "C" int phonlp_softmax_rows(const float *Z, int M, int U, float *P, void *stream) {
    if (!Z || !P || M <= 0 || U <= 0 || U > 1024) return fail(PHONLP_ERR_SHAPE, "softmax_rows: bad argument (U <= 1024)");
    ProfScope ps("bw_softmax", as_stream(stream));
    hipLaunchKernelGGL(bw_softmax_fwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, as_stream(stream), Z, M, U, P);
    return check_launch("bw_softmax_fwd_kernel");
}
extern "C" int phonlp_softmax_rows_backward(const float *P, const float *dP, int M, int U, float *dZ, void *stream) {
    if (!P || !dP || !dZ || M <= 0 || U <= 0 || U > 1024)
        return fail(PHONLP_ERR_SHAPE, "softmax_rows_backward: bad argument (U <= 1024)");
    ProfScope ps("bw_softmax", as_stream(stream));
    hipLaunchKernelGGL(bw_softmax_bwd_kernel, dim3(ceil_div(M, 4)), dim3(256), 0, as_stream(stream), P, dP, M, U, dZ);
    return check_launch("bw_softmax_bwd_kernel");
}
