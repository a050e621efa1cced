// Deep biaffine arc and label scoring (ref: models/common/biaffine.py, annotate_model.py:189-215).
//
// Data flow per call (all fp32, all resident in HBM workspace provided by the caller):
//   hidden [M=B*L, D] --W12 GEMM + bias + ReLU--> XF [M, S*2*KP]   (X1|X2 per scorer, ones column baked
//                                                                   in by the packed bias, zero pad)
//   X1_s [M, KP]      --stage A GEMM (Wbt_s)--> T [M, O*KP]          (T[m][o*KP + d2])
//   T[b], X2[b]       --stage B (per sentence)--> scores[i][j][o], consumed in the epilogue:
//       arc:   3 scorers combined, diagonal masked, row log-softmax staged through LDS
//       label: running arg-max over o kept in registers; the B x L x L x O tensor never exists.
#include <stdlib.h>

#include "mma_f32.h"

namespace phonlp {

int gemm_engine();
int launch_linear(const char *tag, const float *X, long ldx, const float *W, long ldw, const float *bias, float *Y, long ldy, int M,
                  int N, int K, int relu, const int32_t *sentlens, int L, hipStream_t stream, const float *pad_bias = nullptr, int nbatch = 1,
                  long bsx = 0, long bsw = 0, long bsy = 0);

// ------------------------------------------------------------------------------------------ arc
struct ArcArgs {
    const float *T;   // [M, S*KP]
    const float *XF;  // [M, S*2*KP]
    float *out;       // [B, L, L] log-probabilities
    float *raw;       // NULL or [3, B, L, L]
    long ldT, ldx;
    int B, L, KP, S, LP;
    int use_lin, use_dist;
    // ragged batches packed (phonlp_biaffine_arc_scores_packed): T / XF hold only the live words, sentence b's at rows
    // off[b] .. off[b] + sentlens[b]; a padded head column reads the one row `pad` (what the first layer gives a zero
    // hidden row) and padded dependents are not scored at all (their rows of `out` stay untouched)
    const int32_t *off, *sentlens;
    const float *pad;
};

// slab_load with the rows from `live` on taken from the single row `pad` (a padded head's features)
template <int ROWS, int THREADS>
__device__ __forceinline__ void slab_load_pad(Slab<ROWS, THREADS> &s, const float *base, long ld, int rows_valid, int live,
                                              const float *pad, int k0, int K, int tid) {
#pragma unroll
    for (int q = 0; q < Slab<ROWS, THREADS>::N; ++q) {
        const int idx = tid + q * THREADS;
        const int row = idx >> 3;
        const int c4 = (idx & 7) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *rp = row < live ? base + (long)row * ld : pad;
        if (row < rows_valid && k0 + c4 < K) v = *reinterpret_cast<const float4 *>(rp + k0 + c4);
        s.v[q] = v;
    }
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// logsigmoid / softplus exactly as torch defines them (F.logsigmoid; F.softplus beta=1, threshold=20).
__device__ __forceinline__ float log_sigmoid(float x) { return fminf(x, 0.f) - log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// arc[b,i,col] from the three scorers' raw outputs (ref annotate_model.py:192-211), diagonal masked.
__device__ __forceinline__ float arc_combine(const ArcArgs &a, int b, int i, int col, float u, float lin, float dst,
                                             long plane) {
    const int L = a.L;
    float v = u;
    const int off = col - i;
    if (a.S > 1) {
        if (a.raw) {
            const long e = ((long)b * L + i) * L + col;
            a.raw[e] = u;
            a.raw[plane + e] = lin;
            a.raw[2 * plane + e] = dst;
        }
        if (a.use_lin) {
            const float sg = off > 0 ? 1.f : (off < 0 ? -1.f : 0.f);
            v += log_sigmoid(lin * sg);
        }
        if (a.use_dist) {
            const float pred = 1.f + softplus(dst);
            const float d = fabsf((float)off) - pred;
            v += -logf(d * d / 2.f + 1.f);
        }
    } else if (a.raw) {
        a.raw[((long)b * L + i) * L + col] = u;
    }
    if (off == 0) v = -INFINITY;
    return v;
}

// row-wise log-softmax over all L head columns (padding columns included, ref :214), one wave per row
__device__ __forceinline__ void arc_softmax_row(const float *x, float *o, int L, int lane) {
    float m = -INFINITY;
    for (int j = lane; j < L; j += 64) m = fmaxf(m, x[j]);
    m = wave_max(m);
    float s = 0.f;
    for (int j = lane; j < L; j += 64) s += expf(x[j] - m);
    s = wave_sum(s);
    const float ls = logf(s);
    for (int j = lane; j < L; j += 64) o[j] = (x[j] - m) - ls;
}

// One block = 32 dependents (rows i) of one sentence against all L heads.  4 waves; per 128-wide
// column group wave w owns columns [32w, 32w+32).
template <int ENG>
__global__ __launch_bounds__(256) void arc_kernel(ArcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + 32 * PITCH;
    float *rowbuf = smem + 160 * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y, i0 = blockIdx.x * 32;
    const int L = a.L, KP = a.KP, LP = a.LP;
    const int len = a.off ? min(max(a.sentlens[b], 0), L) : L;
    if (i0 >= len) return;  // packed: nothing but padded dependents here (block-uniform)
    const long rowbase = a.off ? a.off[b] : (long)b * L;
    const int iv = len - i0;
    const int nk = (KP + BK - 1) / BK;
    const int total = a.S * nk;
    const float *Tb = a.T + (rowbase + i0) * a.ldT;
    const long plane = (long)a.B * L * L;

    for (int j0 = 0; j0 < L; j0 += 128) {
        const int jv = L - j0, jl = len - j0;
        const float *Xb = a.XF + (rowbase + j0) * a.ldx;
        f32x16 acc0[1][1], acc1[1][1], acc2[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[0][0][r] = acc1[0][0][r] = acc2[0][0][r] = 0.f;

        Slab<32, 256> ra;
        Slab<128, 256> rb;
        slab_load(ra, Tb, a.ldT, iv, 0, KP, tid);
        slab_load_pad(rb, Xb + KP, a.ldx, jv, jl, a.pad + KP, 0, KP, tid);
        __syncthreads();  // previous column group / softmax phase done with the staging buffers
        slab_store(ra, As, tid);
        slab_store(rb, Bs, tid);
        __syncthreads();
        for (int it = 0; it < total; ++it) {
            const int s = it / nk;
            if (it + 1 < total) {
                const int s2 = (it + 1) / nk, k2 = ((it + 1) % nk) * BK;
                slab_load(ra, Tb + s2 * KP, a.ldT, iv, k2, KP, tid);
                slab_load_pad(rb, Xb + (2 * s2 + 1) * KP, a.ldx, jv, jl, a.pad + (2 * s2 + 1) * KP, k2, KP, tid);
            }
            const float *Bw = Bs + wave * 32 * PITCH;
            const int depth = slab_depth((it - s * nk) * BK, KP);
            if (s == 0) mma_slab<1, 1, ENG>(As, Bw, lane, acc0, depth);
            else if (s == 1) mma_slab<1, 1, ENG>(As, Bw, lane, acc1, depth);
            else mma_slab<1, 1, ENG>(As, Bw, lane, acc2, depth);
            __syncthreads();
            if (it + 1 < total) {
                slab_store(ra, As, tid);
                slab_store(rb, Bs, tid);
                __syncthreads();
            }
        }
        // epilogue: combine the scorers (ref annotate_model.py:192-211) into the LDS row buffer
        const int col = j0 + wave * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            const int i = i0 + row;
            if (row < iv && col < L)
                rowbuf[row * LP + col] = arc_combine(a, b, i, col, acc0[0][0][r], acc1[0][0][r], acc2[0][0][r], plane);
        }
    }
    __syncthreads();
    // row-wise log-softmax over all L head columns (padding columns included, ref :214)
    for (int row = wave; row < iv && row < 32; row += 4) {
        arc_softmax_row(rowbuf + row * LP, a.out + ((long)b * L + i0 + row) * L, L, lane);
    }
}

// One scorer's contribution to arc[b,i,col], accumulated in the LDS row buffer in the reference's order
// (unlabeled, + linearization term, + distance term; ref annotate_model.py:192-211).  Same arithmetic as arc_combine.
__device__ __forceinline__ void arc_fold(const ArcArgs &a, int s, int b, int i, int col, float x, long plane, float *cell) {
    const int L = a.L;
    const int off = col - i;
    if (a.raw) a.raw[s * plane + ((long)b * L + i) * L + col] = x;
    float v;
    if (s == 0) {
        v = x;
    } else if (s == 1) {
        v = *cell;
        if (a.use_lin) {
            const float sg = off > 0 ? 1.f : (off < 0 ? -1.f : 0.f);
            v += log_sigmoid(x * sg);
        }
    } else {
        v = *cell;
        if (a.use_dist) {
            const float pred = 1.f + softplus(x);
            const float d = fabsf((float)off) - pred;
            v += -logf(d * d / 2.f + 1.f);
        }
    }
    if (s == a.S - 1 && off == 0) v = -INFINITY;
    *cell = v;
}

// The same for sentences of up to 192 words: one block = 128 dependents against all L heads, 4 waves as 2 x 2, each
// wave 64 rows x 64 columns of a 128-wide column group.  The scorers run one after the other through ONE set of four
// accumulator tiles; after a scorer's last slab its term is added to the LDS row buffer (each lane owns its cells, so
// no barrier is needed between the terms).  A slab of X2 serves 128 dependents instead of 32: 4x less re-reading of
// the head-side features and 8x more MFMA work per wave and barrier.
// ROWS = 128 or 64 dependents per block: 64 halves the LDS footprint (77 KB: two blocks per CU, so one block's slab
// loads and barriers hide behind the other's MFMAs) at the price of reading the head-side slabs twice as often.
template <int ENG, int ROWS>
__global__ __launch_bounds__(256) void arc_kernel128(ArcArgs a) {
    constexpr int TMW = ROWS / 64;  // 32-row accumulator tiles per wave
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *As = smem;
    float *Bs = smem + ROWS * PITCH;
    float *rowbuf = smem + (ROWS + 128) * PITCH + 4 * 32 * 33;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    float *patch = smem + (ROWS + 128) * PITCH + wave * 32 * 33;
    const int b = blockIdx.y, i0 = blockIdx.x * ROWS;
    const int L = a.L, KP = a.KP, LP = a.LP;
    const int len = a.off ? min(max(a.sentlens[b], 0), L) : L;
    if (i0 >= len) return;  // packed: nothing but padded dependents here (block-uniform)
    const long rowbase = a.off ? a.off[b] : (long)b * L;
    const int iv = len - i0;
    const int nk = (KP + BK - 1) / BK;
    const int total = a.S * nk;
    const float *Tb = a.T + (rowbase + i0) * a.ldT;
    const long plane = (long)a.B * L * L;

    for (int j0 = 0; j0 < L; j0 += 128) {
        const int jv = L - j0, jl = len - j0;
        const float *Xb = a.XF + (rowbase + j0) * a.ldx;
        f32x16 acc[TMW][2];
#pragma unroll
        for (int i = 0; i < TMW; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        Slab<ROWS, 256> ra;
        Slab<128, 256> rb;
        slab_load(ra, Tb, a.ldT, iv, 0, KP, tid);
        slab_load_pad(rb, Xb + KP, a.ldx, jv, jl, a.pad + KP, 0, KP, tid);
        __syncthreads();  // previous column group done with the staging buffers
        slab_store(ra, As, tid);
        slab_store(rb, Bs, tid);
        __syncthreads();
        int s = 0, kk = 0;
        for (int it = 0; it < total; ++it) {
            if (it + 1 < total) {
                const int s2 = kk + 1 == nk ? s + 1 : s, k2 = (kk + 1 == nk ? 0 : kk + 1) * BK;
                slab_load(ra, Tb + s2 * KP, a.ldT, iv, k2, KP, tid);
                slab_load_pad(rb, Xb + (2 * s2 + 1) * KP, a.ldx, jv, jl, a.pad + (2 * s2 + 1) * KP, k2, KP, tid);
            }
            const float *Aw = As + wm * (ROWS / 2) * PITCH, *Bw = Bs + wn * 64 * PITCH;
            mma_slab<TMW, 2, ENG>(Aw, Bw, lane, acc, slab_depth(kk * BK, KP));
            __syncthreads();
            if (it + 1 < total) {
                slab_store(ra, As, tid);
                slab_store(rb, Bs, tid);
                __syncthreads();
            }
            if (++kk == nk) {
                // accumulator tile -> this wave's LDS patch (static offsets), then a rolled loop over the patch: the
                // unrolled form hoists 64 cell / raw addresses out of the slab loop and spills.
#pragma unroll
                for (int i = 0; i < TMW; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            patch[acc_row(r, lane) * 33 + (lane & 31)] = acc[i][j][r];
                            acc[i][j][r] = 0.f;
                        }
                        __builtin_amdgcn_wave_barrier();
                        const int col = j0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll 2
                        for (int e = 0; e < 16; ++e) {
                            const int pr = 2 * e + (lane >> 5);
                            const int row = wm * (ROWS / 2) + i * 32 + pr;
                            if (row < iv && col < L)
                                arc_fold(a, s, b, i0 + row, col, patch[pr * 33 + (lane & 31)], plane, rowbuf + row * LP + col);
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                kk = 0;
                ++s;
            }
        }
    }
    __syncthreads();
    for (int row = wave; row < iv && row < ROWS; row += 4)
        arc_softmax_row(rowbuf + row * LP, a.out + ((long)b * L + i0 + row) * L, L, lane);
}

// ---------------------------------------------------------------------------------------- label
struct LabelArgs {
    const float *T;   // [M, O*KP]
    const float *X2;  // [M, ldx] (already offset to the X2 half)
    int32_t *out;     // [B, L, L]
    float *scores;    // NULL or [B, L, O, L]
    long ldT, ldx;
    int B, L, O, KP;
    int tiles;  // ceil(L / 128)
};

// One block = a 128 x 128 (dependent x head) tile of one sentence; 8 waves as 2 x 4, each wave a
// 64 x 32 sub-tile.  The K loop runs over (o, k): after the last slab of relation o the
// accumulators are folded into a running (max, arg-max) pair and cleared.  Strict '>' keeps the
// lowest o on ties (CPU torch.max / np.argmax semantics).
template <int ENG>
__global__ __launch_bounds__(512) void label_kernel(LabelArgs a) {
    constexpr int THREADS = 512;
    __shared__ __attribute__((aligned(16))) float lds[256 * PITCH];
    float *As = lds, *Bs = lds + 128 * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int L = a.L, KP = a.KP, O = a.O;
    const int per_sent = a.tiles * a.tiles;
    const int b = blockIdx.x / per_sent;
    const int rem = blockIdx.x - b * per_sent;
    const int i0 = (rem / a.tiles) * 128, j0 = (rem % a.tiles) * 128;
    const int iv = L - i0, jv = L - j0;
    const float *Tb = a.T + ((long)b * L + i0) * a.ldT;
    const float *Xb = a.X2 + ((long)b * L + j0) * a.ldx;
    const int nk = (KP + BK - 1) / BK;
    const int total = O * nk;

    f32x16 acc[2][1];
    f32x16 best[2];
    int arg[2][16];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[i][0][r] = 0.f;
            best[i][r] = -INFINITY;
            arg[i][r] = 0;
        }

    Slab<128, THREADS> ra, rb;
    slab_load(ra, Tb, a.ldT, iv, 0, KP, tid);
    slab_load(rb, Xb, a.ldx, jv, 0, KP, tid);
    slab_store(ra, As, tid);
    slab_store(rb, Bs, tid);
    __syncthreads();
    int o = 0, kk = 0;
    for (int it = 0; it < total; ++it) {
        int o2 = o, k2 = kk + 1;
        if (k2 == nk) { k2 = 0; ++o2; }
        if (it + 1 < total) {
            slab_load(ra, Tb + (long)o2 * KP, a.ldT, iv, k2 * BK, KP, tid);
            slab_load(rb, Xb, a.ldx, jv, k2 * BK, KP, tid);
        }
        mma_slab<2, 1, ENG>(As + wm * 64 * PITCH, Bs + wn * 32 * PITCH, lane, acc, slab_depth(kk * BK, KP));
        if (kk == nk - 1) {
            const int col = j0 + wn * 32 + (lane & 31);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][0][r];
                    if (v > best[i][r]) { best[i][r] = v; arg[i][r] = o; }
                    if (a.scores) {
                        const int row = i0 + wm * 64 + i * 32 + acc_row(r, lane);
                        if (row < L && col < L) a.scores[(((long)b * L + row) * O + o) * L + col] = v;
                    }
                    acc[i][0][r] = 0.f;
                }
        }
        __syncthreads();
        if (it + 1 < total) {
            slab_store(ra, As, tid);
            slab_store(rb, Bs, tid);
            __syncthreads();
        }
        o = o2;
        kk = k2;
    }
    const int col = j0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * 64 + i * 32 + acc_row(r, lane);
            if (row < L && col < L) a.out[((long)b * L + row) * L + col] = arg[i][r];
        }
}

// rel[b,i] = label_argmax[b, i, heads[b,i]]  (ref annotate_model.py:263-266)
__global__ void gather_labels_kernel(const int32_t *lab, const int32_t *heads, const int32_t *sentlens, int B, int L,
                                     int32_t *rels) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * L) return;
    const int b = idx / L, i = idx % L;
    int v = 0;
    if (i >= 1 && i < sentlens[b]) {
        const int h = heads[idx];
        if (h >= 0 && h < L) v = lab[((long)b * L + i) * L + h];
    }
    rels[idx] = v;
}

// Labels at the decoded heads: rels[b,i] = argmax_o T[b,i,o,:] . h2[b,heads[b,i],:] -- the only entries of the
// B x L x L x O label tensor that annotate() reads (ref annotate_model.py:263-266 picks
// label_argmax[i][head_i]).  One block per word; h2's row is staged in LDS, the four waves take relations
// o = wave, wave+4, ... and stream T's O*KP-float row once (HBM-bound: 4*O*KP bytes per word).  Strict '>' in
// ascending o keeps the lowest o on ties.  Words outside 1 <= i < len get 0, as gather_labels_kernel does.
__global__ __launch_bounds__(256) void label_at_heads_kernel(const float *T, const float *XF, const int32_t *heads,
                                                              const int32_t *sentlens, int L, int O, int KP,
                                                              int32_t *rels, float *best_opt) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    __shared__ float wbest[4];
    __shared__ int warg[4];
    const int word = blockIdx.x;
    const int b = word / L, i = word - b * L;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int h = -1;
    if (i >= 1 && i < sentlens[b]) {
        h = heads[word];
        if (h < 0 || h >= L) h = -1;
    }
    if (h < 0) {  // block-uniform
        if (tid == 0) {
            rels[word] = 0;
            if (best_opt) best_opt[word] = 0.f;
        }
        return;
    }
    const int n4 = KP >> 2;
    const float4 *x2 = reinterpret_cast<const float4 *>(XF + ((long)b * L + h) * 2 * KP + KP);
    float4 *xs4 = reinterpret_cast<float4 *>(xs);
    for (int c = tid; c < n4; c += 256) xs4[c] = x2[c];
    __syncthreads();
    const float4 *Trow = reinterpret_cast<const float4 *>(T + (long)word * O * KP);
    float bestv = -INFINITY;
    int besto = 0;
    for (int o = wave; o < O; o += 12) {  // three relations per trip: their loads are in flight together
        float s[3] = {0.f, 0.f, 0.f};
        for (int c = lane; c < n4; c += 64) {
            const float4 xv = xs4[c];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int ou = o + 4 * u;
                if (ou < O) {
                    const float4 tv = Trow[(long)ou * n4 + c];
                    s[u] = fmaf(tv.x, xv.x, s[u]);
                    s[u] = fmaf(tv.y, xv.y, s[u]);
                    s[u] = fmaf(tv.z, xv.z, s[u]);
                    s[u] = fmaf(tv.w, xv.w, s[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ou = o + 4 * u;
            const float v = wave_sum(s[u]);
            if (ou < O && v > bestv) {
                bestv = v;
                besto = ou;
            }
        }
    }
    if (lane == 0) {
        wbest[wave] = bestv;
        warg[wave] = besto;
    }
    __syncthreads();
    if (tid == 0) {
        float bv = wbest[0];
        int bo = warg[0];
        for (int w = 1; w < 4; ++w)
            if (wbest[w] > bv || (wbest[w] == bv && warg[w] < bo)) {
                bv = wbest[w];
                bo = warg[w];
            }
        rels[word] = bo;
        if (best_opt) best_opt[word] = bv;
    }
}


// The same on the "rows interleaved by 4" image T4 that engine 2's stage A writes (linear_f16x3.hip:
// T4[(m/4)*N + c][m%4] = T[m][c], N = O*KP): one block per group of four consecutive rows m = 4g .. 4g+3.  The four
// heads' h2 rows are staged in LDS interleaved the same way (xs4[d2] = the four words' h2[d2]), so one 16-byte load of
// T4 and one ds_read_b128 feed four words.  Per word: lanes stride over d2 with an fmaf chain, a fixed-order
// cross-lane sum, strict '>' in ascending o (the summation order differs from label_at_heads_kernel's, so the two
// agree to fp32 rounding, not bit for bit).
__global__ __launch_bounds__(256) void label_at_heads4_kernel(const float *T4, const float *XF, const int32_t *heads,
                                                               const int32_t *sentlens, int M, int L, int O, int KP,
                                                               int32_t *rels, float *best_opt) {
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [KP] float4
    __shared__ float wbest[4][4];
    __shared__ int warg[4][4];
    const int grp = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int hrow[4];  // row of XF that holds the head's features, -1 = word without a label
    bool any = false;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const int m = grp * 4 + w;
        hrow[w] = -1;
        if (m < M) {
            const int b = m / L, i = m - b * L;
            if (i >= 1 && i < sentlens[b]) {
                const int h = heads[m];
                if (h >= 0 && h < L) hrow[w] = b * L + h;
            }
        }
        any = any || hrow[w] >= 0;
    }
    if (!any) {  // block-uniform
        if (tid < 4 && grp * 4 + tid < M) {
            rels[grp * 4 + tid] = 0;
            if (best_opt) best_opt[grp * 4 + tid] = 0.f;
        }
        return;
    }
    for (int c = tid; c < KP; c += 256) {
        float4 v;
        v.x = hrow[0] >= 0 ? XF[(long)hrow[0] * 2 * KP + KP + c] : 0.f;
        v.y = hrow[1] >= 0 ? XF[(long)hrow[1] * 2 * KP + KP + c] : 0.f;
        v.z = hrow[2] >= 0 ? XF[(long)hrow[2] * 2 * KP + KP + c] : 0.f;
        v.w = hrow[3] >= 0 ? XF[(long)hrow[3] * 2 * KP + KP + c] : 0.f;
        reinterpret_cast<float4 *>(xs)[c] = v;
    }
    __syncthreads();
    const float4 *xs4 = reinterpret_cast<const float4 *>(xs);
    const float4 *Tg = reinterpret_cast<const float4 *>(T4) + (long)grp * O * KP;
    float bestv[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int besto[4] = {0, 0, 0, 0};
    for (int o = wave; o < O; o += 12) {  // three relations per trip: their loads are in flight together
        float s[3][4];
#pragma unroll
        for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) s[u][w] = 0.f;
        for (int c = lane; c < KP; c += 64) {  // lanes stride over d2: every wave-load is 1 KiB contiguous
            const float4 xv = xs4[c];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int ou = o + 4 * u;
                if (ou < O) {
                    const float4 tv = Tg[(long)ou * KP + c];
                    s[u][0] = fmaf(tv.x, xv.x, s[u][0]);
                    s[u][1] = fmaf(tv.y, xv.y, s[u][1]);
                    s[u][2] = fmaf(tv.z, xv.z, s[u][2]);
                    s[u][3] = fmaf(tv.w, xv.w, s[u][3]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ou = o + 4 * u;
            // four sums over 64 lanes in 7 exchanges: halves trade word pairs, quarters trade words, then a
            // 16-lane butterfly; lane l ends with the total of word (l>>4)&3 ... (fixed order: deterministic)
            const bool hi32 = lane & 32, hi16 = lane & 16;
            float a0 = hi32 ? s[u][2] : s[u][0], a1 = hi32 ? s[u][3] : s[u][1];      // kept pair
            const float g0 = hi32 ? s[u][0] : s[u][2], g1 = hi32 ? s[u][1] : s[u][3];  // given away
            a0 += __shfl_xor(g0, 32);
            a1 += __shfl_xor(g1, 32);
            float v = hi16 ? a1 : a0;
            v += __shfl_xor(hi16 ? a0 : a1, 16);
#pragma unroll
            for (int sft = 8; sft > 0; sft >>= 1) v += __shfl_xor(v, sft);
            // lanes 0-15: word 0, 16-31: word 1, 32-47: word 2, 48-63: word 3
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float t = __shfl(v, 16 * w);
                if (ou < O && t > bestv[w]) {
                    bestv[w] = t;
                    besto[w] = ou;
                }
            }
        }
    }
    if (lane < 4) {
        wbest[wave][lane] = bestv[lane];
        warg[wave][lane] = besto[lane];
    }
    __syncthreads();
    if (tid < 4 && grp * 4 + tid < M) {
        float bv = wbest[0][tid];
        int bo = warg[0][tid];
        for (int w = 1; w < 4; ++w)
            if (wbest[w][tid] > bv || (wbest[w][tid] == bv && warg[w][tid] < bo)) {
                bv = wbest[w][tid];
                bo = warg[w][tid];
            }
        const bool ok = hrow[tid] >= 0;
        rels[grp * 4 + tid] = ok ? bo : 0;
        if (best_opt) best_opt[grp * 4 + tid] = ok ? bv : 0.f;
    }
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_kp(int H) { return kp_of(H); }

// stage A of the 1 or 3 arc scorers + the arc kernel, from the scorers' features XF [M, S*2*KP]
static int arc_from_features(const float *XF, float *T, int B, int L, int KP, int S, int use_lin, int use_dist,
                             const float *Wbt, float *arc_logp, float *raw_opt, hipStream_t st, int m_live = 0,
                             const int32_t *off = nullptr, const int32_t *sentlens = nullptr, const float *pad = nullptr) {
    const int M = off ? m_live : B * L;
    int rc;
    // stage A of the S scorers as ONE batched launch (item s: X1 of scorer s, its bilinear weight, its slice of T).
    // Engine 1 under engine 2 as well: running it on the f16x3 kernel was measured (the GEMM 0.20 -> 0.16 ms at B=256,
    // L=128) and dropped, because the two split passes it needs cost more than that (3.63 vs 3.59 ms per step).
    rc = launch_linear("linear:arc_stage_a", XF, (long)S * 2 * KP, Wbt, KP, nullptr, T, (long)S * KP, M, KP, KP, 0, nullptr, 0,
                       st, nullptr, S, 2L * KP, (long)KP * KP, KP);
    if (rc) return rc;
    ArcArgs a{T, XF, arc_logp, raw_opt, (long)S * KP, (long)S * 2 * KP, B, L, KP, S, 0, use_lin, use_dist, off, sentlens, pad};
    a.LP = (L + 31) / 32 * 32 + 1;
    const size_t lds = (size_t)(160 * PITCH + 32 * a.LP) * sizeof(float);
    static PerDeviceOnce attr_once;
    int attr_dev;
    if (attr_once.pending(&attr_dev)) {
        const void *fns[] = {(const void *)arc_kernel<0>, (const void *)arc_kernel<1>, (const void *)arc_kernel128<0, 128>,
                             (const void *)arc_kernel128<1, 128>, (const void *)arc_kernel128<0, 64>,
                             (const void *)arc_kernel128<1, 64>};
        for (const void *f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return fail(PHONLP_ERR_LAUNCH, "deep_biaffine_arc: cannot raise the dynamic LDS limit");
        attr_once.mark(attr_dev);
    }
    ProfScope ps("arc_stage_b", st);
    // 64 dependents per block (two blocks per CU: measured ahead of 128 per block) up to 192 words, 32 beyond
    constexpr int rows = 64;
    if (L <= 192) {
        if (rows == 64) {
            const size_t lds64 = (size_t)(192 * PITCH + 4 * 32 * 33 + 64 * a.LP) * sizeof(float);
            if (gemm_engine()) hipLaunchKernelGGL((arc_kernel128<1, 64>), dim3(ceil_div(L, 64), B), dim3(256), lds64, st, a);
            else hipLaunchKernelGGL((arc_kernel128<0, 64>), dim3(ceil_div(L, 64), B), dim3(256), lds64, st, a);
            return check_launch("arc_kernel128");
        }
        const size_t lds128 = (size_t)(256 * PITCH + 4 * 32 * 33 + 128 * a.LP) * sizeof(float);
        if (gemm_engine()) hipLaunchKernelGGL((arc_kernel128<1, 128>), dim3(ceil_div(L, 128), B), dim3(256), lds128, st, a);
        else hipLaunchKernelGGL((arc_kernel128<0, 128>), dim3(ceil_div(L, 128), B), dim3(256), lds128, st, a);
        return check_launch("arc_kernel128");
    }
    if (gemm_engine()) hipLaunchKernelGGL(arc_kernel<1>, dim3(ceil_div(L, 32), B), dim3(256), lds, st, a);
    else hipLaunchKernelGGL(arc_kernel<0>, dim3(ceil_div(L, 32), B), dim3(256), lds, st, a);
    return check_launch("arc_kernel");
}

extern "C" size_t phonlp_biaffine_arc_ws_bytes(int B, int L, int H, int n_scorers) {
    const size_t M = (size_t)B * L, KP = kp_of(H);
    return align_up(M * n_scorers * 2 * KP * sizeof(float), 256) + align_up(M * n_scorers * KP * sizeof(float), 256);
}

extern "C" int phonlp_deep_biaffine_arc(const float *hidden, int D, int B, int L, int H, int n_scorers, int use_lin,
                                        int use_dist, const float *W12, const float *b12, const float *Wbt,
                                        float *arc_logp, float *raw_opt, void *ws, size_t ws_bytes, void *stream,
                                        const int32_t *pad_sentlens, const float *b12_pad) {
    if (!hidden || !W12 || !b12 || !Wbt || !arc_logp || B <= 0 || L <= 0 || H <= 0 || D <= 0)
        return fail(PHONLP_ERR_SHAPE, "deep_biaffine_arc: bad argument");
    if (n_scorers != 1 && n_scorers != 3) return fail(PHONLP_ERR_SHAPE, "deep_biaffine_arc: n_scorers must be 1 or 3");
    if (n_scorers == 1 && (use_lin || use_dist))
        return fail(PHONLP_ERR_SHAPE, "deep_biaffine_arc: use_lin/use_dist need n_scorers == 3");
    if (L > 1024) return fail(PHONLP_ERR_UNSUPPORTED, "deep_biaffine_arc: L > 1024 does not fit the LDS row buffer");
    if (ws_bytes < phonlp_biaffine_arc_ws_bytes(B, L, H, n_scorers) || !ws)
        return fail(PHONLP_ERR_WORKSPACE, "deep_biaffine_arc: workspace too small");
    hipStream_t st = as_stream(stream);
    const int M = B * L, KP = kp_of(H), S = n_scorers;
    float *XF = (float *)ws;
    float *T = (float *)((char *)ws + align_up((size_t)M * S * 2 * KP * sizeof(float), 256));
    if ((pad_sentlens == nullptr) != (b12_pad == nullptr))
        return fail(PHONLP_ERR_SHAPE, "deep_biaffine_arc: pad_sentlens and b12_pad go together");
    int rc = launch_linear("linear:arc_w12", hidden, D, W12, D, b12, XF, (long)S * 2 * KP, M, S * 2 * KP, D, 1, pad_sentlens,
                           pad_sentlens ? L : 0, st, b12_pad);
    if (rc) return rc;
    if (rc) return rc;
    return arc_from_features(XF, T, B, L, KP, S, use_lin, use_dist, Wbt, arc_logp, raw_opt, st);
}

/* Arc scores from the scorers' first-layer features (engine 2 computes those with phonlp_linear_split). */
extern "C" size_t phonlp_biaffine_arc_scores_ws_bytes(int B, int L, int H, int n_scorers) {
    return align_up((size_t)B * L * n_scorers * kp_of(H) * sizeof(float), 256);
}

extern "C" int phonlp_biaffine_arc_scores(const float *XF, int B, int L, int H, int n_scorers, int use_lin, int use_dist,
                                          const float *Wbt, float *arc_logp, float *raw_opt, void *ws, size_t ws_bytes,
                                          void *stream) {
    if (!XF || !Wbt || !arc_logp || B <= 0 || L <= 0 || H <= 0) return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores: bad argument");
    if (n_scorers != 1 && n_scorers != 3) return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores: n_scorers must be 1 or 3");
    if (n_scorers == 1 && (use_lin || use_dist))
        return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores: use_lin/use_dist need n_scorers == 3");
    if (L > 1024) return fail(PHONLP_ERR_UNSUPPORTED, "biaffine_arc_scores: L > 1024 does not fit the LDS row buffer");
    if (((uintptr_t)XF & 15)) return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores: XF must be 16-byte aligned");
    if (ws_bytes < phonlp_biaffine_arc_scores_ws_bytes(B, L, H, n_scorers) || !ws)
        return fail(PHONLP_ERR_WORKSPACE, "biaffine_arc_scores: workspace too small");
    return arc_from_features(XF, (float *)ws, B, L, kp_of(H), n_scorers, use_lin, use_dist, Wbt, arc_logp, raw_opt,
                             as_stream(stream));
}

/* The same for a ragged batch whose live words are packed (sentence b's rows at off[b] .. off[b] + sentlens[b] of XF). */
extern "C" int phonlp_biaffine_arc_scores_packed(const float *XF, int m_live, const int32_t *off, const int32_t *sentlens,
                                                 const float *pad_row, int B, int L, int H, int n_scorers, int use_lin,
                                                 int use_dist, const float *Wbt, float *arc_logp, float *raw_opt, void *ws,
                                                 size_t ws_bytes, void *stream) {
    if (!XF || !Wbt || !arc_logp || !off || !sentlens || !pad_row || B <= 0 || L <= 0 || H <= 0 || m_live <= 0 ||
        (long)m_live > (long)B * L)
        return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores_packed: bad argument");
    if (n_scorers != 1 && n_scorers != 3) return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores_packed: n_scorers must be 1 or 3");
    if (n_scorers == 1 && (use_lin || use_dist))
        return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores_packed: use_lin/use_dist need n_scorers == 3");
    if (L > 1024) return fail(PHONLP_ERR_UNSUPPORTED, "biaffine_arc_scores_packed: L > 1024 does not fit the LDS row buffer");
    if (((uintptr_t)XF & 15) || ((uintptr_t)pad_row & 15))
        return fail(PHONLP_ERR_SHAPE, "biaffine_arc_scores_packed: XF, pad_row must be 16-byte aligned");
    if (ws_bytes < align_up((size_t)m_live * n_scorers * kp_of(H) * sizeof(float), 256) || !ws)
        return fail(PHONLP_ERR_WORKSPACE, "biaffine_arc_scores_packed: workspace too small");
    return arc_from_features(XF, (float *)ws, B, L, kp_of(H), n_scorers, use_lin, use_dist, Wbt, arc_logp, raw_opt,
                             as_stream(stream), m_live, off, sentlens, pad_row);
}

extern "C" size_t phonlp_biaffine_label_ws_bytes(int B, int L, int H, int O) {
    const size_t M = (size_t)B * L, KP = kp_of(H);
    return align_up(M * 2 * KP * sizeof(float), 256) + align_up(M * (size_t)O * KP * sizeof(float), 256);
}

extern "C" int phonlp_deep_biaffine_label(const float *hidden, int D, int B, int L, int H, int O, const float *W12,
                                          const float *b12, const float *Wbt, int32_t *label_argmax,
                                          float *scores_opt, void *ws, size_t ws_bytes, void *stream,
                                          const int32_t *pad_sentlens, const float *b12_pad) {
    if (!hidden || !W12 || !b12 || !Wbt || !label_argmax || B <= 0 || L <= 0 || H <= 0 || D <= 0 || O <= 0)
        return fail(PHONLP_ERR_SHAPE, "deep_biaffine_label: bad argument");
    if (ws_bytes < phonlp_biaffine_label_ws_bytes(B, L, H, O) || !ws)
        return fail(PHONLP_ERR_WORKSPACE, "deep_biaffine_label: workspace too small");
    hipStream_t st = as_stream(stream);
    const int M = B * L, KP = kp_of(H);
    float *XF = (float *)ws;
    float *T = (float *)((char *)ws + align_up((size_t)M * 2 * KP * sizeof(float), 256));
    if ((pad_sentlens == nullptr) != (b12_pad == nullptr))
        return fail(PHONLP_ERR_SHAPE, "deep_biaffine_label: pad_sentlens and b12_pad go together");
    int rc = launch_linear("linear:label_w12", hidden, D, W12, D, b12, XF, 2L * KP, M, 2 * KP, D, 1, pad_sentlens,
                           pad_sentlens ? L : 0, st, b12_pad);
    if (rc) return rc;
    rc = launch_linear("linear:label_stage_a", XF, 2L * KP, Wbt, KP, nullptr, T, (long)O * KP, M, O * KP, KP, 0, nullptr, 0, st);
    if (rc) return rc;
    LabelArgs a{T, XF + KP, label_argmax, scores_opt, (long)O * KP, 2L * KP, B, L, O, KP, ceil_div(L, 128)};
    ProfScope ps("label_stage_b", st);
    if (gemm_engine()) hipLaunchKernelGGL(label_kernel<1>, dim3(B * a.tiles * a.tiles), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(label_kernel<0>, dim3(B * a.tiles * a.tiles), dim3(512), 0, st, a);
    return check_launch("label_kernel");
}

extern "C" int phonlp_label_project(const float *hidden, int D, int B, int L, int H, int O, const float *W12,
                                    const float *b12, const float *Wbt, float *XF, float *T, void *stream,
                                    const int32_t *pad_sentlens, const float *b12_pad) {
    if (!hidden || !W12 || !b12 || !Wbt || !XF || !T || B <= 0 || L <= 0 || H <= 0 || D <= 0 || O <= 0)
        return fail(PHONLP_ERR_SHAPE, "label_project: bad argument");
    hipStream_t st = as_stream(stream);
    const int M = B * L, KP = kp_of(H);
    if ((pad_sentlens == nullptr) != (b12_pad == nullptr))
        return fail(PHONLP_ERR_SHAPE, "label_project: pad_sentlens and b12_pad go together");
    int rc = launch_linear("linear:label_w12", hidden, D, W12, D, b12, XF, 2L * KP, M, 2 * KP, D, 1, pad_sentlens,
                           pad_sentlens ? L : 0, st, b12_pad);
    if (rc) return rc;
    return launch_linear("linear:label_stage_a", XF, 2L * KP, Wbt, KP, nullptr, T, (long)O * KP, M, O * KP, KP, 0, nullptr, 0, st);
}

extern "C" int phonlp_label_at_heads(const float *T, const float *XF, const int32_t *heads, const int32_t *sentlens,
                                     int B, int L, int H, int O, int32_t *rels, float *best_opt, void *stream) {
    if (!T || !XF || !heads || !sentlens || !rels || B <= 0 || L <= 0 || H <= 0 || O <= 0)
        return fail(PHONLP_ERR_SHAPE, "label_at_heads: bad argument");
    const int KP = kp_of(H);
    if ((size_t)KP * sizeof(float) > 60 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "label_at_heads: H too large for the LDS row");
    if (((uintptr_t)T & 15) || ((uintptr_t)XF & 15)) return fail(PHONLP_ERR_SHAPE, "label_at_heads: T, XF must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    ProfScope ps("label_at_heads", st);
    hipLaunchKernelGGL(label_at_heads_kernel, dim3(B * L), dim3(256), (size_t)KP * sizeof(float), st, T, XF, heads,
                       sentlens, L, O, KP, rels, best_opt);
    return check_launch("label_at_heads_kernel");
}

extern "C" int phonlp_label_at_heads_rows4(const float *T4, const float *XF, const int32_t *heads,
                                           const int32_t *sentlens, int B, int L, int H, int O, int32_t *rels,
                                           float *best_opt, void *stream) {
    if (!T4 || !XF || !heads || !sentlens || !rels || B <= 0 || L <= 0 || H <= 0 || O <= 0)
        return fail(PHONLP_ERR_SHAPE, "label_at_heads_rows4: bad argument");
    const int KP = kp_of(H);
    if ((size_t)KP * sizeof(float4) > 60 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "label_at_heads_rows4: H too large for the LDS rows");
    if (((uintptr_t)T4 & 15) || ((uintptr_t)XF & 15)) return fail(PHONLP_ERR_SHAPE, "label_at_heads_rows4: T4, XF must be 16-byte aligned");
    hipStream_t st = as_stream(stream);
    ProfScope ps("label_at_heads", st);
    const int M = B * L;
    hipLaunchKernelGGL(label_at_heads4_kernel, dim3((M + 3) / 4), dim3(256), (size_t)KP * sizeof(float4), st, T4, XF, heads,
                       sentlens, M, L, O, KP, rels, best_opt);
    return check_launch("label_at_heads4_kernel");
}

extern "C" int phonlp_gather_labels(const int32_t *label_argmax, const int32_t *heads, const int32_t *sentlens, int B,
                                    int L, int32_t *rels, void *stream) {
    if (!label_argmax || !heads || !sentlens || !rels || B <= 0 || L <= 0)
        return fail(PHONLP_ERR_SHAPE, "gather_labels: bad argument");
    ProfScope ps("gather_labels", as_stream(stream));
    hipLaunchKernelGGL(gather_labels_kernel, dim3(ceil_div(B * L, 256)), dim3(256), 0, as_stream(stream), label_argmax,
                       heads, sentlens, B, L, rels);
    return check_launch("gather_labels_kernel");
}
