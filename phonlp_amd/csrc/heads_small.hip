// POS head tail (softmax / arg-max / soft POS embedding), NER emissions, dep_hid, and the error plumbing.
#include <limits.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <vector>

#include "common.h"

namespace phonlp {

static thread_local char g_err[256] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PHONLP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return PHONLP_OK;
}

// ---------------------------------------------------------------------------------- profiling
namespace {
struct ProfState {
    std::mutex mu;
    bool on = false;
    int cap = 0, used = 0;
    std::vector<hipEvent_t> ev;      // 2 per sample
    std::vector<int> tag;            // tag id per sample
    std::vector<std::string> names;  // tag registry
    std::string only;                // non-empty: sample launches with this tag only
} g_prof;
}  // namespace

ProfScope::ProfScope(const char *tag, hipStream_t s) : slot(-1), stream(s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (!g_prof.on || g_prof.used >= g_prof.cap) return;
    if (!g_prof.only.empty() && g_prof.only != tag) return;
    int id = -1;
    for (size_t i = 0; i < g_prof.names.size(); ++i)
        if (g_prof.names[i] == tag) { id = (int)i; break; }
    if (id < 0) { id = (int)g_prof.names.size(); g_prof.names.emplace_back(tag); }
    slot = g_prof.used++;
    g_prof.tag[slot] = id;
    (void)hipEventRecord(g_prof.ev[2 * slot], stream);
}
ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof.ev[2 * slot + 1], stream);
}

int launch_linear(const char *tag, const float *X, long ldx, const float *W, long ldw, const float *bias, float *Y, long ldy, int M,
                  int N, int K, int relu, const int32_t *sentlens, int L, hipStream_t stream, const float *pad_bias = nullptr, int nbatch = 1,
                  long bsx = 0, long bsw = 0, long bsy = 0);

// engine 2 (linear_f16x3.hip)
struct LinearArgs;
int gemm_engine();
int launch_split_rows(const float *X, long ldx, int rows, int K, void *planes, long pitch, float *inv_scale, hipStream_t stream);
int launch_split_rows2(const float *X, long ldx, int rows, void *planes, long pitch, float *inv_scale, const float *X2,
                       long ldx2, int rows2, void *planes2, float *inv_scale2, int K, hipStream_t stream);
int launch_linear_f16x3(const char *tag, const void *Xp, long pitch_x, const float *x_inv, const void *Wp, long pitch_w,
                        const float *w_inv, const float *bias, float *Y, long ldy, int M, int N, int K16, int relu,
                        const int32_t *sentlens, int L, hipStream_t stream, const float *pad_bias, int rows4 = 0,
                        const LinearArgs *at_heads = nullptr, const LinearArgs *image = nullptr);

// One wavefront per token.  logits [M, U] (U <= 64) -> pos_pred, feat = [enc | softmax(logits) . emb].
// ref annotate_model.py:138-146 / :177-183.  The row is assembled in this wave's LDS buffer; from there it goes out as
// the fp32 feat row (feat != NULL) and / or as its engine-2 split image (planes != NULL: the arithmetic of
// split_rows_kernel, linear_f16x3.hip, so the large layers that follow need no split pass over feat).
// `presplit` (with planes, E % 8 == 0): the enc columns of the image and inv_scale[tok] are already in place, written
// by split_rows_kernel from the enc row alone (pos_head_impl does that to feed upos_hid's engine-2 GEMM).  The row's
// scale depends only on the exponent of its largest magnitude, so as long as the soft POS embedding does not exceed
// the enc part's exponent the image of [enc | emb] keeps that scale and only the emb columns are written here; a row
// whose embedding is larger is redone whole.  Either way the image equals split_rows_kernel's of the feat row.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void pos_tail_kernel(const float *enc, long ld_enc, const float *logits,
                                                       const float *emb, const int32_t *sentlens, int M, int L, int E,
                                                       int U, int TE, float *feat, int32_t *pos_pred,
                                                       unsigned char *planes, float *inv_scale, int presplit) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= M) return;
    const int D = E + TE, K16 = (D + 15) / 16 * 16;
    float *row = rows + (threadIdx.x >> 6) * K16;
    const int b = tok / L, i = tok % L;
    float *frow = feat ? feat + (long)tok * D : nullptr;
    unsigned char *prow = planes ? planes + (long)tok * K16 * 4 : nullptr;
    if (i >= sentlens[b]) {  // padding: pad_packed_sequence zeros, arg-max of a zero row = 0
        if (frow)
            for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<float4 *>(frow + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prow) {
            for (int c = lane * 16; c < K16 * 4; c += 1024) *reinterpret_cast<uint4 *>(prow + c) = make_uint4(0, 0, 0, 0);
            if (lane == 0) inv_scale[tok] = 1.f;
        }
        if (pos_pred && lane == 0) pos_pred[tok] = 0;
        return;
    }
    const float x = lane < U ? logits[(long)tok * U + lane] : -INFINITY;
    float best = x;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o);
        const int i2 = __shfl_xor(bi, o);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (pos_pred && lane == 0) pos_pred[tok] = bi;
    float p = lane < U ? expf(x - best) : 0.f;
    float s = p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    p = p / s;
    unsigned mx = 0;
    for (int d0 = 0; d0 < TE; d0 += 64) {  // every lane runs the loop: the broadcasts read lanes that would otherwise be inactive (TE < U)
        const int d = d0 + lane;
        float acc = 0.f;
        for (int u = 0; u < U; ++u) acc += __shfl(p, u) * (d < TE ? emb[u * TE + d] : 0.f);
        if (d < TE) {
            row[E + d] = acc;
            mx = max(mx, __float_as_uint(acc) & 0x7FFFFFFFu);
        }
    }
    for (int c = D + lane; c < K16; c += 64) row[c] = 0.f;
    bool fast = prow && presplit;  // wave-uniform
    int Ex = 0;
    if (fast) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        int Ee = (int)(mx >> 23);
        Ee = Ee < 16 ? 16 : (Ee > 254 ? 254 : Ee);
        Ex = (int)(__float_as_uint(inv_scale[tok]) >> 23) + 14;  // the enc part's exponent (split_rows_kernel)
        if (Ee > Ex) fast = false;
    }
    if (frow || !fast) {
        const float *erow = enc + (long)tok * ld_enc;
        for (int c = lane * 4; c < E; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(erow + c);
            *reinterpret_cast<float4 *>(row + c) = v;
            mx = max(max(mx, __float_as_uint(v.x) & 0x7FFFFFFFu), max(__float_as_uint(v.y) & 0x7FFFFFFFu,
                     max(__float_as_uint(v.z) & 0x7FFFFFFFu, __float_as_uint(v.w) & 0x7FFFFFFFu)));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (frow)
        for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<float4 *>(frow + c) = *reinterpret_cast<const float4 *>(row + c);
    if (prow) {
        if (!fast) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
            Ex = (int)(mx >> 23);
            Ex = Ex < 16 ? 16 : (Ex > 254 ? 254 : Ex);
            if (lane == 0) inv_scale[tok] = __uint_as_float((unsigned)(Ex - 14) << 23);
        }
        const float scale = __uint_as_float((unsigned)(268 - Ex) << 23);
        for (int g = (fast ? E / 8 : 0) + lane; g * 8 < K16; g += 64) {
            h16x8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sv = row[g * 8 + e] * scale;
                h[e] = (_Float16)sv;
                l[e] = (_Float16)(sv - (float)h[e]);
            }
            *reinterpret_cast<h16x8 *>(prow + g * 32) = h;
            *reinterpret_cast<h16x8 *>(prow + g * 32 + 16) = l;
        }
    }
}

// ---- POS head tail with upos_clf inside (the emb == NULL form of the POS head: the "embedding" columns of feat hold the
// tag probabilities themselves, [enc | softmax(logits) | 0], which is what the scorers read once upos_emb_matrix_dep has
// been folded into their first layers).  One launch instead of upos_clf + pos_tail_kernel.
//
// A block takes 32 tokens.  logits[token][u] = hid[token] . Wclf[u] as an exact-fp32 MFMA product (v_mfma_f32_32x32x2_f32:
// rows = tags, columns = tokens), the k range dealt to the four waves in chunks of eight (chunk c to wave c & 3, a lane's
// float4 of a chunk feeds four MFMAs); the partial tiles are added in wave order through LDS.  The order of a token's sum
// is fixed and does not depend on the batch or on the token's place in the block.  In the 32 x 32 accumulator layout a
// lane then holds ONE token (lane & 31) and sixteen of its tags (u = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)), so arg-max
// (lowest index among equals), softmax and the f16 split of the probabilities are in-lane loops plus one exchange with
// lane ^ 32, for 32 tokens at once.  Fast path (presplit image, the probabilities' exponent within the enc part's): only
// the image's probability columns are written, 8 bytes per lane and piece.  Everything else -- padded rows, a row whose
// scale the probabilities change, no presplit image, the fp32 feat output -- goes token by token through the row buffers
// like pos_tail_kernel, the flagged tokens dealt to the four waves.
__device__ __forceinline__ void pos_probs_slow_row(float *row, const float *pr, const float *enc, long ld_enc,
                                                   const int32_t *sentlens, int tok, int L, int E, int TE, float *feat,
                                                   unsigned char *planes, float *inv_scale, int lane) {
    const int D = E + TE, K16 = (D + 15) / 16 * 16;
    const int b = tok / L, i = tok - b * L;
    float *frow = feat ? feat + (long)tok * D : nullptr;
    unsigned char *prow = planes ? planes + (long)tok * K16 * 4 : nullptr;
    if (i >= sentlens[b]) {
        if (frow)
            for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<float4 *>(frow + c) = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prow) {
            for (int c = lane * 16; c < K16 * 4; c += 1024) *reinterpret_cast<uint4 *>(prow + c) = make_uint4(0, 0, 0, 0);
            if (lane == 0) inv_scale[tok] = 1.f;
        }
        return;
    }
    unsigned mx = 0;
    for (int d = lane; E + d < K16; d += 64) {
        const float v = (d < 32 && d < TE) ? pr[d] : 0.f;
        row[E + d] = v;
        mx = max(mx, __float_as_uint(v) & 0x7FFFFFFFu);
    }
    const float *erow = enc + (long)tok * ld_enc;
    for (int c = lane * 4; c < E; c += 256) {
        const float4 v = *reinterpret_cast<const float4 *>(erow + c);
        *reinterpret_cast<float4 *>(row + c) = v;
        mx = max(max(mx, __float_as_uint(v.x) & 0x7FFFFFFFu), max(__float_as_uint(v.y) & 0x7FFFFFFFu,
                 max(__float_as_uint(v.z) & 0x7FFFFFFFu, __float_as_uint(v.w) & 0x7FFFFFFFu)));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (frow)
        for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<float4 *>(frow + c) = *reinterpret_cast<const float4 *>(row + c);
    if (prow) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        int Ex = (int)(mx >> 23);
        Ex = Ex < 16 ? 16 : (Ex > 254 ? 254 : Ex);
        if (lane == 0) inv_scale[tok] = __uint_as_float((unsigned)(Ex - 14) << 23);
        const float scale = __uint_as_float((unsigned)(268 - Ex) << 23);
        for (int g = lane; g * 8 < K16; g += 64) {
            h16x8 h, l;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sv = row[g * 8 + e] * scale;
                h[e] = (_Float16)sv;
                l[e] = (_Float16)(sv - (float)h[e]);
            }
            *reinterpret_cast<h16x8 *>(prow + g * 32) = h;
            *reinterpret_cast<h16x8 *>(prow + g * 32 + 16) = l;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the row buffer is reused by this wave's next token
}

typedef float pf32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
constexpr int kProbPitch = 33;

// This wave's share of logits[token][u] = hid[token] . Wclf[u] for the block's 32 tokens: k chunks wave, wave + 4, ... of
// eight, lane (n, hf) feeding Wclf[n][8c + 4hf ..] and hid[token n][8c + 4hf ..] to four v_mfma_f32_32x32x2_f32.
__device__ __forceinline__ pf32x16 pos_logits_partial(const float *hid, long ldh, const float *Wclf, int H, int U, int tok,
                                                      bool tv, int n, int hf, int wave) {
    pf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
        const float *hrow = hid + (long)(tv ? tok : 0) * ldh;
        const float *wrow = Wclf + (long)(n < U ? n : 0) * H;
        const int NC = (H + 7) >> 3;
        for (int c0 = wave; c0 < NC; c0 += 16) {  // four chunks' loads in flight (chunks beyond the end: zeros, which add nothing)
            float4 a4[4], b4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 8 * (c0 + 4 * j) + 4 * hf;
                a4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                b4[j] = a4[j];
                if (k < H) {
                    if (n < U) a4[j] = *reinterpret_cast<const float4 *>(wrow + k);
                    if (tv) b4[j] = *reinterpret_cast<const float4 *>(hrow + k);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].x, b4[j].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].y, b4[j].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].z, b4[j].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[j].w, b4[j].w, acc, 0, 0, 0);
            }
        }
    }
    return acc;
}

// Wave 0, after the partial tiles are in LDS: the token's logits (partials added in wave order, + bias), arg-max with the
// lowest index among equals, optional logits / POS id outputs (padded rows: zeros), and the tag probabilities p[r] of
// tag u = (r & 3) + 8 (r >> 2) + 4 hf (0 beyond U).
__device__ __forceinline__ void pos_tile_softmax(const pf32x16 &acc, const float *part, const float *bclf, int U, int hf, int lane,
                                                 int tok, bool tv, bool live, float *logits_out, int32_t *pos_pred,
                                                 float (&p)[16]) {
    float lg[16];
    float best = -INFINITY;
    int bi = INT_MAX;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int u = (r & 3) + 8 * (r >> 2) + 4 * hf;
        const float sum = ((acc[r] + part[r * 64 + lane]) + part[(16 + r) * 64 + lane]) + part[(32 + r) * 64 + lane];
        lg[r] = u < U ? sum + bclf[u] : -INFINITY;
        if (lg[r] > best) { best = lg[r]; bi = u; }
    }
    {
        const float v2 = __shfl_xor(best, 32);
        const int i2 = __shfl_xor(bi, 32);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (bi == INT_MAX) bi = 0;
    if (logits_out && tv) {  // padded rows: zeros, as the masked layer writes them
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u0 = 8 * q + 4 * hf;
            float *dst = logits_out + (long)tok * U + u0;
            if (u0 + 3 < U && !(U & 3) && !((uintptr_t)logits_out & 15)) {
                *reinterpret_cast<float4 *>(dst) = live ? make_float4(lg[4 * q], lg[4 * q + 1], lg[4 * q + 2], lg[4 * q + 3])
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (u0 + e < U) dst[e] = live ? lg[4 * q + e] : 0.f;
            }
        }
    }
    if (pos_pred && tv && hf == 0) pos_pred[tok] = live ? bi : 0;
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        p[r] = lg[r] > -INFINITY ? expf(lg[r] - best) : 0.f;
        s += p[r];
    }
    s += __shfl_xor(s, 32);
#pragma unroll
    for (int r = 0; r < 16; ++r) p[r] = p[r] / s;
}

__global__ __launch_bounds__(256) void pos_clf_tail_kernel(const float *hid, long ldh, const float *Wclf, const float *bclf,
                                                           const float *enc, long ld_enc, const int32_t *sentlens, int M,
                                                           int L, int E, int H, int U, int TE, float *feat,
                                                           int32_t *pos_pred, float *logits_out, unsigned char *planes,
                                                           float *inv_scale, int presplit) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float *part = psm;                       // [3][16][64] partial logit tiles of waves 1-3
    float *prob = part + 3 * 1024;           // [32][kProbPitch]
    int *flags = reinterpret_cast<int *>(prob + 32 * kProbPitch);  // [0] = mask of the tokens left to the slow path
    float *rows = prob + 32 * kProbPitch + 32;                     // [4][K16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = E + TE, K16 = (D + 15) / 16 * 16;
    const int tok0 = blockIdx.x * 32, n = lane & 31, hf = lane >> 5;
    const int tok = tok0 + n;
    const bool tv = tok < M;
    pf32x16 acc = pos_logits_partial(hid, ldh, Wclf, H, U, tok, tv, n, hf, wave);
    if (wave) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave == 0) {
        const int b = tv ? tok / L : 0, i = tok - b * L;
        const bool live = tv && i < sentlens[b];
        float p[16];
        pos_tile_softmax(acc, part, bclf, U, hf, lane, tok, tv, live, logits_out, pos_pred, p);
        unsigned mx = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            mx = max(mx, __float_as_uint(p[r]) & 0x7FFFFFFFu);
            prob[n * kProbPitch + (r & 3) + 8 * (r >> 2) + 4 * hf] = p[r];
        }
        mx = max(mx, (unsigned)__shfl_xor((int)mx, 32));
        int Ee = (int)(mx >> 23);
        Ee = Ee < 16 ? 16 : (Ee > 254 ? 254 : Ee);
        bool fast = live && planes && presplit && !feat;
        int Ex = 0;
        if (fast) {
            Ex = (int)(__float_as_uint(inv_scale[tok]) >> 23) + 14;  // the enc part's exponent (split_rows_kernel)
            if (Ee > Ex) fast = false;
        }
        if (fast) {
            const float scale = __uint_as_float((unsigned)(268 - Ex) << 23);
            unsigned char *prow = planes + (long)tok * K16 * 4 + (E >> 3) * 32 + hf * 8;
            const int G = (K16 - E) >> 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < G) {
                    h16x4 h, l;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float sv = p[4 * q + e] * scale;
                        h[e] = (_Float16)sv;
                        l[e] = (_Float16)(sv - (float)h[e]);
                    }
                    *reinterpret_cast<h16x4 *>(prow + q * 32) = h;
                    *reinterpret_cast<h16x4 *>(prow + q * 32 + 16) = l;
                }
            }
            for (int q = 4; q < G; ++q) {  // (TE <= 32: columns beyond the tags are zero)
                *reinterpret_cast<uint2 *>(prow + q * 32) = make_uint2(0, 0);
                *reinterpret_cast<uint2 *>(prow + q * 32 + 16) = make_uint2(0, 0);
            }
        }
        const unsigned long long slow = __ballot(tv && !fast && hf == 0);
        if (lane == 0) flags[0] = (int)(unsigned)slow;
    }
    __syncthreads();
    unsigned mask = (unsigned)flags[0];
    float *row = rows + wave * K16;
    for (int idx = 0; mask; ++idx) {
        const int n2 = __ffs((int)mask) - 1;
        mask &= mask - 1;
        if ((idx & 3) == wave)
            pos_probs_slow_row(row, prob + n2 * kProbPitch, enc, ld_enc, sentlens, tok0 + n2, L, E, TE, feat, planes, inv_scale, lane);
    }
}

// Tail of the fused tagger head (phonlp_tagger_head): one wavefront per token.  logits [M, U] -> pos_pred and the soft POS
// embedding e = softmax(logits) . emb as in pos_tail_kernel; then the NER emissions
//   ner[t] = relu(hid[tok][H + t] + e . Wne[t] + bner[t]),
// where hid's columns H.. hold ner_tag_clf.weight[:, :E] . enc (computed by the upos_hid GEMM as T extra output
// columns) and Wne = ner_tag_clf.weight[:, E:]: ref annotate_model.py:138-150 without ever forming [enc | e].
__global__ __launch_bounds__(256) void tagger_tail_kernel(const float *hid, long ldh, const float *logits, const float *emb,
                                                          const float *Wne, const float *bner, const int32_t *sentlens,
                                                          int M, int L, int H, int U, int TE, int T, int32_t *pos_pred,
                                                          float *ner) {
    extern __shared__ __attribute__((aligned(16))) float rows[];
    const int lane = threadIdx.x & 63;
    const int tok = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tok >= M) return;
    float *e = rows + (threadIdx.x >> 6) * TE;
    const int b = tok / L, i = tok % L;
    if (i >= sentlens[b]) {  // padding: pad_packed_sequence zeros, arg-max of a zero row = 0
        if (lane == 0) pos_pred[tok] = 0;
        for (int t = lane; t < T; t += 64) ner[(long)tok * T + t] = 0.f;
        return;
    }
    const float x = lane < U ? logits[(long)tok * U + lane] : -INFINITY;
    float best = x;
    int bi = lane;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o);
        const int i2 = __shfl_xor(bi, o);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (lane == 0) pos_pred[tok] = bi;
    float p = lane < U ? expf(x - best) : 0.f;
    float s = p;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    p = p / s;
    for (int d0 = 0; d0 < TE; d0 += 64) {  // every lane runs the loop: the broadcasts read lanes that would otherwise be inactive (TE < U)
        const int d = d0 + lane;
        float acc = 0.f;
        for (int u = 0; u < U; ++u) acc += __shfl(p, u) * (d < TE ? emb[u * TE + d] : 0.f);
        if (d < TE) e[d] = acc;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // e . Wne[t]: G = 64 / T lanes per tag, lane (t, g) sums the d = g (mod G) terms; the G shares of a tag are added in
    // ascending g by one lane (a fixed order: deterministic).  More than 64 tags: passes of 64 with one lane per tag.
    float *share = rows + 4 * TE + (threadIdx.x >> 6) * 64;
    const int Tc = T < 64 ? T : 64, G = 64 / Tc;
    for (int t0 = 0; t0 < T; t0 += Tc) {
        const int tl = lane / G, g = lane - tl * G, t = t0 + tl;
        float part = 0.f;
        if (tl < Tc && t < T) {
            const float *w = Wne + (long)t * TE;
            for (int d = g; d < TE; d += G) part = fmaf(e[d], w[d], part);
        }
        share[lane] = part;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < Tc && t0 + lane < T) {
            float dot = 0.f;
            for (int g2 = 0; g2 < G; ++g2) dot += share[lane * G + g2];
            const float v = (hid[(long)tok * ldh + H + t0 + lane] + dot) + bner[t0 + lane];
            ner[(long)tok * T + t0 + lane] = fmaxf(v, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}


// The fused tagger head's tail with upos_clf inside (the emb == NULL form of phonlp_tagger_head, U <= 32): logits, arg-max and
// softmax as in pos_clf_tail_kernel, then
//   ner[t] = relu(hid[tok][H + t] + sum_u p[u] Wg[t][u] + bner[t]),   Wg = Wne . emb^T  [T, U] (folded by the caller at load),
// i.e. the soft POS embedding e = p . emb contracted with Wne in the other order: no e row, no per-token pass over Wne.
__global__ __launch_bounds__(256) void tagger_clf_tail_kernel(const float *hid, long ldh, const float *Wclf, const float *bclf,
                                                              const float *Wg, const float *bner, const int32_t *sentlens,
                                                              int M, int L, int H, int U, int T, int32_t *pos_pred,
                                                              float *logits_out, float *ner) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    float *part = psm;           // [3][16][64]
    float *G = part + 3 * 1024;  // [U][T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tok0 = blockIdx.x * 32, n = lane & 31, hf = lane >> 5;
    const int tok = tok0 + n;
    const bool tv = tok < M;
    pf32x16 acc = pos_logits_partial(hid, ldh, Wclf, H, U, tok, tv, n, hf, wave);
    for (int e = tid; e < U * T; e += 256) {  // G[u][t] = Wg[t][u]
        const int u = e / T, t = e - u * T;
        G[e] = Wg[(long)t * U + u];
    }
    if (wave) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (wave != 0) return;
    const int b = tv ? tok / L : 0, i = tok - b * L;
    const bool live = tv && i < sentlens[b];
    float p[16];
    pos_tile_softmax(acc, part, bclf, U, hf, lane, tok, tv, live, logits_out, pos_pred, p);
    const float *hrow = hid + (long)(tv ? tok : 0) * ldh + H;
    for (int t = 0; t < T; ++t) {
        float dot = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = (r & 3) + 8 * (r >> 2) + 4 * hf;
            dot = fmaf(p[r], G[(u < U ? u : 0) * T + t], dot);  // p = 0 beyond U
        }
        dot += __shfl_xor(dot, 32);
        if (tv && hf == 0) ner[(long)tok * T + t] = live ? fmaxf((hrow[t] + dot) + bner[t], 0.f) : 0.f;
    }
}

}  // namespace phonlp

using namespace phonlp;

extern "C" int phonlp_version(void) { return 101; }

extern "C" int phonlp_profile_begin(int max_samples) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (max_samples <= 0) return fail(PHONLP_ERR_SHAPE, "profile_begin: max_samples <= 0");
    while ((int)g_prof.ev.size() < 2 * max_samples) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return fail(PHONLP_ERR_LAUNCH, "profile_begin: hipEventCreate failed");
        g_prof.ev.push_back(e);
    }
    g_prof.tag.assign(max_samples, -1);
    g_prof.cap = max_samples;
    g_prof.used = 0;
    g_prof.on = true;
    return PHONLP_OK;
}

extern "C" int phonlp_profile_only(const char *tag) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.only = tag ? tag : "";
    return PHONLP_OK;
}

extern "C" int phonlp_profile_end(float *ms_out_host, int32_t *tag_out_host, int cap) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    int n = g_prof.used < cap ? g_prof.used : cap;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.ev[2 * i + 1]) != hipSuccess ||
            hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]) != hipSuccess)
            return fail(PHONLP_ERR_LAUNCH, "profile_end: event query failed");
        ms_out_host[i] = ms;
        tag_out_host[i] = g_prof.tag[i];
    }
    g_prof.used = 0;
    return n;
}

extern "C" const char *phonlp_profile_tag_name(int id) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    if (id < 0 || id >= (int)g_prof.names.size()) return "";
    return g_prof.names[id].c_str();
}
extern "C" const char *phonlp_last_error(void) { return g_err; }

extern "C" size_t phonlp_pos_head_ws_bytes(int B, int L, int E, int H, int U, int TE) {
    (void)TE;
    const size_t M = (size_t)B * L;
    // hid, logits; then (engine 2) the split images of enc and Whid that feed upos_hid's GEMM
    return align_up(M * H * sizeof(float), 256) + align_up(M * U * sizeof(float), 256) +
           align_up(phonlp_split_bytes((int)M, E), 256) + align_up(M * sizeof(float), 256) +
           align_up(phonlp_split_bytes(H, E), 256) + align_up((size_t)H * sizeof(float), 256);
}

static int pos_head_impl(const float *enc, int ld_enc, const int32_t *sentlens, int B, int L, int E, const float *Whid,
                         const float *bhid, const float *Wclf, const float *bclf, const float *emb, int H, int U, int TE,
                         float *feat, int32_t *pos_pred, float *upos_logits, void *feat_planes, float *feat_inv_scale,
                         void *ws, size_t ws_bytes, void *stream) {
    if (!enc || !sentlens || !Whid || !bhid || !Wclf || !bclf || (!feat && !feat_planes) || B <= 0 || L <= 0)
        return fail(PHONLP_ERR_SHAPE, "pos_head: bad argument");
    if (U > 64 || U <= 0) return fail(PHONLP_ERR_UNSUPPORTED, "pos_head: needs 1 <= U <= 64 (one lane per tag)");
    // emb == NULL: the embedding columns hold the tag probabilities themselves (pos_clf_tail_kernel: upos_clf inside the tail)
    if (!emb && (U > 32 || TE < U || TE > 32))
        return fail(PHONLP_ERR_UNSUPPORTED, "pos_head: emb == NULL (probabilities as the embedding) needs U <= TE <= 32");
    if (!emb && (((uintptr_t)Wclf | (uintptr_t)ws | (uintptr_t)enc) & 15))
        return fail(PHONLP_ERR_SHAPE, "pos_head: emb == NULL needs enc, Wclf and the workspace 16-byte aligned");
    if ((E & 3) || (H & 3) || ((E + TE) & 3) || (ld_enc & 3) || ((uintptr_t)feat & 15) || ((uintptr_t)feat_planes & 15))
        return fail(PHONLP_ERR_SHAPE, "pos_head: E, H, E+TE, ld_enc must be multiples of 4 and feat 16-byte aligned");
    if ((feat_planes != nullptr) != (feat_inv_scale != nullptr))
        return fail(PHONLP_ERR_SHAPE, "pos_head: feat_planes and feat_inv_scale go together");
    if (!ws || ws_bytes < phonlp_pos_head_ws_bytes(B, L, E, H, U, TE))
        return fail(PHONLP_ERR_WORKSPACE, "pos_head: workspace too small");
    hipStream_t st = as_stream(stream);
    const int M = B * L;
    float *hid = (float *)ws;
    float *logits = upos_logits ? upos_logits : (float *)((char *)ws + align_up((size_t)M * H * sizeof(float), 256));
    // Engine 2: upos_hid on split images.  enc is split once -- straight into the enc columns of the feat image when
    // the caller wants one (pos_tail_kernel then only adds the embedding columns), else into the workspace; Whid (H
    // rows) is split per call.  E % 8 == 0 keeps the enc columns on whole 32-byte groups of the image.
    const bool x3 = gemm_engine() == 2 && (E % 8) == 0;
    int rc;
    int presplit = 0;
    if (x3) {
        char *p = (char *)ws + align_up((size_t)M * H * sizeof(float), 256) + align_up((size_t)M * U * sizeof(float), 256);
        void *encp = p;             p += align_up(phonlp_split_bytes(M, E), 256);
        float *enci = (float *)p;   p += align_up((size_t)M * sizeof(float), 256);
        void *whp = p;              p += align_up(phonlp_split_bytes(H, E), 256);
        float *whi = (float *)p;
        const int K16e = (E + 15) / 16 * 16;
        long pitch = 4L * K16e;
        if (feat_planes) {
            encp = feat_planes; enci = feat_inv_scale; pitch = 4L * ((E + TE + 15) / 16 * 16);
            presplit = 1;
        }
        rc = launch_split_rows2(enc, ld_enc, M, encp, pitch, enci, Whid, E, H, whp, whi, E, st);
        if (rc) return rc;
        rc = launch_linear_f16x3("linear:upos_hid", encp, pitch, enci, whp, 4L * K16e, whi, bhid, hid, H, M, H, K16e, 1,
                                 sentlens, L, st, nullptr);
    } else {
        rc = launch_linear("linear:upos_hid", enc, ld_enc, Whid, E, bhid, hid, H, M, H, E, 1, sentlens, L, st);
    }
    if (rc) return rc;
    const size_t lds = 4 * (size_t)((E + TE + 15) / 16 * 16) * sizeof(float);
    const size_t lds_fused = lds + (3 * 1024 + 32 * kProbPitch + 32) * sizeof(float);
    if ((emb ? lds : lds_fused) > 64 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "pos_head: E + TE too large for the row buffers");
    if (!emb) {
        ProfScope ps("pos_tail", st);
        hipLaunchKernelGGL(pos_clf_tail_kernel, dim3(ceil_div(M, 32)), dim3(256), lds_fused, st, hid, (long)H, Wclf, bclf, enc,
                           (long)ld_enc, sentlens, M, L, E, H, U, TE, feat, pos_pred, upos_logits,
                           (unsigned char *)feat_planes, feat_inv_scale, presplit);
        return check_launch("pos_clf_tail_kernel");
    }
    rc = launch_linear("linear:upos_clf", hid, H, Wclf, H, bclf, logits, U, M, U, H, 0, sentlens, L, st);
    if (rc) return rc;
    ProfScope ps("pos_tail", st);
    hipLaunchKernelGGL(pos_tail_kernel, dim3(ceil_div(M, 4)), dim3(256), lds, st, enc, (long)ld_enc, logits, emb, sentlens,
                       M, L, E, U, TE, feat, pos_pred, (unsigned char *)feat_planes, feat_inv_scale, presplit);
    return check_launch("pos_tail_kernel");
}

extern "C" int phonlp_pos_head(const float *enc, int ld_enc, const int32_t *sentlens, int B, int L, int E,
                               const float *Whid, const float *bhid, const float *Wclf, const float *bclf,
                               const float *emb, int H, int U, int TE, float *feat, int32_t *pos_pred,
                               float *upos_logits, void *ws, size_t ws_bytes, void *stream) {
    if (!feat) return fail(PHONLP_ERR_SHAPE, "pos_head: bad argument");
    return pos_head_impl(enc, ld_enc, sentlens, B, L, E, Whid, bhid, Wclf, bclf, emb, H, U, TE, feat, pos_pred, upos_logits,
                         nullptr, nullptr, ws, ws_bytes, stream);
}

extern "C" int phonlp_pos_head_split(const float *enc, int ld_enc, const int32_t *sentlens, int B, int L, int E,
                                     const float *Whid, const float *bhid, const float *Wclf, const float *bclf,
                                     const float *emb, int H, int U, int TE, float *feat_opt, void *feat_planes,
                                     float *feat_inv_scale, int32_t *pos_pred, float *upos_logits, void *ws,
                                     size_t ws_bytes, void *stream) {
    if (!feat_planes) return fail(PHONLP_ERR_SHAPE, "pos_head_split: feat_planes is required");
    return pos_head_impl(enc, ld_enc, sentlens, B, L, E, Whid, bhid, Wclf, bclf, emb, H, U, TE, feat_opt, pos_pred,
                         upos_logits, feat_planes, feat_inv_scale, ws, ws_bytes, stream);
}

/* Fused tagger head: POS head + NER emissions with feat = [enc | pos_emb] never materialised. */
extern "C" size_t phonlp_tagger_head_ws_bytes(int B, int L, int E, int H, int U, int TE, int T) {
    (void)TE;
    const size_t M = (size_t)B * L, HN = (size_t)(H + T + 3) / 4 * 4;
    return align_up(M * HN * sizeof(float), 256) + align_up(M * U * sizeof(float), 256) +
           align_up(phonlp_split_bytes((int)M, E), 256) + align_up(M * sizeof(float), 256) +
           align_up(phonlp_split_bytes(H + T, E), 256) + align_up((size_t)(H + T) * sizeof(float), 256);
}

extern "C" int phonlp_tagger_head(const float *enc, int ld_enc, const int32_t *sentlens, int B, int L, int E,
                                  const float *Whn, const float *bhn, const float *Wclf, const float *bclf,
                                  const float *emb, const float *Wner_emb, const float *bner, int H, int U, int TE, int T,
                                  int32_t *pos_pred, float *ner_logits, float *upos_logits, void *ws, size_t ws_bytes,
                                  void *stream) {
    if (!enc || !sentlens || !Whn || !bhn || !Wclf || !bclf || !Wner_emb || !bner || !pos_pred || !ner_logits ||
        B <= 0 || L <= 0 || T <= 0)
        return fail(PHONLP_ERR_SHAPE, "tagger_head: bad argument");
    if (U > 64 || U <= 0) return fail(PHONLP_ERR_UNSUPPORTED, "tagger_head: needs 1 <= U <= 64 (one lane per tag)");
    if (!emb && (U > 32 || TE != U || (3 * 1024 + (size_t)U * T) * sizeof(float) > 64 * 1024))
        return fail(PHONLP_ERR_UNSUPPORTED, "tagger_head: emb == NULL (identity embedding, Wner_emb [T, U]) needs TE == U <= 32");
    if (!emb && (((uintptr_t)Wclf | (uintptr_t)ws) & 15))
        return fail(PHONLP_ERR_SHAPE, "tagger_head: emb == NULL needs Wclf and the workspace 16-byte aligned");
    if ((E & 3) || (H & 3) || (ld_enc & 3)) return fail(PHONLP_ERR_SHAPE, "tagger_head: E, H, ld_enc must be multiples of 4");
    if (!ws || ws_bytes < phonlp_tagger_head_ws_bytes(B, L, E, H, U, TE, T))
        return fail(PHONLP_ERR_WORKSPACE, "tagger_head: workspace too small");
    if ((size_t)4 * TE * sizeof(float) > 64 * 1024) return fail(PHONLP_ERR_UNSUPPORTED, "tagger_head: TE too large");
    hipStream_t st = as_stream(stream);
    const int M = B * L, N = H + T, HN = (N + 3) / 4 * 4;
    char *p = (char *)ws;
    float *hid = (float *)p;       p += align_up((size_t)M * HN * sizeof(float), 256);
    float *lg_ws = (float *)p;     p += align_up((size_t)M * U * sizeof(float), 256);
    float *logits = upos_logits ? upos_logits : lg_ws;
    int rc;
    // upos_hid and the enc part of ner_tag_clf as ONE GEMM: N = H + T output columns, ReLU on the first H only
    if (gemm_engine() == 2 && (E % 8) == 0) {
        void *encp = p;             p += align_up(phonlp_split_bytes(M, E), 256);
        float *enci = (float *)p;   p += align_up((size_t)M * sizeof(float), 256);
        void *whp = p;              p += align_up(phonlp_split_bytes(N, E), 256);
        float *whi = (float *)p;
        const int K16e = (E + 15) / 16 * 16;
        rc = launch_split_rows2(enc, ld_enc, M, encp, 0, enci, Whn, E, N, whp, whi, E, st);
        if (rc) return rc;
        rc = launch_linear_f16x3("linear:tagger_hid", encp, 4L * K16e, enci, whp, 4L * K16e, whi, bhn, hid, HN, M, N, K16e, -H,
                                 sentlens, L, st, nullptr);
    } else {
        rc = launch_linear("linear:upos_hid", enc, ld_enc, Whn, E, bhn, hid, HN, M, N, E, -H, sentlens, L, st);
    }
    if (rc) return rc;
    if (!emb) {  // identity embedding, Wner_emb [T, U]: upos_clf inside the tail
        ProfScope ps("tagger_tail", st);
        hipLaunchKernelGGL(tagger_clf_tail_kernel, dim3(ceil_div(M, 32)), dim3(256), (3 * 1024 + (size_t)U * T) * sizeof(float), st,
                           hid, (long)HN, Wclf, bclf, Wner_emb, bner, sentlens, M, L, H, U, T, pos_pred, upos_logits, ner_logits);
        return check_launch("tagger_clf_tail_kernel");
    }
    rc = launch_linear("linear:upos_clf", hid, HN, Wclf, H, bclf, logits, U, M, U, H, 0, sentlens, L, st);
    if (rc) return rc;
    ProfScope ps("tagger_tail", st);
    hipLaunchKernelGGL(tagger_tail_kernel, dim3(ceil_div(M, 4)), dim3(256), 4 * (size_t)(TE + 64) * sizeof(float), st, hid, (long)HN,
                       logits, emb, Wner_emb, bner, sentlens, M, L, H, U, TE, T, pos_pred, ner_logits);
    return check_launch("tagger_tail_kernel");
}

extern "C" int phonlp_ner_emission(const float *feat, int D, const int32_t *sentlens, int B, int L, const float *Wner,
                                   const float *bner, int T, float *logits, void *stream) {
    if (!feat || !sentlens || !Wner || !logits || B <= 0 || L <= 0 || T <= 0)
        return fail(PHONLP_ERR_SHAPE, "ner_emission: bad argument");
    return launch_linear("linear:ner_emission", feat, D, Wner, D, bner, logits, T, B * L, T, D, 1, sentlens, L, as_stream(stream));
}

extern "C" int phonlp_dep_hidden(const float *feat, int D, const int32_t *sentlens, int B, int L, const float *Wdep,
                                 const float *bdep, float *hidden, void *stream) {
    if (!feat || !sentlens || !Wdep || !hidden || B <= 0 || L <= 0)
        return fail(PHONLP_ERR_SHAPE, "dep_hidden: bad argument");
    return launch_linear("linear:dep_hid", feat, D, Wdep, D, bdep, hidden, D, B * L, D, D, 0, sentlens, L, as_stream(stream));
}
