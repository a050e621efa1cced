/* TEST INFRASTRUCTURE ONLY -- plain-C CPU oracle for the sequence/tree decoders on PhoNLP's
 * task-head hot path (Viterbi, CRF forward algorithm, Chu-Liu-Edmonds single root).
 *
 * A from-scratch restatement of the reference's algorithms (citations below are paths under
 * /root/reference/phonlp), used only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg as the checker for the HIP kernels.  The product never links or calls it.
 * Pinned against outputs of the reference's own code via tests/golden/{viterbi,crf,mst}.npz
 * (tests/test_oracle_c.py).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off -o oracle/_build/liboracle.so oracle/decoders.c -lm
 * (-ffp-contract=off: every arithmetic step of the reference is a separately rounded add.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG_INF (-INFINITY)

/* ---------------------------------------------------------------------------------------
 * Viterbi -- ref models/common/crf.py:103-126.  scores[l][T], trans[prev][cur]; float32
 * adds only; first-occurrence argmax.  tags_out[l]; returns best score via *best_out.
 * bp scratch is allocated here.  Returns 0, or -1 on allocation failure.
 */
int oracle_viterbi(const float *scores, const float *trans, int l, int T, int32_t *tags_out, float *best_out) {
    if (l <= 0) return 0;
    int32_t *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)l * T);
    float *prev = (float *)malloc(sizeof(float) * T * 2);
    if (!bp || !prev) { free(bp); free(prev); return -1; }
    float *cur = prev + T;
    memcpy(prev, scores, sizeof(float) * T);
    for (int t = 1; t < l; ++t) {
        for (int c = 0; c < T; ++c) {
            float best = prev[0] + trans[c];
            int arg = 0;
            for (int p = 1; p < T; ++p) {
                float v = prev[p] + trans[p * T + c];
                if (v > best) { best = v; arg = p; }
            }
            cur[c] = scores[(size_t)t * T + c] + best;
            bp[(size_t)t * T + c] = arg;
        }
        float *tmp = prev; prev = cur; cur = tmp;
    }
    int last = 0;
    for (int c = 1; c < T; ++c) if (prev[c] > prev[last]) last = c;
    if (best_out) *best_out = prev[last];
    tags_out[l - 1] = last;
    for (int t = l - 1; t >= 1; --t) { last = bp[(size_t)t * T + last]; tags_out[t - 1] = last; }
    free(bp);
    free(prev < cur ? prev : cur);
    return 0;
}

/* batch form: logits[B][L][T], sentlens[B]; tags[B][L] (entries >= len left untouched) */
int oracle_viterbi_batch(const float *logits, const float *trans, const int32_t *sentlens, int B, int L, int T,
                         int32_t *tags, float *best) {
    for (int b = 0; b < B; ++b) {
        int rc = oracle_viterbi(logits + (size_t)b * L * T, trans, sentlens[b], T, tags + (size_t)b * L,
                                best ? best + b : NULL);
        if (rc) return rc;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * CRF log-partition -- ref crf.py:79-100 with log_sum_exp :129-138.  inputs[B][L][T],
 * pad[B][L] (1 = padding: keep the previous alphas, :95-98), trans[prev][cur].
 * float32 arithmetic with expf/logf (the reference's torch kernels are not bit-reproducible
 * across libraries; compare within tolerance).
 */
int oracle_crf_log_norm(const float *inputs, const uint8_t *pad, const float *trans, int B, int L, int T,
                        float *log_norm) {
    float *alpha = (float *)malloc(sizeof(float) * T * 2);
    if (!alpha) return -1;
    float *nxt = alpha + T;
    for (int b = 0; b < B; ++b) {
        const float *x = inputs + (size_t)b * L * T;
        memcpy(alpha, x, sizeof(float) * T);
        for (int t = 1; t < L; ++t) {
            if (pad[(size_t)b * L + t]) continue;
            for (int c = 0; c < T; ++c) {
                float m = NEG_INF;
                for (int p = 0; p < T; ++p) { float v = alpha[p] + trans[p * T + c]; if (v > m) m = v; }
                float s = 0.f;
                for (int p = 0; p < T; ++p) s += expf((alpha[p] + trans[p * T + c]) - m);
                nxt[c] = x[(size_t)t * T + c] + (m + logf(s));
            }
            memcpy(alpha, nxt, sizeof(float) * T);
        }
        float m = NEG_INF;
        for (int c = 0; c < T; ++c) if (alpha[c] > m) m = alpha[c];
        float s = 0.f;
        for (int c = 0; c < T; ++c) s += expf(alpha[c] - m);
        log_norm[b] = m + logf(s);
    }
    free(alpha);
    return 0;
}

/* ---------------------------------------------------------------------------------------
 * Chu-Liu-Edmonds -- ref models/common/chuliu_edmonds.py.
 */

/* numpy's float64 add-reduction order (pairwise summation with an 8-way unrolled block of up
 * to 128 elements), needed because ref :75 ``cycle_scores.sum()`` and :158 ``tree_probs.sum()``
 * feed comparisons.  Verified against numpy 2.2 for n = 1..1000. */
static double np_sum(const double *a, int n) {
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

static int argmax_row(const double *row, int n) {
    int arg = 0;
    double best = row[0];
    for (int j = 1; j < n; ++j) if (row[j] > best) { best = row[j]; arg = j; }
    return arg;
}

/* The cycle that ref :67 ``cycles.pop()`` picks: tarjan (:6-49) lists cycles in ascending order
 * of their smallest member (a cycle of a head-pointer graph is only reachable from its own
 * members), so the last one has the largest minimum.  mark[v]=1 for its members; returns its
 * size (0 = no cycle of size > 1). */
static int last_listed_cycle(const int *head, int n, uint8_t *mark, int *path, uint8_t *state) {
    memset(state, 0, n);
    memset(mark, 0, n);
    int best_min = -1, best_start = -1;
    for (int s = 0; s < n; ++s) {
        int len = 0, v = s;
        while (state[v] == 0) { state[v] = 1; path[len++] = v; v = head[v]; }
        if (state[v] == 1) {
            int k = 0;
            while (path[k] != v) ++k;
            if (len - k > 1) {
                int mn = path[k];
                for (int q = k; q < len; ++q) if (path[q] < mn) mn = path[q];
                if (mn > best_min) { best_min = mn; best_start = v; }
            }
        }
        for (int q = 0; q < len; ++q) state[path[q]] = 2;
    }
    if (best_start < 0) return 0;
    int cnt = 0, v = best_start;
    do { mark[v] = 1; ++cnt; v = head[v]; } while (v != best_start);
    return cnt;
}

/* ref :52-129.  s: n x n row-major (rows = dependents, cols = heads), edited in place at this
 * level like the reference (:55-57).  tree[n] out.  Returns 0 / -1 (allocation). */
static int cle(double *s, int n, int *tree) {
    for (int i = 0; i < n; ++i) s[(size_t)i * n + i] = NEG_INF;
    for (int j = 0; j < n; ++j) s[j] = NEG_INF;
    s[0] = 0.0;
    for (int i = 0; i < n; ++i) tree[i] = argmax_row(s + (size_t)i * n, n);

    uint8_t *mark = (uint8_t *)malloc(2 * (size_t)n);
    int *path = (int *)malloc(sizeof(int) * (size_t)n);
    if (!mark || !path) { free(mark); free(path); return -1; }
    int c = last_listed_cycle(tree, n, mark, path, mark + n);
    free(path);
    if (c == 0) { free(mark); return 0; }

    int k = n - c;
    int *cyc = (int *)malloc(sizeof(int) * (size_t)(n + 3 * (k + 1)));
    double *edge = (double *)malloc(sizeof(double) * (size_t)c);
    double *sub = (double *)calloc((size_t)(k + 1) * (k + 1), sizeof(double));
    if (!cyc || !edge || !sub) { free(mark); free(cyc); free(edge); free(sub); return -1; }
    int *rest = cyc + c, *enter_pick = rest + k, *leave_pick = enter_pick + (k + 1), *small = leave_pick + (k + 1);
    int a = 0, r = 0;
    for (int v = 0; v < n; ++v) { if (mark[v]) cyc[a++] = v; else rest[r++] = v; }
    for (int q = 0; q < c; ++q) edge[q] = s[(size_t)cyc[q] * n + tree[cyc[q]]];
    double total = np_sum(edge, c);
    for (int x = 0; x < k; ++x) {
        for (int y = 0; y < k; ++y) sub[(size_t)x * (k + 1) + y] = s[(size_t)rest[x] * n + rest[y]];
        /* ref :84,:88,:97 -- best way INTO the cycle from outside head rest[x] */
        double best = (s[(size_t)cyc[0] * n + rest[x]] - edge[0]) + total;
        int arg = 0;
        for (int q = 1; q < c; ++q) {
            double v = (s[(size_t)cyc[q] * n + rest[x]] - edge[q]) + total;
            if (v > best) { best = v; arg = q; }
        }
        enter_pick[x] = arg;
        sub[(size_t)k * (k + 1) + x] = best;
        /* ref :86,:90,:99 -- best head INSIDE the cycle for outside dependent rest[x] */
        best = s[(size_t)rest[x] * n + cyc[0]];
        arg = 0;
        for (int q = 1; q < c; ++q) {
            double v = s[(size_t)rest[x] * n + cyc[q]];
            if (v > best) { best = v; arg = q; }
        }
        leave_pick[x] = arg;
        sub[(size_t)x * (k + 1) + k] = best;
    }
    int rc = cle(sub, k + 1, small);
    if (rc == 0) {
        int meta_head = small[k];
        int *old = (int *)malloc(sizeof(int) * (size_t)n);
        if (!old) rc = -1;
        else {
            memcpy(old, tree, sizeof(int) * (size_t)n);
            for (int x = 0; x < k; ++x)
                tree[rest[x]] = small[x] < k ? rest[small[x]] : cyc[leave_pick[x]];
            for (int q = 0; q < c; ++q) tree[cyc[q]] = old[cyc[q]];
            tree[cyc[enter_pick[meta_head]]] = rest[meta_head];
            free(old);
        }
    }
    free(mark); free(cyc); free(edge); free(sub);
    return rc;
}

/* ref :133-169.  scores: l x l float32 with row stride ld (the caller's adj[:l,:l] slice).
 * heads[l] out (heads[0] = 0 is ROOT's self loop).  Returns 0 ok, 1 = no finite single-root
 * tree (the reference hits its broken assert there), -1 allocation failure. */
int oracle_mst_one_root(const float *scores, int ld, int l, int32_t *heads) {
    if (l <= 0) return 0;
    size_t nn = (size_t)l * l;
    double *work = (double *)malloc(sizeof(double) * nn * 2);
    int *tree = (int *)malloc(sizeof(int) * (size_t)l * 3);
    double *picked = (double *)malloc(sizeof(double) * (size_t)l);
    if (!work || !tree || !picked) { free(work); free(tree); free(picked); return -1; }
    double *trial = work + nn;
    int *cand = tree + l, *best_tree = cand + l;
    for (int i = 0; i < l; ++i)
        for (int j = 0; j < l; ++j) work[(size_t)i * l + j] = (double)scores[(size_t)i * ld + j];
    int rc = cle(work, l, tree);
    int nroots = 0;
    for (int i = 1; i < l && rc == 0; ++i) if (tree[i] == 0) ++nroots;
    if (rc == 0 && nroots != 1) {
        double best_score = NEG_INF;
        int have = 0;
        for (int root = 1; root < l && rc == 0; ++root) {
            if (tree[root] != 0) continue;
            memcpy(trial, work, sizeof(double) * nn);
            double root_score = trial[(size_t)root * l];
            for (int i = 1; i < l; ++i) trial[(size_t)i * l] = NEG_INF;
            for (int j = 0; j < l; ++j) trial[(size_t)root * l + j] = NEG_INF;
            trial[(size_t)root * l] = 0.0;
            rc = cle(trial, l, cand);
            if (rc) break;
            int finite = 1;
            for (int i = 0; i < l; ++i) {
                picked[i] = trial[(size_t)i * l + cand[i]];
                if (!(picked[i] > NEG_INF)) finite = 0;
            }
            double score = finite ? np_sum(picked, l) + root_score : NEG_INF;
            if (score > best_score) { best_score = score; have = 1; memcpy(best_tree, cand, sizeof(int) * (size_t)l); }
        }
        if (rc == 0) {
            if (!have) rc = 1;
            else memcpy(tree, best_tree, sizeof(int) * (size_t)l);
        }
    }
    if (rc == 0) for (int i = 0; i < l; ++i) heads[i] = tree[i];
    free(work); free(tree); free(picked);
    return rc;
}

/* batch form over arc[B][L][L] with per-sentence lengths; heads[B][L]; status[B] */
int oracle_mst_batch(const float *arc, const int32_t *sentlens, int B, int L, int32_t *heads, int32_t *status) {
    int worst = 0;
    for (int b = 0; b < B; ++b) {
        int rc = oracle_mst_one_root(arc + (size_t)b * L * L, L, sentlens[b], heads + (size_t)b * L);
        if (status) status[b] = rc;
        if (rc < 0) return rc;
        if (rc > worst) worst = rc;
    }
    return worst;
}
