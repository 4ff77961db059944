/*
 * vl_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See vl_oracle.h for the role of this file and how its parity is pinned.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).  No
 * -march: the reference's f64 loops are separate multiply and add, strictly
 * in index order (rustc never contracts or reassociates without fast-math),
 * so FMA contraction and vector reassociation must stay off here too.
 */
#include "vl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * Distance math (src/lib.rs:380-572)
 * ---------------------------------------------------------------------- */

/* src/lib.rs:425-444: one pass, three accumulators starting at 0.0, then
 * dot / (sqrt(na) * sqrt(nb)); 0.0 if either norm is exactly 0. */
double vlo_cosine(const double *a, const double *b, size_t n)
{
    double dot = 0.0, na = 0.0, nb = 0.0;
    for (size_t i = 0; i < n; ++i) {
        double x = a[i], y = b[i];
        dot += x * y;
        na += x * x;
        nb += y * y;
    }
    double norm_a = sqrt(na), norm_b = sqrt(nb);
    if (norm_a == 0.0 || norm_b == 0.0)
        return 0.0;
    return dot / (norm_a * norm_b);
}

/* `.sum::<f64>()` folds from the additive identity -0.0 (core::iter Sum for
 * floats, Rust >= 1.83; the crate is edition 2024, i.e. rustc >= 1.85).  The
 * only observable effect is the sign of an all-(-0.0) sum. */
#define VLO_SUM_IDENTITY (-0.0)

/* src/lib.rs:476-489: (x - y).powi(2) summed in order; 1/(1+sqrt(s)). */
double vlo_euclidean(const double *a, const double *b, size_t n)
{
    double s = VLO_SUM_IDENTITY;
    for (size_t i = 0; i < n; ++i) {
        double d = a[i] - b[i];
        s += d * d; /* powi(2) lowers to one multiply */
    }
    return 1.0 / (1.0 + sqrt(s));
}

/* src/lib.rs:521-532: |x - y| summed in order; 1/(1+s). */
double vlo_manhattan(const double *a, const double *b, size_t n)
{
    double s = VLO_SUM_IDENTITY;
    for (size_t i = 0; i < n; ++i)
        s += fabs(a[i] - b[i]);
    return 1.0 / (1.0 + s);
}

/* src/lib.rs:565-572: x*y summed in order, returned unbounded. */
double vlo_dot(const double *a, const double *b, size_t n)
{
    double s = VLO_SUM_IDENTITY;
    for (size_t i = 0; i < n; ++i)
        s += a[i] * b[i];
    return s;
}

/* src/lib.rs:380-391 */
double vlo_calculate(int metric, const double *a, const double *b, size_t n)
{
    switch (metric) {
    case VLO_COSINE: return vlo_cosine(a, b, n);
    case VLO_EUCLIDEAN: return vlo_euclidean(a, b, n);
    case VLO_MANHATTAN: return vlo_manhattan(a, b, n);
    default: return vlo_dot(a, b, n);
    }
}

/* ------------------------------------------------------------------------
 * Ingest step (src/embeddings.rs:169-181): what EmbeddingGenerator does to the
 * model's f32 output before the row reaches FlatIndex::add
 * ---------------------------------------------------------------------- */
void vlo_embed_f32(const float *emb, size_t n, int normalize, double *out)
{
    double s = VLO_SUM_IDENTITY;
    for (size_t i = 0; i < n; ++i) {
        out[i] = (double)emb[i]; /* :172 `x as f64` */
        s += out[i] * out[i];    /* :175 map(x * x).sum() */
    }
    const double norm = sqrt(s);
    if (normalize && norm > 0.0) /* :176-180 */
        for (size_t i = 0; i < n; ++i)
            out[i] = out[i] / norm;
}

/* ------------------------------------------------------------------------
 * HNSW boundary (src/index/hnsw.rs)
 * ---------------------------------------------------------------------- */

/* Rust `f64 as u64`: truncate toward zero, saturating; NaN -> 0. */
static uint64_t rust_f64_as_u64(double v)
{
    if (!(v > 0.0)) /* negative, zero, NaN */
        return 0;
    if (v >= 18446744073709551616.0)
        return UINT64_MAX;
    return (uint64_t)v;
}

/* f64::clamp(min, max): NaN stays NaN. */
static double rust_clamp(double v, double lo, double hi)
{
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}

/* src/index/hnsw.rs:113-174 */
uint64_t vlo_hnsw_distance(int metric, const double *a, const double *b, size_t n)
{
    switch (metric) {
    case VLO_EUCLIDEAN: { /* :116-122 */
        double s = VLO_SUM_IDENTITY;
        for (size_t i = 0; i < n; ++i) {
            double d = a[i] - b[i];
            s += d * d;
        }
        return rust_f64_as_u64(sqrt(s) * 1000.0);
    }
    case VLO_COSINE: { /* :128-147: fold from (0.0, 0.0, 0.0) */
        double dot = 0.0, na = 0.0, nb = 0.0;
        for (size_t i = 0; i < n; ++i) {
            double x = a[i], y = b[i];
            dot = dot + x * y;
            na = na + x * x;
            nb = nb + y * y;
        }
        double norm_a = sqrt(na), norm_b = sqrt(nb);
        if (norm_a == 0.0 || norm_b == 0.0)
            return 1000;
        double cosine_sim = dot / (norm_a * norm_b);
        return rust_f64_as_u64((1.0 - cosine_sim) * 1000.0);
    }
    case VLO_MANHATTAN: { /* :153-159 */
        double s = VLO_SUM_IDENTITY;
        for (size_t i = 0; i < n; ++i)
            s += fabs(a[i] - b[i]);
        return rust_f64_as_u64(s * 1000.0);
    }
    default: { /* DotProduct :165-173 */
        double s = VLO_SUM_IDENTITY;
        for (size_t i = 0; i < n; ++i)
            s += a[i] * b[i];
        return rust_f64_as_u64(1000.0 - rust_clamp(s, -1000.0, 1000.0));
    }
    }
}

/* src/index/hnsw.rs:51-75 */
double vlo_convert_distance_to_similarity(double distance, int metric)
{
    switch (metric) {
    case VLO_EUCLIDEAN: return 1.0 / (1.0 + distance);
    case VLO_COSINE: {
        double cos_distance = distance / 1000.0;
        return 1.0 - cos_distance;
    }
    case VLO_MANHATTAN: return 1.0 / (1.0 + distance);
    default: return rust_clamp((1000.0 - distance) / 1000.0, 0.0, 1.0);
    }
}

/* src/index/hnsw.rs:478-479 */
double vlo_hnsw_score(uint64_t d, int metric)
{
    double distance = (double)d / 1000.0;
    return vlo_convert_distance_to_similarity(distance, metric);
}

/* ------------------------------------------------------------------------
 * Stable descending sort of (score, payload) records:
 *   sort_by(|a, b| b.score.partial_cmp(&a.score).unwrap())
 * (src/index/flat.rs:116, src/index/hnsw.rs:493).  `a` stays in front of `b`
 * unless b.score > a.score; equal scores keep storage order.
 * ---------------------------------------------------------------------- */
typedef struct {
    uint64_t id;
    double score;
    char *text;     /* e.text.clone(): one heap block per materialised result */
    void *metadata; /* Option<Value>::None */
} vlo_result;

static void merge_sort_desc(vlo_result *v, vlo_result *tmp, size_t n)
{
    for (size_t w = 1; w < n; w *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * w) {
            size_t mid = lo + w < n ? lo + w : n;
            size_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                /* take the right element only if it is strictly greater */
                if (v[j].score > v[i].score) tmp[o++] = v[j++];
                else tmp[o++] = v[i++];
            }
            while (i < mid) tmp[o++] = v[i++];
            while (j < hi) tmp[o++] = v[j++];
        }
        memcpy(v, tmp, n * sizeof(vlo_result));
    }
}

/* ------------------------------------------------------------------------
 * FlatIndex (src/index/flat.rs:60-135), reference-faithful memory layout:
 * Vec<Vector> where each Vector owns a separate heap block for `values`
 * and another for `text` (src/lib.rs:164-174).
 * ---------------------------------------------------------------------- */
typedef struct {
    uint64_t id;
    double *values;
    size_t len;
    char *text;
} vlo_vector;

struct vlo_flat {
    size_t dim;
    vlo_vector *data;
    size_t n, cap;
};

static int push_row(vlo_flat *f, uint64_t id, const double *values, size_t len)
{
    if (f->n == f->cap) {
        size_t cap = f->cap ? f->cap * 2 : 16;
        vlo_vector *d = (vlo_vector *)realloc(f->data, cap * sizeof(vlo_vector));
        if (!d) return -1;
        f->data = d;
        f->cap = cap;
    }
    vlo_vector *r = &f->data[f->n];
    r->id = id;
    r->len = len;
    r->values = (double *)malloc((len ? len : 1) * sizeof(double));
    r->text = (char *)malloc(24);
    if (!r->values || !r->text) return -1;
    memcpy(r->values, values, len * sizeof(double));
    memcpy(r->text, "synthetic row text....\0", 24);
    f->n++;
    return 0;
}

vlo_flat *vlo_flat_new(size_t dim, const uint64_t *ids, const double *values, size_t n)
{
    vlo_flat *f = (vlo_flat *)calloc(1, sizeof(vlo_flat));
    if (!f) return NULL;
    f->dim = dim;
    for (size_t i = 0; i < n; ++i) {
        if (push_row(f, ids[i], values + i * dim, dim) != 0) {
            vlo_flat_free(f);
            return NULL;
        }
    }
    return f;
}

/* More rows appended as FlatIndex::new would hold them (no validation, src/index/flat.rs:68-73): lets a caller build
 * a corpus too large to stage twice in host memory chunk by chunk (bench.py's full-size CPU query). */
int vlo_flat_extend(vlo_flat *f, const uint64_t *ids, const double *values, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (push_row(f, ids[i], values + i * f->dim, f->dim) != 0) return -1;
    return VLO_OK;
}

void vlo_flat_free(vlo_flat *f)
{
    if (!f) return;
    for (size_t i = 0; i < f->n; ++i) {
        free(f->data[i].values);
        free(f->data[i].text);
    }
    free(f->data);
    free(f);
}

/* src/index/flat.rs:82-91 */
int vlo_flat_add(vlo_flat *f, uint64_t id, const double *values, size_t len)
{
    if (len != f->dim) return VLO_DIM_MISMATCH;
    for (size_t i = 0; i < f->n; ++i)
        if (f->data[i].id == id) return VLO_DUP_ID;
    return push_row(f, id, values, len) == 0 ? VLO_OK : -1;
}

/* src/index/flat.rs:93-96: retain(|e| e.id != id); always Ok. */
int vlo_flat_delete(vlo_flat *f, uint64_t id)
{
    size_t o = 0;
    for (size_t i = 0; i < f->n; ++i) {
        if (f->data[i].id == id) {
            free(f->data[i].values);
            free(f->data[i].text);
        } else {
            f->data[o++] = f->data[i];
        }
    }
    f->n = o;
    return VLO_OK;
}

size_t vlo_flat_len(const vlo_flat *f) { return f->n; }
size_t vlo_flat_dim(const vlo_flat *f) { return f->dim; }

/* src/index/flat.rs:129-131: first row with that id. */
int vlo_flat_get(const vlo_flat *f, uint64_t id, double *out)
{
    for (size_t i = 0; i < f->n; ++i) {
        if (f->data[i].id == id) {
            memcpy(out, f->data[i].values, f->data[i].len * sizeof(double));
            return VLO_OK;
        }
    }
    return VLO_NOT_FOUND;
}

/* src/index/flat.rs:98-119 */
int vlo_flat_search(const vlo_flat *f, const double *q, size_t q_len, size_t k, int metric,
                    uint64_t *out_ids, double *out_scores, size_t *out_n)
{
    *out_n = 0;
    if (f->n != 0 && q_len != f->dim) { /* :99-104 */
        *out_n = f->dim;
        return VLO_DIM_MISMATCH;
    }
    size_t n = f->n;
    if (n == 0) return VLO_OK;

    /* :106-114  one SearchResult per stored row, text cloned */
    vlo_result *res = (vlo_result *)malloc(n * sizeof(vlo_result));
    vlo_result *tmp = (vlo_result *)malloc(n * sizeof(vlo_result));
    if (!res || !tmp) {
        free(res);
        free(tmp);
        return -1;
    }
    int has_nan = 0;
    for (size_t i = 0; i < n; ++i) {
        const vlo_vector *e = &f->data[i];
        res[i].id = e->id;
        res[i].score = vlo_calculate(metric, e->values, q, q_len);
        res[i].text = (char *)malloc(24);
        if (res[i].text) memcpy(res[i].text, e->text, 24);
        res[i].metadata = NULL;
        if (res[i].score != res[i].score) has_nan = 1;
    }
    int rc = VLO_OK;
    if (has_nan && n >= 2) {
        rc = VLO_NAN_PANIC; /* :116 partial_cmp(..).unwrap() on a NaN */
    } else {
        merge_sort_desc(res, tmp, n); /* :116 */
        size_t m = k < n ? k : n;     /* :117 */
        for (size_t i = 0; i < m; ++i) {
            out_ids[i] = res[i].id;
            out_scores[i] = res[i].score;
        }
        *out_n = m;
    }
    for (size_t i = 0; i < n; ++i) free(res[i].text);
    free(res);
    free(tmp);
    return rc;
}

/* src/index/hnsw.rs:468-495 applied to neighbours a graph walk returned. */
size_t vlo_hnsw_postprocess(uint64_t *ids, const uint64_t *dists, double *scores, size_t n,
                            size_t k, int metric)
{
    if (n == 0) return 0;
    vlo_result *res = (vlo_result *)malloc(n * sizeof(vlo_result));
    vlo_result *tmp = (vlo_result *)malloc(n * sizeof(vlo_result));
    for (size_t i = 0; i < n; ++i) {
        res[i].id = ids[i];
        res[i].score = vlo_hnsw_score(dists[i], metric);
        res[i].text = NULL;
        res[i].metadata = NULL;
    }
    merge_sort_desc(res, tmp, n);
    size_t m = k < n ? k : n;
    for (size_t i = 0; i < m; ++i) {
        ids[i] = res[i].id;
        scores[i] = res[i].score;
    }
    free(res);
    free(tmp);
    return m;
}
