/* vl_fair.c -- "fair CPU" baseline for bench.py's cpu_baseline leg (mode ii).
 * TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE: vectorlite_amd never links or loads it.
 *
 * What a well-written CPU flat scan would do with the same data the GPU path holds: contiguous f32
 * slab, cached inverse norms (cosine = dot * inv_norm(row) * inv_norm(query)), SIMD-friendly
 * accumulation, a small per-thread top-k, OpenMP over every host core.  It is NOT a restatement of
 * the reference (src/index/flat.rs:98-119 walks AoS f64 rows one by one on one thread and sorts N
 * records -- that is oracle/vl_oracle.c); it exists so the GPU/CPU ratio is not inflated by the
 * reference's allocation pattern.  Scores are f32-accumulated: bench.py reports its recall against
 * the exact answer beside the rate.
 *
 * Built by bench.py on the machine it runs on:  gcc -O3 -march=native -fopenmp -shared -fPIC
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <omp.h>

#define LANES 64

static inline float dot_f32(const float* restrict a, const float* restrict b, size_t dim)
{
    float acc[LANES] = {0};
    size_t i = 0;
    for (; i + LANES <= dim; i += LANES)
        for (int l = 0; l < LANES; ++l) acc[l] += a[i + l] * b[i + l];
    float s = 0.f;
    for (int l = 0; l < LANES; ++l) s += acc[l];
    for (; i < dim; ++i) s += a[i] * b[i];
    return s;
}

int vlf_threads(void) { return omp_get_max_threads(); }
void vlf_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

/* inv_norm[i] = 1/|row i| (0 for a zero row, so its cosine is 0.0 like src/lib.rs:439-440) */
void vlf_inv_norms(const float* slab, size_t n, size_t dim, float* inv_norm)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const float s = dot_f32(slab + i * dim, slab + i * dim, dim);
        inv_norm[i] = s > 0.f ? 1.0f / sqrtf(s) : 0.f;
    }
}

typedef struct { float score; uint32_t pos; } ent;

static inline int ent_better(ent a, ent b) { return a.score > b.score || (a.score == b.score && a.pos < b.pos); }

/* cosine top-k of one query over the slab; out sorted by (score desc, pos asc); returns min(k, n) */
size_t vlf_search_cosine(const float* slab, const float* inv_norm, size_t n, size_t dim, const float* q, size_t k,
                         uint32_t* out_pos, float* out_score)
{
    if (k > n) k = n;
    if (k == 0) return 0;
    const int nt = omp_get_max_threads();
    ent* all = (ent*)malloc(sizeof(ent) * k * (size_t)nt);
    size_t* cnt = (size_t*)calloc((size_t)nt, sizeof(size_t));
    const float qs = dot_f32(q, q, dim);
    const float qinv = qs > 0.f ? 1.0f / sqrtf(qs) : 0.f;
#pragma omp parallel
    {
        const int t = omp_get_thread_num();
        ent* top = all + (size_t)t * k; /* sorted, best first */
        size_t m = 0;
#pragma omp for schedule(static)
        for (size_t i = 0; i < n; ++i) {
            ent e = {dot_f32(slab + i * dim, q, dim) * inv_norm[i] * qinv, (uint32_t)i};
            if (m == k && !ent_better(e, top[k - 1])) continue;
            size_t j = m < k ? m++ : k - 1;
            while (j > 0 && ent_better(e, top[j - 1])) { top[j] = top[j - 1]; --j; }
            top[j] = e;
        }
        cnt[t] = m;
    }
    size_t got = 0;
    for (size_t r = 0; r < k; ++r) { /* k-way pick over the per-thread sorted lists */
        int bt = -1;
        for (int t = 0; t < nt; ++t)
            if (cnt[t] && (bt < 0 || ent_better(all[(size_t)t * k], all[(size_t)bt * k]))) bt = t;
        if (bt < 0) break;
        ent* top = all + (size_t)bt * k;
        out_pos[got] = top[0].pos;
        out_score[got] = top[0].score;
        ++got;
        for (size_t j = 1; j < cnt[bt]; ++j) top[j - 1] = top[j];
        --cnt[bt];
    }
    free(all);
    free(cnt);
    return got;
}
