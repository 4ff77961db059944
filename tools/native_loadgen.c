/* tools/native_loadgen.c -- a closed loop of T native threads calling vl_index_search_cap on ONE handle: the reference's
 * many-readers usage (tokio workers under RwLock::read, src/client.rs:398) without Python's GIL between a caller's return
 * and its next call.  Built by tools/concurrent_native.py (gcc -shared); the search entry point arrives as a function
 * pointer, so this file links against nothing.  Each thread answers per_thread queries (its own slice of Q) one after
 * the other; per-query wall time goes to lat_s, the ids of each thread's FIRST query to first_ids (checked by the caller
 * against the lone search). */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

typedef int (*search_cap_fn)(const void *, const double *, uint64_t, uint64_t, int, uint64_t, uint64_t *, double *, uint64_t *);

typedef struct {
    search_cap_fn fn;
    const void *h;
    const double *q;
    uint64_t dim, k;
    int metric, per_thread, t;
    double *lat_s;
    uint64_t *first_ids;
    double *first_scores;
    pthread_barrier_t *bar;
    int rc;
} job_t;

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    uint64_t *ids = (uint64_t *)malloc(j->k * sizeof(uint64_t));
    double *sc = (double *)malloc(j->k * sizeof(double));
    pthread_barrier_wait(j->bar);
    for (int i = 0; i < j->per_thread && j->rc == 0; ++i) {
        const uint64_t qi = (uint64_t)j->t * (uint64_t)j->per_thread + (uint64_t)i;
        uint64_t n = 0;
        const double t0 = now_s();
        j->rc = j->fn(j->h, j->q + qi * j->dim, j->dim, j->k, j->metric, j->k, ids, sc, &n);
        j->lat_s[qi] = now_s() - t0;
        if (i == 0)
            for (uint64_t c = 0; c < j->k; ++c) {
                j->first_ids[(uint64_t)j->t * j->k + c] = c < n ? ids[c] : ~0ull;
                j->first_scores[(uint64_t)j->t * j->k + c] = c < n ? sc[c] : 0.0;
            }
    }
    free(ids);
    free(sc);
    return NULL;
}

/* returns 0 or the first failing status; *elapsed_s = barrier release .. last thread done */
int vl_loadgen(void *fn, const void *h, const double *q, uint64_t dim, uint64_t k, int metric, int threads, int per_thread,
               double *lat_s, uint64_t *first_ids, double *first_scores, double *elapsed_s)
{
    pthread_t *th = (pthread_t *)malloc((size_t)threads * sizeof(pthread_t));
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_barrier_t bar;
    pthread_barrier_init(&bar, NULL, (unsigned)threads + 1u);
    for (int t = 0; t < threads; ++t) {
        jobs[t] = (job_t){(search_cap_fn)fn, h, q, dim, k, metric, per_thread, t, lat_s, first_ids, first_scores, &bar, 0};
        pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    pthread_barrier_wait(&bar);
    const double t0 = now_s();
    int rc = 0;
    for (int t = 0; t < threads; ++t) {
        pthread_join(th[t], NULL);
        if (jobs[t].rc != 0 && rc == 0) rc = jobs[t].rc;
    }
    *elapsed_s = now_s() - t0;
    pthread_barrier_destroy(&bar);
    free(th);
    free(jobs);
    return rc;
}
