/* A plain-C caller that sizes its result buffers from an EARLIER vl_index_len() while another thread keeps adding
 * rows (the reference would hold RwLock::write for add, src/client.rs:333; a C caller has no such lock).  With
 * vl_index_search the number of entries written follows the index length at search time; vl_index_search_cap bounds
 * it by the caller's own capacity (truncate(k), src/index/flat.rs:117, once more).  Canary words behind the buffers
 * must survive, every answer must be a prefix of the full answer's order (scores descending).
 * The multi-GPU handle (two row shards on one card) goes through the same loop. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "vectorlite_amd.h"

#define DIM 16
#define CANARY 0xC0FFEE0DDEADBEEFull

static vl_index *g_idx;
static volatile int g_stop;

static void fill(double *v, unsigned long long seed)
{
    for (int i = 0; i < DIM; ++i) {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        v[i] = (double)((seed >> 33) % 2001) / 1000.0 - 1.0;
    }
}

static void *grower(void *arg)
{
    (void)arg;
    double v[DIM];
    unsigned long long id = 1000000;
    while (!g_stop) {
        fill(v, id);
        if (vl_index_add(g_idx, id, v, DIM) != VL_OK) {
            fprintf(stderr, "add failed: %s\n", vl_last_error());
            exit(3);
        }
        ++id;
    }
    return NULL;
}

static int run(vl_index *idx, const char *what)
{
    g_idx = idx;
    g_stop = 0;
    double v[DIM], q[DIM];
    for (unsigned long long i = 0; i < 5; ++i) {
        fill(v, i);
        if (vl_index_add(idx, i, v, DIM) != VL_OK) return 1;
    }
    pthread_t th;
    if (pthread_create(&th, NULL, grower, NULL) != 0) return 1;
    int bad = 0;
    uint64_t most = 0;
    for (int it = 0; it < 300 && !bad; ++it) {
        const uint64_t cap = vl_index_len(idx); /* the index keeps growing after this line */
        uint64_t *ids = (uint64_t *)malloc((cap + 1) * sizeof(uint64_t));
        double *scores = (double *)malloc((cap + 1) * sizeof(double));
        ids[cap] = CANARY;
        memcpy(&scores[cap], &ids[cap], sizeof(double));
        fill(q, 777 + (unsigned long long)it);
        uint64_t n = 0;
        const int rc = vl_index_search_cap(idx, q, DIM, (uint64_t)1 << 40, VL_DOTPRODUCT, cap, ids, scores, &n);
        uint64_t tail;
        memcpy(&tail, &scores[cap], sizeof tail);
        if (rc != VL_OK || n > cap || ids[cap] != CANARY || tail != CANARY) {
            fprintf(stderr, "%s: rc %d n %llu cap %llu canaries %d %d: %s\n", what, rc, (unsigned long long)n,
                    (unsigned long long)cap, ids[cap] == CANARY, tail == CANARY, vl_last_error());
            bad = 1;
        }
        for (uint64_t j = 1; j < n && !bad; ++j)
            if (scores[j] > scores[j - 1]) bad = 1;
        if (n > most) most = n;
        free(ids);
        free(scores);
    }
    g_stop = 1;
    pthread_join(th, NULL);
    printf("%s: %llu rows at the end, largest answer %llu, %s\n", what, (unsigned long long)vl_index_len(idx),
           (unsigned long long)most, bad ? "BAD" : "fine");
    return bad;
}

int main(void)
{
    vl_index *one = NULL, *two = NULL;
    const int devs[2] = {0, 0};
    if (vl_flat_create(DIM, 0, &one) != VL_OK) return 2;
    if (vl_flat_create_multi(DIM, devs, 2, VL_MULTI_ROW_SHARDS, &two) != VL_OK) return 2;
    int bad = run(one, "single-GPU handle");
    bad |= run(two, "two row shards");
    vl_index_destroy(one);
    vl_index_destroy(two);
    puts(bad ? "FAILED" : "ok");
    return bad;
}
