/* Plain-C caller of libvectorlite_amd.so: the reference's own flat-search test
 * (src/index/flat.rs:187-201: unit basis vectors, q = [1,0,0], k = 2 -> first id 1, score 1)
 * followed by a batched search and an HNSW round trip.  Doubles as the check that
 * include/vectorlite_amd.h is valid C99 (tests/test_cabi_cpu.py compiles it with gcc -std=c99 -pedantic).
 *
 *   gcc -std=c99 -I include integration/c/example.c -L vectorlite_amd -lvectorlite_amd -Wl,-rpath,$PWD/vectorlite_amd -o example
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "vectorlite_amd.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != VL_OK) {                                                                \
            fprintf(stderr, "%s -> status %d: %s\n", #call, rc_, vl_last_error());        \
            return 1;                                                                      \
        }                                                                                  \
    } while (0)

int main(void)
{
    int n_dev = 0, abi = 0;
    vl_runtime_info(&n_dev, &abi);
    printf("devices %d, abi %d\n", n_dev, abi);
    if (n_dev == 0) {
        fprintf(stderr, "no HIP device: this library has no CPU fallback\n");
        return 2;
    }

    vl_index *idx = NULL;
    CHECK(vl_flat_create(3, 0, &idx));
    const double rows[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (uint64_t i = 0; i < 3; ++i) CHECK(vl_index_add(idx, i + 1, rows[i], 3));
    if (vl_index_add(idx, 1, rows[0], 3) != VL_ERR_DUP_ID) return 1;          /* "Vector ID 1 already exists" */
    if (vl_index_add(idx, 9, rows[0], 2) != VL_ERR_DIM_MISMATCH) return 1;    /* "Vector dimension mismatch" */

    const double q[3] = {1, 0, 0};
    uint64_t ids[2], n = 0;
    double scores[2];
    CHECK(vl_index_search(idx, q, 3, 2, VL_COSINE, ids, scores, &n));
    printf("flat: n=%llu first id=%llu score=%.17g\n", (unsigned long long)n, (unsigned long long)ids[0], scores[0]);
    if (n != 2 || ids[0] != 1 || fabs(scores[0] - 1.0) > 1e-10) return 1;

    uint64_t e = 0, a = 0;
    if (vl_index_search(idx, q, 2, 2, VL_COSINE, ids, scores, &n) != VL_ERR_DIM_MISMATCH) return 1;
    vl_last_dim_mismatch(&e, &a);
    if (e != 3 || a != 2) return 1;

    const double qs[2][3] = {{0, 1, 0}, {0.1, 0.1, 1.1}};
    uint64_t bids[2 * 2], bn[2];
    double bscores[2 * 2];
    CHECK(vl_index_search_batch(idx, &qs[0][0], 2, 3, 2, VL_EUCLIDEAN, bids, bscores, bn));
    printf("batch: first ids %llu %llu\n", (unsigned long long)bids[0], (unsigned long long)bids[2]);
    if (bids[0] != 2 || bids[2] != 3) return 1;
    CHECK(vl_index_delete(idx, 2));
    CHECK(vl_index_delete(idx, 2)); /* absent id: still Ok for the flat index (src/index/flat.rs:93-96) */
    if (vl_index_len(idx) != 2) return 1;
    vl_index_destroy(idx);

    /* the ingest step in front of add: f32 model output, widened and L2-normalised on the device exactly as
     * src/embeddings.rs:169-181 does on the host ({3,4} -> {0.6,0.8}; a zero row stays zero) */
    vl_index *emb = NULL;
    CHECK(vl_flat_create(2, 0, &emb));
    const float model_out[3][2] = {{3.f, 4.f}, {0.f, 0.f}, {-5.f, 12.f}};
    const uint64_t emb_ids[3] = {7, 8, 9};
    CHECK(vl_index_add_embeddings_f32(emb, emb_ids, &model_out[0][0], 3, /*normalize=*/1, /*validate=*/1, /*on_device=*/0));
    double stored[2];
    CHECK(vl_index_get_vector(emb, 7, stored));
    printf("embedding: stored row of id 7 = {%.17g, %.17g}\n", stored[0], stored[1]);
    if (stored[0] != 3.0 / 5.0 || stored[1] != 4.0 / 5.0) return 1;
    CHECK(vl_index_get_vector(emb, 8, stored));
    if (stored[0] != 0.0 || stored[1] != 0.0) return 1;
    if (vl_index_add_embeddings_f32(emb, emb_ids, &model_out[0][0], 1, 1, 1, 0) != VL_ERR_DUP_ID) return 1;
    vl_index_destroy(emb);

    vl_index *hn = NULL;
    CHECK(vl_hnsw_create(3, VL_EUCLIDEAN, 0, &hn));
    const double hrows[4][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 1}};
    for (uint64_t i = 0; i < 4; ++i) CHECK(vl_index_add(hn, 100 + i, hrows[i], 3));
    const double hq[3] = {1.1, 0.1, 0.1};
    CHECK(vl_index_search(hn, hq, 3, 2, VL_EUCLIDEAN, ids, scores, &n)); /* src/index/hnsw.rs:628-633: first id 100 */
    printf("hnsw: n=%llu first id=%llu score=%.17g\n", (unsigned long long)n, (unsigned long long)ids[0], scores[0]);
    if (n == 0 || ids[0] != 100) return 1;
    if (vl_index_search(hn, hq, 3, 2, VL_COSINE, ids, scores, &n) != VL_ERR_METRIC_MISMATCH) return 1;
    if (vl_index_delete(hn, 999) != VL_ERR_NOT_FOUND) return 1; /* "Vector ID 999 does not exist" */
    vl_index_destroy(hn);
    printf("ok\n");
    return 0;
}
