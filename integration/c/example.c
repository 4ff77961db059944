/* Plain-C caller of libvectorlite_amd.so: the reference's own flat-search test
 * (src/index/flat.rs:187-201: unit basis vectors, q = [1,0,0], k = 2 -> first id 1, score 1)
 * followed by a batched search, an HNSW round trip and a row-sharded batch over the library's own RCCL communicator.  Doubles as the check that
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
    /* k = 10 asked, two slots offered: min(k, len, capacity) results are written (vl_index_search_cap) */
    CHECK(vl_index_search_cap(idx, q, 3, 10, VL_COSINE, 2, ids, scores, &n));
    printf("flat: n=%llu first id=%llu score=%.17g\n", (unsigned long long)n, (unsigned long long)ids[0], scores[0]);
    if (n != 2 || ids[0] != 1 || fabs(scores[0] - 1.0) > 1e-10) return 1;

    uint64_t e = 0, a = 0;
    if (vl_index_search(idx, q, 2, 2, VL_COSINE, ids, scores, &n) != VL_ERR_DIM_MISMATCH) return 1;
    vl_last_dim_mismatch(&e, &a);
    if (e != 3 || a != 2) return 1;

    const double qs[2][3] = {{0, 1, 0}, {0.1, 0.1, 1.1}};
    uint64_t bids[2 * 2], bn[2];
    double bscores[2 * 2];
    CHECK(vl_index_search_batch_cap(idx, &qs[0][0], 2, 3, 5, VL_EUCLIDEAN, 2, bids, bscores, bn)); /* rows 2 apart */
    printf("batch: first ids %llu %llu\n", (unsigned long long)bids[0], (unsigned long long)bids[2]);
    if (bids[0] != 2 || bids[2] != 3) return 1;
    CHECK(vl_index_delete(idx, 2));
    CHECK(vl_index_delete(idx, 2)); /* absent id: still Ok for the flat index (src/index/flat.rs:93-96) */
    if (vl_index_len(idx) != 2) return 1;

    /* the same index over two GPUs of the node in this one process (here: the same card twice), rows sharded:
     * identical answers, no rank processes, no id handshake */
    {
        vl_index *multi = NULL;
        const int devs[2] = {0, 0};
        CHECK(vl_flat_create_multi(3, devs, 2, VL_MULTI_ROW_SHARDS, &multi));
        for (uint64_t i = 0; i < 3; ++i) CHECK(vl_index_add(multi, i + 1, rows[i], 3));
        if (vl_index_add(multi, 3, rows[0], 3) != VL_ERR_DUP_ID) return 1; /* wherever id 3 lives */
        uint64_t mids[3], mn = 0;
        double msc[3];
        CHECK(vl_index_search_cap(multi, q, 3, 3, VL_COSINE, 3, mids, msc, &mn));
        int parts = 0, mode = -1;
        uint64_t prow[2] = {0, 0};
        CHECK(vl_index_parts(multi, &parts, &mode, prow, NULL, 2));
        printf("multi: %d parts (mode %d) holding %llu + %llu rows; best id %llu\n", parts, mode, (unsigned long long)prow[0],
               (unsigned long long)prow[1], (unsigned long long)mids[0]);
        if (mn != 3 || mids[0] != 1 || mids[1] != 2 || mids[2] != 3 || parts != 2 || prow[0] + prow[1] != 3) return 1;
        vl_index_destroy(multi);
    }
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

    /* row shards (north_star's config 3): this process is rank 0 of a world of 1 -- the same calls a rank of 8 makes.
     * vl_comm_unique_id on rank 0, the 128 bytes to the others by the host's own channel, vl_comm_create everywhere
     * (ncclCommInitRank), vl_shard_sync (each rank learns its row offset), then batches: local scan, ONE ncclAllGather
     * of the per-shard top-k records inside the library, device merge.  The answer equals vl_index_search_batch on
     * one index holding every shard's rows. */
    {
        enum { N = 9000, D = 8, NQ = 5, K = 4 };
        static double srows[N][D];
        static uint64_t sids[N];
        uint64_t seed = 88172645463325252ull;
        for (int i = 0; i < N; ++i) {
            sids[i] = 1000 + (uint64_t)i;
            for (int d = 0; d < D; ++d) {
                seed ^= seed << 13; seed ^= seed >> 7; seed ^= seed << 17;      /* xorshift: any data will do */
                srows[i][d] = (double)(seed % 2001) / 1000.0 - 1.0;
            }
        }
        for (int d = 0; d < D; ++d) srows[7000][d] = srows[12][d];              /* an exact duplicate: position breaks the tie */
        vl_index *shard = NULL;
        CHECK(vl_flat_from_rows(D, sids, &srows[0][0], N, 0, &shard));
        uint8_t id[VL_COMM_ID_BYTES];
        vl_comm *comm = NULL;
        CHECK(vl_comm_unique_id(id));
        CHECK(vl_comm_create(id, /*world=*/1, /*rank=*/0, /*device=*/0, &comm));
        uint64_t off = 99, total = 0;
        CHECK(vl_shard_sync(shard, comm, &off, &total));
        if (off != 0 || total != N || vl_comm_world(comm) != 1 || vl_comm_rank(comm) != 0) return 1;
        double sq[NQ][D];
        for (int q = 0; q < NQ; ++q)
            for (int d = 0; d < D; ++d) sq[q][d] = srows[12 + 100 * q][d] * (q == 0 ? 1.0 : 1.01);
        uint64_t gpos[NQ * K], gids[NQ * K], gn[NQ], lids[NQ * K], ln[NQ];
        double gsc[NQ * K], lsc[NQ * K];
        CHECK(vl_shard_search_batch(shard, comm, &sq[0][0], NQ, D, K, VL_EUCLIDEAN, gpos, gids, gsc, gn));
        CHECK(vl_index_search_batch(shard, &sq[0][0], NQ, D, K, VL_EUCLIDEAN, lids, lsc, ln));
        for (int q = 0; q < NQ; ++q) {
            if (gn[q] != K || ln[q] != K) return 1;
            for (int j = 0; j < K; ++j)
                if (gids[q * K + j] != lids[q * K + j] || gsc[q * K + j] != lsc[q * K + j] || gids[q * K + j] != 1000 + gpos[q * K + j]) return 1;
        }
        printf("shards: query 0 -> ids %llu %llu (rows 12 and 7000 are equal: the earlier one first), scores %.17g %.17g\n",
               (unsigned long long)gids[0], (unsigned long long)gids[1], gsc[0], gsc[1]);
        if (gids[0] != 1012 || gids[1] != 8000 || gsc[0] != 1.0 || gsc[1] != 1.0) return 1;
        /* a query of the wrong length fails on every rank alike (the status travels inside the exchange) */
        if (vl_shard_search_batch(shard, comm, &sq[0][0], 1, D - 1, K, VL_EUCLIDEAN, gpos, gids, gsc, gn) != VL_ERR_DIM_MISMATCH) return 1;
        /* a mutation without a re-sync is reported, not silently merged */
        CHECK(vl_index_delete(shard, 1000));
        if (vl_shard_search_batch(shard, comm, &sq[0][0], 1, D, K, VL_EUCLIDEAN, gpos, gids, gsc, gn) != VL_ERR_INVALID_ARG) return 1;
        CHECK(vl_shard_sync(shard, comm, &off, &total));
        if (total != N - 1) return 1;
        CHECK(vl_shard_search_batch(shard, comm, &sq[0][0], 1, D, K, VL_EUCLIDEAN, gpos, gids, gsc, gn));
        if (gn[0] != K || gids[0] != 1012 || gpos[0] != 11) return 1;          /* positions moved up by one */
        vl_comm_destroy(comm);
        vl_index_destroy(shard);
    }
    printf("ok\n");
    return 0;
}
