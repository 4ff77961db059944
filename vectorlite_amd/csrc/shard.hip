// shard.hip -- device merge of per-shard exact top-k lists (row-sharded batched flat search, shard.hpp).
//
// Every shard's list for a query is already in the reference's order -- score descending, GLOBAL storage
// position ascending on ties (src/index/flat.rs:116; -0.0 == +0.0 like partial_cmp) -- and global positions
// are unique across shards, so the merged rank of entry j of shard r is
//     j + sum over the other shards r' of |{entries of r' that precede it}|
// and each term is one binary search.  One workgroup per query, one thread per candidate: no sort, no
// atomics, world * ks * log2(ks) * world compares per query (config 3: 8 shards x 10 entries -> 80 threads).
// The work is a few hundred KB per batch: latency-bound by construction, neither roofline applies.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "shard.hpp"

namespace vl {
namespace {

constexpr int MERGE_THREADS = 256;

// does (sa, pa) come before (sb, pb) in the reference's result order?
__device__ __forceinline__ bool precedes(double sa, unsigned long long pa, double sb, unsigned long long pb)
{
    return sa > sb || (sa == sb && pa < pb);
}

__global__ __launch_bounds__(MERGE_THREADS) void k_shard_merge(const unsigned long long* __restrict__ gathered,
                                                               uint32_t world, uint32_t nq, uint32_t ks, uint32_t k_out,
                                                               unsigned long long* __restrict__ out_gpos,
                                                               unsigned long long* __restrict__ out_ids,
                                                               double* __restrict__ out_scores,
                                                               unsigned long long* __restrict__ out_n,
                                                               ShardMergeOut* __restrict__ out_status)
{
    __shared__ uint32_t s_cnt[SHARD_MAX_WORLD];
    const uint32_t q = blockIdx.x;
    const size_t words = (size_t)SHARD_HDR_WORDS + nq + 3ull * nq * ks;
    const size_t plane = (size_t)nq * ks;

    if (q == 0 && threadIdx.x == 0) {  // first failing rank, in rank order (every rank computes the same answer)
        unsigned long long st = 0, rk = 0;
        for (uint32_t r = 0; r < world; ++r) {
            const unsigned long long s = gathered[r * words];
            if (s != 0) {
                st = s;
                rk = r;
                break;
            }
        }
        out_status->status = st;
        out_status->rank = rk;
    }
    if (threadIdx.x < world) {
        const unsigned long long c = gathered[threadIdx.x * words + SHARD_HDR_WORDS + q];
        s_cnt[threadIdx.x] = (uint32_t)(c < ks ? c : ks);
    }
    __syncthreads();

    uint32_t total = 0;
    for (uint32_t r = 0; r < world; ++r) total += s_cnt[r];
    const uint32_t n_out = total < k_out ? total : k_out;
    if (threadIdx.x == 0) out_n[q] = n_out;

    for (uint32_t c = threadIdx.x; c < world * ks; c += MERGE_THREADS) {
        const uint32_t r = c / ks, j = c - r * ks;
        if (j >= s_cnt[r]) continue;
        const unsigned long long* base = gathered + r * words + SHARD_HDR_WORDS + nq + (size_t)q * ks;
        const double s = __longlong_as_double((long long)base[j]);
        const unsigned long long p = base[plane + j];
        uint32_t rank = j;
        for (uint32_t o = 0; o < world && rank < k_out; ++o) {
            if (o == r) continue;
            const unsigned long long* ob = gathered + o * words + SHARD_HDR_WORDS + nq + (size_t)q * ks;
            uint32_t lo = 0, hi = s_cnt[o];  // first entry of shard o that does NOT precede (s, p)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const double so = __longlong_as_double((long long)ob[mid]);
                if (precedes(so, ob[plane + mid], s, p))
                    lo = mid + 1;
                else
                    hi = mid;
            }
            rank += lo;
        }
        if (rank < k_out) {
            const size_t o = (size_t)q * k_out + rank;
            out_scores[o] = s;
            out_gpos[o] = p;
            out_ids[o] = base[2 * plane + j];
        }
    }
}

}  // namespace

hipError_t launch_shard_merge(hipStream_t stream, const unsigned long long* gathered, uint32_t world, uint32_t nq,
                              uint32_t ks, uint32_t k_out, unsigned long long* out_gpos, unsigned long long* out_ids,
                              double* out_scores, unsigned long long* out_n, ShardMergeOut* out_status)
{
    // shapes the kernel and its grid assume, checked on the host before anything is launched
    if (!gathered || !out_gpos || !out_ids || !out_scores || !out_n || !out_status) return hipErrorInvalidValue;
    if (world == 0 || world > (uint32_t)SHARD_MAX_WORLD || nq == 0 || ks == 0 || k_out == 0) return hipErrorInvalidValue;
    if ((uint64_t)world * ks > 0xFFFFFFFFull || k_out > (uint64_t)world * ks) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_shard_merge, dim3(nq), dim3(MERGE_THREADS), 0, stream, gathered, world, nq, ks, k_out, out_gpos,
                       out_ids, out_scores, out_n, out_status);
    return hipGetLastError();
}

}  // namespace vl
