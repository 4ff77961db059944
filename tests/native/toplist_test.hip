// Unit test of the TopList machinery on hardware (debug tool, not shipped).
#include "../../vectorlite_amd/csrc/kernels.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <random>
using namespace vl;
template <typename K>
__global__ void k_offer(const K* keys, const uint32_t* pos, int n, int active_per_step, K* out_k, uint32_t* out_p)
{
    TopList<K> L; L.init();
    int lane = threadIdx.x;
    for (int s = 0; s < n; s += 64) {
        int i = s + lane;
        bool valid = i < n && lane < active_per_step;
        K k = valid ? keys[i] : (K)0;
        L.offer(k, valid ? pos[i] : 0u, valid);
    }
    out_k[lane] = L.key; out_p[lane] = L.pos;
}
template <typename K>
int run(int n, int active, int seed, int mode)
{
    std::mt19937 rng(seed);
    std::vector<K> keys(n); std::vector<uint32_t> pos(n);
    for (int i = 0; i < n; ++i) { pos[i] = i; keys[i] = mode == 0 ? (K)((rng() & 1) ? 1.0 : -1.0) : (K)((int)(rng() % 1000) - 500) / (K)7; }
    K* dk; uint32_t* dp; K* ok; uint32_t* op;
    (void)hipMalloc(&dk, n * sizeof(K)); (void)hipMalloc(&dp, n * 4); (void)hipMalloc(&ok, 64 * sizeof(K)); (void)hipMalloc(&op, 64 * 4);
    (void)hipMemcpy(dk, keys.data(), n * sizeof(K), hipMemcpyHostToDevice); (void)hipMemcpy(dp, pos.data(), n * 4, hipMemcpyHostToDevice);
    k_offer<K><<<1, 64>>>(dk, dp, n, active, ok, op);
    std::vector<K> hk(64); std::vector<uint32_t> hp(64);
    (void)hipMemcpy(hk.data(), ok, 64 * sizeof(K), hipMemcpyDeviceToHost); (void)hipMemcpy(hp.data(), op, 64 * 4, hipMemcpyDeviceToHost);
    std::vector<int> idx;
    for (int s = 0; s < n; s += 64) for (int l = 0; l < 64 && s + l < n; ++l) if (l < active) idx.push_back(s + l);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return keys[a] > keys[b] || (keys[a] == keys[b] && pos[a] < pos[b]); });
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        uint32_t wp = i < (int)idx.size() ? pos[idx[i]] : 0xFFFFFFFFu;
        if (hp[i] != wp) { if (bad < 3) printf("  n=%d active=%d mode=%d lane %d got pos %u key %g want pos %u\n", n, active, mode, i, hp[i], (double)hk[i], wp); bad++; }
    }
    (void)hipFree(dk); (void)hipFree(dp); (void)hipFree(ok); (void)hipFree(op);
    return bad;
}
int main()
{
    int total = 0;
    for (int mode = 0; mode < 2; ++mode)
        for (int n : {1, 5, 64, 65, 128, 1000})
            for (int active : {64, 1, 33}) {
                int a = run<float>(n, active, n * 7 + active, mode);
                int b = run<double>(n, active, n * 7 + active, mode);
                if (a || b) printf("n=%d active=%d mode=%d: float bad=%d double bad=%d\n", n, active, mode, a, b);
                total += a + b;
            }
    printf("total bad %d\n", total);
    return 0;
}
