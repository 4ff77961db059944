// Microbenchmark: the inner loop of k_mfma_scan in isolation (A fragments from LDS, one ds_read_b128 per
// MFMA, B fragments in registers, no global traffic, no barrier).  What fraction of the MFMA peak does this
// instruction structure reach?   hipcc --offload-arch=gfx950 -O3 mfma_loop.hip -o mfma_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KSTEPS = 24, GS = 4, NG = KSTEPS / GS, ROW = KSTEPS * 32 + 16;

template <int NWAVES, int AHEAD>
__global__ __launch_bounds__(NWAVES * 64) void k(const __bf16* q, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char tile[2][32 * ROW];
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    for (int i = threadIdx.x; i < 2 * 32 * ROW / 4; i += NWAVES * 64) ((unsigned*)tile)[i] = 0x3f803f80u + i * 0x10001u;  // bf16 ~1.0 noise
    bf16x8 b[KSTEPS];
    for (int s = 0; s < KSTEPS; ++s) b[s] = *(const bf16x8*)(q + (size_t)((blockIdx.x * NWAVES * 32 + (threadIdx.x >> 6) * 32 + col) & 4095) * KSTEPS * 16 + 16 * s + 8 * half);
    for (int s = 0; s < KSTEPS; ++s) asm volatile("" : "+v"(b[s]));
    __syncthreads();
    float keep = 0.f;
    bf16x8 a[AHEAD + 1][GS];
    const unsigned char* arow = &tile[0][col * ROW + half * 16];
    for (int p = 0; p < AHEAD; ++p)
        for (int jj = 0; jj < GS; ++jj) a[p][jj] = *(const bf16x8*)(arow + (p * GS + jj) * 32);
    for (int it = 0; it < iters; ++it) {
        const unsigned char* ar = &tile[it & 1][col * ROW + half * 16];
        const unsigned char* an = &tile[(it + 1) & 1][col * ROW + half * 16];
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) {
                const int gg = g + AHEAD;
                a[gg % (AHEAD + 1)][jj] = gg < NG ? *(const bf16x8*)(ar + (gg * GS + jj) * 32) : *(const bf16x8*)(an + ((gg - NG) * GS + jj) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < GS; ++jj) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g % (AHEAD + 1)][jj], b[g * GS + jj], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float m = acc[0];
        for (int r = 1; r < 16; ++r) m = fmaxf(m, acc[r]);
        keep = fmaxf(keep, m);
    }
    out[blockIdx.x * NWAVES * 64 + threadIdx.x] = keep;
}


typedef float f32x4v __attribute__((ext_vector_type(4)));
// same output tile per wave (32 rows x 32 queries x K) with v_mfma_f32_16x16x32_bf16: 2 row blocks x 2 query blocks
template <int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k16(const __bf16* q, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char tile[2][32 * ROW];
    const int lane = threadIdx.x & 63, r16 = lane & 15, kg = lane >> 4;
    for (int i = threadIdx.x; i < 2 * 32 * ROW / 4; i += NWAVES * 64) ((unsigned*)tile)[i] = 0x3f803f80u + i * 0x10001u;
    constexpr int KS = KSTEPS / 2;  // K = 32 per MFMA
    bf16x8 b[2][KS];
    for (int qb = 0; qb < 2; ++qb)
        for (int s = 0; s < KS; ++s)
            b[qb][s] = *(const bf16x8*)(q + (size_t)((blockIdx.x * NWAVES * 32 + (threadIdx.x >> 6) * 32 + qb * 16 + r16) & 4095) * KSTEPS * 16 + 32 * s + 8 * kg);
    for (int qb = 0; qb < 2; ++qb)
        for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(b[qb][s]));
    __syncthreads();
    float keep = 0.f;
    constexpr int G2 = 2, NG2 = KS / G2;  // 2 K-steps (= 4 ds_reads, 8 MFMAs) per group
    bf16x8 a[2][G2][2];
    for (int j = 0; j < G2; ++j)
        for (int rb = 0; rb < 2; ++rb) a[0][j][rb] = *(const bf16x8*)(&tile[0][(rb * 16 + r16) * ROW + kg * 16] + j * 64);
    for (int it = 0; it < iters; ++it) {
        const unsigned char* ar = &tile[it & 1][r16 * ROW + kg * 16];
        const unsigned char* an = &tile[(it + 1) & 1][r16 * ROW + kg * 16];
        f32x4v acc[2][2];
        for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 4; ++r) acc[x][y][r] = 0.f;
#pragma unroll
        for (int g = 0; g < NG2; ++g) {
#pragma unroll
            for (int j = 0; j < G2; ++j)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const int gg = g + 1;
                    a[gg & 1][j][rb] = gg < NG2 ? *(const bf16x8*)(ar + rb * 16 * ROW + (gg * G2 + j) * 64) : *(const bf16x8*)(an + rb * 16 * ROW + j * 64);
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < G2; ++j)
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int qb = 0; qb < 2; ++qb)
                        acc[rb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[g & 1][j][rb], b[qb][g * G2 + j], acc[rb][qb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float m = acc[0][0][0];
        for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) for (int r = 0; r < 4; ++r) m = fmaxf(m, acc[x][y][r]);
        keep = fmaxf(keep, m);
    }
    out[blockIdx.x * NWAVES * 64 + threadIdx.x] = keep;
}

template <int NW>
void run16(const char* name, int wgs, int iters, const __bf16* q, float* out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k16<NW>), dim3(wgs), dim3(NW * 64), 0, 0, q, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k16<NW>), dim3(wgs), dim3(NW * 64), 0, 0, q, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * KSTEPS * (double)iters * NW * wgs;
    printf("%-34s %4d WGs x %d waves: %8.3f ms  %.3f PFLOP/s\n", name, wgs, NW, ms, flop / (ms * 1e-3) / 1e15);
}

template <int NW, int AH>
void run(const char* name, int wgs, int iters, const __bf16* q, float* out)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NW, AH>), dim3(wgs), dim3(NW * 64), 0, 0, q, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NW, AH>), dim3(wgs), dim3(NW * 64), 0, 0, q, out, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 32 * 32 * 16 * KSTEPS * (double)iters * NW * wgs;
    printf("%-34s %4d WGs x %d waves: %8.3f ms  %.3f PFLOP/s\n", name, wgs, NW, ms, flop / (ms * 1e-3) / 1e15);
}

int main()
{
    __bf16* q; float* out;
    hipMalloc(&q, 1 << 24); hipMalloc(&out, 1 << 22);  // 4096 queries x 384 bf16 = 3 MB used; 131072 floats written
    std::vector<unsigned short> h((1 << 23), 0x3f80);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3f00 + (unsigned short)((i * 2654435761u) >> 25);
    hipMemcpy(q, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int it = 4000;
    run<8, 1>("8 waves, frags 1 group ahead", 256, it, q, out);
    run<8, 2>("8 waves, frags 2 groups ahead", 256, it, q, out);
    run<4, 1>("4 waves, frags 1 group ahead", 256, it, q, out);
    run<4, 2>("4 waves, frags 2 groups ahead", 256, it, q, out);
    run<4, 1>("4 waves x 2 WGs/CU, 1 ahead", 512, it, q, out);
    run<4, 2>("4 waves x 2 WGs/CU, 2 ahead", 512, it, q, out);
    run16<8>("16x16x32: 8 waves", 256, it, q, out);
    run16<4>("16x16x32: 4 waves", 256, it, q, out);
    run16<4>("16x16x32: 4 waves x 2 WGs/CU", 512, it, q, out);
    run<8, 1>("32x32x16 again: 8 waves, 1 ahead", 256, it, q, out);
    return 0;
}
