#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out_inplace, int* out_outofplace, int* out_shfl)
{
    int lane = threadIdx.x;
    int v = lane * 10 + 1;
    int a = __builtin_amdgcn_update_dpp(v, v, 0x138, 0xF, 0xF, false);
    int b = __builtin_amdgcn_update_dpp(-7, v, 0x138, 0xF, 0xF, false);
    int c = __shfl_up(v, 1);
    out_inplace[lane] = a;
    out_outofplace[lane] = b;
    out_shfl[lane] = c;
}
int main()
{
    int *d; hipMalloc(&d, 3 * 64 * sizeof(int));
    k<<<1, 64>>>(d, d + 64, d + 128);
    int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad_a = 0, bad_b = 0, bad_c = 0;
    for (int i = 0; i < 64; ++i) {
        int want = i == 0 ? 1 : (i - 1) * 10 + 1;
        if (h[i] != want) { if (!bad_a) printf("inplace lane %d got %d want %d\n", i, h[i], want); bad_a++; }
        int wantb = i == 0 ? -7 : (i - 1) * 10 + 1;
        if (h[64 + i] != wantb) { if (!bad_b) printf("outofplace lane %d got %d want %d\n", i, h[64 + i], wantb); bad_b++; }
        if (h[128 + i] != want) { if (!bad_c) printf("shfl lane %d got %d want %d\n", i, h[128 + i], want); bad_c++; }
    }
    printf("bad inplace=%d outofplace=%d shfl=%d\n", bad_a, bad_b, bad_c);
    return 0;
}
