// lab: semantics of v_permlane16_swap / v_permlane32_swap through the builtins
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o)
{
    unsigned v = threadIdx.x;
    auto r = __builtin_amdgcn_permlane16_swap(v, v + 100, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
    auto r2 = __builtin_amdgcn_permlane32_swap(v, v + 100, false, false);
    o[128 + threadIdx.x] = r2[0];
    o[192 + threadIdx.x] = r2[1];
}
int main()
{
    unsigned* d;
    hipMalloc(&d, 256 * 4);
    k<<<1, 64>>>(d);
    unsigned h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int j = 0; j < 4; ++j)
    {
        for (int i = 0; i < 64; i += 4)
            printf("%3u ", h[64 * j + i]);
        printf("\n");
    }
    return 0;
}
