// lab: semantics + issue rate of v_cvt_scalef32_pk_{bf16,f16}_fp8 vs v_and_or_b32 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

__global__ void sem(unsigned* out, float sc)
{
    unsigned t = threadIdx.x; // 0..255: byte value
    unsigned w = t | (t << 8);
    out[t] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, sc, false));
    out[256 + t] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w, sc, false));
}

template <int MODE>
__global__ void rate(unsigned* out, int iters, float sc)
{
    unsigned a = threadIdx.x * 2654435761u, b = a ^ 0x1234567u, c = a + 77u, d = b + 99u;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
            if (MODE == 0)
            {
                a = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a, sc, false));
                b = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(b, sc, true));
                c = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(c, sc, false));
                d = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(d, sc, true));
            }
            else if (MODE == 1)
            {
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(b) : "v"(c), "v"(d));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(c) : "v"(d), "v"(a));
                asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(d) : "v"(a), "v"(b));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 1 << 22);
    sem<<<1, 256>>>(d, 512.f);
    std::vector<unsigned> h(512);
    hipMemcpy(h.data(), d, 2048, hipMemcpyDeviceToHost);
    for (int t = 0; t < 20; ++t)
    {
        unsigned bf = h[t] & 0xffff, hf = h[256 + t] & 0xffff;
        unsigned fb = bf << 16;
        float f;
        memcpy(&f, &fb, 4);
        printf("byte %2d -> bf16 0x%04x (%g)  f16 0x%04x\n", t, bf, f, hf);
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 4096;
    for (int mode = 0; mode < 2; ++mode)
    {
        for (int rep = 0; rep < 2; ++rep)
        {
            hipEventRecord(e0);
            if (mode == 0)
                rate<0><<<256 * 8, 256>>>(d, iters, 512.f);
            else
                rate<1><<<256 * 8, 256>>>(d, iters, 512.f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            // per SIMD: 8 blocks * 4 waves per CU / 4 SIMDs = 8 waves per SIMD, each iters*64 ops
            double ops_per_simd = 8.0 * iters * 64;
            printf("mode %d: %.3f ms  -> %.2f cycles/op/SIMD @2.4GHz\n", mode, ms, ms * 1e-3 * 2.4e9 / ops_per_simd);
        }
    }
    return 0;
}
