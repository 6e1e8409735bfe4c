// lab: a register-only int8 / bf16 wave-tile loop -- NA x NB distinct random fragments, one accumulator per pair, like the
// inner loop of a GEMM with every memory instruction removed.  Varies the wave tile and the waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lab_mfma_tile.hip -o tools/bin/lab_mfma_tile
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// I8: v_mfma_i32_16x16x64_i8, else v_mfma_f32_16x16x32_bf16.  ROT: rotate the fragments every k-step (new operand bits
// each MFMA, as when fragments stream in from LDS) by xor-ing a per-step register into them (1 VALU per fragment).
template <int NA, int NB, bool I8, bool ROT>
__global__ __launch_bounds__(256) void tile_loop(const u32x4* in, float* out, int iters)
{
    u32x4 a[NA], b[NB];
    for (int i = 0; i < NA; ++i)
        a[i] = in[(i * 256 + threadIdx.x) & 4095];
    for (int j = 0; j < NB; ++j)
        b[j] = in[((NA + j) * 256 + threadIdx.x) & 4095];
    f32x4 c[NA][NB] = {};
    unsigned r = threadIdx.x * 2654435761u;
    for (int it = 0; it < iters; ++it)
    {
        if constexpr (ROT)
        {
            r = r * 1664525u + 1013904223u;
#pragma unroll
            for (int i = 0; i < NA; ++i)
                a[i].x ^= r, a[i].z ^= r >> 3;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                b[j].y ^= r, b[j].w ^= r >> 5;
            if constexpr (!I8) // keep bf16 exponents moderate: clear the top exponent bits of both halves
            {
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    a[i].x &= 0xbfffbfffu, a[i].z &= 0xbfffbfffu;
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    b[j].y &= 0xbfffbfffu, b[j].w &= 0xbfffbfffu;
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
            {
                if constexpr (I8)
                    asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c[i][j]) : "v"(a[i]), "v"(b[j]));
                else
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c[i][j]) : "v"(a[i]), "v"(b[j]));
            }
    }
    float sink = 0.f;
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < NB; ++j)
            sink += c[i][j][0] + c[i][j][3];
    if (sink == 12345.678f)
        out[0] = sink;
}

template <int NA, int NB, bool I8, bool ROT>
void run(const char* name, int wg_per_cu, const u32x4* zero, const u32x4* rnd, float* out)
{
    const int wgs = 256 * wg_per_cu, iters = 160000 / (NA * NB);
    for (int kind = 0; kind < 2; ++kind)
    {
        const u32x4* in = kind ? rnd : zero;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        tile_loop<NA, NB, I8, ROT><<<wgs, 256>>>(in, out, 50);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep)
        {
            hipEventRecord(e0);
            tile_loop<NA, NB, I8, ROT><<<wgs, 256>>>(in, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double per = I8 ? 2.0 * 16 * 16 * 64 : 2.0 * 16 * 16 * 32;
        const double total = (double) wgs * 4 * iters * NA * NB * per;
        printf("%-34s wg/CU %d %-7s %8.3f ms  %8.1f T(FL)OP/s\n", name, wg_per_cu, kind ? "random" : "zeros", best,
            total / (best * 1e-3) / 1e12);
    }
}

int main()
{
    std::vector<unsigned> h(4096 * 4);
    srand(1);
    for (auto& x : h)
        x = (unsigned) rand() ^ ((unsigned) rand() << 16);
    u32x4 *zero, *rnd;
    float* out;
    hipMalloc(&zero, 4096 * 16);
    hipMalloc(&rnd, 4096 * 16);
    hipMalloc(&out, 64);
    hipMemset(zero, 0, 4096 * 16);
    hipMemcpy(rnd, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    run<8, 8, true, false>("i8 8x8 frags (128x128 wave tile)", 1, zero, rnd, out);
    run<8, 8, true, true>("i8 8x8 frags, operands rotate", 1, zero, rnd, out);
    run<8, 4, true, false>("i8 8x4 frags (128x64 wave tile)", 2, zero, rnd, out);
    run<8, 4, true, true>("i8 8x4 frags, operands rotate", 2, zero, rnd, out);
    run<4, 4, true, true>("i8 4x4 frags, operands rotate", 2, zero, rnd, out);
    for (auto& x : h)
    {
        const unsigned lo = 0x3e00u + (rand() & 0x3ff) + ((rand() & 1) << 15), hi = 0x3e00u + (rand() & 0x3ff) + ((rand() & 1) << 15);
        x = lo | (hi << 16);
    }
    hipMemcpy(rnd, h.data(), 4096 * 16, hipMemcpyHostToDevice);
    run<8, 4, false, false>("bf16 8x4 frags", 2, zero, rnd, out);
    run<8, 4, false, true>("bf16 8x4 frags, operands rotate", 2, zero, rnd, out);
    return 0;
}
