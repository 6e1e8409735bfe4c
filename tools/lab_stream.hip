// lab_stream.hip -- development microbenchmark: how fast can ANY kernel stream S bytes of cold HBM?
// Sets the achievable floor for the small (8-24 MB) decode GEMV launches.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/lab_stream.hip -o gpurun_out/lab_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int LOADS, bool NT>
__global__ __launch_bounds__(512) void stream_kernel(const uint4* __restrict__ src, float* out, size_t n16)
{
    // each wave reads LOADS x 1 KiB contiguous; all loads issued up front
    const size_t wave = (size_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t base = wave * LOADS * 64 + (threadIdx.x & 63);
    uint4 v[LOADS];
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
    {
        const size_t idx = base + (size_t) i * 64;
        if (idx < n16)
        {
            if (NT)
            {
                typedef __attribute__((ext_vector_type(4))) unsigned int u4;
                u4 t = __builtin_nontemporal_load(reinterpret_cast<const u4*>(src + idx));
                v[i] = make_uint4(t.x, t.y, t.z, t.w);
            }
            else
                v[i] = src[idx];
        }
        else
            v[i] = make_uint4(0, 0, 0, 0);
    }
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
        acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (acc == 0x12345678u)
        out[threadIdx.x] = 1.f;
}

template <int LOADS, bool NT>
void run(const char* name, char* buf, size_t set_bytes, int sets, int wg_threads, float* out)
{
    const size_t n16 = set_bytes / 16;
    const size_t waves = (n16 + (size_t) LOADS * 64 - 1) / ((size_t) LOADS * 64);
    const int wpb = wg_threads / 64;
    const unsigned grid = (unsigned) ((waves + wpb - 1) / wpb);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t graph;
    hipGraphExec_t exec;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int s = 0; s < sets; ++s)
        hipLaunchKernelGGL((stream_kernel<LOADS, NT>), dim3(grid), dim3(wg_threads), 0, st,
            reinterpret_cast<const uint4*>(buf + (size_t) s * set_bytes), out, n16);
    hipStreamEndCapture(st, &graph);
    hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphLaunch(exec, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    const int reps = 5;
    for (int r = 0; r < reps; ++r)
        hipGraphLaunch(exec, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / (reps * sets);
    printf("%-28s set %6.1f MB grid %5u x %4d loads/wave %2d : %7.2f us/launch  %7.1f GB/s\n", name, set_bytes / 1e6, grid,
        wg_threads, LOADS, us, set_bytes / us / 1e3);
}

int main(int argc, char** argv)
{
    const size_t set_bytes = argc > 1 ? (size_t) atof(argv[1]) * 1000000 : 8912896;
    const int sets = (int) (640e6 / set_bytes) + 1;
    char* buf;
    float* out;
    hipMalloc(&buf, set_bytes * sets);
    hipMalloc(&out, 4096);
    hipMemset(buf, 1, set_bytes * sets);
    run<1, false>("plain 1 KiB/wave", buf, set_bytes, sets, 256, out);
    run<4, false>("plain 4 KiB/wave", buf, set_bytes, sets, 256, out);
    run<8, false>("plain 8 KiB/wave", buf, set_bytes, sets, 256, out);
    run<8, false>("plain 8 KiB/wave wg512", buf, set_bytes, sets, 512, out);
    run<8, true>("nt 8 KiB/wave wg512", buf, set_bytes, sets, 512, out);
    run<16, false>("plain 16 KiB/wave wg512", buf, set_bytes, sets, 512, out);
    run<16, true>("nt 16 KiB/wave wg512", buf, set_bytes, sets, 512, out);
    run<4, true>("nt 4 KiB/wave wg256", buf, set_bytes, sets, 256, out);
    return 0;
}
