// norm_quant.hip -- the feeders of the SmoothQuant GEMM, HBM-bound, one pass over the row:
//   * RMSNorm / LayerNorm with fused int8 quantisation (static per-tensor scale or dynamic per-token scale):
//     kernels/rmsnormKernels.cu:52-150 generalRmsNorm, kernels/layernormKernels.cu:60-192 generalLayerNorm
//     (plugins rmsnormQuantizationPlugin.cpp:125-165, layernormQuantizationPlugin.cpp:128-170);
//   * static per-tensor quantisation: kernels/quantization.cu:30-86 invokeQuantization (quantizeTensorPlugin.cpp:99-127).
// Arithmetic follows the reference statement by statement: statistics in fp32, `val = T((x [- mean]) * rstd * gamma
// + beta)` rounded to the activation type T, amax over |val| in T starting from 1e-6 (rounded to T), int8 =
// cvt.rni.sat(float(val) * scale) with scale = *scale_orig_quant or 127 / amax, per-token scale out = amax / 127.
// Layout: one workgroup of 256 threads per row; the row stays in registers between the statistics pass and the
// output pass (16-byte loads, 8 elements per chunk), so HBM traffic is the algorithmic minimum: the row read once,
// the int8 (or T) row written once.  Rows longer than 16384 or not a multiple of 8 take a re-reading scalar kernel.
#include "kernels.h"
#include <cstdlib>

namespace mi355q
{
namespace
{

template <typename T>
struct NormIO;
template <>
struct NormIO<float>
{
    static __device__ __forceinline__ void load8(const void* base, size_t idx, float (&x)[8])
    {
        const float4* p = reinterpret_cast<const float4*>(static_cast<const float*>(base) + idx);
        const float4 a = p[0], b = p[1];
        x[0] = a.x, x[1] = a.y, x[2] = a.z, x[3] = a.w, x[4] = b.x, x[5] = b.y, x[6] = b.z, x[7] = b.w;
    }
    static __device__ __forceinline__ float load1(const void* base, size_t idx)
    {
        return static_cast<const float*>(base)[idx];
    }
    static __device__ __forceinline__ float round(float v)
    {
        return v;
    }
    static __device__ __forceinline__ void store8(void* base, size_t idx, const float (&x)[8])
    {
        float4* p = reinterpret_cast<float4*>(static_cast<float*>(base) + idx);
        p[0] = make_float4(x[0], x[1], x[2], x[3]);
        p[1] = make_float4(x[4], x[5], x[6], x[7]);
    }
    static __device__ __forceinline__ void store1(void* base, size_t idx, float v)
    {
        static_cast<float*>(base)[idx] = v;
    }
};
template <typename T>
struct NormIO16
{
    static __device__ __forceinline__ void load8(const void* base, size_t idx, float (&x)[8])
    {
        const uint4 w = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(base) + idx);
        const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            x[2 * i] = T16<T>::lo(u[i]);
            x[2 * i + 1] = T16<T>::hi(u[i]);
        }
    }
    static __device__ __forceinline__ float load1(const void* base, size_t idx)
    {
        return T16<T>::to_float(static_cast<const uint16_t*>(base)[idx]);
    }
    static __device__ __forceinline__ float round(float v)
    {
        return T16<T>::to_float(T16<T>::from_float(v));
    }
    static __device__ __forceinline__ void store8(void* base, size_t idx, const float (&x)[8])
    {
        *reinterpret_cast<uint4*>(static_cast<uint16_t*>(base) + idx)
            = make_uint4(pack2<T>(x[0], x[1]), pack2<T>(x[2], x[3]), pack2<T>(x[4], x[5]), pack2<T>(x[6], x[7]));
    }
    static __device__ __forceinline__ void store1(void* base, size_t idx, float v)
    {
        static_cast<uint16_t*>(base)[idx] = T16<T>::from_float(v);
    }
};
template <>
struct NormIO<f16_t> : NormIO16<f16_t>
{
};
template <>
struct NormIO<bf16_t> : NormIO16<bf16_t>
{
};

__device__ __forceinline__ float quant_rni_sat_f(float x)
{
    // cvt.rni.sat.s8.f32 (common/cudaTypeUtils.cuh cuda_cast<int8_t, float>) in two operations: clamp to [-128, 127],
    // then add 1.5 * 2^23 -- the sum's unit in the last place is 1, so the addition itself rounds to nearest-even, and
    // the low byte of the result's bit pattern is the two's-complement int8 (clamping first or rounding first gives
    // the same integer: the bounds are integers)
    return __builtin_amdgcn_fmed3f(x, -128.f, 127.f) + 12582912.f;
}

__device__ __forceinline__ uint32_t quant_rni_sat(float x)
{
    return __builtin_bit_cast(uint32_t, quant_rni_sat_f(x)) & 0xffu;
}

// four products -> four int8 in one dword (v_perm_b32 gathers the low bytes: 2.75 VALU operations per element)
__device__ __forceinline__ uint32_t quant4_rni_sat(float a, float b, float c, float d)
{
    const uint32_t za = __builtin_bit_cast(uint32_t, quant_rni_sat_f(a)), zb = __builtin_bit_cast(uint32_t, quant_rni_sat_f(b));
    const uint32_t zc = __builtin_bit_cast(uint32_t, quant_rni_sat_f(c)), zd = __builtin_bit_cast(uint32_t, quant_rni_sat_f(d));
    uint32_t w = __builtin_amdgcn_perm(zb, za, 0x0c0c0400u); // byte 0 <- za.0, byte 1 <- zb.0
    w = __builtin_amdgcn_perm(zc, w, 0x0c040100u);           // keep bytes 0, 1; byte 2 <- zc.0
    return __builtin_amdgcn_perm(zd, w, 0x04020100u);        // keep bytes 0..2; byte 3 <- zd.0
}

struct NormArgs
{
    const void* input;
    const void* gamma;
    const void* beta; // or null
    void* out;        // T [tokens, hidden] or null
    int8_t* out_quant;
    const float* scale_per_tensor; // or null
    float* scale_per_token;        // or null
    float eps;
    int tokens, hidden;
};

// block-wide sum / max over NT threads (1, 2 or 4 waves); every thread gets the result.  One wave: shuffles only.
template <bool MAX, int NT = 256>
__device__ __forceinline__ float block_reduce(float v, float* sm)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
    {
        const float o = __shfl_xor(v, off);
        v = MAX ? fmaxf(v, o) : v + o;
    }
    if constexpr (NT == 64)
        return v;
    else
    {
        const int wave = threadIdx.x >> 6;
        __syncthreads(); // sm may still be read from the previous reduction
        if ((threadIdx.x & 63) == 0)
            sm[wave] = v;
        __syncthreads();
        if constexpr (NT == 128)
            return MAX ? fmaxf(sm[0], sm[1]) : sm[0] + sm[1];
        else
            return MAX ? fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3])) : (sm[0] + sm[1]) + (sm[2] + sm[3]);
    }
}

typedef __attribute__((ext_vector_type(4))) unsigned int norm_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int norm_u32x2_t;

// buffer descriptor over `bytes` bytes at p (wave-uniform): loads past the end return 0, stores past the end are dropped
__device__ __forceinline__ __amdgpu_buffer_rsrc_t norm_rsrc(const void* p, uint32_t bytes)
{
    const uint64_t u = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (u >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*) (((uint64_t) hi << 32) | lo), 0, bytes, 0x00020000);
}

// 8 consecutive elements at element offset i0 of the buffer (zeros past the end): the raw load, and its conversion
template <typename T>
struct NormRaw
{
    norm_u32x4_t v[sizeof(T) == 4 ? 2 : 1];
};

template <typename T, int AUX = 0>
__device__ __forceinline__ NormRaw<T> norm_load8_raw(__amdgpu_buffer_rsrc_t rs, int i0)
{
    NormRaw<T> r;
    if constexpr (sizeof(T) == 4)
    {
        r.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, i0 * 4, 0, AUX);
        r.v[1] = __builtin_amdgcn_raw_buffer_load_b128(rs, i0 * 4 + 16, 0, AUX);
    }
    else
        r.v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, i0 * 2, 0, AUX);
    return r;
}

template <typename T>
__device__ __forceinline__ void norm_cvt8(const NormRaw<T>& r, float (&x)[8])
{
    if constexpr (sizeof(T) == 4)
    {
        const uint32_t u[8] = {r.v[0].x, r.v[0].y, r.v[0].z, r.v[0].w, r.v[1].x, r.v[1].y, r.v[1].z, r.v[1].w};
#pragma unroll
        for (int j = 0; j < 8; ++j)
            x[j] = __builtin_bit_cast(float, u[j]);
    }
    else
    {
        const uint32_t u[4] = {r.v[0].x, r.v[0].y, r.v[0].z, r.v[0].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            x[2 * j] = T16<T>::lo(u[j]);
            x[2 * j + 1] = T16<T>::hi(u[j]);
        }
    }
}

// MODE: 0 RMSNorm, 1 LayerNorm two-pass variance, 2 LayerNorm E[x^2] - E[x]^2 (use_diff_of_squares)
// No load or store sits under a branch (buffer descriptors bound the row instead): hipcc counts vmcnt statically, and
// with each chunk's load and first use inside `if (live)` the chunks of a row were fetched one HBM round trip after
// the other.
template <typename T, int MODE, int VPT, int NT>
__global__ __launch_bounds__(NT) void norm_quant_kernel(NormArgs a)
{
    __shared__ float sm[4];
    const size_t row = blockIdx.x;
    const size_t base = row * a.hidden;
    const uint32_t row_bytes = (uint32_t) a.hidden * sizeof(T);
    const __amdgpu_buffer_rsrc_t in_rs = norm_rsrc(static_cast<const char*>(a.input) + base * sizeof(T), row_bytes);
    const __amdgpu_buffer_rsrc_t g_rs = norm_rsrc(a.gamma, row_bytes);
    const __amdgpu_buffer_rsrc_t b_rs = norm_rsrc(a.beta ? a.beta : a.gamma, a.beta ? row_bytes : 0);
    float x[VPT][8];
    float s1 = 0.f, s2 = 0.f;
    {
        NormRaw<T> raw[VPT];
#pragma unroll
        for (int c = 0; c < VPT; ++c)
            raw[c] = norm_load8_raw<T, 2>(in_rs, (c * NT + threadIdx.x) * 8); // nontemporal: read once
#pragma unroll
        for (int c = 0; c < VPT; ++c)
        {
            norm_cvt8<T>(raw[c], x[c]);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                s1 += x[c][j]; // elements past the row are 0
                s2 += x[c][j] * x[c][j];
            }
        }
    }
    float mean = 0.f, rstd;
    if (MODE == 0)
        rstd = rsqrtf(block_reduce<false, NT>(s2, sm) / a.hidden + a.eps);
    else if (MODE == 2)
    {
        mean = block_reduce<false, NT>(s1, sm) / a.hidden;
        rstd = rsqrtf(block_reduce<false, NT>(s2, sm) / a.hidden - mean * mean + a.eps);
    }
    else
    {
        mean = block_reduce<false, NT>(s1, sm) / a.hidden;
        float v = 0.f;
#pragma unroll
        for (int c = 0; c < VPT; ++c)
        {
            const bool live = (c * NT + (int) threadIdx.x) * 8 < a.hidden;
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const float d = x[c][j] - mean;
                v += live ? d * d : 0.f;
            }
        }
        rstd = rsqrtf(block_reduce<false, NT>(v, sm) / a.hidden + a.eps);
    }
    // normalise, round to T (gamma / beta past the row read as 0, so the dead elements become 0 and leave amax alone)
    float amax = NormIO<T>::round(1e-6f);
    {
        NormRaw<T> graw[VPT], braw[VPT];
#pragma unroll
        for (int c = 0; c < VPT; ++c)
            graw[c] = norm_load8_raw<T>(g_rs, (c * NT + threadIdx.x) * 8);
#pragma unroll
        for (int c = 0; c < VPT; ++c) // without beta: a zero-sized descriptor, the loads return 0 and cost no memory access
            braw[c] = norm_load8_raw<T>(b_rs, (c * NT + threadIdx.x) * 8);
#pragma unroll
        for (int c = 0; c < VPT; ++c)
        {
            float g[8], b[8];
            norm_cvt8<T>(graw[c], g);
            norm_cvt8<T>(braw[c], b);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                float v = (MODE == 0 ? x[c][j] : x[c][j] - mean) * rstd * g[j];
                if (a.beta)
                    v = v + b[j];
                x[c][j] = NormIO<T>::round(v);
                amax = fmaxf(amax, fabsf(x[c][j]));
            }
        }
    }
    float scale = 0.f;
    if (a.scale_per_token)
    {
        amax = block_reduce<true, NT>(amax, sm);
        scale = 127.f / amax;
        if (threadIdx.x == 0)
            a.scale_per_token[row] = amax / 127.f;
    }
    else if (a.scale_per_tensor)
        scale = a.scale_per_tensor[0];
    if (a.scale_per_token || a.scale_per_tensor)
    {
        const __amdgpu_buffer_rsrc_t q_rs = norm_rsrc(a.out_quant + base, (uint32_t) a.hidden);
#pragma unroll
        for (int c = 0; c < VPT; ++c)
        {
            norm_u32x2_t w;
            w.x = quant4_rni_sat(x[c][0] * scale, x[c][1] * scale, x[c][2] * scale, x[c][3] * scale);
            w.y = quant4_rni_sat(x[c][4] * scale, x[c][5] * scale, x[c][6] * scale, x[c][7] * scale);
            __builtin_amdgcn_raw_buffer_store_b64(w, q_rs, (c * NT + threadIdx.x) * 8, 0, 2);
        }
    }
    else
    {
#pragma unroll
        for (int c = 0; c < VPT; ++c)
            if ((c * NT + (int) threadIdx.x) * 8 < a.hidden)
                NormIO<T>::store8(a.out, base + (c * NT + threadIdx.x) * 8, x[c]);
    }
}


// ---- row kernel without statistics: per-token int8 quantisation of x itself (OP 0; quantization.cu:93-127
// invokePerTokenQuantization) or of the gated activation of the SmoothQuant MLP (OP 1; quantization/layers.py:922-941
// SmoothQuantGatedMLP.forward: inter = silu(fc(x)); y = inter * gate(x); y = y / smoother; quantize_tensor |
// quantize_per_token), every step rounded to T as the reference's layer boundaries do.  Same layout as the norm
// kernel: the row stays in registers between the amax and the output, one read and one write per element.
struct RowQuantArgs
{
    const void* x;        // [rows, cols], row stride ld_x elements; OP 1: the tensor that goes through SiLU
    const void* gate;     // OP 1: the multiplier, row stride ld_gate
    const void* smoother; // OP 1: T [cols] divisor, or null
    void* out;            // T [rows, cols] or null
    int8_t* out_quant;    // int8 [rows, cols] or null
    const float* scale_per_tensor;
    float* scale_per_token;
    int rows, cols;
    int64_t ld_x, ld_gate;
};

template <typename T, int OP, int VPT, int NT>
__global__ __launch_bounds__(NT) void row_quant_kernel(RowQuantArgs a)
{
    __shared__ float sm[4];
    const size_t row = blockIdx.x;
    const uint32_t row_bytes = (uint32_t) a.cols * sizeof(T);
    const __amdgpu_buffer_rsrc_t x_rs = norm_rsrc(static_cast<const char*>(a.x) + row * a.ld_x * sizeof(T), row_bytes);
    float x[VPT][8];
    {
        NormRaw<T> raw[VPT];
#pragma unroll
        for (int c = 0; c < VPT; ++c)
            raw[c] = norm_load8_raw<T, 2>(x_rs, (c * NT + threadIdx.x) * 8);
        if constexpr (OP == 1)
        {
            const __amdgpu_buffer_rsrc_t g_rs
                = norm_rsrc(static_cast<const char*>(a.gate) + row * a.ld_gate * sizeof(T), row_bytes);
            const __amdgpu_buffer_rsrc_t s_rs = norm_rsrc(a.smoother ? a.smoother : a.x, a.smoother ? row_bytes : 0);
            NormRaw<T> graw[VPT], sraw[VPT];
#pragma unroll
            for (int c = 0; c < VPT; ++c)
                graw[c] = norm_load8_raw<T, 2>(g_rs, (c * NT + threadIdx.x) * 8);
#pragma unroll
            for (int c = 0; c < VPT; ++c)
                sraw[c] = norm_load8_raw<T>(s_rs, (c * NT + threadIdx.x) * 8);
#pragma unroll
            for (int c = 0; c < VPT; ++c)
            {
                float g[8], sd[8];
                norm_cvt8<T>(raw[c], x[c]);
                norm_cvt8<T>(graw[c], g);
                norm_cvt8<T>(sraw[c], sd);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                {
                    // silu(v) = v / (1 + exp(-v)); elements past the row are 0 -> 0
                    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * x[c][j]);
                    float v = NormIO<T>::round(x[c][j] * __builtin_amdgcn_rcpf(1.f + e));
                    v = NormIO<T>::round(v * g[j]);
                    if (a.smoother)
                        v = NormIO<T>::round(v * __builtin_amdgcn_rcpf(sd[j]));
                    x[c][j] = v;
                }
            }
        }
        else
        {
#pragma unroll
            for (int c = 0; c < VPT; ++c)
                norm_cvt8<T>(raw[c], x[c]);
        }
    }
    float scale = 0.f;
    if (a.scale_per_token)
    {
        float amax = NormIO<T>::round(1e-6f);
#pragma unroll
        for (int c = 0; c < VPT; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                amax = fmaxf(amax, fabsf(x[c][j]));
        amax = block_reduce<true, NT>(amax, sm);
        scale = 127.f / amax;
        if (threadIdx.x == 0)
            a.scale_per_token[row] = amax / 127.f;
    }
    else if (a.scale_per_tensor)
        scale = a.scale_per_tensor[0];
    if (a.out_quant)
    {
        const __amdgpu_buffer_rsrc_t q_rs = norm_rsrc(a.out_quant + row * a.cols, (uint32_t) a.cols);
#pragma unroll
        for (int c = 0; c < VPT; ++c)
        {
            norm_u32x2_t w;
            w.x = quant4_rni_sat(x[c][0] * scale, x[c][1] * scale, x[c][2] * scale, x[c][3] * scale);
            w.y = quant4_rni_sat(x[c][4] * scale, x[c][5] * scale, x[c][6] * scale, x[c][7] * scale);
            __builtin_amdgcn_raw_buffer_store_b64(w, q_rs, (c * NT + threadIdx.x) * 8, 0, 2);
        }
    }
    if (a.out)
    {
#pragma unroll
        for (int c = 0; c < VPT; ++c)
            if ((c * NT + (int) threadIdx.x) * 8 < a.cols)
                NormIO<T>::store8(a.out, row * a.cols + (c * NT + threadIdx.x) * 8, x[c]);
    }
}

template <typename T, int OP>
void launch_row_quant(const RowQuantArgs& a, hipStream_t stream)
{
    const int nt = a.cols <= 8192 ? 128 : 256;
    const int vpt = (a.cols + nt * 8 - 1) / (nt * 8);
#define MQ_ROW_CASE(V, N)                                                                                              \
    if (vpt <= V && nt == N)                                                                                           \
    {                                                                                                                  \
        hipLaunchKernelGGL((row_quant_kernel<T, OP, V, N>), dim3(a.rows), dim3(N), 0, stream, a);                     \
        MQ_HIP(hipGetLastError());                                                                                     \
        return;                                                                                                        \
    }
    MQ_ROW_CASE(1, 128) MQ_ROW_CASE(2, 128) MQ_ROW_CASE(3, 128) MQ_ROW_CASE(4, 128) MQ_ROW_CASE(6, 128) MQ_ROW_CASE(8, 128)
    MQ_ROW_CASE(6, 256) MQ_ROW_CASE(8, 256) MQ_ROW_CASE(12, 256) MQ_ROW_CASE(16, 256)
#undef MQ_ROW_CASE
    MQ_CHECK(false, "row kernel: %d columns exceed 32768", a.cols);
}

// any hidden size: scalar, re-reads the row (three or four passes served by L2)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void norm_quant_generic_kernel(NormArgs a)
{
    __shared__ float sm[4];
    const size_t row = blockIdx.x;
    const size_t base = row * a.hidden;
    float s1 = 0.f, s2 = 0.f;
    for (int i = threadIdx.x; i < a.hidden; i += 256)
    {
        const float v = NormIO<T>::load1(a.input, base + i);
        s1 += v;
        s2 += v * v;
    }
    float mean = 0.f, rstd;
    if (MODE == 0)
        rstd = rsqrtf(block_reduce<false>(s2, sm) / a.hidden + a.eps);
    else if (MODE == 2)
    {
        mean = block_reduce<false>(s1, sm) / a.hidden;
        rstd = rsqrtf(block_reduce<false>(s2, sm) / a.hidden - mean * mean + a.eps);
    }
    else
    {
        mean = block_reduce<false>(s1, sm) / a.hidden;
        float v = 0.f;
        for (int i = threadIdx.x; i < a.hidden; i += 256)
        {
            const float d = NormIO<T>::load1(a.input, base + i) - mean;
            v += d * d;
        }
        rstd = rsqrtf(block_reduce<false>(v, sm) / a.hidden + a.eps);
    }
    auto normed = [&](int i)
    {
        float v = (MODE == 0 ? NormIO<T>::load1(a.input, base + i) : NormIO<T>::load1(a.input, base + i) - mean) * rstd
            * NormIO<T>::load1(a.gamma, i);
        if (a.beta)
            v = v + NormIO<T>::load1(a.beta, i);
        return NormIO<T>::round(v);
    };
    float scale = 0.f;
    if (a.scale_per_token)
    {
        float amax = NormIO<T>::round(1e-6f);
        for (int i = threadIdx.x; i < a.hidden; i += 256)
            amax = fmaxf(amax, fabsf(normed(i)));
        amax = block_reduce<true>(amax, sm);
        scale = 127.f / amax;
        if (threadIdx.x == 0)
            a.scale_per_token[row] = amax / 127.f;
    }
    else if (a.scale_per_tensor)
        scale = a.scale_per_tensor[0];
    for (int i = threadIdx.x; i < a.hidden; i += 256)
    {
        const float v = normed(i);
        if (a.scale_per_token || a.scale_per_tensor)
            a.out_quant[base + i] = (int8_t) quant_rni_sat(v * scale);
        else
            NormIO<T>::store1(a.out, base + i, v);
    }
}

template <typename T, int MODE>
void launch_norm(const NormArgs& a, hipStream_t stream)
{
    const dim3 grid(a.tokens), block(256);
    // threads per row: 128 up to 8192 columns, 256 above (<= 8 chunks of 8 elements per thread).  Measured at
    // 16384 x 4096 fp16, dynamic scale: 64 threads 49.0 us, 128 threads 43.8 us, 256 threads 46.6 us -- the row
    // width hardly matters, and neither did halving the VALU work of the quantisation (a plain torch copy of the same
    // footprint reaches 5.8 TB/s, this kernel 4.6)
    static const int lab_nt = lab_env("MI355Q_NORM_NT") ? atoi(lab_env("MI355Q_NORM_NT")) : 0;
    const int nt = lab_nt ? lab_nt : (a.hidden <= 8192 ? 128 : 256);
    const int vpt = (a.hidden + nt * 8 - 1) / (nt * 8);
    if (a.hidden % 8 != 0 || a.hidden > 16384 || vpt > 8)
        hipLaunchKernelGGL((norm_quant_generic_kernel<T, MODE>), grid, block, 0, stream, a);
    else
    {
#define MQ_NORM_CASE(V, N)                                                                                             \
    if (vpt <= V && nt == N)                                                                                           \
    {                                                                                                                  \
        hipLaunchKernelGGL((norm_quant_kernel<T, MODE, V, N>), grid, dim3(N), 0, stream, a);                          \
        MQ_HIP(hipGetLastError());                                                                                     \
        return;                                                                                                        \
    }
        MQ_NORM_CASE(1, 64) MQ_NORM_CASE(2, 64) MQ_NORM_CASE(3, 64) MQ_NORM_CASE(4, 64) MQ_NORM_CASE(6, 64) MQ_NORM_CASE(8, 64)
        MQ_NORM_CASE(1, 128) MQ_NORM_CASE(2, 128) MQ_NORM_CASE(3, 128) MQ_NORM_CASE(4, 128) MQ_NORM_CASE(6, 128) MQ_NORM_CASE(8, 128)
        MQ_NORM_CASE(1, 256) MQ_NORM_CASE(2, 256) MQ_NORM_CASE(3, 256) MQ_NORM_CASE(4, 256) MQ_NORM_CASE(6, 256) MQ_NORM_CASE(8, 256)
#undef MQ_NORM_CASE
        MQ_CHECK(false, "norm: no kernel for hidden %d with %d threads per row", a.hidden, nt);
    }
    MQ_HIP(hipGetLastError());
}

template <int MODE>
void dispatch_norm(const NormArgs& a, int type, hipStream_t stream)
{
    switch (type)
    {
    case MI355Q_FLOAT: launch_norm<float, MODE>(a, stream); break;
    case MI355Q_HALF: launch_norm<f16_t, MODE>(a, stream); break;
    case MI355Q_BF16: launch_norm<bf16_t, MODE>(a, stream); break;
    default: MQ_CHECK(false, "norm: unsupported data type %d (float, half, bf16)", type);
    }
}

NormArgs make_norm_args(void* out, const void* input, const void* gamma, const void* beta, float eps, int tokens,
    int hidden, const float* scale, float* dynamic_scale, int8_t* out_quant)
{
    MQ_CHECK(input && gamma, "null input / gamma");
    MQ_CHECK(tokens >= 0 && hidden > 0, "tokens %d hidden %d", tokens, hidden);
    if (scale || dynamic_scale)
        MQ_CHECK(out_quant, "a quantised output buffer is needed with scale / dynamic_scale");
    else
        MQ_CHECK(out, "null output");
    NormArgs a{};
    a.input = input;
    a.gamma = gamma;
    a.beta = beta;
    a.out = out;
    a.out_quant = out_quant;
    a.scale_per_tensor = dynamic_scale ? nullptr : scale; // per-token scaling wins, generalRmsNorm :110-126
    a.scale_per_token = dynamic_scale;
    a.eps = eps;
    a.tokens = tokens;
    a.hidden = hidden;
    return a;
}

template <typename T>
__global__ __launch_bounds__(256) void quantize_tensor_kernel(int8_t* dst, const void* src, int64_t chunks, const float* scale_ptr)
{
    const float scale = scale_ptr[0];
    for (int64_t c = (int64_t) blockIdx.x * 256 + threadIdx.x; c < chunks; c += (int64_t) gridDim.x * 256)
    {
        float x[8];
        NormIO<T>::load8(src, (size_t) c * 8, x);
        *reinterpret_cast<uint2*>(dst + c * 8)
            = make_uint2(quant4_rni_sat(x[0] * scale, x[1] * scale, x[2] * scale, x[3] * scale),
                quant4_rni_sat(x[4] * scale, x[5] * scale, x[6] * scale, x[7] * scale));
    }
}

template <typename T>
__global__ void quantize_tensor_tail_kernel(int8_t* dst, const void* src, int64_t from, int64_t size, const float* scale_ptr)
{
    const int64_t i = from + threadIdx.x;
    if (i < size)
        dst[i] = (int8_t) quant_rni_sat(NormIO<T>::load1(src, (size_t) i) * scale_ptr[0]);
}

template <typename T>
void launch_quantize_tensor(int8_t* dst, const void* src, int64_t size, const float* scale, hipStream_t stream)
{
    const int64_t chunks = size / 8;
    if (chunks > 0)
    {
        const int blocks = (int) std::min<int64_t>((chunks + 255) / 256, 256 * 32);
        hipLaunchKernelGGL(quantize_tensor_kernel<T>, dim3(blocks), dim3(256), 0, stream, dst, src, chunks, scale);
    }
    if (size % 8)
        hipLaunchKernelGGL(quantize_tensor_tail_kernel<T>, dim3(1), dim3(64), 0, stream, dst, src, chunks * 8, size, scale);
    MQ_HIP(hipGetLastError());
}

} // namespace

bool row_quant_supported(int64_t cols)
{
    return cols % 8 == 0 && cols > 0 && cols <= 32768;
}

void per_token_quantization_rows(int8_t* dst, const void* src, int64_t rows, int64_t cols, float* scales, int type, hipStream_t stream)
{
    RowQuantArgs a{};
    a.x = src;
    a.out_quant = dst;
    a.scale_per_token = scales;
    a.rows = (int) rows;
    a.cols = (int) cols;
    a.ld_x = cols;
    switch (type)
    {
    case MI355Q_FLOAT: launch_row_quant<float, 0>(a, stream); break;
    case MI355Q_HALF: launch_row_quant<f16_t, 0>(a, stream); break;
    case MI355Q_BF16: launch_row_quant<bf16_t, 0>(a, stream); break;
    default: MQ_CHECK(false, "per-token quantisation of dtype %d", type);
    }
}

void gated_silu(const mi355q_gated_act_params& p, hipStream_t stream)
{
    MQ_CHECK(p.rows >= 0 && p.cols > 0, "rows %d cols %d", p.rows, p.cols);
    if (p.rows == 0)
        return;
    MQ_CHECK(p.act_in && p.gate_in, "null input");
    MQ_CHECK(p.cols % 8 == 0 && p.cols <= 32768, "cols must be a multiple of 8 and <= 32768, got %d", p.cols);
    MQ_CHECK(p.ld_act >= p.cols && p.ld_gate >= p.cols && p.ld_act % 8 == 0 && p.ld_gate % 8 == 0,
        "row strides must be >= cols and multiples of 8 elements");
    if (p.scale || p.dynamic_scale)
        MQ_CHECK(p.out_quant, "a quantised output buffer is needed with scale / dynamic_scale");
    else
        MQ_CHECK(p.out, "null output");
    RowQuantArgs a{};
    a.x = p.act_in;
    a.gate = p.gate_in;
    a.smoother = p.smoother;
    a.out = p.out;
    a.out_quant = (p.scale || p.dynamic_scale) ? p.out_quant : nullptr;
    a.scale_per_tensor = p.dynamic_scale ? nullptr : p.scale;
    a.scale_per_token = p.dynamic_scale;
    a.rows = p.rows;
    a.cols = p.cols;
    a.ld_x = p.ld_act;
    a.ld_gate = p.ld_gate;
    switch (p.type)
    {
    case MI355Q_FLOAT: launch_row_quant<float, 1>(a, stream); break;
    case MI355Q_HALF: launch_row_quant<f16_t, 1>(a, stream); break;
    case MI355Q_BF16: launch_row_quant<bf16_t, 1>(a, stream); break;
    default: MQ_CHECK(false, "gated activation: unsupported data type %d (float, half, bf16)", p.type);
    }
}

void rms_norm(void* out, const void* input, const void* gamma, const void* beta, float eps, int tokens, int hidden,
    const float* scale, float* dynamic_scale, int8_t* out_quant, int type, hipStream_t stream)
{
    if (tokens == 0) // empty batch: the tensors may have no storage
        return;
    const NormArgs a = make_norm_args(out, input, gamma, beta, eps, tokens, hidden, scale, dynamic_scale, out_quant);
    dispatch_norm<0>(a, type, stream);
}

void layer_norm(void* out, const void* input, const void* gamma, const void* beta, float eps, int tokens, int hidden,
    bool use_diff_of_squares, const float* scale, float* dynamic_scale, int8_t* out_quant, int type, hipStream_t stream)
{
    if (tokens == 0)
        return;
    const NormArgs a = make_norm_args(out, input, gamma, beta, eps, tokens, hidden, scale, dynamic_scale, out_quant);
    if (use_diff_of_squares)
        dispatch_norm<2>(a, type, stream);
    else
        dispatch_norm<1>(a, type, stream);
}

void quantize_tensor(int8_t* dst, const void* src, int64_t size, const float* scale, int type, hipStream_t stream)
{
    MQ_CHECK(size % 4 == 0, "[ERROR][invokeQuantization] size should be a multiple of 4.");
    if (size == 0)
        return;
    MQ_CHECK(dst && src && scale, "null argument");
    switch (type)
    {
    case MI355Q_FLOAT: launch_quantize_tensor<float>(dst, src, size, scale, stream); break;
    case MI355Q_HALF: launch_quantize_tensor<f16_t>(dst, src, size, scale, stream); break;
    case MI355Q_BF16: launch_quantize_tensor<bf16_t>(dst, src, size, scale, stream); break;
    default: MQ_CHECK(false, "quantize_tensor: unsupported data type %d", type);
    }
}

} // namespace mi355q
