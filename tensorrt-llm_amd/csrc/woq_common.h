// woq_common.h -- device helpers shared by the weight-only kernels (woq_gemm.hip: decode GEMV,
// dequantise, split-K reduce; woq_prefill.hip: prefill GEMM).  Everything lives in an anonymous
// namespace: each translation unit gets its own copy.
#pragma once
#include "kernels.h"

namespace mi355q
{
namespace
{

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;


struct WoqArgs
{
    const uint8_t* qweight;
    const uint16_t* scales;
    const uint16_t* zeros;
    const uint16_t* act;
    const uint16_t* act_scale; // per-channel pre-quant scale [k] applied while the decode kernel stages the rows, or null
    const uint16_t* bias;
    const float* alpha; // device [1] or null: out = alpha * acc + bias (the W4A8 alpha, fpA_intB groupwise gemm overload)
    uint16_t* out;
    float* slabs; // [ksplit][m][n] fp32 partials when ksplit > 1
    size_t slab_stride; // elements between two slabs in the prefill kernel (m * n; grouped GEMM: total rows * n)
    int m, n, k;
    int group;  // group size in k; == k for per-channel
    int group_shift; // log2(group); 31 for per-channel (k >> 31 == 0)
    int ktiles; // full k-tiles: k / KTILE
    int ksplit;
    int red_buffers; // decode kernel: cross-wave reduce buffers in rotation (2: one barrier per n-block; 1: two)
    // grouped (MoE) form of the prefill kernel: rows sorted by expert, moe_rows[e] = rows of experts 0..e (inclusive
    // cumulative, total_rows_before_expert of moe_kernels.cu); every expert has its own weights / scales / bias
    const int64_t* moe_rows; // device [moe_experts] or null
    int moe_experts;
    size_t moe_wbytes;       // bytes of one expert's weight matrix
    int act_fn;              // epilogue activation (ActivationType of moe_gemm_kernels.h: 0 gelu, 1 relu, 2 silu, 5 identity)
    // grouped form, rows gathered on the fly (run_moe: no expandInputRows pass): row r of the expert-sorted order is row
    // moe_row_map[r] % moe_row_mod of `act`; null = `act` is already in expert order
    const int32_t* moe_row_map;
    int moe_row_mod;
    // the grouped launch in two parts (256-row tiles): moe_full_only = this launch takes an expert's whole TM-row tiles
    // only; moe_tail_mod = 256: this launch takes what such a launch with 256-row tiles left, rows >= (rows / 256) * 256
    // of every expert, in its own (shorter) tiles, 256 / TM of them per expert
    int moe_full_only, moe_tail_mod;
    // prefill kernel: tiles are walked in groups of pf_group_m m-tiles x all n-tiles, m fastest inside a group
    // (0: m fastest over the whole launch)
    int pf_group_m;
    // gated activation fused into the epilogue (run_moe's w1 GEMM; doGatedActivationKernel, moe_kernels.cu:795-817): the
    // n columns are [linear | gate] halves of glu_inter columns each; a wave takes matching n-blocks of both halves and
    // stores T(T(linear) * T(act(T(gate)))) to out[m][glu_inter] -- no [m][2 inter] buffer, no second pass.  0 = off;
    // glu_fn: woq_activation's code of the gate's activation (2 SiLU for Swiglu, 0 GELU for Geglu)
    int glu_inter, glu_fn;
};

// moeGemmBiasAct's epilogues (cutlass_extensions/epilogue_helpers.h: LinearCombination{Relu, Silu, Generic<GELU_taylor>})
__device__ __forceinline__ float woq_activation(float x, int fn)
{
    switch (fn)
    {
    case 0: return 0.5f * x * (1.f + tanhf(0.7978845608028654f * x * (1.f + 0.044715f * x * x))); // GELU_taylor
    case 1: return fmaxf(x, 0.f);
    case 2: return x / (1.f + __expf(-x));
    default: return x;
    }
}

template <typename T>
__device__ __forceinline__ f32x4_t mfma16(const uint4& a, const uint4& b, f32x4_t c);
template <>
__device__ __forceinline__ f32x4_t mfma16<bf16_t>(const uint4& a, const uint4& b, f32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(
        __builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4_t mfma16<f16_t>(const uint4& a, const uint4& b, f32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(
        __builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

__device__ __forceinline__ float sbyte(uint32_t w, int i)
{
    return (float) (int8_t) (w >> (8 * i)); // v_cvt_f32_i32_sdwa sext(BYTE_i)
}

// GFX950 int4 dword (preprocess.cpp Gfx950Map): element j of the lane's 8-row fragment is stored
// biased (u = q + 8) at nibble j < 4 ? 2j : 2(j-4)+1, i.e. `w & 0x0f0f0f0f` holds elements 0..3 as
// bytes and `(w >> 4) & 0x0f0f0f0f` elements 4..7.
//
// Exact dequantisation: 8 int4 -> 8 values round16(fma(q, scale, zero)) (4 packed dwords).
// After the xor the nibbles are two's complement again; lo/hi bytes carry 16*q exactly, so the fma
// uses scale/16 (a power-of-two rescale: exact).
template <typename T>
__device__ __forceinline__ uint4 dequant8_int4(uint32_t w, float s16, float z)
{
    w ^= 0x88888888u;
    const uint32_t lo = (w << 4) & 0xf0f0f0f0u; // byte i = 16 * q[i]
    const uint32_t hi = w & 0xf0f0f0f0u;        // byte i = 16 * q[4+i]
    uint4 r;
    r.x = pack2<T>(fmaf(sbyte(lo, 0), s16, z), fmaf(sbyte(lo, 1), s16, z));
    r.y = pack2<T>(fmaf(sbyte(lo, 2), s16, z), fmaf(sbyte(lo, 3), s16, z));
    r.z = pack2<T>(fmaf(sbyte(hi, 0), s16, z), fmaf(sbyte(hi, 1), s16, z));
    r.w = pack2<T>(fmaf(sbyte(hi, 2), s16, z), fmaf(sbyte(hi, 3), s16, z));
    return r;
}

// Integer-domain unpack for the decode GEMV: 8 biased int4 -> the 8 integers u = 0..15 as exact
// 16-bit floats.  A byte 0x00..0x0f read as OCP e4m3 is u * 2^-9 (denormals 0..7, then exponent 1),
// so v_cvt_scalef32_pk_{bf16,f16}_fp8 with scale 2^9 converts two nibbles per instruction:
// 3 mask/shift + 4 convert VALU ops per 8 weights instead of 23 for the exact dequantisation.
template <typename T>
__device__ __forceinline__ uint32_t cvt2_u4(uint32_t bytes, bool hi_word);
template <>
__device__ __forceinline__ uint32_t cvt2_u4<bf16_t>(uint32_t bytes, bool hi_word)
{
    return hi_word ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(bytes, 512.f, true))
                   : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(bytes, 512.f, false));
}
template <>
__device__ __forceinline__ uint32_t cvt2_u4<f16_t>(uint32_t bytes, bool hi_word)
{
    return hi_word ? __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(bytes, 512.f, true))
                   : __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(bytes, 512.f, false));
}

template <typename T>
__device__ __forceinline__ uint4 unpack8_u4(uint32_t w)
{
    const uint32_t lo = w & 0x0f0f0f0fu, hi = (w >> 4) & 0x0f0f0f0fu;
    uint4 r;
    r.x = cvt2_u4<T>(lo, false);
    r.y = cvt2_u4<T>(lo, true);
    r.z = cvt2_u4<T>(hi, false);
    r.w = cvt2_u4<T>(hi, true);
    return r;
}

// sixteen 1.0s: the B operand that turns an MFMA into the row sums of A
template <typename T>
__device__ __forceinline__ uint4 ones8()
{
    constexpr uint32_t one2 = T16<T>::is_bf16 ? 0x3f803f80u : 0x3c003c00u;
    return make_uint4(one2, one2, one2, one2);
}

// 8 signed int8 (two dwords) -> 8 dequantised 16-bit values
template <typename T>
__device__ __forceinline__ uint4 dequant8_int8(uint32_t w0, uint32_t w1, float s, float z)
{
    uint4 r;
    r.x = pack2<T>(fmaf(sbyte(w0, 0), s, z), fmaf(sbyte(w0, 1), s, z));
    r.y = pack2<T>(fmaf(sbyte(w0, 2), s, z), fmaf(sbyte(w0, 3), s, z));
    r.z = pack2<T>(fmaf(sbyte(w1, 0), s, z), fmaf(sbyte(w1, 1), s, z));
    r.w = pack2<T>(fmaf(sbyte(w1, 2), s, z), fmaf(sbyte(w1, 3), s, z));
    return r;
}

// 8 signed int8 (two dwords) -> the 8 integers as exact 16-bit floats (|q| <= 128 fits either mantissa): the fragment
// of the deferred-scale form, whose per-column scale meets the accumulator in the epilogue
template <typename T>
__device__ __forceinline__ uint4 cvt8_i8(uint32_t w0, uint32_t w1)
{
    uint4 r;
    r.x = pack2<T>(sbyte(w0, 0), sbyte(w0, 1));
    r.y = pack2<T>(sbyte(w0, 2), sbyte(w0, 3));
    r.z = pack2<T>(sbyte(w1, 0), sbyte(w1, 1));
    r.w = pack2<T>(sbyte(w1, 2), sbyte(w1, 3));
    return r;
}

template <int WBITS>
struct Geo
{
    static constexpr int KTILE = WBITS == 4 ? 128 : 64;
    static constexpr int KSTEPS = KTILE / 32;      // MFMA k-steps per tile (4 or 2)
    static constexpr int NSZ = WBITS == 4 ? 2 : 1; // (scale, zero) pairs a tile may span (group 64 on int4)
};

// k-step ST of the lane fragment `w` -> dequantised B operand.  s / z: this k-step's group.
template <typename T, int WBITS, int ST>
__device__ __forceinline__ uint4 dequant_step(const uint4& w, float s, float z)
{
    const uint32_t* wd = reinterpret_cast<const uint32_t*>(&w);
    if constexpr (WBITS == 4)
        return dequant8_int4<T>(wd[ST], s * 0.0625f, z);
    else
        return dequant8_int8<T>(wd[ST * 2], wd[ST * 2 + 1], s, z);
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// SRD over `bytes` bytes at p (wave-uniform inputs only, so hipcc never wraps the loads in a
// waterfall loop); out-of-range offsets return 0.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, size_t bytes)
{
    const uint64_t u = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (u >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t) (bytes > 0xffffffffull ? 0xffffffffull : bytes));
    return __builtin_amdgcn_make_buffer_rsrc((void*) (((uint64_t) hi << 32) | lo), 0, nb, 0x00020000);
}

} // namespace
} // namespace mi355q
