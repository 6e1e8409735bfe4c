// decode_common.h -- shared by the decode-attention kernels (decode_attn.hip: VALU kernel for MHA and
// small GQA groups; decode_attn_gqa.hip: MFMA kernel for GQA groups of 4..16 query heads).
#pragma once
#include "kernels.h"
#include <cfloat>

namespace mi355q
{
namespace
{

constexpr int NW = 4; // waves per workgroup
constexpr float NEG_BIG = -1e30f;

struct DecodeArgs
{
    const void* qkv;
    const void* qkv_bias;
    void* out;
    const int32_t* seq_lens;
    const int64_t* block_ptrs;
    const float* kv_scale_orig_quant;
    const float* kv_scale_quant_orig;
    float* part_acc; // [B, Hq, splits, DH]
    float* part_ml;  // [B, Hq, splits, 2]
    int batch, num_heads, num_kv_heads;
    int max_blocks, tpb_log2;
    int splits;
    int pos_type, rot_dim, rot_scale_type, rot_max_pos;
    float rot_base, rot_scale;
    float score_scale; // 1 / (sqrt(Dh) * q_scaling)
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

template <typename T>
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c);
template <>
__device__ __forceinline__ float dot2<bf16_t>(uint32_t a, uint32_t b, float c)
{
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
template <>
__device__ __forceinline__ float dot2<f16_t>(uint32_t a, uint32_t b, float c)
{
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
}

// sum over the LPT lanes that hold one token row; every lane of the group ends with the total.
template <int LPT>
__device__ __forceinline__ float group_sum(float v)
{
#define MQ_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    if (LPT >= 2)
        MQ_DPP_ADD(0xB1); // quad_perm [1,0,3,2]
    if (LPT >= 4)
        MQ_DPP_ADD(0x4E); // quad_perm [2,3,0,1]
    if (LPT >= 8)
        MQ_DPP_ADD(0x141); // row_half_mirror
    if (LPT >= 16)
        MQ_DPP_ADD(0x140); // row_mirror
#undef MQ_DPP_ADD
    if (LPT >= 32)
        v += __shfl_xor(v, 16);
    if (LPT >= 64)
        v += __shfl_xor(v, 32);
    return v;
}

// KVT: 0 = cache holds T, 1 = int8, 2 = fp8 e4m3 (OCP)
template <int KVT>
__device__ __forceinline__ float kv8_to_float(uint32_t w, int byte);
template <>
__device__ __forceinline__ float kv8_to_float<1>(uint32_t w, int byte)
{
    return (float) (int8_t) (w >> (8 * byte));
}
template <>
__device__ __forceinline__ float kv8_to_float<2>(uint32_t w, int byte)
{
    switch (byte)
    {
    case 0: return __builtin_amdgcn_cvt_f32_fp8((int) w, 0);
    case 1: return __builtin_amdgcn_cvt_f32_fp8((int) w, 1);
    case 2: return __builtin_amdgcn_cvt_f32_fp8((int) w, 2);
    default: return __builtin_amdgcn_cvt_f32_fp8((int) w, 3);
    }
}

__device__ __forceinline__ uint8_t float_to_int8_sat(float x)
{
    // cvt.rni.sat.s8.f32 (decoderMaskedMultiheadAttentionUtils.h:3629-3639)
    float r = rintf(x);
    r = fminf(fmaxf(r, -128.f), 127.f);
    return (uint8_t) (int8_t) (int) r;
}

__device__ __forceinline__ uint8_t float_to_fp8(float x)
{
    // OCP e4m3fn, RNE, saturating to +-448 like __nv_fp8_e4m3's default conversion
    x = fminf(fmaxf(x, -448.f), 448.f);
    return (uint8_t) (__builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false) & 0xff);
}

// Stage q (G heads), k, v of the new token into LDS as floats already rounded to T: bias add in T,
// RoPE in fp32 (decoderMaskedMultiheadAttentionTemplate.h:1391-1527).  smem: [(G + 2) * DH] floats
// = q[G][DH], k[DH], v[DH].  Ends with a barrier.
template <typename T, int DH, int G>
__device__ __forceinline__ void stage_new_token(const DecodeArgs& p, int b, int hk, int tl, float* smem)
{
    const int tid = threadIdx.x;
    const int Hq = p.num_heads, Hkv = p.num_kv_heads;
    // ---- stage q (G heads), k, v of the new token: bias add in T, RoPE in fp32, round to T
    {
        const uint16_t* qkv = reinterpret_cast<const uint16_t*>(p.qkv) + (size_t) b * (Hq + 2 * Hkv) * DH;
        const uint16_t* bias = reinterpret_cast<const uint16_t*>(p.qkv_bias);
        // 16-byte loads, all of them issued before the first use: the scalar form (one 2-byte load per trip of a
        // 5-trip loop, each waited for) cost five cold HBM round trips -- most of a workgroup's fixed ~8 us
        constexpr int NV = (G + 2) * DH / 8;                     // 16-byte vectors of q (G heads), k, v
        constexpr int TRIPS = (NV + NW * 64 - 1) / (NW * 64);
        uint4 xv[TRIPS], bv[TRIPS];
        int offv[TRIPS];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t)
        {
            const int v = min(tid + t * NW * 64, NV - 1); // past the end: a valid vector, not stored
            const int vec = (v * 8) / DH, d = (v * 8) % DH;
            offv[t] = vec < G ? (hk * G + vec) * DH + d : (vec == G ? (Hq + hk) * DH + d : (Hq + Hkv + hk) * DH + d);
            xv[t] = *reinterpret_cast<const uint4*>(qkv + offv[t]);
            bv[t] = make_uint4(0, 0, 0, 0);
        }
        if (bias)
        {
#pragma unroll
            for (int t = 0; t < TRIPS; ++t)
                bv[t] = *reinterpret_cast<const uint4*>(bias + offv[t]);
        }
#pragma unroll
        for (int t = 0; t < TRIPS; ++t)
        {
            const int v = tid + t * NW * 64;
            if (v < NV)
            {
                const uint32_t xw[4] = {xv[t].x, xv[t].y, xv[t].z, xv[t].w};
                const uint32_t bw[4] = {bv[t].x, bv[t].y, bv[t].z, bv[t].w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                {
                    float x = (j & 1) ? T16<T>::hi(xw[j >> 1]) : T16<T>::lo(xw[j >> 1]);
                    if (bias)
                    {
                        const float bb = (j & 1) ? T16<T>::hi(bw[j >> 1]) : T16<T>::lo(bw[j >> 1]);
                        x = T16<T>::to_float(T16<T>::from_float(x + bb));
                    }
                    smem[v * 8 + j] = x;
                }
            }
        }
        // cos / sin of the token's angles depend on the pair index only: one table per block (rot_dim / 2 entries)
        // instead of powf + sincosf for every (head, pair) -- with 8 query heads per kv head that was 9x the
        // transcendental work, in every split of the sequence
        __shared__ float rope_tab[DH];
        const bool rope = p.pos_type == 1 || p.pos_type == 2; // kROPE_GPTJ / kROPE_GPT_NEOX
        const int half = p.rot_dim >> 1;
        if (rope && tid < half)
        {
            float base = p.rot_base, scale = p.rot_scale;
            // update_rotary_base_n_scale (decoderMaskedMultiheadAttentionUtils.h:2552-2579)
            if (p.rot_scale_type == 2)
            {
                if (tl > p.rot_max_pos)
                {
                    float bb = tl * (scale / p.rot_max_pos) - (scale - 1.f);
                    base = base * powf(bb, (float) p.rot_dim / (float) (p.rot_dim - 2));
                }
                scale = 1.f;
            }
            else if (p.rot_scale_type == 1)
                scale = 1.f / scale;
            const float ang = ((float) tl * scale) / powf(base, (float) (2 * tid) / (float) p.rot_dim);
            float sn, cs;
            sincosf(ang, &sn, &cs);
            rope_tab[2 * tid] = cs;
            rope_tab[2 * tid + 1] = sn;
        }
        __syncthreads();
        if (rope)
        {
            for (int i = tid; i < (G + 1) * half; i += NW * 64)
            {
                int vec = i / half, j = i % half;
                float* x = smem + vec * DH;
                int i0 = p.pos_type == 2 ? j : 2 * j;
                int i1 = p.pos_type == 2 ? j + half : 2 * j + 1;
                const float cs = rope_tab[2 * j], sn = rope_tab[2 * j + 1];
                float a = x[i0], c = x[i1];
                x[i0] = T16<T>::to_float(T16<T>::from_float(cs * a - sn * c));
                x[i1] = T16<T>::to_float(T16<T>::from_float(cs * c + sn * a));
            }
            __syncthreads();
        }
    }
}

} // namespace
} // namespace mi355q
