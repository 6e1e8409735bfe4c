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
    // split-KV (the reference's multi-block mode, decoderMaskedMultiheadAttentionTemplate.h:2614-2786): one group of
    // records per (sequence, kv head): [splits][G][DH + 4] floats (acc | m | l | pad), group stride `group_floats` (a multiple
    // of 32 floats: no 128-byte line is shared between groups), and one arrival word per group (arrival_ring.hip:
    // zero at launch, ARRIVAL_WORD_STRIDE words apart)
    float* partials;
    uint32_t* arrivals;
    int group_floats;
    int batch, num_heads, num_kv_heads;
    int max_blocks, tpb_log2;
    int splits;
    // cache addressing (kvCacheUtils.h:36-153, 241-338).  Slots read: [0, min(kv length, kv_window)); the new token goes
    // to kv_slot(kv length).  Without a window these are the identity (kv_window = capacity, sink = bubble = 0).
    int kv_window, sink, bubble, cyc_len;
    int head_tokens;          // tokens of one head inside a block: tokens_per_block, or linear_max_seq_len
    const char* linear_base;  // KVLinearBuffer::data or null; one (sequence, K|V) plane is linear_plane_bytes
    size_t linear_plane_bytes;
    // the general kernel only (one workgroup per QUERY head): beams, any q / kv head ratio
    int q_per_kv, beam_width, indir_stride;
    const int32_t* cache_indir;
    const int32_t* input_lengths;
    int hs;                       // the head size in memory when a kernel runs a head padded to its DH (PAD form): hs <= DH
    const uint16_t* alibi_slopes; // T [Hq] or null (decode_attn_kernel only): score += slope[h] * (t - tlength), Template.h:2026
    int pos_type, rot_dim, rot_scale_type, rot_max_pos;
    float rot_base, rot_scale;
    float score_scale; // 1 / (sqrt(Dh) * q_scaling)
    // POS_SHIFT (general kernel only): keys are read from `shift_k`, this step's rotated copy of the K cache in the
    // cache's own block structure ([sequence][block] blocks of shift_block_bytes), at the slot of the token that holds
    // window position t; q / the new k are rotated to position kv_len; the new k is NOT appended here (the shift
    // kernel stores it un-rotated)
    const char* shift_k;
    size_t shift_block_bytes;
    // relative attention bias (decode_attn_kernel only): T [Hq][stride][stride] (max_distance == 0) or the bucket table
    // T [Hq][stride] (Template.h:1773-1797, 1960-1985)
    const uint16_t* rel_bias;
    int rel_stride, max_distance;
#ifdef MI355Q_LAB
    unsigned long long* stamps; // lab: [workgroup][16] s_memrealtime stamps (100 MHz), or null
#endif
};

#ifdef MI355Q_LAB
#define MQ_STAMP(p, i)                                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
        if ((p).stamps && threadIdx.x == 0)                                                                            \
            (p).stamps[(size_t) blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                            \
    } while (0)
#else
#define MQ_STAMP(p, i)                                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
    } while (0)
#endif

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

// the cyclic slot map of a sequence's cache: KVBlockArray ctor (kvCacheUtils.h:71-94), KVLinearBuffer ctor (:277-295)
struct CyclicLayout
{
    int window, sink, bubble, cyc_len;
};
inline CyclicLayout cyclic_layout(bool linear, int capacity, int max_blocks, int tokens_per_block, int cyclic_window, int sink)
{
    CyclicLayout c{};
    c.window = cyclic_window > 0 ? cyclic_window : capacity;
    c.sink = sink;
    MQ_CHECK(c.sink >= 0 && c.sink < c.window, "sink_token_length %d against a window of %d", c.sink, c.window);
    if (linear)
    {
        c.bubble = 0;
        c.cyc_len = c.window - c.sink;
    }
    else
    {
        const int in_last = c.sink % tokens_per_block;
        c.bubble = in_last == 0 ? 0 : tokens_per_block - in_last;
        const bool one_more = (max_blocks - 1) * tokens_per_block >= c.window + c.bubble;
        c.cyc_len = one_more ? c.window + tokens_per_block - c.sink : c.window - c.sink;
    }
    MQ_CHECK(c.sink + c.bubble + c.cyc_len <= capacity, "the cyclic cache (sink %d + bubble %d + %d slots) exceeds the %d slots of a sequence",
        c.sink, c.bubble, c.cyc_len, capacity);
    MQ_CHECK(c.window <= capacity, "attention window %d exceeds the %d cache slots of a sequence", c.window, capacity);
    return c;
}

// KVBlockArray / KVLinearBuffer::getKVTokenIdx (kvCacheUtils.h:109-117, 308-316): the cache slot of token t
__device__ __forceinline__ int kv_slot(const DecodeArgs& p, int t)
{
    return t < p.sink ? t : p.sink + p.bubble + (t - p.sink) % p.cyc_len;
}

// index into the relative-attention-bias input for head h, key step t, kv length tl (Template.h:1775-1782 explicit,
// :1960-1977 implicit: the kernel halves num_buckets, takes max_exact = half of that and a logarithmic bucket beyond)
__device__ __forceinline__ size_t rel_bias_index(const DecodeArgs& p, int h, int t, int tl)
{
    if (p.max_distance <= 0)
        return ((size_t) h * p.rel_stride + tl) * p.rel_stride + t;
    const int nb = p.rel_stride / 2;
    int rp = t - tl, rb = rp > 0 ? nb : 0;
    rp = abs(rp);
    const int max_exact = nb / 2;
    int large = max_exact
        + (int) (logf(rp * 1.0f / max_exact) / logf((float) p.max_distance / max_exact) * (nb - max_exact));
    large = min(large, nb - 1);
    rb += rp < max_exact ? rp : large;
    return (size_t) h * p.rel_stride + rb;
}

// POS_SHIFT: the cache slot of the token at window position x when `tl` tokens are cached: the sink tokens keep their
// places, position x >= sink is sequence token cyc_start + (x - sink) with cyc_start the oldest live non-sink token
// (shiftKCache's token_idx, unfusedAttentionKernels.cu:1790-1795; Template.h:1918-1927 reads the same slots)
__device__ __forceinline__ int kv_pos_slot(const DecodeArgs& p, int x, int tl)
{
    if (x < p.sink)
        return x;
    const int cyc_start = tl > p.kv_window ? p.sink + tl - p.kv_window : p.sink;
    return kv_slot(p, cyc_start + (x - p.sink));
}

// base of the K (kv = 0) or V (kv = 1) block `blk` of sequence `seq` (getKBlockPtr / getVBlockPtr, :101-137, :302-331)
__device__ __forceinline__ char* kv_block_ptr(const DecodeArgs& p, int seq, int kv, int blk)
{
    if (p.linear_base)
        return const_cast<char*>(p.linear_base) + ((size_t) seq * 2 + kv) * p.linear_plane_bytes;
    return reinterpret_cast<char*>(p.block_ptrs[((size_t) seq * 2 + kv) * p.max_blocks + blk]);
}

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

// The new token's q (G heads), k, v and their biases as loaded: 16-byte vectors, all of them issued before the first
// use (the scalar form -- one 2-byte load per trip of a 5-trip loop, each waited for -- cost five cold HBM round trips).
// No load sits under a branch (a missing bias re-reads qkv and is ignored): the compiler's vmcnt bookkeeping stays
// exact, so loads the caller issues AFTER these are still in flight while the token is staged.
template <int G, int DH>
struct NewTokenRegs
{
    static constexpr int NV = (G + 2) * DH / 8; // 16-byte vectors of q (G heads), k, v
    static constexpr int TRIPS = (NV + NW * 64 - 1) / (NW * 64);
    uint4 x[TRIPS], bias[TRIPS];
};

// PAD: the head occupies hs = p.hs <= DH elements in memory; the registers / LDS keep the DH layout, elements past hs
// are zero.
template <typename T, int DH, int G, bool PAD = false>
__device__ __forceinline__ void issue_new_token(const DecodeArgs& p, int b, int hq0, int hk, NewTokenRegs<G, DH>& r)
{
    const int tid = threadIdx.x;
    const int Hq = p.num_heads, Hkv = p.num_kv_heads;
    const int hs = PAD ? p.hs : DH;
    const uint16_t* qkv = reinterpret_cast<const uint16_t*>(p.qkv) + (size_t) b * (Hq + 2 * Hkv) * hs;
    const uint16_t* bias = p.qkv_bias ? reinterpret_cast<const uint16_t*>(p.qkv_bias) : qkv;
#pragma unroll
    for (int t = 0; t < NewTokenRegs<G, DH>::TRIPS; ++t)
    {
        const int v = min(tid + t * NW * 64, NewTokenRegs<G, DH>::NV - 1); // past the end: a valid vector, not stored
        const int vec = (v * 8) / DH;
        const int d = PAD ? min((v * 8) % DH, hs - 8) : (v * 8) % DH; // past the head: a valid vector, zeroed when staged
        const int off = vec < G ? (hq0 + vec) * hs + d : (vec == G ? (Hq + hk) * hs + d : (Hq + Hkv + hk) * hs + d);
        r.x[t] = *reinterpret_cast<const uint4*>(qkv + off);
        r.bias[t] = *reinterpret_cast<const uint4*>(bias + off);
    }
}

// base^(2 i / rot_dim) of rotary pair i = threadIdx.x: needs no sequence length unless the base is rescaled by it
// (dynamic NTK), so a kernel computes it while its first loads are in flight (powf is a few hundred instructions).
__device__ __forceinline__ float rope_early_pow(const DecodeArgs& p)
{
    const bool rope = p.pos_type == 1 || p.pos_type == 2;
    if (!rope || p.rot_scale_type == 2 || (int) threadIdx.x >= (p.rot_dim >> 1))
        return 1.f;
    return powf(p.rot_base, (float) (2 * threadIdx.x) / (float) p.rot_dim);
}

// Stage q (G heads), k, v of the new token into LDS as floats already rounded to T: bias add in T,
// RoPE in fp32 (decoderMaskedMultiheadAttentionTemplate.h:1391-1527).  smem: [(G + 2) * DH] floats
// = q[G][DH], k[DH], v[DH].  Ends with a barrier.  The cos / sin table is built before the token's registers are
// touched, i.e. while their loads are still in flight.
template <typename T, int DH, int G, bool PAD = false>
__device__ __forceinline__ void stage_new_token(
    const DecodeArgs& p, const NewTokenRegs<G, DH>& r, int tl, float* smem, float early_pow)
{
    const int tid = threadIdx.x;
    if (p.shift_k) // POS_SHIFT: the position of the current step is the kv loop length (Template.h:1500)
        tl = min(tl, p.kv_window);
    {
        // cos / sin of the token's angles depend on the pair index only: one table per block (rot_dim / 2 entries)
        // instead of powf + sincosf for every (head, pair) -- with 8 query heads per kv head that was 9x the
        // transcendental work, in every split of the sequence
        __shared__ float rope_tab[DH];
        const bool rope = p.pos_type == 1 || p.pos_type == 2; // kROPE_GPTJ / kROPE_GPT_NEOX
        const int half = p.rot_dim >> 1;
        if (rope && tid < half)
        {
            float scale = p.rot_scale;
            float pw = early_pow;
            // update_rotary_base_n_scale (decoderMaskedMultiheadAttentionUtils.h:2552-2579)
            if (p.rot_scale_type == 2)
            {
                float base = p.rot_base;
                if (tl > p.rot_max_pos)
                {
                    float bb = tl * (scale / p.rot_max_pos) - (scale - 1.f);
                    base = base * powf(bb, (float) p.rot_dim / (float) (p.rot_dim - 2));
                }
                scale = 1.f;
                pw = powf(base, (float) (2 * tid) / (float) p.rot_dim);
            }
            else if (p.rot_scale_type == 1)
                scale = 1.f / scale;
            const float ang = ((float) tl * scale) / pw;
            float sn, cs;
            sincosf(ang, &sn, &cs);
            rope_tab[2 * tid] = cs;
            rope_tab[2 * tid + 1] = sn;
        }
        const bool bias = p.qkv_bias != nullptr;
#pragma unroll
        for (int t = 0; t < NewTokenRegs<G, DH>::TRIPS; ++t)
        {
            const int v = tid + t * NW * 64;
            if (v < NewTokenRegs<G, DH>::NV)
            {
                const uint32_t xw[4] = {r.x[t].x, r.x[t].y, r.x[t].z, r.x[t].w};
                const uint32_t bw[4] = {r.bias[t].x, r.bias[t].y, r.bias[t].z, r.bias[t].w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                {
                    float x = (j & 1) ? T16<T>::hi(xw[j >> 1]) : T16<T>::lo(xw[j >> 1]);
                    if (bias)
                    {
                        const float bb = (j & 1) ? T16<T>::hi(bw[j >> 1]) : T16<T>::lo(bw[j >> 1]);
                        x = T16<T>::to_float(T16<T>::from_float(x + bb));
                    }
                    if (PAD && (v * 8) % DH >= p.hs)
                        x = 0.f;
                    smem[v * 8 + j] = x;
                }
            }
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

template <typename T, int DH, int G, bool PAD = false>
__device__ __forceinline__ void stage_new_token(const DecodeArgs& p, int b, int hq0, int hk, int tl, float* smem)
{
    NewTokenRegs<G, DH> r;
    issue_new_token<T, DH, G, PAD>(p, b, hq0, hk, r);
    stage_new_token<T, DH, G, PAD>(p, r, tl, smem, rope_early_pow(p));
}

typedef __attribute__((address_space(1))) uint32_t dec_gu32;
typedef __attribute__((address_space(1))) float dec_gf32;
typedef __attribute__((ext_vector_type(4))) float dec_f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t dec_u32x4;

constexpr int PART_PAD = 4; // a partial record is [DH | m | l | 2 pad] floats: 16-byte aligned rows

__device__ __forceinline__ __amdgpu_buffer_rsrc_t dec_rsrc(const void* p, size_t bytes)
{
    const uint64_t u = (uint64_t) p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t) u);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t) (u >> 32));
    const uint32_t nb = __builtin_amdgcn_readfirstlane((uint32_t) bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*) (((uint64_t) hi << 32) | lo), 0, nb, 0x00020000);
}

// The end of a decode workgroup: merge its NSTATES partial states (LDS `red`: [NSTATES][G][DH + 2] = acc | m | l) and
// (`group` = the workgroups that share the K/V stream, `out_head0` = row of the first output head) either write the output (no split) or take part in the split-KV merge INSIDE the launch (the reference does the same
// with its block_counter, decoderMaskedMultiheadAttentionTemplate.h:2614-2786; a second kernel used to do it here).
// A thread owns 4 consecutive d of one head.  Split-KV: the workgroup publishes its G merged states into its records
// of the (sequence, kv head) group -- 16-byte write-through (sc1) stores --, every storing wave drains vmcnt, barrier,
// ONE lane counts the workgroup on the group's arrival word, and the workgroup whose arrival completes the count
// merges the splits in split order (deterministic) with sc1 loads, all of a thread's loads issued before the first
// use.  Hand-off form: MI355X_MICROARCH.md, Valid forms, row "one lane of each storing workgroup ... the workgroup
// whose add came last".  The arrival word comes from arrival_ring.hip (zero at launch, one per 128-byte line, put
// back to zero by the last arrival's own wrapping increment -- one round trip, no reset pass, graph replay safe).
// `red` is reused as scratch (>= 1 + G * 65 floats).
template <typename T, int G, int DH, int THREADS, int NSTATES, bool PAD = false>
__device__ __forceinline__ void finish_states(const DecodeArgs& p, int group, size_t out_head0, int split, float* red)
{
    constexpr int RS = DH + 2;           // LDS row stride
    constexpr int REC = DH + PART_PAD;   // global record stride (floats)
    const int tid = threadIdx.x;
    const int hs = PAD ? p.hs : DH;      // output row: hs elements per head (columns past hs are padding, not stored)
    const size_t grp_bytes = (size_t) p.group_floats * sizeof(float);
    const __amdgpu_buffer_rsrc_t grp
        = dec_rsrc(p.partials + (size_t) group * p.group_floats, p.splits > 1 ? grp_bytes : 0);
    for (int i = tid; i < G * DH / 4; i += THREADS)
    {
        const int g = i / (DH / 4), d = (i % (DH / 4)) * 4;
        float mm = NEG_BIG;
#pragma unroll
        for (int w = 0; w < NSTATES; ++w)
            mm = fmaxf(mm, red[(w * G + g) * RS + DH]);
        float ll = 0.f;
        dec_f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NSTATES; ++w)
        {
            const float* r = red + (w * G + g) * RS;
            const float c = __builtin_amdgcn_exp2f(r[DH] - mm);
            ll += r[DH + 1] * c;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j] += r[d + j] * c;
        }
        if (p.splits == 1)
        {
            // inv_sum = 1 / (sum + 1e-6)  (decoderMaskedMultiheadAttentionTemplate.h:2285-2290)
            const float inv = 1.f / (ll + 1e-6f);
            const uint2 v = make_uint2(pack2<T>(o[0] * inv, o[1] * inv), pack2<T>(o[2] * inv, o[3] * inv));
            if (!PAD || d < hs)
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (out_head0 + g) * hs + d) = v;
        }
        else
        {
            const uint32_t rec = (uint32_t) ((split * G + g) * REC) * 4;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dec_u32x4, o), grp, rec + d * 4, 0, 16 /* sc1 */);
            if (d == 0)
            {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(mm), grp, rec + DH * 4, 0, 16);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ll), grp, rec + DH * 4 + 4, 0, 16);
            }
        }
    }
    if (p.splits == 1)
        return;
    MQ_STAMP(p, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave, before the barrier the counting lane passes
    __syncthreads();                                 // (also: everyone is done reading `red`)
    MQ_STAMP(p, 7);
    int* last = reinterpret_cast<int*>(red);
    if (tid == 0)
    {
        // global_atomic_inc with wrap value splits - 1: the arrival that finds splits - 1 is the last one, and its
        // own increment puts the word back to zero for the next launch (arrival_ring.hip)
        dec_gu32* word = (dec_gu32*) (p.arrivals + (size_t) group * ARRIVAL_WORD_STRIDE);
        const uint32_t seen = __builtin_amdgcn_atomic_inc32(word, (uint32_t) (p.splits - 1), __ATOMIC_RELAXED, "agent");
        *last = seen == (uint32_t) (p.splits - 1);
    }
    __syncthreads();
    MQ_STAMP(p, 8);
    if (!*last)
        return;
    const int S = p.splits; // <= 64
    if (S <= 8)
    {
        // one round trip: a thread loads the (m, l) pairs and its 4 columns of every split together
        typedef __attribute__((ext_vector_type(2))) uint32_t dec_u32x2;
        for (int i = tid; i < G * DH / 4; i += THREADS)
        {
            const int g = i / (DH / 4), d = (i % (DH / 4)) * 4;
            dec_u32x4 v[8];
            dec_u32x2 ml[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const uint32_t rec = (uint32_t) ((min(j, S - 1) * G + g) * REC) * 4;
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(grp, rec + d * 4, 0, 16);
                ml[j] = __builtin_amdgcn_raw_buffer_load_b64(grp, rec + DH * 4, 0, 16);
            }
            float mm = NEG_BIG;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                mm = fmaxf(mm, j < S ? __uint_as_float(ml[j].x) : NEG_BIG);
            float ll = 0.f;
            dec_f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const float c = j < S ? __builtin_amdgcn_exp2f(__uint_as_float(ml[j].x) - mm) : 0.f;
                ll = fmaf(__uint_as_float(ml[j].y), c, ll);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = fmaf(__uint_as_float(v[j][e]), c, o[e]);
            }
            const float inv = 1.f / (ll + 1e-6f); // decoderMaskedMultiheadAttentionTemplate.h:2771-2777
            const uint2 w = make_uint2(pack2<T>(o[0] * inv, o[1] * inv), pack2<T>(o[2] * inv, o[3] * inv));
            if (!PAD || d < hs)
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (out_head0 + g) * hs + d) = w;
        }
        MQ_STAMP(p, 9);
        return;
    }
    float* c_s = red + 1;      // [G][64] rescale factors
    float* l_s = c_s + G * 64; // [G] merged sums
    __syncthreads();           // everyone has read *last
    // (m, l) of every (head, split) once: one wave per head, lane = split
    for (int i = tid; i < G * 64; i += THREADS)
    {
        const int g = i >> 6, s = i & 63;
        const uint32_t rec = (uint32_t) ((min(s, S - 1) * G + g) * REC) * 4;
        const float m_ld = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grp, rec + DH * 4, 0, 16));
        const float l_ld = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(grp, rec + DH * 4 + 4, 0, 16));
        const float m = s < S ? m_ld : NEG_BIG, l = s < S ? l_ld : 0.f;
        float mm = m;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            mm = fmaxf(mm, __shfl_xor(mm, o));
        const float c = __builtin_amdgcn_exp2f(m - mm);
        float ll = l * c;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1)
            ll += __shfl_xor(ll, o);
        c_s[i] = c;
        if (s == 0)
            l_s[g] = ll;
    }
    __syncthreads();
    for (int i = tid; i < G * DH / 4; i += THREADS)
    {
        const int g = i / (DH / 4), d = (i % (DH / 4)) * 4;
        dec_f32x4 o = {0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < S; s0 += 8) // 8 splits' loads in flight, summed in split order
        {
            dec_u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(grp, (uint32_t) ((min(s0 + j, S - 1) * G + g) * REC + d) * 4, 0, 16);
#pragma unroll
            for (int j = 0; j < 8; ++j)
            {
                const float c = s0 + j < S ? c_s[g * 64 + min(s0 + j, 63)] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = fmaf(__uint_as_float(v[j][e]), c, o[e]);
            }
        }
        // inv_sum = 1 / (sum + 1e-6)  (decoderMaskedMultiheadAttentionTemplate.h:2771-2777)
        const float inv = 1.f / (l_s[g] + 1e-6f);
        const uint2 v = make_uint2(pack2<T>(o[0] * inv, o[1] * inv), pack2<T>(o[2] * inv, o[3] * inv));
        if (!PAD || d < hs)
            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + (out_head0 + g) * hs + d) = v;
    }
}

} // namespace
} // namespace mi355q
