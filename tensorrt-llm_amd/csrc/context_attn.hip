// context_attn.hip -- the context (prefill) phase of GPTAttention over the paged cache:
//   1. ctx_prepare_kernel: bias + RoPE on q, k of every input token, K / V appended to the KVBlockArray
//      cache (quantised for int8 / fp8 caches), processed q, k, v written HEAD-MAJOR to the workspace
//      ([head][token row][Dh]: a head's rows are contiguous, which is what the attention kernel streams).
//      Replaces invokeAddFusedQKVBiasTranspose / invokeApplyBiasRopeUpdateKVCache
//      (kernels/unfusedAttentionKernels.cu; called from gptAttentionCommon.cpp:930-990).
//   2. ctx_attn_kernel: causal flash attention among the input tokens of each sequence (the reference's
//      context FMHA, gptAttentionCommon.cpp:1000-1075 -> contextFusedMultiHeadAttention/fmhaRunner.cpp; cubins
//      only, behaviour spec = unfused softmax(QK^T / sqrt(d)) V with the causal mask), on MFMA:
//        workgroup = (128-row query tile, q head, sequence), 4 waves x 32 query rows, 2 workgroups per CU;
//        K / V tiles of 64 keys stream through a 2-stage LDS ring filled by LDS-DMA (global_load_lds_dwordx4,
//        swizzle applied on the source address), shared by the 4 waves; one barrier per tile;
//        S^T = K . Q^T (K fragments by ds_read_b128, Q^T fragments resident in registers), online softmax per
//        query column in the log2 domain, the probability registers ARE the B fragment of O^T += V^T . P^T,
//        V^T fragments by the transposing LDS read (ds_read_b64_tr_b16) from the row-major V tile.
// Scope: causal mask (optionally the sliding window, gptKernels.cu:174-177), ALiBi slopes, fp16 / bf16, head size
// 64 / 128, MHA / GQA / MQA, padded or packed input, paged or linear cache; the attention uses the un-quantised K / V
// of the input tokens (as the reference's context FMHA does).
// Chunked context (use_paged_context_fmha, gptAttentionCommon.cpp:825-864): when a sequence already has `past` tokens
// in its cache, the input tokens take positions past .. past + len - 1, are appended there, and the attention kernel
// (CACHED form) streams K / V of ALL kv_len = past + len tokens out of the cache itself (16-bit caches only, as in the
// reference: "Paged Context FMHA doesn't work with fp8/int8 kv cache").
#include "decode_common.h"

namespace mi355q
{

namespace
{

typedef __attribute__((ext_vector_type(8))) __bf16 cbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 cf16x8_t;
typedef __attribute__((ext_vector_type(4))) float cf32x4_t;
typedef __attribute__((ext_vector_type(16))) float cf32x16_t;
typedef __attribute__((ext_vector_type(4))) short cs16x4_t;
typedef __attribute__((address_space(3))) cs16x4_t lds_cs16x4_t;
typedef __attribute__((address_space(3))) void ctx_lds_void_t;
typedef __attribute__((address_space(1))) const void ctx_gbl_void_t;

// LDS-DMA issued from inline asm: the compiler's waitcnt pass makes every LDS read that follows a
// __builtin_amdgcn_global_load_lds wait for vmcnt(0) (it cannot tell the ring stages apart), which would serialise
// the prefetch with the compute.  Issued this way the DMA is ordered only by the explicit s_waitcnt vmcnt + barrier
// of the tile loop.  lds_off: wave-uniform LDS byte address; lane l's 16 bytes land at lds_off + 16 l.
__device__ __forceinline__ void ctx_dma16(const void* gptr, uint32_t lds_off)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_off) : "memory", "m0");
}

// fmaxf on MFMA outputs makes hipcc emit a canonicalising v_max_f32 per operand first; the scores are never sNaN
__device__ __forceinline__ float ctx_max3(float a, float b, float c)
{
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max over lane ^ 16 (W = 16) or lane ^ 32 (W = 32) through v_permlane{16,32}_swap: with both operands holding x
// the two results are [x0 x0 x2 x2] / [x1 x1 x3 x3] by rows of 16 lanes (W = 16) or [lo lo] / [hi hi] by halves
// (W = 32), so their max is the pairwise all-reduce.  The swap and the max sit in ONE asm statement: given the builtin,
// hipcc treats the swap's two results as the same value when its operands are copies of each other and emits
// max(r0, r0) -- every lane then ends up with the first row's maximum instead of the column's (found in round 2: a
// consistent but too small running maximum, harmless until a sliding window leaves that row with nothing but masked
// scores).  s_nop 1 either side: a VALU write needs 2 wait states before the swap reads it, and the swap's results
// 2 before a VALU reads them; the compiler cannot see inside the statement to add them.
template <int W>
__device__ __forceinline__ float ctx_swap_max(float x)
{
    float a = x, b = x;
    if constexpr (W == 16)
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    else
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
    return a;
}

struct CtxArgs
{
    const uint16_t* qkv;      // input rows [(Hq + 2 Hkv) * Dh]
    const uint16_t* qkv_bias; // or null
    uint16_t* proc;           // workspace: q [Hq][rows][Dh] | k [Hkv][rows][Dh] | v [Hkv][rows][Dh]
    uint16_t* out;            // [rows, Hq * Dh]
    const int32_t* ctx_lens;  // device [B]
    const int32_t* offsets;   // device [B]: first token row of each sequence (packed input) or null (padded: b * S)
    const int64_t* block_ptrs;
    const float* kv_scale_orig_quant;
    int batch, max_len, num_heads, num_kv_heads, head_size;
    int rows;                 // total token rows of qkv (padded: B * S)
    int max_blocks, tpb_log2, kv_type; // 0 = T, 1 = int8, 2 = fp8
    int head_tokens;          // tokens of one head inside a block: tokens_per_block, or linear_max_seq_len
    char* linear_base;        // KVLinearBuffer::data (kvCacheUtils.h:241-338) or null; one (sequence, K|V) plane is linear_plane_bytes
    size_t linear_plane_bytes;
    int pos_type, rot_dim, rot_scale_type, rot_max_pos;
    float rot_base, rot_scale, score_scale;
    const int32_t* kv_lens;   // device [B]: tokens of the sequence INCLUDING this call's (past + len), or null (no past)
    int mask_window;          // sliding window: key t is visible to the query at position a iff a - mask_window <= t <= a
    int kv_window, sink, bubble, cyc_len; // cache slot map of the append (kvCacheUtils.h:71-117); identity without a window
    const uint16_t* alibi_slopes;         // T [Hq] or null: score += slope[h] * (t - a)
    // relative attention bias (kRELATIVE; invokeAddRelativeAttentionBiasUnaligned, unfusedAttentionKernels.cu:1688-1751,
    // called with bidirectional = false from gptAttentionCommon.cpp:1006-1016): T values added to the scaled scores.
    // max_distance == 0: [Hq, rel_stride, rel_stride], entry [h][a][t]; > 0: the table [Hq, rel_stride = num_buckets]
    const uint16_t* rel_bias;
    int rel_stride, max_distance;
    int dh_pad; // row length of `proc`: head_size, or the 64 / 128 it is padded to (zeros) for the attention kernel
    int q_from_input;  // the attention kernel takes q straight from the qkv rows (bias add + rotation in its prologue, the
                       // angles from cs_table): the fast prepare kernel then skips the q vectors altogether
    float2* cs_table;  // [rows][64] (cos, sin) of pair j at the row's position, written by the fast prepare kernel for
                       // q_from_input (it lives in the q part of `proc`, which nothing else uses then)
    int kv_from_cache; // the attention kernel reads K / V from the cache this call appends to (16-bit cache, every token at its
                       // own slot): the fast prepare kernel then writes K / V once, not to the workspace as well
    int raw_k_cache; // POS_SHIFT: K goes to the cache without the position embedding (k_wo_pos,
                     // unfusedAttentionKernels_2_template.h:284-347); generic prepare kernel only
};

__device__ __forceinline__ int ctx_kv_slot(const CtxArgs& p, int t)
{
    return t < p.sink ? t : p.sink + p.bubble + (t - p.sink) % p.cyc_len;
}
// tokens already in the cache before this call's (chunked context: kv_seq_lengths - q_seq_lengths,
// unfusedAttentionKernels_2_template.h:252-256)
__device__ __forceinline__ int ctx_past(const CtxArgs& p, int b, int len)
{
    return p.kv_lens ? max(p.kv_lens[b] - len, 0) : 0;
}
// the append keeps the sink tokens and the last (window - sink) tokens of the sequence (:336-339; with a past the
// bound is taken on the whole kv length, where the reference takes the chunk's and lets two tokens race for a slot)
__device__ __forceinline__ bool ctx_keeps(const CtxArgs& p, int pos, int kv_len)
{
    return pos < p.sink || pos >= max(kv_len - p.kv_window + p.sink, p.sink);
}

// base of the K (kv = 0) or V (kv = 1) block `blk` of sequence b (getKBlockPtr / getVBlockPtr)
__device__ __forceinline__ char* ctx_block_ptr(const CtxArgs& p, int b, int kv, int blk)
{
    if (p.linear_base)
        return p.linear_base + ((size_t) b * 2 + kv) * p.linear_plane_bytes;
    return reinterpret_cast<char*>(p.block_ptrs[((size_t) b * 2 + kv) * p.max_blocks + blk]);
}
__device__ __forceinline__ bool ctx_has_cache(const CtxArgs& p)
{
    return p.block_ptrs != nullptr || p.linear_base != nullptr;
}

__global__ void ctx_offsets_kernel(const int32_t* lens, int32_t* offsets, int batch)
{
    if (threadIdx.x == 0 && blockIdx.x == 0)
    {
        int acc = 0;
        for (int b = 0; b < batch; ++b)
        {
            offsets[b] = acc;
            acc += lens[b];
        }
    }
}

// grid (max_len, B), block 256: one input token per block
template <typename T>
__global__ __launch_bounds__(256) void ctx_prepare_kernel(CtxArgs p)
{
    const int b = blockIdx.y, s = blockIdx.x;
    const int len = p.ctx_lens[b];
    if (s >= len)
        return;
    const int Hq = p.num_heads, Hkv = p.num_kv_heads, D = p.head_size;
    const int hidden = (Hq + 2 * Hkv) * D;
    const size_t row = (size_t) (p.offsets ? p.offsets[b] : b * p.max_len) + s;
    const uint16_t* src = p.qkv + row * hidden;
    extern __shared__ float xs[]; // [hidden] floats already rounded to T, then [rot_dim] cos | sin
    float* cs_tab = xs + hidden;
    for (int i = threadIdx.x; i < hidden; i += 256)
    {
        float x = T16<T>::to_float(src[i]);
        if (p.qkv_bias)
            x = T16<T>::to_float(T16<T>::from_float(x + T16<T>::to_float(p.qkv_bias[i]))); // add in T
        xs[i] = x;
    }
    const bool rope = p.pos_type == 1 || p.pos_type == 2; // kROPE_GPTJ / kROPE_GPT_NEOX on q and k heads, position pos
    const int half = p.rot_dim >> 1;
    const int past = ctx_past(p, b, len);
    const int pos = past + s;
    float* k_raw = cs_tab + p.rot_dim; // [Hkv * D], POS_SHIFT only: the keys before the rotation
    if (p.raw_k_cache)
    {
        __syncthreads();
        for (int i = threadIdx.x; i < Hkv * D; i += 256)
            k_raw[i] = xs[Hq * D + i];
    }
    if (rope && (int) threadIdx.x < half) // the angles only depend on (position, pair index): one table per token
    {
        const int j = threadIdx.x;
        float base = p.rot_base, scale = p.rot_scale;
        if (p.rot_scale_type == 2) // dynamic NTK (decoderMaskedMultiheadAttentionUtils.h:2552-2579) at the sequence length
        {
            if (len > p.rot_max_pos)
            {
                const float bb = len * (scale / p.rot_max_pos) - (scale - 1.f);
                base = base * powf(bb, (float) p.rot_dim / (float) (p.rot_dim - 2));
            }
            scale = 1.f;
        }
        else if (p.rot_scale_type == 1)
            scale = 1.f / scale;
        const float ang = ((float) pos * scale) / powf(base, (float) (2 * j) / (float) p.rot_dim);
        float sn, cs;
        sincosf(ang, &sn, &cs);
        cs_tab[2 * j] = cs;
        cs_tab[2 * j + 1] = sn;
    }
    __syncthreads();
    if (rope)
    {
        for (int i = threadIdx.x; i < (Hq + Hkv) * half; i += 256)
        {
            const int vec = i / half, j = i % half;
            float* x = xs + vec * D;
            const int i0 = p.pos_type == 2 ? j : 2 * j;
            const int i1 = p.pos_type == 2 ? j + half : 2 * j + 1;
            const float cs = cs_tab[2 * j], sn = cs_tab[2 * j + 1];
            const float a = x[i0], c = x[i1];
            x[i0] = T16<T>::to_float(T16<T>::from_float(cs * a - sn * c));
            x[i1] = T16<T>::to_float(T16<T>::from_float(cs * c + sn * a));
        }
        __syncthreads();
    }
    // processed rows, head-major: vector v of the row goes to proc[(v' * rows + row) * dh_pad] with q heads first;
    // a head size the attention kernel is not built for is padded with zeros to the next one that is
    const int DP = p.dh_pad;
    for (int i = threadIdx.x; i < (Hq + 2 * Hkv) * DP / 2; i += 256)
    {
        const int e = 2 * i, vec = e / DP, d = e % DP;
        const uint32_t w = d < D ? pack2<T>(xs[vec * D + d], xs[vec * D + d + 1]) : 0u;
        *reinterpret_cast<uint32_t*>(p.proc + ((size_t) vec * p.rows + row) * DP + d) = w;
    }
    // cache append at the slot of position pos: block [Hkv][tokens_per_block][Dh]
    if (ctx_has_cache(p) && ctx_keeps(p, pos, past + len))
    {
        const int tpb = 1 << p.tpb_log2;
        const int slot = ctx_kv_slot(p, pos);
        const int blk = slot >> p.tpb_log2, tin = slot & (tpb - 1);
        const float qs = p.kv_type ? p.kv_scale_orig_quant[0] : 1.f;
        for (int i = threadIdx.x; i < 2 * Hkv * D / 4; i += 256) // 4 consecutive d per thread
        {
            const int e4 = 4 * i;
            const int kv = e4 / (Hkv * D), rem = e4 % (Hkv * D), hk = rem / D, d = rem % D;
            const float* x = (p.raw_k_cache && kv == 0) ? k_raw + hk * D + d : xs + (Hq + kv * Hkv + hk) * D + d;
            const size_t e = ((size_t) hk * p.head_tokens + tin) * D + d;
            char* base = ctx_block_ptr(p, b, kv, blk);
            if (p.kv_type == 0)
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + e) = make_uint2(pack2<T>(x[0], x[1]), pack2<T>(x[2], x[3]));
            else if (p.kv_type == 1)
                *reinterpret_cast<uint32_t*>(base + e) = (uint32_t) float_to_int8_sat(x[0] * qs) | ((uint32_t) float_to_int8_sat(x[1] * qs) << 8)
                    | ((uint32_t) float_to_int8_sat(x[2] * qs) << 16) | ((uint32_t) float_to_int8_sat(x[3] * qs) << 24);
            else
                *reinterpret_cast<uint32_t*>(base + e) = (uint32_t) float_to_fp8(x[0] * qs) | ((uint32_t) float_to_fp8(x[1] * qs) << 8)
                    | ((uint32_t) float_to_fp8(x[2] * qs) << 16) | ((uint32_t) float_to_fp8(x[3] * qs) << 24);
        }
    }
}

// Fast form of the prepare kernel for the common layouts (no RoPE, GPT-J pairs, or GPT-NeoX with
// rotary_embedding_dim == head_size): 16-byte loads / stores, no LDS staging of the row.  A work item is one head
// vector's chunk pair (c, c + Dh/16): the two 16-byte chunks that NeoX rotates against each other.  Same arithmetic,
// in the same order, as ctx_prepare_kernel (bias add in T, rotation in fp32, one rounding to T).
// grid (max_len, B), block 256.
template <typename T>
__global__ __launch_bounds__(256) void ctx_prepare_fast_kernel(CtxArgs p)
{
    const int b = blockIdx.y, s = blockIdx.x;
    const int len = p.ctx_lens[b];
    if (s >= len)
        return;
    const int Hq = p.num_heads, Hkv = p.num_kv_heads, D = p.head_size;
    const int nvec = Hq + 2 * Hkv, hidden = nvec * D;
    const size_t row = (size_t) (p.offsets ? p.offsets[b] : b * p.max_len) + s;
    const uint16_t* src = p.qkv + row * hidden;
    __shared__ float cs_tab[128]; // [j] cos | [64 + j] sin, j < rot_dim / 2 <= 64
    const bool rope = p.pos_type == 1 || p.pos_type == 2;
    const int half = p.rot_dim >> 1;
    const int past = ctx_past(p, b, len);
    const int pos = past + s;
    if (rope)
    {
        if ((int) threadIdx.x < half)
        {
            const int j = threadIdx.x;
            float base = p.rot_base, scale = p.rot_scale;
            if (p.rot_scale_type == 2)
            {
                if (len > p.rot_max_pos)
                {
                    const float bb = len * (scale / p.rot_max_pos) - (scale - 1.f);
                    base = base * powf(bb, (float) p.rot_dim / (float) (p.rot_dim - 2));
                }
                scale = 1.f;
            }
            else if (p.rot_scale_type == 1)
                scale = 1.f / scale;
            const float ang = ((float) pos * scale) / powf(base, (float) (2 * j) / (float) p.rot_dim);
            float sn, cs;
            sincosf(ang, &sn, &cs);
            cs_tab[j] = cs;
            cs_tab[64 + j] = sn;
            if (p.q_from_input)
                p.cs_table[row * 64 + j] = make_float2(cs, sn);
        }
        __syncthreads();
    }
    const int cpv = D / 16; // chunk pairs per vector: 8 / 4
    const int tpb = 1 << p.tpb_log2;
    const int slot = ctx_kv_slot(p, pos);
    const int blk = slot >> p.tpb_log2, tin = slot & (tpb - 1);
    const bool bp = ctx_has_cache(p) && ctx_keeps(p, pos, past + len);
    const float qs = (bp && p.kv_type) ? p.kv_scale_orig_quant[0] : 1.f;
    for (int it = (p.q_from_input ? Hq * cpv : 0) + threadIdx.x; it < nvec * cpv; it += 256)
    {
        const int vec = it / cpv, c = it % cpv;
        const int d0 = 8 * c, d1 = 8 * c + D / 2;
        const uint4 r0 = *reinterpret_cast<const uint4*>(src + vec * D + d0);
        const uint4 r1 = *reinterpret_cast<const uint4*>(src + vec * D + d1);
        const uint32_t w0[4] = {r0.x, r0.y, r0.z, r0.w}, w1[4] = {r1.x, r1.y, r1.z, r1.w};
        float x0[8], x1[8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            x0[2 * i] = T16<T>::lo(w0[i]);
            x0[2 * i + 1] = T16<T>::hi(w0[i]);
            x1[2 * i] = T16<T>::lo(w1[i]);
            x1[2 * i + 1] = T16<T>::hi(w1[i]);
        }
        if (p.qkv_bias)
        {
            const uint4 b0 = *reinterpret_cast<const uint4*>(p.qkv_bias + vec * D + d0);
            const uint4 b1 = *reinterpret_cast<const uint4*>(p.qkv_bias + vec * D + d1);
            const uint32_t v0[4] = {b0.x, b0.y, b0.z, b0.w}, v1[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                x0[2 * i] = T16<T>::to_float(T16<T>::from_float(x0[2 * i] + T16<T>::lo(v0[i])));
                x0[2 * i + 1] = T16<T>::to_float(T16<T>::from_float(x0[2 * i + 1] + T16<T>::hi(v0[i])));
                x1[2 * i] = T16<T>::to_float(T16<T>::from_float(x1[2 * i] + T16<T>::lo(v1[i])));
                x1[2 * i + 1] = T16<T>::to_float(T16<T>::from_float(x1[2 * i + 1] + T16<T>::hi(v1[i])));
            }
        }
        if (rope && vec < Hq + Hkv)
        {
            if (p.pos_type == 2) // NeoX, rot_dim == D: element d0 + i pairs with d1 + i, angle index d0 + i
            {
#pragma unroll
                for (int i = 0; i < 8; ++i)
                {
                    const float cs = cs_tab[d0 + i], sn = cs_tab[64 + d0 + i];
                    const float a = x0[i], c2 = x1[i];
                    x0[i] = cs * a - sn * c2;
                    x1[i] = cs * c2 + sn * a;
                }
            }
            else // GPT-J: pairs (2j, 2j+1) inside a chunk, elements >= rot_dim pass through
            {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                {
                    const int j0 = (d0 >> 1) + i, j1 = (d1 >> 1) + i;
                    if (j0 < half)
                    {
                        const float cs = cs_tab[j0], sn = cs_tab[64 + j0];
                        const float a = x0[2 * i], c2 = x0[2 * i + 1];
                        x0[2 * i] = cs * a - sn * c2;
                        x0[2 * i + 1] = cs * c2 + sn * a;
                    }
                    if (j1 < half)
                    {
                        const float cs = cs_tab[j1], sn = cs_tab[64 + j1];
                        const float a = x1[2 * i], c2 = x1[2 * i + 1];
                        x1[2 * i] = cs * a - sn * c2;
                        x1[2 * i + 1] = cs * c2 + sn * a;
                    }
                }
            }
        }
        const uint4 o0 = make_uint4(pack2<T>(x0[0], x0[1]), pack2<T>(x0[2], x0[3]), pack2<T>(x0[4], x0[5]), pack2<T>(x0[6], x0[7]));
        const uint4 o1 = make_uint4(pack2<T>(x1[0], x1[1]), pack2<T>(x1[2], x1[3]), pack2<T>(x1[4], x1[5]), pack2<T>(x1[6], x1[7]));
        if (vec < Hq || !p.kv_from_cache)
        {
            uint16_t* dst = p.proc + ((size_t) vec * p.rows + row) * D;
            *reinterpret_cast<uint4*>(dst + d0) = o0;
            *reinterpret_cast<uint4*>(dst + d1) = o1;
        }
        if (bp && vec >= Hq)
        {
            const int kv = (vec - Hq) / Hkv, hk = (vec - Hq) % Hkv;
            const size_t e = ((size_t) hk * p.head_tokens + tin) * D;
            char* base = ctx_block_ptr(p, b, kv, blk);
            if (p.kv_type == 0)
            {
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + e + d0) = o0;
                *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(base) + e + d1) = o1;
            }
            else
            {
                // quantise the T-rounded values, like the generic kernel
                const uint32_t q0[4] = {o0.x, o0.y, o0.z, o0.w}, q1[4] = {o1.x, o1.y, o1.z, o1.w};
                uint32_t c0[2] = {0, 0}, c1[2] = {0, 0};
#pragma unroll
                for (int i = 0; i < 8; ++i)
                {
                    const float a = (i & 1) ? T16<T>::hi(q0[i >> 1]) : T16<T>::lo(q0[i >> 1]);
                    const float c2 = (i & 1) ? T16<T>::hi(q1[i >> 1]) : T16<T>::lo(q1[i >> 1]);
                    const uint32_t ca = p.kv_type == 1 ? float_to_int8_sat(a * qs) : float_to_fp8(a * qs);
                    const uint32_t cb = p.kv_type == 1 ? float_to_int8_sat(c2 * qs) : float_to_fp8(c2 * qs);
                    c0[i >> 2] |= ca << (8 * (i & 3));
                    c1[i >> 2] |= cb << (8 * (i & 3));
                }
                *reinterpret_cast<uint2*>(base + e + d0) = make_uint2(c0[0], c0[1]);
                *reinterpret_cast<uint2*>(base + e + d1) = make_uint2(c1[0], c1[1]);
            }
        }
    }
}

template <typename T>
__device__ __forceinline__ cf32x4_t ctx_mfma(const uint4& a, const uint4& b, cf32x4_t c);
template <>
__device__ __forceinline__ cf32x4_t ctx_mfma<bf16_t>(const uint4& a, const uint4& b, cf32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cbf16x8_t, a), __builtin_bit_cast(cbf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ cf32x4_t ctx_mfma<f16_t>(const uint4& a, const uint4& b, cf32x4_t c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(cf16x8_t, a), __builtin_bit_cast(cf16x8_t, b), c, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ cf32x16_t ctx_mfma32(const uint4& a, const uint4& b, cf32x16_t c);
template <>
__device__ __forceinline__ cf32x16_t ctx_mfma32<bf16_t>(const uint4& a, const uint4& b, cf32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(cbf16x8_t, a), __builtin_bit_cast(cbf16x8_t, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ cf32x16_t ctx_mfma32<f16_t>(const uint4& a, const uint4& b, cf32x16_t c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cf16x8_t, a), __builtin_bit_cast(cf16x8_t, b), c, 0, 0, 0);
}

constexpr int CTX_KT = 64;   // keys per tile
constexpr int CTX_QW = 32;   // query rows per wave

// grid (Hq, ceil(max_len / (32 NWV)), B), block 64 NWV.  blockIdx.y counts query tiles from the END of the sequence so the
// long (many key tiles) workgroups are dispatched first.
// CACHED: K / V of keys 0 .. kv_len - 1 come from the (16-bit) cache, the query rows sit at positions past .. (chunked
// context); else from the workspace rows of this call (past = 0).  BIAS 1: ALiBi, 2: relative attention bias; the
// scores are biased before the softmax.
// W32: the 32x32x16 MFMA form (lane = one query column of the wave's 32, see compute32 below); else 16x16x32.
template <typename T, int DH, int NWV, bool CACHED, int BIAS, bool W32 = false>
__global__ __launch_bounds__(64 * NWV) __attribute__((amdgpu_waves_per_eu(DH > 128 ? 1 : 2))) void ctx_attn_kernel(CtxArgs p)
{
    constexpr int CTX_QB = CTX_QW * NWV;    // query rows per workgroup: 128 (4 waves, short sequences) / 256 (8 waves)
    constexpr int KS = DH / 32;             // QK k-steps
    constexpr int DT = DH / 16;             // output d-tiles
    constexpr int ROWB = DH * 2;            // bytes of a K / V row
    constexpr int CH = ROWB / 16;           // 16-byte chunks per row: 16 / 8
    constexpr int TILE_BYTES = CTX_KT * ROWB; // one K (or V) tile: 16 / 8 KiB
    constexpr int STAGE_BYTES = 2 * TILE_BYTES;
    constexpr int RPD = 1024 / ROWB;        // rows per DMA instruction (64 lanes x 16 B): 4 / 8
    constexpr int DPT = CTX_KT / RPD;       // DMA instructions per tile: 16 / 8
    constexpr int DPW = DPT / NWV;          // per wave
    static_assert(DPT % NWV == 0, "tile pieces do not split evenly over the waves");
    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    // blockIdx.x enumerates the q heads kv-head-fastest: workgroups are dealt to the 8 XCDs round-robin in linear
    // order, so an XCD gets the heads of kv-heads = its index (mod 8) for EVERY query tile -- equal work per XCD
    // (the causal tiles differ 16:1 in length) and a kv-head's K / V stay in one XCD's L2
    const int b = blockIdx.z;
    const int hk = blockIdx.x % p.num_kv_heads;
    const int h = hk * (p.num_heads / p.num_kv_heads) + blockIdx.x / p.num_kv_heads;
    const int len = p.ctx_lens[b];
    const int qtile = (int) gridDim.y - 1 - (int) blockIdx.y;
    const int Q0 = qtile * CTX_QB;
    if (Q0 >= len) // whole workgroup past the sequence
        return;
    const int q0 = Q0 + wave * CTX_QW; // this wave's first query row
    const bool wave_live = q0 < len;
    const int kv_len = CACHED ? max(p.kv_lens[b], len) : len; // keys of the sequence
    const int past = kv_len - len;                            // position of query row 0
    const int window = p.mask_window;
    const size_t row0 = (size_t) (p.offsets ? p.offsets[b] : b * p.max_len);
    const uint16_t* qbase = p.proc + ((size_t) h * p.rows + row0) * DH;
    const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.proc + ((size_t) (p.num_heads + hk) * p.rows + row0) * DH);
    const uint8_t* vbase
        = reinterpret_cast<const uint8_t*>(p.proc + ((size_t) (p.num_heads + p.num_kv_heads + hk) * p.rows + row0) * DH);

    // swizzles of the 16-byte chunk index by tile row.  K: 16 consecutive rows x one logical chunk (a ds_read_b128
    // fragment) cover all 64 banks once.  V: the transposing 8-byte reads of 4 row groups do.
    auto ksw = [](int row) { return CH >= 16 ? (row & 15) : ((row >> 1) & 7); };
    auto vsw = [](int row) { return (((row & 3) << 2) | ((row >> 2) & 3)) & (CH - 1); };

    // ---- DMA staging: instruction d of a tile fills LDS bytes [1024 d, +1024): lane -> (row, stored chunk)
    const int drow = lane / CH, dch = lane % CH;
    const uint32_t lds_base = __builtin_amdgcn_readfirstlane((uint32_t) (uintptr_t) (ctx_lds_void_t*) lds);
    uint32_t krel[DPW], vrel[DPW]; // byte offset of this lane's 16 bytes inside a full tile
#pragma unroll
    for (int i = 0; i < DPW; ++i)
    {
        const int row = (wave * DPW + i) * RPD + drow;
        krel[i] = row * ROWB + 16 * (dch ^ ksw(row));
        vrel[i] = row * ROWB + 16 * (dch ^ vsw(row));
    }
    // CACHED: the lane's K / V block bases of the tile staged NEXT, loaded one tile ahead so that their latency hides
    // behind a tile of compute (the values are only touched after the loop's own vmcnt(0))
    uint64_t kblk[CACHED ? DPW : 1], vblk[CACHED ? DPW : 1];
    // Blocks of at least a tile (64 tokens; the linear buffer is one block): a 64-key tile lies inside ONE block, its rows
    // are contiguous there like the workspace's, and the block base is wave-uniform -- one scalar load per tile and the
    // same DMA addressing as the non-cached form, instead of a pointer load and 64-bit address arithmetic per lane and
    // DMA (that difference was why this form ran ~10 % behind the other per key, tools/lab_ctx_from_cache.py)
    const bool uni_blocks = CACHED && p.tpb_log2 >= 6;
    uint64_t kblk_u = 0, vblk_u = 0;
    auto fetch_blocks = [&](int tile)
    {
        if constexpr (CACHED)
        {
            if (uni_blocks)
            {
                const int blk = min(tile * CTX_KT, kv_len - 1) >> p.tpb_log2;
                if (p.linear_base)
                {
                    kblk_u = (uint64_t) (p.linear_base + ((size_t) b * 2) * p.linear_plane_bytes);
                    vblk_u = kblk_u + p.linear_plane_bytes;
                    // (no return: the last tile of a plane whose length is not a multiple of 64 takes the per-lane,
                    // clamped form below -- its rows past the plane are not this buffer's)
                }
                else
                {
                    kblk_u = (uint64_t) p.block_ptrs[((size_t) b * 2) * p.max_blocks + blk];
                    vblk_u = (uint64_t) p.block_ptrs[((size_t) b * 2 + 1) * p.max_blocks + blk];
                    return;
                }
            }
#pragma unroll
            for (int i = 0; i < DPW; ++i)
            {
                const int row = (wave * DPW + i) * RPD + drow;
                const int t = min(tile * CTX_KT + row, kv_len - 1);
                const int blk = t >> p.tpb_log2;
                if (p.linear_base)
                {
                    kblk[i] = (uint64_t) (p.linear_base + ((size_t) b * 2) * p.linear_plane_bytes);
                    vblk[i] = kblk[i] + p.linear_plane_bytes;
                }
                else
                {
                    kblk[i] = (uint64_t) p.block_ptrs[((size_t) b * 2) * p.max_blocks + blk];
                    vblk[i] = (uint64_t) p.block_ptrs[((size_t) b * 2 + 1) * p.max_blocks + blk];
                }
            }
        }
    };
    auto stage = [&](int tile, int buf)
    {
        const int t0 = tile * CTX_KT;
        if constexpr (CACHED)
        {
            const int tmask = (1 << p.tpb_log2) - 1;
            if (uni_blocks && (!p.linear_base || t0 + CTX_KT <= p.head_tokens))
            {
                // rows past the sequence are rows of the same (allocated) block: read as they are, masked later
                const size_t in_blk = ((size_t) hk * p.head_tokens + (t0 & tmask)) * ROWB;
                const uint8_t* kt = reinterpret_cast<const uint8_t*>(kblk_u) + in_blk;
                const uint8_t* vt = reinterpret_cast<const uint8_t*>(vblk_u) + in_blk;
#pragma unroll
                for (int i = 0; i < DPW; ++i)
                {
                    const uint32_t kdst = lds_base + buf * STAGE_BYTES + (wave * DPW + i) * 1024;
                    ctx_dma16(kt + krel[i], kdst);
                    ctx_dma16(vt + vrel[i], kdst + TILE_BYTES);
                }
                return;
            }
#pragma unroll
            for (int i = 0; i < DPW; ++i)
            {
                const int d = wave * DPW + i;
                const int row = d * RPD + drow;
                const int t = min(t0 + row, kv_len - 1); // rows past the sequence re-read its last row (masked)
                const size_t in_blk = ((size_t) hk * p.head_tokens + (t & tmask)) * ROWB;
                const uint32_t kdst = lds_base + buf * STAGE_BYTES + d * 1024;
                ctx_dma16(reinterpret_cast<const uint8_t*>(kblk[i]) + in_blk + 16 * (dch ^ ksw(row)), kdst);
                ctx_dma16(reinterpret_cast<const uint8_t*>(vblk[i]) + in_blk + 16 * (dch ^ vsw(row)), kdst + TILE_BYTES);
            }
        }
        else if (t0 + CTX_KT <= len) // wave-uniform: the whole tile is inside the sequence
        {
            const uint8_t* kt = kbase + (size_t) t0 * ROWB;
            const uint8_t* vt = vbase + (size_t) t0 * ROWB;
#pragma unroll
            for (int i = 0; i < DPW; ++i)
            {
                const uint32_t kdst = lds_base + buf * STAGE_BYTES + (wave * DPW + i) * 1024;
                ctx_dma16(kt + krel[i], kdst);
                ctx_dma16(vt + vrel[i], kdst + TILE_BYTES);
            }
        }
        else
        {
#pragma unroll
            for (int i = 0; i < DPW; ++i)
            {
                const int d = wave * DPW + i;
                const int row = d * RPD + drow;
                const size_t grow = (size_t) min(t0 + row, len - 1) * ROWB; // rows past the sequence re-read its last row (masked)
                const uint32_t kdst = lds_base + buf * STAGE_BYTES + d * 1024;
                ctx_dma16(kbase + grow + 16 * (dch ^ ksw(row)), kdst);
                ctx_dma16(vbase + grow + 16 * (dch ^ vsw(row)), kdst + TILE_BYTES);
            }
        }
    };

    const int kv_end = min(kv_len, past + Q0 + CTX_QB); // causal: keys up to the last query row of the workgroup
    const int ntiles = (kv_end + CTX_KT - 1) / CTX_KT;
    const int tile_lo = max(past + Q0 - window, 0) / CTX_KT; // sliding window: first key the workgroup's first row sees

    // the first K / V tile goes out before the q prologue: its round trip then runs under the q loads (and, with
    // q_from_input, under the bias add and the rotation) instead of after them.  vmcnt retires in order, so the waits
    // hipcc places for the q loads also cover these DMAs -- which the loop needs before its first tile anyway.
    fetch_blocks(tile_lo);
    stage(tile_lo, tile_lo & 1);
    fetch_blocks(min(tile_lo + 1, ntiles - 1));

    // Q^T fragments (B operand).  16x16x32: column = query row q0 + 16*qt + lr, k = d = 32*ks + 8*lq + j.
    // 32x32x16: column = query row q0 + L, k = d = 16*ks + 8*hh + j  (L = lane & 31, hh = lane >> 5)
    constexpr int KS16 = DH / 16, DB = DH / 32;
    const int L = lane & 31, hh = lane >> 5;
    uint4 qf[W32 ? 1 : 2][W32 ? 1 : KS];
    uint4 qg[W32 ? KS16 : 1];
    if constexpr (W32)
    {
        const int qr = min(q0 + L, len - 1); // rows past the sequence re-read its last row; never stored
#pragma unroll
        for (int ks = 0; ks < KS16; ++ks)
            qg[ks] = *reinterpret_cast<const uint4*>(qbase + (size_t) qr * DH + 16 * ks + 8 * hh);
#pragma unroll
        for (int ks = 0; ks < KS16; ++ks)
            asm volatile("" : "+v"(qg[ks].x), "+v"(qg[ks].y), "+v"(qg[ks].z), "+v"(qg[ks].w));
    }
    else
    {
        if (p.q_from_input) // wave-uniform
        {
            // q straight from the qkv rows: the prepare pass's arithmetic for a q vector, in its order (bias add in T,
            // rotation in fp32 with the angles the prepare pass tabulated, one rounding to T) -- the same bits it
            // would have written to the workspace, without the round trip.  A lane's 8 elements of k-step ks are
            // d = 32 ks + 8 lq + i: NeoX pairs (d, d + Dh/2) are k-steps (ks, ks + KS/2) of the same lane, GPT-J pairs
            // (2 j, 2 j + 1) neighbours inside the chunk.
            const int hidden = (p.num_heads + 2 * p.num_kv_heads) * DH;
            const bool rope = p.pos_type == 1 || p.pos_type == 2;
            const int half = p.rot_dim >> 1;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
            {
                const int qr = min(q0 + 16 * qt + lr, len - 1);
                const uint16_t* src = p.qkv + (row0 + qr) * (size_t) hidden + (size_t) h * DH + 8 * lq;
                float x[KS][8];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                {
                    const uint4 r = *reinterpret_cast<const uint4*>(src + 32 * ks);
                    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                    {
                        x[ks][2 * i] = T16<T>::lo(w[i]);
                        x[ks][2 * i + 1] = T16<T>::hi(w[i]);
                    }
                    if (p.qkv_bias)
                    {
                        const uint4 bb = *reinterpret_cast<const uint4*>(p.qkv_bias + (size_t) h * DH + 32 * ks + 8 * lq);
                        const uint32_t bw[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                        {
                            x[ks][2 * i] = T16<T>::to_float(T16<T>::from_float(x[ks][2 * i] + T16<T>::lo(bw[i])));
                            x[ks][2 * i + 1] = T16<T>::to_float(T16<T>::from_float(x[ks][2 * i + 1] + T16<T>::hi(bw[i])));
                        }
                    }
                }
                if (rope)
                {
                    const float2* tab = p.cs_table + (row0 + qr) * 64;
                    if (p.pos_type == 2) // NeoX, rot_dim == Dh
                    {
#pragma unroll
                        for (int ks = 0; ks < KS / 2; ++ks)
                        {
                            const float4* t4 = reinterpret_cast<const float4*>(tab + 32 * ks + 8 * lq);
#pragma unroll
                            for (int i2 = 0; i2 < 4; ++i2)
                            {
                                const float4 cs2 = t4[i2]; // (cos, sin) of pairs 2 i2, 2 i2 + 1
                                const float cs[2] = {cs2.x, cs2.z}, sn[2] = {cs2.y, cs2.w};
#pragma unroll
                                for (int e = 0; e < 2; ++e)
                                {
                                    const int i = 2 * i2 + e;
                                    const float a = x[ks][i], c2 = x[ks + KS / 2][i];
                                    x[ks][i] = cs[e] * a - sn[e] * c2;
                                    x[ks + KS / 2][i] = cs[e] * c2 + sn[e] * a;
                                }
                            }
                        }
                    }
                    else // GPT-J: pair j = (32 ks + 8 lq) / 2 + i2, elements past rot_dim pass through
                    {
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks)
                        {
                            const int j0 = 16 * ks + 4 * lq;
                            const float4* t4 = reinterpret_cast<const float4*>(tab + min(j0, 60));
#pragma unroll
                            for (int i2 = 0; i2 < 2; ++i2)
                            {
                                const float4 cs2 = t4[i2];
                                const float cs[2] = {cs2.x, cs2.z}, sn[2] = {cs2.y, cs2.w};
#pragma unroll
                                for (int e = 0; e < 2; ++e)
                                {
                                    const int jj = 2 * i2 + e;
                                    if (j0 + jj < half)
                                    {
                                        const float a = x[ks][2 * jj], c2 = x[ks][2 * jj + 1];
                                        x[ks][2 * jj] = cs[e] * a - sn[e] * c2;
                                        x[ks][2 * jj + 1] = cs[e] * c2 + sn[e] * a;
                                    }
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    qf[qt][ks] = make_uint4(pack2<T>(x[ks][0], x[ks][1]), pack2<T>(x[ks][2], x[ks][3]), pack2<T>(x[ks][4], x[ks][5]),
                        pack2<T>(x[ks][6], x[ks][7]));
            }
        }
        else
        {
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
            {
                const int qr = min(q0 + 16 * qt + lr, len - 1); // rows past the sequence re-read its last row; never stored
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    qf[qt][ks] = *reinterpret_cast<const uint4*>(qbase + (size_t) qr * DH + 32 * ks + 8 * lq);
            }
        }
        // make the compiler's own vmcnt bookkeeping for the Q loads end here, before any (invisible to it) DMA is in
        // flight: otherwise it places its waits for them at their first use inside the loop, where they would drain the
        // prefetch
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                asm volatile("" : "+v"(qf[qt][ks].x), "+v"(qf[qt][ks].y), "+v"(qf[qt][ks].z), "+v"(qf[qt][ks].w));
    }
    const float sscale_in = p.score_scale * 1.4426950408889634f;
    // BIAS: the scores are scaled and biased in place (log2 domain), the softmax then runs with scale 1
    const float sscale = BIAS ? 1.f : sscale_in;
    float slope2 = 0.f;
    if constexpr (BIAS == 1)
        slope2 = T16<T>::to_float(p.alibi_slopes[h]) * 1.4426950408889634f;
    // BIAS 2: the bias of (query position a, key t), log2 domain.  Implicit: the decoder's (one-directional) bucket of
    // a - t, the same float expression as the reference's kernel; explicit: indices clamped to the tensor (entries past
    // it only ever belong to masked keys or rows that are not stored)
    auto rel_bias_at = [&](int a, int t) -> float
    {
        size_t idx;
        if (p.max_distance > 0)
        {
            const int nb = p.rel_stride, max_exact = nb / 2;
            const int rp = max(a - t, 0);
            int large = max_exact
                + (int) (logf(rp * 1.0f / max_exact) / logf((float) p.max_distance / max_exact) * (nb - max_exact));
            large = min(large, nb - 1);
            idx = (size_t) h * nb + (rp < max_exact ? rp : large);
        }
        else
            idx = ((size_t) h * p.rel_stride + min(a, p.rel_stride - 1)) * p.rel_stride + min(t, p.rel_stride - 1);
        return T16<T>::to_float(p.rel_bias[idx]) * 1.4426950408889634f;
    };

    float m_run[2] = {NEG_BIG, NEG_BIG}, l_run[2] = {0.f, 0.f};
    cf32x4_t o[W32 ? 1 : 2][W32 ? 1 : DT];
    cf32x16_t o2[W32 ? DB : 1];
    if constexpr (W32)
    {
#pragma unroll
        for (int db = 0; db < DB; ++db)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                o2[db][i] = 0.f;
    }
    else
    {
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                o[qt][dt] = cf32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    const int rq = lr >> 2, rp = lr & 3;
    auto compute = [&](int buf, int t0)
    {
        if constexpr (!W32)
        {
            const uint8_t* kt = lds + buf * STAGE_BYTES;
            const uint8_t* vt = kt + TILE_BYTES;
            // ---- S^T tiles [key tile x][query tile qt]
            cf32x4_t s[4][2];
    #pragma unroll
            for (int x = 0; x < 4; ++x)
            {
                const int row = 16 * x + lr;
                uint4 kf[KS];
    #pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    kf[ks] = *reinterpret_cast<const uint4*>(kt + ROWB * row + 16 * ((4 * ks + lq) ^ ksw(row)));
    #pragma unroll
                for (int qt = 0; qt < 2; ++qt)
                {
                    s[x][qt] = cf32x4_t{0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
                        s[x][qt] = ctx_mfma<T>(kf[ks], qf[qt][ks], s[x][qt]);
                }
            }
            // ---- mask (only tiles that cross the diagonal or the end of the sequence) + online softmax per query column.
            // The running maximum is kept on the RAW scores (sscale > 0) and the scale rides in the exponent's fma:
            // p = exp2(s * sscale - m * sscale)
            const bool need_mask = t0 + CTX_KT - 1 > past + q0 || t0 + CTX_KT > kv_len
                || t0 < past + q0 + CTX_QW - 1 - window; // wave-uniform: diagonal, end of the sequence, start of a row's window
            uint4 pf[2][2];
    #pragma unroll
            for (int qt = 0; qt < 2; ++qt)
            {
                const int qpos = past + q0 + 16 * qt + lr;
                if constexpr (BIAS == 1)
                {
    #pragma unroll
                    for (int x = 0; x < 4; ++x)
    #pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            s[x][qt][rr] = __builtin_fmaf(s[x][qt][rr], sscale_in, slope2 * (float) (t0 + 16 * x + 4 * lq + rr - qpos));
                }
                if constexpr (BIAS == 2)
                {
    #pragma unroll
                    for (int x = 0; x < 4; ++x)
    #pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            s[x][qt][rr] = __builtin_fmaf(s[x][qt][rr], sscale_in, rel_bias_at(qpos, t0 + 16 * x + 4 * lq + rr));
                }
                if (need_mask)
                {
                    // -inf, not NEG_BIG: a row whose keys of this tile are all masked (sliding window) keeps m_run = NEG_BIG,
                    // and exp2(-inf - m_run) is still 0
    #pragma unroll
                    for (int x = 0; x < 4; ++x)
    #pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                        {
                            const int t = t0 + 16 * x + 4 * lq + rr;
                            s[x][qt][rr] = (t <= qpos && t < kv_len && t >= qpos - window) ? s[x][qt][rr] : -__builtin_inff();
                        }
                }
                float mx = ctx_max3(s[0][qt][0], s[0][qt][1], s[0][qt][2]);
                float my = ctx_max3(s[0][qt][3], s[1][qt][0], s[1][qt][1]);
                mx = ctx_max3(mx, s[1][qt][2], s[1][qt][3]);
                my = ctx_max3(my, s[2][qt][0], s[2][qt][1]);
                mx = ctx_max3(mx, s[2][qt][2], s[2][qt][3]);
                my = ctx_max3(my, s[3][qt][0], s[3][qt][1]);
                mx = ctx_max3(mx, s[3][qt][2], s[3][qt][3]);
                // all-reduce over the 4 lanes of the column (lane ^ 16, lane ^ 32) with the row-swap permutes: no LDS trip
                mx = ctx_swap_max<16>(ctx_max3(mx, my, my));
                const float m_new = ctx_max3(ctx_swap_max<32>(mx), m_run[qt], m_run[qt]);
                if (__any(m_new > m_run[qt])) // wave-uniform: rescale only when some column's max moved
                {
                    const float corr = __builtin_amdgcn_exp2f((m_run[qt] - m_new) * sscale);
                    m_run[qt] = m_new;
                    l_run[qt] *= corr;
    #pragma unroll
                    for (int dt = 0; dt < DT; ++dt)
    #pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            o[qt][dt][rr] *= corr;
                }
                // a masked score is -inf: exp2(-inf + anything finite) = 0, also while m_run is still NEG_BIG
                const float msc = -m_run[qt] * sscale;
                float pr[4][4];
                float ls = 0.f;
    #pragma unroll
                for (int x = 0; x < 4; ++x)
    #pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                    {
                        pr[x][rr] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[x][qt][rr], sscale, msc));
                        ls += pr[x][rr];
                    }
                l_run[qt] += ls;
                // P^T fragments (B operand), one per 32 keys: k index j < 4 -> key 4*lq + j of tile 2kh, j >= 4 -> of tile 2kh+1
    #pragma unroll
                for (int kh = 0; kh < 2; ++kh)
                    pf[qt][kh] = make_uint4(pack2<T>(pr[2 * kh][0], pr[2 * kh][1]), pack2<T>(pr[2 * kh][2], pr[2 * kh][3]),
                        pack2<T>(pr[2 * kh + 1][0], pr[2 * kh + 1][1]), pack2<T>(pr[2 * kh + 1][2], pr[2 * kh + 1][3]));
            }
            // ---- O^T += V^T . P^T  (A operand rows d = 16*dt + lr).  Transposing read: lane 4q'+p' of a 16-lane group
            // supplies row r0 + q', columns 4p' .. 4p'+3 of the 16-column block; lane i receives column i of rows r0 .. r0+3
    #pragma unroll
            for (int kh = 0; kh < 2; ++kh)
            {
                const int ra = 32 * kh + 4 * lq + rq, rb = ra + 16;
    #pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                {
                    const int ch = 2 * dt + (rp >> 1), sub = 8 * (rp & 1);
                    const cs16x4_t va
                        = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_cs16x4_t*) (vt + ROWB * ra + 16 * (ch ^ vsw(ra)) + sub));
                    const cs16x4_t vb
                        = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_cs16x4_t*) (vt + ROWB * rb + 16 * (ch ^ vsw(rb)) + sub));
                    const uint2 a2 = __builtin_bit_cast(uint2, va), b2 = __builtin_bit_cast(uint2, vb);
                    const uint4 vf = make_uint4(a2.x, a2.y, b2.x, b2.y);
                    o[0][dt] = ctx_mfma<T>(vf, pf[0][kh], o[0][dt]);
                    o[1][dt] = ctx_mfma<T>(vf, pf[1][kh], o[1][dt]);
                }
            }
            }
    };

    // ---- the 32x32x16 form.  S^T = K . Q^T as two 32-key blocks x2: lane (L, hh) holds, for query column L, the keys
    // 32*x2 + 8*(i/4) + 4*hh + i%4 (i < 16).  The 8 values i = 8*g2 .. 8*g2+7 are the B fragment (k = 8*hh + j) of
    // k-step (x2, g2) of O^T += V^T . P^T, whose A fragment takes V rows r0 .. r0+3 and r0+8 .. r0+11,
    // r0 = 32*x2 + 16*g2 + 4*hh, by two transposing reads.  O^T block db: d = 32*db + 8*(i/4) + 4*hh + i%4, column L.
    auto compute32 = [&](int buf, int t0)
    {
        if constexpr (W32)
        {
            const uint8_t* kt = lds + buf * STAGE_BYTES;
            const uint8_t* vt = kt + TILE_BYTES;
            cf32x16_t s[2];
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    s[x2][i] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS16; ++ks)
#pragma unroll
                for (int x2 = 0; x2 < 2; ++x2)
                {
                    const int row = 32 * x2 + L;
                    const uint4 kf = *reinterpret_cast<const uint4*>(kt + ROWB * row + 16 * ((2 * ks + hh) ^ ksw(row)));
                    s[x2] = ctx_mfma32<T>(kf, qg[ks], s[x2]);
                }
            const bool need_mask = t0 + CTX_KT - 1 > past + q0 || t0 + CTX_KT > kv_len
                || t0 < past + q0 + CTX_QW - 1 - window; // wave-uniform: diagonal, end of the sequence, start of a row's window
            const int qpos = past + q0 + L;
            if constexpr (BIAS == 1)
            {
#pragma unroll
                for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        s[x2][i] = __builtin_fmaf(
                            s[x2][i], sscale_in, slope2 * (float) (t0 + 32 * x2 + 8 * (i >> 2) + 4 * hh + (i & 3) - qpos));
            }
            if constexpr (BIAS == 2)
            {
#pragma unroll
                for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        s[x2][i] = __builtin_fmaf(
                            s[x2][i], sscale_in, rel_bias_at(qpos, t0 + 32 * x2 + 8 * (i >> 2) + 4 * hh + (i & 3)));
            }
            if (need_mask)
            {
#pragma unroll
                for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                    {
                        const int t = t0 + 32 * x2 + 8 * (i >> 2) + 4 * hh + (i & 3);
                        s[x2][i] = (t <= qpos && t < kv_len && t >= qpos - window) ? s[x2][i] : -__builtin_inff();
                    }
            }
            float mx = s[0][0], my = s[1][0];
#pragma unroll
            for (int i = 1; i < 15; i += 2)
            {
                mx = ctx_max3(mx, s[0][i], s[0][i + 1]);
                my = ctx_max3(my, s[1][i], s[1][i + 1]);
            }
            mx = ctx_max3(mx, my, s[0][15]);
            mx = ctx_max3(mx, s[1][15], s[1][15]);
            const float m_new = ctx_max3(ctx_swap_max<32>(mx), m_run[0], m_run[0]);
            if (__any(m_new > m_run[0])) // wave-uniform: rescale only when some column's max moved
            {
                const float corr = __builtin_amdgcn_exp2f((m_run[0] - m_new) * sscale);
                m_run[0] = m_new;
                l_run[0] *= corr;
#pragma unroll
                for (int db = 0; db < DB; ++db)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        o2[db][i] *= corr;
            }
            const float msc = -m_run[0] * sscale;
            float ls = 0.f;
            uint4 pf[2][2];
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
            {
                float pr[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                {
                    pr[i] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[x2][i], sscale, msc));
                    ls += pr[i];
                }
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2)
                    pf[x2][g2] = make_uint4(pack2<T>(pr[8 * g2], pr[8 * g2 + 1]), pack2<T>(pr[8 * g2 + 2], pr[8 * g2 + 3]),
                        pack2<T>(pr[8 * g2 + 4], pr[8 * g2 + 5]), pack2<T>(pr[8 * g2 + 6], pr[8 * g2 + 7]));
            }
            l_run[0] += ls;
            const int cbl = (lane >> 4) & 1;
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2)
                {
                    const int ra = 32 * x2 + 16 * g2 + 4 * hh + rq, rb = ra + 8;
#pragma unroll
                    for (int db = 0; db < DB; ++db)
                    {
                        const int ch = 2 * (2 * db + cbl) + (rp >> 1), sub = 8 * (rp & 1);
                        const cs16x4_t va
                            = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_cs16x4_t*) (vt + ROWB * ra + 16 * (ch ^ vsw(ra)) + sub));
                        const cs16x4_t vb
                            = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_cs16x4_t*) (vt + ROWB * rb + 16 * (ch ^ vsw(rb)) + sub));
                        const uint2 a2 = __builtin_bit_cast(uint2, va), b2 = __builtin_bit_cast(uint2, vb);
                        o2[db] = ctx_mfma32<T>(make_uint4(a2.x, a2.y, b2.x, b2.y), pf[x2][g2], o2[db]);
                    }
                }
        }
    };

    for (int i = tile_lo; i < ntiles; ++i)
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // my share of tile i has landed (and the next tile's block bases)
        __syncthreads();                                  // everyone's has, and everyone is done with tile i-1
        if (i + 1 < ntiles)
        {
            stage(i + 1, (i + 1) & 1);
            fetch_blocks(min(i + 2, ntiles - 1));
        }
        const int t0 = i * CTX_KT;
        // wave-uniform: tiles past this wave's last row, or before its first row's window, are fully masked
        if (wave_live && t0 < past + q0 + CTX_QW && t0 + CTX_KT > past + q0 - window)
        {
            if constexpr (W32)
                compute32(i & 1, t0);
            else
                compute(i & 1, t0);
        }
    }

    // ---- finish: sum over the 4 lanes of a column, normalise, store 4 consecutive d per lane
    if (!wave_live)
        return;

    if constexpr (W32)
    {
        float l = l_run[0];
        l += __shfl_xor(l, 32);
        const float inv = 1.f / (l + 1e-6f);
        const int qr = q0 + L;
        if (qr < len)
        {
            const int hs = p.head_size; // <= DH: columns past it are the zero padding of `proc`
            uint16_t* orow = p.out + (row0 + qr) * (size_t) (p.num_heads * hs) + (size_t) h * hs;
#pragma unroll
            for (int db = 0; db < DB; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                {
                    const int d = 32 * db + 8 * g + 4 * hh;
                    if (d < hs)
                        *reinterpret_cast<uint2*>(orow + d) = make_uint2(pack2<T>(o2[db][4 * g] * inv, o2[db][4 * g + 1] * inv),
                            pack2<T>(o2[db][4 * g + 2] * inv, o2[db][4 * g + 3] * inv));
                }
        }
    }
    else
    {
    #pragma unroll
        for (int qt = 0; qt < 2; ++qt)
        {
            float l = l_run[qt];
            l += __shfl_xor(l, 16);
            l += __shfl_xor(l, 32);
            const float inv = 1.f / (l + 1e-6f);
            const int qr = q0 + 16 * qt + lr;
            if (qr < len)
            {
                const int hs = p.head_size; // <= DH: columns past it are the zero padding of `proc`
                uint16_t* orow = p.out + (row0 + qr) * (size_t) (p.num_heads * hs) + (size_t) h * hs;
    #pragma unroll
                for (int dt = 0; dt < DT; ++dt)
                    if (16 * dt + 4 * lq < hs)
                        *reinterpret_cast<uint2*>(orow + 16 * dt + 4 * lq) = make_uint2(
                            pack2<T>(o[qt][dt][0] * inv, o[qt][dt][1] * inv), pack2<T>(o[qt][dt][2] * inv, o[qt][dt][3] * inv));
            }
        }
    }
}


int ctx_ilog2(int x)
{
    int l = 0;
    while ((1 << l) < x)
        ++l;
    return l;
}

} // namespace

size_t context_attention_workspace_size(const mi355q_context_attention_params& q)
{
    const size_t hidden
        = (size_t) (q.num_heads + 2 * q.num_kv_heads) * (q.head_size <= 64 ? 64 : (q.head_size <= 128 ? 128 : 256)); // padded rows
    const size_t tokens = q.remove_padding ? (size_t) q.num_tokens : (size_t) q.batch * q.max_context_length;
    return tokens * hidden * 2 + ((size_t) q.batch * sizeof(int32_t) + 255) / 256 * 256;
}

void context_attention(const mi355q_context_attention_params& q, hipStream_t stream)
{
    MQ_CHECK(q.act_type == MI355Q_HALF || q.act_type == MI355Q_BF16, "act_type %d", q.act_type);
    MQ_CHECK(q.qkv && q.out && q.context_lengths, "null tensor");
    MQ_CHECK(q.batch >= 0 && q.num_heads > 0 && q.num_kv_heads > 0 && q.num_heads % q.num_kv_heads == 0, "bad sizes");
    // 64 / 128 run as they are; other multiples of 8 up to 128 are zero-padded to the next of the two in the workspace
    // (the reference's context FMHA is built for 32, 40, 64, 80, 128, 160, 256; its unfused path takes any size)
    // (256: one workgroup per CU -- 128 KiB of K / V stages and ~300 registers per lane -- a functional path, not a tuned one)
    MQ_CHECK(q.head_size >= 8 && q.head_size <= 256 && q.head_size % 8 == 0,
        "context attention: head_size %d not built (multiples of 8 up to 256)", q.head_size);
    const int dh_pad = q.head_size <= 64 ? 64 : (q.head_size <= 128 ? 128 : 256);
    if (q.batch == 0 || q.max_context_length == 0)
        return;
    const size_t need = context_attention_workspace_size(q);
    MQ_CHECK(q.workspace && q.workspace_bytes >= need, "workspace too small: %zu < %zu", q.workspace_bytes, need);
    CtxArgs a{};
    a.qkv = reinterpret_cast<const uint16_t*>(q.qkv);
    a.qkv_bias = reinterpret_cast<const uint16_t*>(q.qkv_bias);
    a.out = reinterpret_cast<uint16_t*>(q.out);
    a.ctx_lens = q.context_lengths;
    a.block_ptrs = q.block_pointers;
    a.kv_scale_orig_quant = q.kv_scale_orig_quant;
    a.batch = q.batch;
    a.max_len = q.max_context_length;
    a.num_heads = q.num_heads;
    a.num_kv_heads = q.num_kv_heads;
    a.head_size = q.head_size;
    a.max_blocks = q.max_blocks_per_seq;
    a.kv_type = q.kv_type == q.act_type ? 0 : (q.kv_type == MI355Q_INT8 ? 1 : 2);
    // chunked context: the sequences already hold sequence_lengths - context_lengths tokens in the cache
    const bool cached = q.sequence_lengths != nullptr;
    const int max_kv = cached ? std::max(q.max_kv_length, q.max_context_length) : q.max_context_length;
    const bool has_cache = q.kv_linear_buffer || q.block_pointers;
    const int capacity = q.kv_linear_buffer ? q.linear_max_seq_len : q.max_blocks_per_seq * q.tokens_per_block;
    // sliding window (cyclic_attention_window_size): mask rule of gptKernels.cu:174-177, and the cache keeps the sink
    // tokens + the last window - sink tokens at their cyclic slots (unfusedAttentionKernels_2_template.h:336-340)
    const bool windowed = q.cyclic_attention_window_size > 0;
    a.mask_window = windowed ? q.cyclic_attention_window_size : (1 << 29);
    a.kv_window = 1 << 29;
    a.sink = a.bubble = 0;
    a.cyc_len = 1 << 29;
    MQ_CHECK(q.sink_token_length >= 0, "sink_token_length %d", q.sink_token_length);
    if (has_cache && (windowed || q.sink_token_length > 0))
    {
        const CyclicLayout cl = cyclic_layout(q.kv_linear_buffer != nullptr, capacity, q.max_blocks_per_seq, q.tokens_per_block,
            q.cyclic_attention_window_size, q.sink_token_length);
        a.kv_window = cl.window;
        a.sink = cl.sink;
        a.bubble = cl.bubble;
        a.cyc_len = cl.cyc_len;
    }
    if (cached)
    {
        MQ_CHECK(has_cache, "sequence_lengths (chunked context) needs a cache to read the earlier tokens from");
        MQ_CHECK(a.kv_type == 0, "chunked context reads K / V from the cache: 16-bit caches only (the reference: \"Paged Context "
                                 "FMHA doesn't work with fp8/int8 kv cache\")");
        // the kernel indexes the cache by position: every token of the sequence must still sit at its own slot
        MQ_CHECK(q.sink_token_length == 0 && max_kv <= a.kv_window,
            "chunked context over a wrapped cyclic cache is not built (kv length %d, window %d, sink %d)", max_kv, a.kv_window,
            q.sink_token_length);
        a.kv_lens = q.sequence_lengths;
    }
    if (q.alibi_slopes)
        a.alibi_slopes = reinterpret_cast<const uint16_t*>(q.alibi_slopes);
    if (q.kv_linear_buffer)
    {
        MQ_CHECK(windowed || q.linear_max_seq_len >= max_kv, "kv length %d exceeds the linear cache (%d tokens per sequence)",
            max_kv, q.linear_max_seq_len);
        MQ_CHECK(q.kv_type == q.act_type || q.kv_type == MI355Q_INT8 || q.kv_type == MI355Q_FP8, "kv cache type %d", q.kv_type);
        if (a.kv_type)
            MQ_CHECK(q.kv_scale_orig_quant, "8-bit kv cache needs kv_scale_orig_quant");
        a.block_ptrs = nullptr;
        a.max_blocks = 1;
        a.tpb_log2 = 30; // one "block" holds the whole sequence
        a.head_tokens = q.linear_max_seq_len;
        a.linear_base = static_cast<char*>(const_cast<void*>(q.kv_linear_buffer));
        a.linear_plane_bytes = (size_t) q.num_kv_heads * q.linear_max_seq_len * q.head_size * (a.kv_type ? 1 : 2);
    }
    else if (q.block_pointers)
    {
        a.head_tokens = q.tokens_per_block;
        MQ_CHECK(q.tokens_per_block > 0 && (q.tokens_per_block & (q.tokens_per_block - 1)) == 0,
            "tokens_per_block must be a power of 2 (kvCacheUtils.h:80-82), got %d", q.tokens_per_block);
        MQ_CHECK(windowed || max_kv <= q.max_blocks_per_seq * q.tokens_per_block,
            "kv length %d exceeds the block table (%d x %d)", max_kv, q.max_blocks_per_seq, q.tokens_per_block);
        MQ_CHECK(q.kv_type == q.act_type || q.kv_type == MI355Q_INT8 || q.kv_type == MI355Q_FP8, "kv cache type %d", q.kv_type);
        if (a.kv_type)
            MQ_CHECK(q.kv_scale_orig_quant, "8-bit kv cache needs kv_scale_orig_quant");
        a.tpb_log2 = ctx_ilog2(q.tokens_per_block);
    }
    a.pos_type = q.position_embedding_type;
    a.rot_dim = q.rotary_embedding_dim;
    a.rot_scale_type = q.rotary_embedding_scale_type;
    a.rot_max_pos = q.rotary_embedding_max_positions;
    a.rot_base = q.rotary_embedding_base;
    a.rot_scale = q.rotary_embedding_scale;
    a.score_scale = 1.f / (sqrtf((float) q.head_size) * (q.q_scaling == 0.f ? 1.f : q.q_scaling));
    if (q.position_embedding_type == 1 || q.position_embedding_type == 2)
        MQ_CHECK(q.rotary_embedding_dim > 0 && q.rotary_embedding_dim <= q.head_size && q.rotary_embedding_dim % 2 == 0,
            "rotary_embedding_dim %d", q.rotary_embedding_dim);
    else if (q.position_embedding_type == 3 || q.position_embedding_type == 4) // kALIBI / kALIBI_WITH_SCALE: the scale is in the slopes
        MQ_CHECK(q.alibi_slopes, "position_embedding_type %d needs alibi_slopes", q.position_embedding_type);
    else if (q.position_embedding_type == 5) // kRELATIVE
    {
        MQ_CHECK(q.relative_attention_bias && q.relative_attention_bias_stride > 0 && q.max_distance >= 0,
            "position_embedding_type 5 needs relative_attention_bias (stride %d, max_distance %d)",
            q.relative_attention_bias_stride, q.max_distance);
        if (q.max_distance > 0) // max_exact = num_buckets / 2 divides inside a logarithm
            MQ_CHECK(q.relative_attention_bias_stride >= 2, "implicit relative attention bias with %d buckets",
                q.relative_attention_bias_stride);
        else
            MQ_CHECK(q.relative_attention_bias_stride >= max_kv,
                "the relative attention bias [heads, %d, %d] does not cover kv length %d", q.relative_attention_bias_stride,
                q.relative_attention_bias_stride, max_kv);
        a.rel_bias = reinterpret_cast<const uint16_t*>(q.relative_attention_bias);
        a.rel_stride = q.relative_attention_bias_stride;
        a.max_distance = q.max_distance;
    }
    else
        MQ_CHECK(q.position_embedding_type == 0, "position_embedding_type %d not supported in context (0 - 5)",
            q.position_embedding_type);
    const size_t hidden = (size_t) (q.num_heads + 2 * q.num_kv_heads) * q.head_size;
    const size_t hidden_pad = (size_t) (q.num_heads + 2 * q.num_kv_heads) * dh_pad;
    const size_t tokens = q.remove_padding ? (size_t) q.num_tokens : (size_t) q.batch * q.max_context_length;
    MQ_CHECK(tokens > 0 && tokens < (1u << 31), "token rows %zu", tokens);
    a.rows = (int) tokens;
    a.dh_pad = dh_pad;
    a.proc = reinterpret_cast<uint16_t*>(q.workspace);
    if (dh_pad != q.head_size)
        MQ_CHECK(!cached, "chunked context with head_size %d is not built (the kernel streams cache rows of 64 / 128)", q.head_size);
    if (q.remove_padding)
    {
        int32_t* offs = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(q.workspace) + tokens * hidden_pad * 2);
        hipLaunchKernelGGL(ctx_offsets_kernel, dim3(1), dim3(64), 0, stream, q.context_lengths, offs, q.batch);
        a.offsets = offs;
    }
    const bool bf = q.act_type == MI355Q_BF16;
    const dim3 pgrid(q.max_context_length, q.batch);
    a.raw_k_cache = q.position_shift_enabled ? 1 : 0;
    if (a.raw_k_cache)
        MQ_CHECK(q.position_embedding_type == 1 || q.position_embedding_type == 2,
            "position_shift_enabled needs a rotary position embedding (type %d)", q.position_embedding_type);
    const size_t plds = (hidden + q.head_size + (a.raw_k_cache ? (size_t) q.num_kv_heads * q.head_size : 0)) * sizeof(float);
    MQ_CHECK(plds <= 64 * 1024, "qkv hidden size %zu too large for the prepare kernel", hidden);
    const bool fast = !a.raw_k_cache && dh_pad == q.head_size && q.head_size <= 128 // (the fast kernel's cos / sin table holds 64 pairs)
        && (q.position_embedding_type == 0 || q.position_embedding_type == 1 || q.position_embedding_type == 3
            || q.position_embedding_type == 4 || q.position_embedding_type == 5
            || (q.position_embedding_type == 2 && q.rotary_embedding_dim == q.head_size));
    // One copy of K / V instead of two: with a 16-bit cache that keeps every token at its own slot (no window, no sink
    // tokens) the attention kernel can read the keys and values this call has just appended -- its chunked-context form
    // with an empty past -- and the prepare pass then writes them to the cache only (a third less traffic in that
    // pass; the values are the same T-rounded bits either way).  tools/lab_ctx_from_cache.py, interleaved: B8 S512
    // 89.6 -> 76.9 us, B4 S2048 D64 165.3 -> 142.8, B1 S4096 206.9 -> 190.8, B4 S2048 272.3 -> 267.5.  Not with many
    // query heads per kv head: K / V are a small part of the pass there, and the scattered 16 KiB block pieces read
    // a little slower than the workspace's contiguous megabyte per head (64q / 8kv S4096: 615 -> 640 us)
    const bool from_cache = !cached && fast && has_cache && a.kv_type == 0 && !windowed && q.sink_token_length == 0
        && (q.head_size == 64 || q.head_size == 128) && (q.num_heads <= 2 * q.num_kv_heads || lab_env("MI355Q_CTX_ONE_COPY"))
        && !lab_env("MI355Q_CTX_TWO_COPIES");
    // q without the round trip through the workspace: the attention kernel's prologue does the prepare pass's work for its
    // own q rows (context_attn.hip, q_from_input); the rotation angles come from a table the prepare pass writes into the
    // (then unused) q part of the workspace
    bool w32_lab = false;
#ifdef MI355Q_LAB
    w32_lab = lab_env("MI355Q_CTX_W32") != nullptr;
#endif
    if (fast && (q.head_size == 64 || q.head_size == 128) && (size_t) q.num_heads * q.head_size * 2 >= 64 * sizeof(float2) && !w32_lab
        && !lab_env("MI355Q_CTX_Q_PASS"))
    {
        a.q_from_input = 1;
        a.cs_table = reinterpret_cast<float2*>(a.proc);
    }
    if (from_cache)
    {
        a.kv_from_cache = 1;
        a.kv_lens = q.context_lengths; // kv length = context length: past 0
    }
    const bool attn_cached = cached || from_cache;
    if (fast && bf)
        hipLaunchKernelGGL(ctx_prepare_fast_kernel<bf16_t>, pgrid, dim3(256), 0, stream, a);
    else if (fast)
        hipLaunchKernelGGL(ctx_prepare_fast_kernel<f16_t>, pgrid, dim3(256), 0, stream, a);
    else if (bf)
        hipLaunchKernelGGL(ctx_prepare_kernel<bf16_t>, pgrid, dim3(256), plds, stream, a);
    else
        hipLaunchKernelGGL(ctx_prepare_kernel<f16_t>, pgrid, dim3(256), plds, stream, a);
    // 4 waves x 32 query rows per workgroup.  (8-wave workgroups -- lock-stepped, or the two halves run one segment
    // apart so that a SIMD's two waves alternate matrix and VALU segments -- measured 3-8 % slower: DESIGN.md 4.4)
    const dim3 agrid(q.num_heads, (q.max_context_length + 32 * 4 - 1) / (32 * 4), q.batch);
    const int bias = a.rel_bias ? 2 : a.alibi_slopes ? 1 : 0;
#ifdef MI355Q_LAB
    // lab: the 32x32x16 form of the same kernel (measured: within -3 .. +3 % of the 16x16x32 form)
    const bool w32 = lab_env("MI355Q_CTX_W32") != nullptr && dh_pad <= 128;
#define MQ_CTX_LAUNCH2(TT, DD, CC, BB)                                                                                 \
    do                                                                                                                 \
    {                                                                                                                  \
        if (DD <= 128 && w32)                                                                                          \
            hipLaunchKernelGGL((ctx_attn_kernel<TT, (DD <= 128 ? DD : 128), 4, CC, BB, true>), agrid, dim3(256), 0, stream, a); \
        else                                                                                                           \
            hipLaunchKernelGGL((ctx_attn_kernel<TT, DD, 4, CC, BB, false>), agrid, dim3(256), 0, stream, a);           \
    } while (0)
#else
#define MQ_CTX_LAUNCH2(TT, DD, CC, BB) hipLaunchKernelGGL((ctx_attn_kernel<TT, DD, 4, CC, BB>), agrid, dim3(256), 0, stream, a)
#endif
#define MQ_CTX_LAUNCH(TT, DD)                                                                                          \
    do                                                                                                                 \
    {                                                                                                                  \
        if (attn_cached && bias == 2)                                                                                  \
            MQ_CTX_LAUNCH2(TT, DD, true, 2);                                                                           \
        else if (attn_cached && bias)                                                                                  \
            MQ_CTX_LAUNCH2(TT, DD, true, 1);                                                                           \
        else if (attn_cached)                                                                                          \
            MQ_CTX_LAUNCH2(TT, DD, true, 0);                                                                           \
        else if (bias == 2)                                                                                            \
            MQ_CTX_LAUNCH2(TT, DD, false, 2);                                                                          \
        else if (bias)                                                                                                 \
            MQ_CTX_LAUNCH2(TT, DD, false, 1);                                                                          \
        else                                                                                                           \
            MQ_CTX_LAUNCH2(TT, DD, false, 0);                                                                          \
    } while (0)
    if (bf && dh_pad == 256)
        MQ_CTX_LAUNCH(bf16_t, 256);
    else if (dh_pad == 256)
        MQ_CTX_LAUNCH(f16_t, 256);
    else if (bf && dh_pad == 128)
        MQ_CTX_LAUNCH(bf16_t, 128);
    else if (bf)
        MQ_CTX_LAUNCH(bf16_t, 64);
    else if (dh_pad == 128)
        MQ_CTX_LAUNCH(f16_t, 128);
    else
        MQ_CTX_LAUNCH(f16_t, 64);
#undef MQ_CTX_LAUNCH
#undef MQ_CTX_LAUNCH2
    MQ_HIP(hipGetLastError());
}

} // namespace mi355q
