// context_attn.hip -- the context (prefill) phase of GPTAttention over the paged cache:
//   1. ctx_prepare_kernel: bias + RoPE on q, k of every input token, K / V appended to the KVBlockArray
//      cache (quantised for int8 / fp8 caches), processed q | k | v rows written to the workspace.
//      Replaces invokeAddFusedQKVBiasTranspose / invokeApplyBiasRopeUpdateKVCache
//      (kernels/unfusedAttentionKernels.cu; called from gptAttentionCommon.cpp:930-990).
//   2. ctx_attn_kernel: causal flash attention among the input tokens of each sequence (the reference's
//      context FMHA, gptAttentionCommon.cpp:1000-1075 -> contextFusedMultiHeadAttention/fmhaRunner.cpp; cubins
//      only, behaviour spec = unfused softmax(QK^T / sqrt(d)) V with the causal mask), on MFMA:
//        S^T = K . Q^T per 32-token step (K rows straight to A-fragment layout), online softmax per query
//        column in the log2 domain, the probability registers are the B fragment of O^T += V^T . P^T,
//        V through a per-wave swizzled LDS tile read back transposed (ds_read_b64_tr_b16) -- the decode GQA
//        kernel's structure with the 16 "columns" being query rows instead of query heads.
//      workgroup = (128-row query tile, q head, sequence), 4 waves x 32 query rows.
// Scope: causal mask, fp16 / bf16, head size 64 / 128, MHA / GQA / MQA, padded or packed input, paged cache;
// the attention uses the un-quantised K / V of the input tokens (as the reference's context FMHA does).
#include "decode_common.h"

namespace mi355q
{

namespace
{

typedef __attribute__((ext_vector_type(8))) __bf16 cbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 cf16x8_t;
typedef __attribute__((ext_vector_type(4))) float cf32x4_t;
typedef __attribute__((ext_vector_type(4))) short cs16x4_t;
typedef __attribute__((address_space(3))) cs16x4_t lds_cs16x4_t;

struct CtxArgs
{
    const uint16_t* qkv;      // input rows [(Hq + 2 Hkv) * Dh]
    const uint16_t* qkv_bias; // or null
    uint16_t* proc;           // workspace: processed rows, same indexing as qkv
    uint16_t* out;            // [tokens, Hq * Dh]
    const int32_t* ctx_lens;  // device [B]
    const int32_t* offsets;   // device [B]: first token row of each sequence (packed input) or null (padded: b * S)
    const int64_t* block_ptrs;
    const float* kv_scale_orig_quant;
    int batch, max_len, num_heads, num_kv_heads, head_size;
    int max_blocks, tpb_log2, kv_type; // 0 = T, 1 = int8, 2 = fp8
    int pos_type, rot_dim, rot_scale_type, rot_max_pos;
    float rot_base, rot_scale, score_scale;
};

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
    uint16_t* dst = p.proc + row * hidden;
    extern __shared__ float xs[]; // [hidden] floats already rounded to T
    for (int i = threadIdx.x; i < hidden; i += 256)
    {
        float x = T16<T>::to_float(src[i]);
        if (p.qkv_bias)
            x = T16<T>::to_float(T16<T>::from_float(x + T16<T>::to_float(p.qkv_bias[i]))); // add in T
        xs[i] = x;
    }
    __syncthreads();
    if (p.pos_type == 1 || p.pos_type == 2) // kROPE_GPTJ / kROPE_GPT_NEOX on q and k heads, position s
    {
        const int half = p.rot_dim >> 1;
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
        for (int i = threadIdx.x; i < (Hq + Hkv) * half; i += 256)
        {
            const int vec = i / half, j = i % half;
            float* x = xs + vec * D;
            const int i0 = p.pos_type == 2 ? j : 2 * j;
            const int i1 = p.pos_type == 2 ? j + half : 2 * j + 1;
            const float ang = ((float) s * scale) / powf(base, (float) (2 * j) / (float) p.rot_dim);
            float sn, cs;
            sincosf(ang, &sn, &cs);
            const float a = x[i0], c = x[i1];
            x[i0] = T16<T>::to_float(T16<T>::from_float(cs * a - sn * c));
            x[i1] = T16<T>::to_float(T16<T>::from_float(cs * c + sn * a));
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < hidden; i += 256)
        dst[i] = T16<T>::from_float(xs[i]);
    // cache append at position s: block [Hkv][tokens_per_block][Dh]
    if (p.block_ptrs)
    {
        const int tpb = 1 << p.tpb_log2;
        const int64_t* bp = p.block_ptrs + (size_t) b * 2 * p.max_blocks;
        const int blk = s >> p.tpb_log2, tin = s & (tpb - 1);
        const float qs = p.kv_type ? p.kv_scale_orig_quant[0] : 1.f;
        for (int i = threadIdx.x; i < 2 * Hkv * D; i += 256)
        {
            const int kv = i / (Hkv * D), rem = i % (Hkv * D), hk = rem / D, d = rem % D;
            const float x = xs[(Hq + kv * Hkv + hk) * D + d];
            const size_t e = ((size_t) hk * tpb + tin) * D + d;
            char* base = reinterpret_cast<char*>(bp[kv * p.max_blocks + blk]);
            if (p.kv_type == 0)
                reinterpret_cast<uint16_t*>(base)[e] = T16<T>::from_float(x);
            else if (p.kv_type == 1)
                reinterpret_cast<uint8_t*>(base)[e] = float_to_int8_sat(x * qs);
            else
                reinterpret_cast<uint8_t*>(base)[e] = float_to_fp8(x * qs);
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

constexpr int CTX_STEP = 32; // key tokens per step
constexpr int CTX_QW = 32;   // query rows per wave
constexpr int CTX_QB = 128;  // query rows per workgroup

// grid (ceil(max_len / 128), Hq, B), block 256
template <typename T, int DH>
__global__ __launch_bounds__(256) void ctx_attn_kernel(CtxArgs p)
{
    constexpr int KS = DH / 32;      // QK k-steps
    constexpr int DT = DH / 16;      // output d-tiles
    constexpr int ROWB = DH * 2;     // bytes of a V row in the LDS tile
    constexpr int CH = ROWB / 16;    // 16-byte chunks per row: 16 / 8
    __shared__ __attribute__((aligned(16))) uint8_t vt_all[4 * CTX_STEP * ROWB];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int b = blockIdx.z, h = blockIdx.y;
    const int hk = h / (p.num_heads / p.num_kv_heads);
    const int len = p.ctx_lens[b];
    const int q0 = blockIdx.x * CTX_QB + wave * CTX_QW; // this wave's first query row
    if (q0 >= len) // wave-uniform; no barriers below
        return;
    const int hidden = (p.num_heads + 2 * p.num_kv_heads) * DH;
    const size_t row0 = (size_t) (p.offsets ? p.offsets[b] : b * p.max_len);
    const uint16_t* qbase = p.proc + row0 * hidden + (size_t) h * DH;
    const uint16_t* kbase = p.proc + row0 * hidden + (size_t) (p.num_heads + hk) * DH;
    const uint16_t* vbase = p.proc + row0 * hidden + (size_t) (p.num_heads + p.num_kv_heads + hk) * DH;
    uint8_t* vt = vt_all + wave * CTX_STEP * ROWB;
    auto voff = [](int row, int ch) { return ROWB * row + 16 * (ch ^ ((((row & 3) << 2) | ((row >> 2) & 3)) & (CH - 1))); };

    // Q^T fragments (B operand): column = query row q0 + 16*qt + lr, k = d = 32*ks + 8*lq + j
    uint4 qf[2][KS];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
    {
        const int qr = min(q0 + 16 * qt + lr, len - 1); // rows past the sequence re-read its last row; never stored
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            qf[qt][ks] = *reinterpret_cast<const uint4*>(qbase + (size_t) qr * hidden + 32 * ks + 8 * lq);
    }
    const float sscale = p.score_scale * 1.4426950408889634f;

    float m_run[2] = {NEG_BIG, NEG_BIG}, l_run[2] = {0.f, 0.f};
    cf32x4_t o[2][DT];
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            o[qt][dt] = cf32x4_t{0.f, 0.f, 0.f, 0.f};

    const int t_end = min(len, q0 + CTX_QW); // causal: keys 0 .. last query row of the wave
    for (int t0 = 0; t0 < t_end; t0 += CTX_STEP)
    {
        const int tmax = len - 1 - t0;
        // ---- S^T tiles [key tile x][query tile qt]
        cf32x4_t s[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
        {
            const uint16_t* krow = kbase + (size_t) (t0 + min(16 * x + lr, tmax)) * hidden + 8 * lq;
            uint4 kf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                kf[ks] = *reinterpret_cast<const uint4*>(krow + 32 * ks);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt)
            {
                s[x][qt] = cf32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    s[x][qt] = ctx_mfma<T>(kf[ks], qf[qt][ks], s[x][qt]);
            }
        }
        // ---- V tile -> LDS: rows (64 / CH) * i + lane / CH, chunk lane % CH
        {
            constexpr int RPI = 64 / CH; // rows per load instruction: 4 / 8
#pragma unroll
            for (int i = 0; i < CTX_STEP / RPI; ++i)
            {
                const int r = RPI * i + lane / CH, ch = lane % CH;
                *reinterpret_cast<uint4*>(vt + voff(r, ch))
                    = *reinterpret_cast<const uint4*>(vbase + (size_t) (t0 + min(r, tmax)) * hidden + 8 * ch);
            }
        }
        // ---- causal mask + online softmax per query column
        uint4 pf[2];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
        {
            const int qpos = q0 + 16 * qt + lr;
            float mx = NEG_BIG;
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                {
                    const int t = t0 + 16 * x + 4 * lq + rr;
                    s[x][qt][rr] = (t <= qpos && t < len) ? s[x][qt][rr] * sscale : NEG_BIG;
                    mx = fmaxf(mx, s[x][qt][rr]);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[qt], mx);
            const float corr = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
            m_run[qt] = m_new;
            l_run[qt] *= corr;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    o[qt][dt][rr] *= corr;
            float pA[4], pB[4];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
            {
                // a fully masked column keeps m_new = NEG_BIG: exp2(NEG_BIG - NEG_BIG) = 1 would count masked keys
                pA[rr] = s[0][qt][rr] > NEG_BIG * 0.5f ? __builtin_amdgcn_exp2f(s[0][qt][rr] - m_new) : 0.f;
                pB[rr] = s[1][qt][rr] > NEG_BIG * 0.5f ? __builtin_amdgcn_exp2f(s[1][qt][rr] - m_new) : 0.f;
                l_run[qt] += pA[rr] + pB[rr];
            }
            pf[qt] = make_uint4(pack2<T>(pA[0], pA[1]), pack2<T>(pA[2], pA[3]), pack2<T>(pB[0], pB[1]), pack2<T>(pB[2], pB[3]));
        }
        // ---- O^T += V^T . P^T  (A operand rows d = 16*dt + lr; k index j < 4 -> key 4*lq + j, j >= 4 -> key 16 + 4*lq + j - 4)
        const int rq = lr >> 2, rp = lr & 3;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
        {
            const int ch = 2 * dt + (rp >> 1), sub = 8 * (rp & 1);
            const cs16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_cs16x4_t*) (vt + voff(4 * lq + rq, ch) + sub));
            const cs16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_cs16x4_t*) (vt + voff(16 + 4 * lq + rq, ch) + sub));
            const uint2 a2 = __builtin_bit_cast(uint2, va), b2 = __builtin_bit_cast(uint2, vb);
            const uint4 vf = make_uint4(a2.x, a2.y, b2.x, b2.y);
            o[0][dt] = ctx_mfma<T>(vf, pf[0], o[0][dt]);
            o[1][dt] = ctx_mfma<T>(vf, pf[1], o[1][dt]);
        }
    }

    // ---- finish: sum over the 4 lanes of a column, normalise, store 4 consecutive d per lane
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
            uint16_t* orow = p.out + (row0 + qr) * (size_t) (p.num_heads * DH) + (size_t) h * DH;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                *reinterpret_cast<uint2*>(orow + 16 * dt + 4 * lq) = make_uint2(
                    pack2<T>(o[qt][dt][0] * inv, o[qt][dt][1] * inv), pack2<T>(o[qt][dt][2] * inv, o[qt][dt][3] * inv));
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
    const size_t hidden = (size_t) (q.num_heads + 2 * q.num_kv_heads) * q.head_size;
    const size_t tokens = q.remove_padding ? (size_t) q.num_tokens : (size_t) q.batch * q.max_context_length;
    return tokens * hidden * 2 + ((size_t) q.batch * sizeof(int32_t) + 255) / 256 * 256;
}

void context_attention(const mi355q_context_attention_params& q, hipStream_t stream)
{
    MQ_CHECK(q.act_type == MI355Q_HALF || q.act_type == MI355Q_BF16, "act_type %d", q.act_type);
    MQ_CHECK(q.qkv && q.out && q.context_lengths, "null tensor");
    MQ_CHECK(q.batch >= 0 && q.num_heads > 0 && q.num_kv_heads > 0 && q.num_heads % q.num_kv_heads == 0, "bad sizes");
    MQ_CHECK(q.head_size == 64 || q.head_size == 128, "context attention: head_size %d not built (64, 128)", q.head_size);
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
    if (q.block_pointers)
    {
        MQ_CHECK(q.tokens_per_block > 0 && (q.tokens_per_block & (q.tokens_per_block - 1)) == 0,
            "tokens_per_block must be a power of 2 (kvCacheUtils.h:80-82), got %d", q.tokens_per_block);
        MQ_CHECK(q.max_context_length <= q.max_blocks_per_seq * q.tokens_per_block,
            "max_context_length %d exceeds the block table (%d x %d)", q.max_context_length, q.max_blocks_per_seq,
            q.tokens_per_block);
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
    else
        MQ_CHECK(q.position_embedding_type == 0, "position_embedding_type %d not supported in context (0,1,2)",
            q.position_embedding_type);
    const size_t hidden = (size_t) (q.num_heads + 2 * q.num_kv_heads) * q.head_size;
    const size_t tokens = q.remove_padding ? (size_t) q.num_tokens : (size_t) q.batch * q.max_context_length;
    a.proc = reinterpret_cast<uint16_t*>(q.workspace);
    if (q.remove_padding)
    {
        int32_t* offs = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(q.workspace) + tokens * hidden * 2);
        hipLaunchKernelGGL(ctx_offsets_kernel, dim3(1), dim3(64), 0, stream, q.context_lengths, offs, q.batch);
        a.offsets = offs;
    }
    const bool bf = q.act_type == MI355Q_BF16;
    const dim3 pgrid(q.max_context_length, q.batch);
    const size_t plds = hidden * sizeof(float);
    MQ_CHECK(plds <= 64 * 1024, "qkv hidden size %zu too large for the prepare kernel", hidden);
    if (bf)
        hipLaunchKernelGGL(ctx_prepare_kernel<bf16_t>, pgrid, dim3(256), plds, stream, a);
    else
        hipLaunchKernelGGL(ctx_prepare_kernel<f16_t>, pgrid, dim3(256), plds, stream, a);
    const dim3 agrid((q.max_context_length + CTX_QB - 1) / CTX_QB, q.num_heads, q.batch);
    if (bf && q.head_size == 128)
        hipLaunchKernelGGL((ctx_attn_kernel<bf16_t, 128>), agrid, dim3(256), 0, stream, a);
    else if (bf)
        hipLaunchKernelGGL((ctx_attn_kernel<bf16_t, 64>), agrid, dim3(256), 0, stream, a);
    else if (q.head_size == 128)
        hipLaunchKernelGGL((ctx_attn_kernel<f16_t, 128>), agrid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL((ctx_attn_kernel<f16_t, 64>), agrid, dim3(256), 0, stream, a);
    MQ_HIP(hipGetLastError());
}

} // namespace mi355q
