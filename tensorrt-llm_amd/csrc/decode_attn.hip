// decode_attn.hip -- one generation step of masked multi-head / grouped-query attention over the
// KVBlockArray paged cache (replaces kernels/decoderMaskedMultiheadAttention/* and the XQA cubins,
// reference: decoderMaskedMultiheadAttentionTemplate.h:1303-2800, kvCacheUtils.h:36-153).
//
// Design (gfx950, HBM-bound):
//   * one workgroup = (sequence b, kv head hk, kv split s), 4 waves; ALL q heads of the kv group are
//     served by the same K/V stream (XQA-style), so the cache is read exactly once.
//   * K/V rows go straight to VGPRs with 16-byte-per-lane loads: LPT = Dh*elem/16 lanes cover one
//     token row, one wave-instruction reads TPW = 64/LPT consecutive tokens = 1 KiB contiguous
//     (tokens of one head are contiguous inside a block: [Hkv][tokens_per_block][Dh]).
//   * q.k partial dots use v_dot2c_f32_{bf16,f16}; the LPT-lane reduction uses DPP only.
//   * every (wave, token slot) keeps an independent online-softmax state (m, l, acc) so the K/V
//     loop has no cross-lane traffic besides the dot reduction; states are merged once at the end
//     (in-wave shuffles, then LDS across waves, then -- for multi-block -- a tiny merge kernel that
//     keeps partials in fp32).
//   * the new token's bias + RoPE + cache append (quantised for int8/fp8 caches) is fused in.
#include "decode_common.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace mi355q
{

namespace
{


// GEN = the general form: one workgroup per QUERY head (G = 1; any q / kv head ratio), every lane resolves its own
// token's block pointer, so beams (cache_indirection, decoderMaskedMultiheadAttentionTemplate.h:1873-1879, 2076, 2441)
// and tiles that straddle blocks work; the kv group's first query head appends the new token.
template <typename T, int KVT, int DH, int G, int U, bool GEN = false, bool PAD = false>
__global__ __launch_bounds__(NW * 64) void decode_attn_kernel(DecodeArgs p)
{
    static_assert(!PAD || GEN, "padded head sizes run in the general form");
    static_assert(!GEN || (G == 1 && U == 1), "the general form serves one query head per workgroup");
    constexpr int KVB = KVT == 0 ? 2 : 1;  // bytes per cache element
    constexpr int E = 16 / KVB;            // cache elements per lane per load
    constexpr int LPT = DH / E;            // lanes per token row
    constexpr int TPW = 64 / LPT;          // token rows per wave-load
    constexpr int TILE = TPW * U;          // tokens per wave iteration
    constexpr int EP = E / 2;              // packed 16-bit pairs per lane
    static_assert(DH % E == 0 && LPT <= 64 && LPT >= 2, "unsupported head size");

    // LDS: phase 1 holds q (G heads), k, v of the new token as floats already rounded to T;
    //      phase 2 (after the loop) holds the per-wave partial states.
    constexpr int RED_FLOATS = NW * G * (DH + 2);
    constexpr int QKV_FLOATS = (G + 2) * DH;
    __shared__ float smem[RED_FLOATS > QKV_FLOATS ? RED_FLOATS : QKV_FLOATS];
    float* q_s = smem;              // [G][DH]
    float* k_s = smem + G * DH;     // [DH]
    float* v_s = k_s + DH;          // [DH]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x % p.splits;
    const int bh = blockIdx.x / p.splits;
    const int Hq = p.num_heads;
    const int heads_in_grid = GEN ? Hq : p.num_kv_heads;
    const int hg = bh % heads_in_grid;
    const int b = bh / heads_in_grid;
    const int hk = GEN ? hg / p.q_per_kv : hg;   // kv head
    const int hq0 = GEN ? hg : hg * G;           // first query head of this workgroup
    const int tl = p.seq_lens[b] - 1; // cached tokens; the current token sits at position tl
    const int kv_len = min(tl, p.kv_window); // slots [0, kv_len) are read (Template.h:1493)
    const int tpb = 1 << p.tpb_log2;
    const bool last_split = split == p.splits - 1;
    const bool appends = !GEN || hg % p.q_per_kv == 0;

    stage_new_token<T, DH, G, PAD>(p, b, hq0, hk, tl, smem);

    const int ts = lane / LPT;  // token slot inside a wave-load
    const int ch = lane % LPT;  // 16-byte chunk of the row
    // PAD (head sizes that are not a power of two, run as the next one): a cache row is hs elements, lanes whose chunk
    // lies past it hold zeros and touch no memory
    const int hs = PAD ? p.hs : DH;
    const bool ch_live = !PAD || ch * E < hs;
    const float k_deq = KVT ? p.kv_scale_quant_orig[0] : 1.f;

    // q chunk of this lane, packed 16-bit pairs (exact: q was rounded to T above)
    uint32_t qp[G][EP];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < EP; ++e)
            qp[g][e] = pack2<T>(q_s[g * DH + ch * E + 2 * e], q_s[g * DH + ch * E + 2 * e + 1]);

    float m[G], l[G], acc[G][E];
#pragma unroll
    for (int g = 0; g < G; ++g)
    {
        m[g] = NEG_BIG;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e)
            acc[g][e] = 0.f;
    }
    // logits are kept in the log2 domain: exp(x) = exp2(x * log2(e))
    const float sscale = p.score_scale * 1.4426950408889634f * k_deq;
    // ALiBi (Template.h:1891-1895, :2026): the slot index stands for the time step, as in the reference's loop
    const bool alibi = p.alibi_slopes != nullptr;
    float slope2[G];
#pragma unroll
    for (int g = 0; g < G; ++g)
        slope2[g] = alibi ? T16<T>::to_float(p.alibi_slopes[hq0 + g]) * 1.4426950408889634f : 0.f;

    // ---- the new token (registers, un-quantised), handled by wave 0 / slot 0 of the last split
    if (last_split && wave == 0 && ts == 0)
    {
        float kf[E], vf[E];
#pragma unroll
        for (int e = 0; e < E; ++e)
        {
            kf[e] = k_s[ch * E + e];
            vf[e] = v_s[ch * E + e];
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
        {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < EP; ++e)
                s = dot2<T>(qp[g][e], pack2<T>(kf[2 * e], kf[2 * e + 1]), s);
            s = group_sum<LPT>(s) * (p.score_scale * 1.4426950408889634f);
            if (p.rel_bias) // the current step's own entry (Template.h:1787-1797)
                s += T16<T>::to_float(p.rel_bias[rel_bias_index(p, hq0 + g, tl, tl)]) * 1.4426950408889634f;
            m[g] = s;
            l[g] = 1.f;
#pragma unroll
            for (int e = 0; e < E; ++e)
                acc[g][e] = vf[e];
        }
    }

    // ---- the cache append, at slot getKVTokenIdx(tl) (Template.h:1485, 2221-2223, 2478-2480).  When the cyclic cache
    // has wrapped, that slot still holds the oldest token of the window, which this step attends to as well (the
    // reference reads kv_loop_length slots and overwrites afterwards, :2207, :2469): then the split that reads the
    // slot writes it, after its own loop and a barrier; otherwise nobody reads the slot and the last split writes early.
    const int slot = kv_slot(p, tl);
    // POS_SHIFT (general form): the live tokens are read at kv_pos_slot(position); the slot about to be overwritten is
    // live iff the token cyc_len before this one is still inside the window, and it is then read at position `sink`
    const bool pos_shift = GEN && p.shift_k != nullptr;
    const int old_token = tl - p.cyc_len;
    const bool wrapped = pos_shift
        ? (old_token >= p.sink && old_token >= (tl > p.kv_window ? p.sink + tl - p.kv_window : p.sink))
        : slot < kv_len;
    const int wrapped_pos = pos_shift ? p.sink : slot; // the loop index at which that slot is read
    auto append_new_token = [&]()
    {
        if (wave == 0 && ts == 0 && appends && ch_live)
        {
            const int blk = slot >> p.tpb_log2;
            const size_t row = ((size_t) hk * p.head_tokens + (slot & (tpb - 1))) * hs * KVB + (size_t) ch * 16;
            uint4 kw, vw;
            if constexpr (KVT == 0)
            {
                uint32_t* kk = reinterpret_cast<uint32_t*>(&kw);
                uint32_t* vv = reinterpret_cast<uint32_t*>(&vw);
#pragma unroll
                for (int e = 0; e < EP; ++e)
                {
                    kk[e] = pack2<T>(k_s[ch * E + 2 * e], k_s[ch * E + 2 * e + 1]);
                    vv[e] = pack2<T>(v_s[ch * E + 2 * e], v_s[ch * E + 2 * e + 1]);
                }
            }
            else
            {
                const float qs = p.kv_scale_orig_quant[0];
                uint8_t* kk = reinterpret_cast<uint8_t*>(&kw);
                uint8_t* vv = reinterpret_cast<uint8_t*>(&vw);
#pragma unroll
                for (int e = 0; e < E; ++e)
                {
                    kk[e] = KVT == 1 ? float_to_int8_sat(k_s[ch * E + e] * qs) : float_to_fp8(k_s[ch * E + e] * qs);
                    vv[e] = KVT == 1 ? float_to_int8_sat(v_s[ch * E + e] * qs) : float_to_fp8(v_s[ch * E + e] * qs);
                }
            }
            if (!pos_shift) // POS_SHIFT: the shift kernel has stored the un-rotated key
                *reinterpret_cast<uint4*>(kv_block_ptr(p, b, 0, blk) + row) = kw;
            *reinterpret_cast<uint4*>(kv_block_ptr(p, b, 1, blk) + row) = vw;
        }
    };
    if (!wrapped && last_split)
        append_new_token();

    // ---- stream the cached tokens of this split
    bool writes_after_loop = false;
    {
        const int ntiles = (kv_len + TILE - 1) / TILE;
        const int tps = (ntiles + p.splits - 1) / p.splits;
        const int tile_end = min(ntiles, (split + 1) * tps);
        writes_after_loop = wrapped && (wrapped_pos / TILE) / tps == split;
        const size_t head_off = (size_t) hk * p.head_tokens * hs * KVB + (size_t) ch * 16;
        // beams: tokens before beam0_context_length come from beam 0 of the batch entry, later ones from the beam
        // cache_indir names; with a wrapped cyclic cache the context may be overwritten: 0 (Template.h:1497-1498)
        const int beam_w = GEN ? max(p.beam_width, 1) : 1;
        const int seq0 = (b / beam_w) * beam_w;
        const int beam0_ctx = (GEN && beam_w > 1) ? (tl > p.kv_window ? 0 : p.input_lengths[b]) : 0;
        const int32_t* indir = (GEN && beam_w > 1) ? p.cache_indir + (size_t) b * p.indir_stride : nullptr;

        for (int tile = split * tps + wave; tile < tile_end; tile += NW)
        {
            const int t0 = tile * TILE;
            uint4 kreg[U], vreg[U];
            bool valid[U];
            if constexpr (GEN)
            {
                const int t = t0 + ts;
                valid[0] = t < kv_len;
                kreg[0] = make_uint4(0, 0, 0, 0);
                vreg[0] = make_uint4(0, 0, 0, 0);
                if (valid[0] && ch_live)
                {
                    const int seq = beam_w > 1 ? seq0 + (t < beam0_ctx ? 0 : indir[t]) : b;
                    const int sl = pos_shift ? kv_pos_slot(p, t, tl) : t;
                    const int blk = sl >> p.tpb_log2;
                    const size_t off = head_off + (size_t) (sl & (tpb - 1)) * hs * KVB;
                    const char* kblock = pos_shift ? p.shift_k + ((size_t) seq * p.max_blocks + blk) * p.shift_block_bytes
                                                   : kv_block_ptr(p, seq, 0, blk);
                    kreg[0] = *reinterpret_cast<const uint4*>(kblock + off);
                    vreg[0] = *reinterpret_cast<const uint4*>(kv_block_ptr(p, seq, 1, blk) + off);
                }
            }
            else
            {
                const int blk = t0 >> p.tpb_log2;
                const char* kbase = kv_block_ptr(p, b, 0, blk) + head_off;
                const char* vbase = kv_block_ptr(p, b, 1, blk) + head_off;
                const int tin = (t0 & (tpb - 1)) + ts;
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    valid[u] = t0 + u * TPW + ts < kv_len;
                    kreg[u] = make_uint4(0, 0, 0, 0);
                    if (valid[u])
                        kreg[u] = *reinterpret_cast<const uint4*>(kbase + (size_t) (tin + u * TPW) * DH * KVB);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    vreg[u] = make_uint4(0, 0, 0, 0);
                    if (valid[u])
                        vreg[u] = *reinterpret_cast<const uint4*>(vbase + (size_t) (tin + u * TPW) * DH * KVB);
                }
            }

            float s[G][U];
#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                uint32_t kp[EP];
                const uint32_t* w = reinterpret_cast<const uint32_t*>(&kreg[u]);
                if constexpr (KVT == 0)
                {
#pragma unroll
                    for (int e = 0; e < EP; ++e)
                        kp[e] = w[e];
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < EP; ++e) // 8-bit values are exact in fp16 and in bf16
                        kp[e] = pack2<T>(kv8_to_float<KVT>(w[e / 2], (2 * e) & 3),
                            kv8_to_float<KVT>(w[e / 2], (2 * e + 1) & 3));
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
                {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < EP; ++e)
                        d = dot2<T>(qp[g][e], kp[e], d);
                    s[g][u] = group_sum<LPT>(d) * sscale;
                }
            }
            if (alibi) // wave-uniform
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    const float dist = (float) (t0 + (GEN ? 0 : u * TPW) + ts - tl);
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        s[g][u] = __builtin_fmaf(slope2[g], dist, s[g][u]);
                }
            }

            if (p.rel_bias) // wave-uniform.  The loop index stands for the time step, as in the reference's loop
            {
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    const int t = t0 + (GEN ? 0 : u * TPW) + ts;
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (valid[u])
                            s[g][u] += T16<T>::to_float(p.rel_bias[rel_bias_index(p, hq0 + g, t, tl)]) * 1.4426950408889634f;
                }
            }

            float pr[G][U];
#pragma unroll
            for (int g = 0; g < G; ++g)
            {
                float mx = m[g];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    mx = valid[u] ? fmaxf(mx, s[g][u]) : mx;
                const float corr = __builtin_amdgcn_exp2f(m[g] - mx);
                m[g] = mx;
                float ps = 0.f;
#pragma unroll
                for (int u = 0; u < U; ++u)
                {
                    pr[g][u] = valid[u] ? __builtin_amdgcn_exp2f(s[g][u] - mx) : 0.f;
                    ps += pr[g][u];
                }
                l[g] = l[g] * corr + ps;
#pragma unroll
                for (int e = 0; e < E; ++e)
                    acc[g][e] *= corr;
            }

#pragma unroll
            for (int u = 0; u < U; ++u)
            {
                float vf[E];
                const uint32_t* w = reinterpret_cast<const uint32_t*>(&vreg[u]);
                if constexpr (KVT == 0)
                {
#pragma unroll
                    for (int e = 0; e < EP; ++e)
                    {
                        vf[2 * e] = T16<T>::lo(w[e]);
                        vf[2 * e + 1] = T16<T>::hi(w[e]);
                    }
                }
                else
                {
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        vf[e] = kv8_to_float<KVT>(w[e / 4], e & 3);
                }
#pragma unroll
                for (int g = 0; g < G; ++g)
                {
                    // cached V of an 8-bit cache is dequantised as float(v8) * kv_scale_quant_orig
                    const float pv = KVT ? pr[g][u] * k_deq : pr[g][u];
#pragma unroll
                    for (int e = 0; e < E; ++e)
                        acc[g][e] = fmaf(pv, vf[e], acc[g][e]);
                }
            }
        }
    }

    if (wrapped) // wave-uniform, the same in every wave of the workgroup
    {
        __syncthreads(); // every wave of this split has read the slot
        if (writes_after_loop)
            append_new_token();
    }

    // ---- merge the TPW token slots of the wave (lanes with equal `ch`)
#pragma unroll
    for (int g = 0; g < G; ++g)
    {
        float mw = m[g];
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1)
            mw = fmaxf(mw, __shfl_xor(mw, o));
        const float c = __builtin_amdgcn_exp2f(m[g] - mw);
        l[g] *= c;
#pragma unroll
        for (int e = 0; e < E; ++e)
            acc[g][e] *= c;
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1)
        {
            l[g] += __shfl_xor(l[g], o);
#pragma unroll
            for (int e = 0; e < E; ++e)
                acc[g][e] += __shfl_xor(acc[g][e], o);
        }
        m[g] = mw;
    }

    // ---- merge the waves through LDS
    __syncthreads(); // everyone is done with q_s/k_s/v_s
    float* red = smem; // [NW][G][DH + 2]
    if (ts == 0)
    {
#pragma unroll
        for (int g = 0; g < G; ++g)
        {
            float* r = red + (wave * G + g) * (DH + 2);
#pragma unroll
            for (int e = 0; e < E; ++e)
                r[ch * E + e] = acc[g][e];
            if (ch == 0)
            {
                r[DH] = m[g];
                r[DH + 1] = l[g];
            }
        }
    }
    __syncthreads();
    finish_states<T, G, DH, NW * 64, NW, PAD>(p, b * heads_in_grid + hg, (size_t) b * Hq + hq0, split, red);
}

// ---- POS_SHIFT (StreamingLLM): the step's rotated copy of the live keys.  The reference runs invokeShiftKCache
// (unfusedAttentionKernels.cu:1776-1947; gptAttentionCommon.cpp:1276-1290) in front of every generation step: the K cache
// holds keys without position embedding, the kernel rotates the key at window position x (the sink tokens, then the
// last window - sink tokens in age order) to position x and writes it, in T, to a scratch cache the attention kernel
// reads instead.  Same here, with the scratch in the cache's own block structure (`shift_k`) so that the attention
// kernel keeps its addressing; and this kernel also stores the NEW token's un-rotated key (bias added, rounded to T)
// into the real cache (k_wo_pos, Template.h:1745-1752, 2216-2233) -- in the workgroup that has just read the slot when
// the slot still holds a live token, else in the extra workgroup x == kv_len.
// grid (positions + 1, batch), block 256.  16-bit caches.
template <typename T>
__global__ __launch_bounds__(256) void shift_k_cache_kernel(DecodeArgs p)
{
    const int x = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int tl = p.seq_lens[b] - 1;
    const int kv_len = min(tl, p.kv_window);
    if (x > kv_len)
        return;
    const int Hq = p.num_heads, Hkv = p.num_kv_heads, D = p.hs;
    const int tpb = 1 << p.tpb_log2;
    const int cyc_start = tl > p.kv_window ? p.sink + tl - p.kv_window : p.sink;
    const int old_token = tl - p.cyc_len;
    const bool old_live = old_token >= p.sink && old_token >= cyc_start;
    const int slot_new = kv_slot(p, tl);
    auto store_new_key = [&]()
    {
        const uint16_t* qkv = reinterpret_cast<const uint16_t*>(p.qkv) + (size_t) b * (Hq + 2 * Hkv) * D + (size_t) Hq * D;
        const uint16_t* bias = p.qkv_bias ? reinterpret_cast<const uint16_t*>(p.qkv_bias) + (size_t) Hq * D : nullptr;
        uint16_t* dst = reinterpret_cast<uint16_t*>(kv_block_ptr(p, b, 0, slot_new >> p.tpb_log2));
        for (int e = tid; e < Hkv * D; e += 256)
        {
            const int hk = e / D, d = e % D;
            uint16_t w = qkv[e];
            if (bias)
                w = T16<T>::from_float(T16<T>::to_float(w) + T16<T>::to_float(bias[e])); // the add is done in T
            dst[((size_t) hk * p.head_tokens + (slot_new & (tpb - 1))) * D + d] = w;
        }
    };
    if (x == kv_len)
    {
        if (!old_live)
            store_new_key();
        return;
    }
    __shared__ float tab[256]; // cos | sin of pair j at [2j], [2j + 1]; rot_dim / 2 <= 128
    const int half = p.rot_dim >> 1;
    if (tid < half)
    {
        // update_rotary_base_n_scale at the cache length (unfusedAttentionKernels.cu:1826-1827), angle of position x
        float base = p.rot_base, scale = p.rot_scale;
        if (p.rot_scale_type == 2)
        {
            if (kv_len > p.rot_max_pos)
            {
                const float bb = kv_len * (scale / p.rot_max_pos) - (scale - 1.f);
                base = base * powf(bb, (float) p.rot_dim / (float) (p.rot_dim - 2));
            }
            scale = 1.f;
        }
        else if (p.rot_scale_type == 1)
            scale = 1.f / scale;
        const float ang = ((float) x * scale) / powf(base, (float) (2 * tid) / (float) p.rot_dim);
        float sn, cs;
        sincosf(ang, &sn, &cs);
        tab[2 * tid] = cs;
        tab[2 * tid + 1] = sn;
    }
    __syncthreads();
    const int tok = x < p.sink ? x : cyc_start + (x - p.sink);
    const int sl = kv_slot(p, tok);
    const int blk = sl >> p.tpb_log2, tin = sl & (tpb - 1);
    const uint16_t* src = reinterpret_cast<const uint16_t*>(kv_block_ptr(p, b, 0, blk));
    uint16_t* dst = reinterpret_cast<uint16_t*>(const_cast<char*>(p.shift_k) + ((size_t) b * p.max_blocks + blk) * p.shift_block_bytes);
    for (int i = tid; i < Hkv * (D / 2); i += 256)
    {
        const int hk = i / (D / 2), j = i % (D / 2);
        const size_t row = ((size_t) hk * p.head_tokens + tin) * D;
        if (j < half)
        {
            const int i0 = p.pos_type == 2 ? j : 2 * j;
            const int i1 = p.pos_type == 2 ? j + half : 2 * j + 1;
            const float cs = tab[2 * j], sn = tab[2 * j + 1];
            const float a = T16<T>::to_float(src[row + i0]), c = T16<T>::to_float(src[row + i1]);
            dst[row + i0] = T16<T>::from_float(cs * a - sn * c);
            dst[row + i1] = T16<T>::from_float(cs * c + sn * a);
        }
        else // elements past rotary_embedding_dim pass through
        {
            const int e0 = p.rot_dim + 2 * (j - half);
            dst[row + e0] = src[row + e0];
            dst[row + e0 + 1] = src[row + e0 + 1];
        }
    }
    if (old_live && tok == old_token) // this workgroup has read the slot the new key goes to
    {
        __syncthreads();
        store_new_key();
    }
}

int ilog2(int x)
{
    int l = 0;
    while ((1 << l) < x)
        ++l;
    return l;
}

struct Plan;
int choose_splits(const mi355q_decode_attention_params& q, const Plan& pl);
int choose_splits_impl(const mi355q_decode_attention_params& q, int tile, int wgs, int max_kv_len)
{
    // every path ends at min(64, tiles): the merge holds 64 rescale factors, and a split without a tile is wasted
    const int max_tiles = std::max(1, (max_kv_len + tile - 1) / tile);
    const int cap = std::min(64, max_tiles);
    if (q.multi_block_mode > 0)
        return std::min(q.multi_block_mode, cap);
    static const int env_splits = lab_env("MI355Q_ATTN_SPLITS") ? atoi(lab_env("MI355Q_ATTN_SPLITS")) : 0; // lab
    if (env_splits > 0)
        return std::min(env_splits, cap);
    // aim for 2 workgroups per CU, never less than NW tiles per workgroup (a workgroup's fixed cost -- staging the new
    // token, the cross-wave merge, its partial state -- is ~8 us: at batch 64 x 1 kv head, ctx 2048, 8 splits take
    // 16 us, 16 splits 23 us)
    int s = (512 + wgs - 1) / wgs;
    s = std::min(s, std::max(1, max_tiles / NW));
    return std::max(1, std::min(s, 64));
}

template <int KVB>
constexpr int tile_tokens(int dh, int u)
{
    return 64 / (dh / (16 / KVB)) * u;
}

int pick_unroll(int dh, int kvb, int tpb, int g)
{
    const int tpw = 64 / (dh / (16 / kvb));
    int u = g >= 8 ? 2 : (g >= 4 ? 4 : 8); // bound the G x U score registers
    while (u > 1 && tpw * u > tpb)
        u >>= 1;
    return u;
}

template <typename T, int KVT, int DH, int G, int U>
void launch(const mi355q_decode_attention_params& q, const DecodeArgs& a, hipStream_t stream)
{
    const int grid = q.batch * q.num_kv_heads * a.splits;
    hipLaunchKernelGGL((decode_attn_kernel<T, KVT, DH, G, U>), dim3(grid), dim3(NW * 64), 0, stream, a);
}

template <typename T, int KVT, int DH, int G>
void dispatch_u(const mi355q_decode_attention_params& q, const DecodeArgs& a, int u, hipStream_t s)
{
    switch (u)
    {
    case 8: launch<T, KVT, DH, G, 8>(q, a, s); break;
    case 4: launch<T, KVT, DH, G, 4>(q, a, s); break;
    case 2: launch<T, KVT, DH, G, 2>(q, a, s); break;
    default: launch<T, KVT, DH, G, 1>(q, a, s); break;
    }
}

template <typename T, int KVT, int DH>
void dispatch_g(const mi355q_decode_attention_params& q, const DecodeArgs& a, int u, hipStream_t s)
{
    const int g = q.num_heads / q.num_kv_heads;
    switch (g)
    {
    case 1: dispatch_u<T, KVT, DH, 1>(q, a, u, s); break;
    case 2: dispatch_u<T, KVT, DH, 2>(q, a, u, s); break;
    case 4: dispatch_u<T, KVT, DH, 4>(q, a, u, s); break;
    case 8: dispatch_u<T, KVT, DH, 8>(q, a, u, s); break;
    default: MQ_UNSUPPORTED("decode attention: q heads per kv head must be 1, 2, 4, 8 (or 16 with head_size 128, tokens_per_block >= 32); got %d", g);
    }
}

template <typename T, int KVT>
void dispatch_dh(const mi355q_decode_attention_params& q, const DecodeArgs& a, int u, hipStream_t s)
{
    switch (q.head_size)
    {
    case 64: dispatch_g<T, KVT, 64>(q, a, u, s); break;
    case 128: dispatch_g<T, KVT, 128>(q, a, u, s); break;
    default: MQ_UNSUPPORTED("decode attention: head_size %d not built (64, 128)", q.head_size);
    }
}

template <typename T>
void dispatch_kv(const mi355q_decode_attention_params& q, const DecodeArgs& a, int u, hipStream_t s)
{
    if (q.kv_type == q.act_type)
        dispatch_dh<T, 0>(q, a, u, s);
    else if (q.kv_type == MI355Q_INT8)
        dispatch_dh<T, 1>(q, a, u, s);
    else if (q.kv_type == MI355Q_FP8)
        dispatch_dh<T, 2>(q, a, u, s);
    else
        MQ_UNSUPPORTED("decode attention: kv cache type %d", q.kv_type);
}

// the general form (decode_attn_kernel<..., GEN = true>): head sizes 32 .. 256, one workgroup per query head
template <typename T, int KVT, int DH>
void launch_general(const mi355q_decode_attention_params& q, const DecodeArgs& a, hipStream_t stream)
{
    const int grid = q.batch * q.num_heads * a.splits;
    if (q.head_size == DH)
        hipLaunchKernelGGL((decode_attn_kernel<T, KVT, DH, 1, 1, true>), dim3(grid), dim3(NW * 64), 0, stream, a);
    else // 48, 80, 96, 104, 112, 144, 160, 192, 224 ... (decoderMaskedMultiheadAttention.cu dispatches on 13 sizes)
        hipLaunchKernelGGL((decode_attn_kernel<T, KVT, DH, 1, 1, true, true>), dim3(grid), dim3(NW * 64), 0, stream, a);
}

int padded_head_size(int hs)
{
    return hs <= 32 ? 32 : (hs <= 64 ? 64 : (hs <= 128 ? 128 : 256));
}

template <typename T, int KVT>
void dispatch_general_dh(const mi355q_decode_attention_params& q, const DecodeArgs& a, hipStream_t s)
{
    switch (padded_head_size(q.head_size))
    {
    case 32: launch_general<T, KVT, 32>(q, a, s); break;
    case 64: launch_general<T, KVT, 64>(q, a, s); break;
    case 128: launch_general<T, KVT, 128>(q, a, s); break;
    default: launch_general<T, KVT, 256>(q, a, s); break;
    }
}

template <typename T>
void dispatch_general(const mi355q_decode_attention_params& q, const DecodeArgs& a, hipStream_t s)
{
    if (q.kv_type == q.act_type)
        dispatch_general_dh<T, 0>(q, a, s);
    else if (q.kv_type == MI355Q_INT8)
        dispatch_general_dh<T, 1>(q, a, s);
    else if (q.kv_type == MI355Q_FP8)
        dispatch_general_dh<T, 2>(q, a, s);
    else
        MQ_UNSUPPORTED("decode attention: kv cache type %d", q.kv_type);
}

#include "decode_attn_gqa.inc"

struct Plan
{
    int kvb, u, tile, splits;
    bool gqa, general;
    int groups;   // workgroup groups that share one K/V stream = arrival words = partial-record groups
    int capacity; // cache slots per sequence
    int window, sink, bubble, cyc_len; // KVBlockArray / KVLinearBuffer constructor (kvCacheUtils.h:71-94, 277-295)
};

int choose_splits(const mi355q_decode_attention_params& q, const Plan& pl)
{
    return choose_splits_impl(q, pl.tile, pl.groups, std::min(q.max_seq_len, pl.window));
}

Plan make_plan(const mi355q_decode_attention_params& q)
{
    MQ_CHECK(q.act_type == MI355Q_HALF || q.act_type == MI355Q_BF16, "act_type %d", q.act_type);
    MQ_CHECK(q.batch > 0 && q.num_heads > 0 && q.num_kv_heads > 0, "bad sizes");
    MQ_CHECK(q.num_heads % q.num_kv_heads == 0, "num_heads %% num_kv_heads");
    const bool linear = q.kv_linear_buffer != nullptr;
    Plan pl{};
    if (linear)
    {
        MQ_CHECK(q.linear_max_seq_len > 0, "linear_max_seq_len %d", q.linear_max_seq_len);
        pl.capacity = q.linear_max_seq_len;
    }
    else
    {
        MQ_CHECK(q.tokens_per_block > 0 && (q.tokens_per_block & (q.tokens_per_block - 1)) == 0,
            "tokens_per_block must be a power of 2 (kvCacheUtils.h:80-82), got %d", q.tokens_per_block);
        MQ_CHECK(q.max_blocks_per_seq > 0, "max_blocks_per_seq %d", q.max_blocks_per_seq);
        pl.capacity = q.max_blocks_per_seq * q.tokens_per_block;
    }
    // the cyclic map (KVBlockArray ctor :89-93, KVLinearBuffer ctor :291-294)
    const CyclicLayout cl = cyclic_layout(linear, pl.capacity, q.max_blocks_per_seq, q.tokens_per_block,
        q.cyclic_attention_window_size, q.sink_token_length);
    pl.window = cl.window;
    pl.sink = cl.sink;
    pl.bubble = cl.bubble;
    pl.cyc_len = cl.cyc_len;
    // without a window every token needs its own slot
    if (q.cyclic_attention_window_size <= 0)
        MQ_CHECK(q.max_seq_len >= 1 && q.max_seq_len <= pl.capacity, "max_seq_len %d exceeds the cache (%d slots per sequence)",
            q.max_seq_len, pl.capacity);
    else
        MQ_CHECK(q.max_seq_len >= 1, "max_seq_len %d", q.max_seq_len);
    const int beams = q.beam_width > 1 ? q.beam_width : 1;
    if (beams > 1)
    {
        MQ_CHECK(q.cache_indirection && q.input_lengths, "beam_width %d needs cache_indirection and input_lengths", beams);
        MQ_CHECK(q.batch % beams == 0, "batch %d is not a multiple of beam_width %d", q.batch, beams);
        MQ_CHECK(q.max_attention_window_size >= std::min(q.max_seq_len - 1, pl.window),
            "cache_indirection rows of %d entries are shorter than the %d slots read", q.max_attention_window_size,
            std::min(q.max_seq_len - 1, pl.window));
    }
    pl.kvb = q.kv_type == q.act_type ? 2 : 1;
    // a lane holds 16 bytes of a cache row: head sizes in multiples of 8 (16-bit cache) / 16 (8-bit cache), up to 256;
    // sizes that are not 32 / 64 / 128 / 256 run padded to the next of those in the general form
    MQ_CHECK(q.head_size >= 16 && q.head_size <= 256 && q.head_size % (16 / pl.kvb) == 0,
        "head_size %d not built (multiples of %d up to 256)", q.head_size, 16 / pl.kvb);
    const int dhp = padded_head_size(q.head_size);
    const int g = q.num_heads / q.num_kv_heads;
    const int tpw = 64 / (dhp / (16 / pl.kvb));
    const int tpb = linear ? (1 << 30) : q.tokens_per_block;
    const bool pos_shift = q.position_shift_enabled != 0;
    if (pos_shift)
    {
        MQ_CHECK(q.position_embedding_type == 1 || q.position_embedding_type == 2,
            "position_shift_enabled needs a rotary position embedding (type %d)", q.position_embedding_type);
        if (q.kv_type != q.act_type)
            MQ_UNSUPPORTED("position_shift_enabled with an 8-bit kv cache is not built (the rotated keys of a step are kept in T)");
        if (beams > 1)
            MQ_UNSUPPORTED("position_shift_enabled with beam search is not built");
    }
    // ALiBi / relative attention bias: the dot-product kernels add the bias
    pl.gqa = beams == 1 && !q.alibi_slopes && !q.relative_attention_bias && !pos_shift && gqa_mfma_applies(q);
    // the general form: beams, head sizes / head ratios the grouped kernels are not built for, tiny blocks, and the
    // position shift (a token's slot is then a function of its position: per-lane addressing)
    pl.general = !pl.gqa
        && (beams > 1 || pos_shift || !(g == 1 || g == 2 || g == 4 || g == 8) || !(q.head_size == 64 || q.head_size == 128)
            || tpb < tpw);
    pl.groups = q.batch * (pl.general ? q.num_heads : q.num_kv_heads);
    if (pl.gqa)
    {
        pl.u = 1;
        pl.tile = GQA_STEP;
    }
    else if (pl.general)
    {
        pl.u = 1;
        pl.tile = tpw;
    }
    else
    {
        pl.u = pick_unroll(q.head_size, pl.kvb, tpb, g);
        pl.tile = tpw * pl.u;
    }
    pl.splits = choose_splits(q, pl);
    return pl;
}

} // namespace

namespace
{
// floats of one group (the query heads one workgroup serves: the kv group, or one head in the general form): its
// partial records padded to a 128-byte line
size_t group_floats(const mi355q_decode_attention_params& q, const Plan& pl)
{
    const size_t g = pl.general ? 1 : (size_t) (q.num_heads / q.num_kv_heads);
    return ((size_t) pl.splits * g * (padded_head_size(q.head_size) + PART_PAD) + 31) / 32 * 32;
}
// bytes of the split-KV area of one call
size_t partials_bytes(const mi355q_decode_attention_params& q, const Plan& pl)
{
    return (size_t) pl.groups * group_floats(q, pl) * sizeof(float);
}
// POS_SHIFT: one 16-bit K block per cache block of every sequence (a linear cache: one "block" per sequence)
size_t shift_block_bytes(const mi355q_decode_attention_params& q)
{
    const size_t tokens = q.kv_linear_buffer ? (size_t) q.linear_max_seq_len : (size_t) q.tokens_per_block;
    return (size_t) q.num_kv_heads * tokens * q.head_size * 2;
}
size_t shift_k_bytes(const mi355q_decode_attention_params& q)
{
    return (size_t) q.batch * (q.kv_linear_buffer ? 1 : q.max_blocks_per_seq) * shift_block_bytes(q);
}
} // namespace

size_t decode_attention_workspace_size(const mi355q_decode_attention_params& q)
{
    make_plan(q);
    // batch' * splits(batch') <= 1024 / Hkv + batch' for every batch' <= batch, so this bound holds
    // for any smaller batch enqueued later (TensorRT asks once, at configure time, with max shapes)
    size_t rows = q.multi_block_mode > 0 ? (size_t) q.batch * q.multi_block_mode
                                         : (size_t) (1024 + q.num_kv_heads - 1) / q.num_kv_heads + q.batch;
    if (lab_env("MI355Q_ATTN_SPLITS")) // lab: the override of choose_splits_impl
        rows = std::max(rows, (size_t) q.batch * std::max(1, atoi(lab_env("MI355Q_ATTN_SPLITS"))));
    const size_t groups = (size_t) q.batch * q.num_heads; // the general form has one group per query head
    // records + the per-group padding to a line
    size_t bytes = rows * q.num_heads * (padded_head_size(q.head_size) + PART_PAD) * sizeof(float) + groups * 32 * sizeof(float);
    if (q.position_shift_enabled)
        bytes = (bytes + 255) / 256 * 256 + shift_k_bytes(q); // the step's rotated keys (shift_k_cache, gptAttentionCommon.cpp:1260-1262)
    return bytes;
}

#ifdef MI355Q_LAB
namespace
{
unsigned long long* g_lab_stamps = nullptr;
unsigned long long* lab_decode_stamps()
{
    if (!lab_env("MI355Q_ATTN_STAMPS"))
        return nullptr;
    if (!g_lab_stamps)
    {
        MQ_HIP(hipHostMalloc((void**) &g_lab_stamps, 4096 * 16 * sizeof(unsigned long long), hipHostMallocMapped));
        std::memset(g_lab_stamps, 0, 4096 * 16 * sizeof(unsigned long long));
    }
    return g_lab_stamps;
}
} // namespace
extern "C" const unsigned long long* mi355q_lab_decode_stamps()
{
    return g_lab_stamps;
}
#endif

void masked_multihead_attention(const mi355q_decode_attention_params& q, hipStream_t stream)
{
    Plan pl = make_plan(q);
    const bool linear = q.kv_linear_buffer != nullptr;
    MQ_CHECK(q.qkv && q.out && q.sequence_lengths && (linear || q.block_pointers), "null tensor");
    MQ_CHECK(((uintptr_t) q.qkv & 15) == 0 && ((uintptr_t) q.qkv_bias & 15) == 0,
        "qkv / qkv_bias must be 16-byte aligned (the new token is staged with 16-byte loads)");
    if (pl.kvb == 1)
        MQ_CHECK(q.kv_scale_orig_quant && q.kv_scale_quant_orig, "8-bit kv cache needs both scales");
    DecodeArgs a{};
    a.qkv = q.qkv;
    a.qkv_bias = q.qkv_bias;
    a.out = q.out;
    a.seq_lens = q.sequence_lengths;
    a.block_ptrs = q.block_pointers;
    a.kv_scale_orig_quant = q.kv_scale_orig_quant;
    a.kv_scale_quant_orig = q.kv_scale_quant_orig;
    a.batch = q.batch;
    a.num_heads = q.num_heads;
    a.num_kv_heads = q.num_kv_heads;
    a.max_blocks = linear ? 1 : q.max_blocks_per_seq;
    a.tpb_log2 = linear ? 30 : ilog2(q.tokens_per_block); // one "block" holds the whole sequence of a linear cache
    a.head_tokens = linear ? q.linear_max_seq_len : q.tokens_per_block;
    a.linear_base = static_cast<const char*>(q.kv_linear_buffer);
    a.hs = q.head_size;
    a.linear_plane_bytes = linear ? (size_t) q.num_kv_heads * q.linear_max_seq_len * q.head_size * pl.kvb : 0;
    a.kv_window = pl.window;
    a.sink = pl.sink;
    a.bubble = pl.bubble;
    a.cyc_len = pl.cyc_len;
    a.q_per_kv = q.num_heads / q.num_kv_heads;
    a.beam_width = q.beam_width > 1 ? q.beam_width : 1;
    a.indir_stride = q.max_attention_window_size;
    a.cache_indir = q.cache_indirection;
    a.input_lengths = q.input_lengths;
    a.splits = pl.splits;
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
    else if (q.position_embedding_type == 3 || q.position_embedding_type == 4) // kALIBI / kALIBI_WITH_SCALE: scale is in the slopes
    {
        MQ_CHECK(q.alibi_slopes, "position_embedding_type %d needs alibi_slopes", q.position_embedding_type);
        a.alibi_slopes = reinterpret_cast<const uint16_t*>(q.alibi_slopes);
    }
    else if (q.position_embedding_type == 5) // kRELATIVE
    {
        MQ_CHECK(q.relative_attention_bias && q.relative_attention_bias_stride > 0,
            "position_embedding_type 5 needs relative_attention_bias and its stride");
        if (q.max_distance > 0)
            MQ_CHECK(q.relative_attention_bias_stride >= 4, "relative attention: %d buckets", q.relative_attention_bias_stride);
        else
            MQ_CHECK(q.relative_attention_bias_stride >= q.max_seq_len, "relative_attention_bias [Hq, %d, %d] is smaller than max_seq_len %d",
                q.relative_attention_bias_stride, q.relative_attention_bias_stride, q.max_seq_len);
        a.rel_bias = reinterpret_cast<const uint16_t*>(q.relative_attention_bias);
        a.rel_stride = q.relative_attention_bias_stride;
        a.max_distance = q.max_distance;
    }
    else
        MQ_CHECK(q.position_embedding_type == 0, "position_embedding_type %d not supported in decode (0 - 5)",
            q.position_embedding_type);
    const size_t need_partials = pl.splits > 1 ? partials_bytes(q, pl) : 0;
    if (pl.splits > 1)
    {
        MQ_CHECK(q.workspace && q.workspace_bytes >= need_partials, "workspace too small: %zu < %zu", q.workspace_bytes, need_partials);
        MQ_CHECK(((uintptr_t) q.workspace & 127) == 0, "workspace must be 128-byte aligned");
        a.group_floats = (int) group_floats(q, pl);
        a.partials = reinterpret_cast<float*>(q.workspace);
        a.arrivals = arrival_lines((size_t) pl.groups, stream);
    }
    if (q.position_shift_enabled)
    {
        // the step's rotated keys behind the partials; the shift kernel fills them (and stores the new un-rotated key)
        const size_t off = (need_partials + 255) / 256 * 256, need = off + shift_k_bytes(q);
        MQ_CHECK(q.workspace && q.workspace_bytes >= need,
            "position_shift_enabled: workspace too small: %zu < %zu (mi355q_decode_attention_workspace_size)", q.workspace_bytes, need);
        MQ_CHECK(((uintptr_t) q.workspace & 15) == 0, "workspace must be 16-byte aligned");
        a.shift_k = static_cast<const char*>(q.workspace) + off;
        a.shift_block_bytes = shift_block_bytes(q);
        const dim3 grid(std::min(q.max_seq_len - 1, pl.window) + 1, q.batch);
        if (q.act_type == MI355Q_BF16)
            hipLaunchKernelGGL(shift_k_cache_kernel<bf16_t>, grid, dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(shift_k_cache_kernel<f16_t>, grid, dim3(256), 0, stream, a);
    }
#ifdef MI355Q_LAB
    a.stamps = lab_decode_stamps();
#endif
    if (pl.gqa)
    {
        if (q.act_type == MI355Q_BF16)
            dispatch_gqa<bf16_t>(q, a, stream);
        else
            dispatch_gqa<f16_t>(q, a, stream);
    }
    else if (pl.general)
    {
        if (q.act_type == MI355Q_BF16)
            dispatch_general<bf16_t>(q, a, stream);
        else
            dispatch_general<f16_t>(q, a, stream);
    }
    else if (q.act_type == MI355Q_BF16)
        dispatch_kv<bf16_t>(q, a, pl.u, stream);
    else
        dispatch_kv<f16_t>(q, a, pl.u, stream);
    MQ_HIP(hipGetLastError());
}

// ---- the entry point that allocates nothing and takes no workspace (mi355q_masked_multihead_attention_static).
// The split-KV partials live in a file-scope device array of the code object: every GPU the library runs on holds its
// own copy at its own address until the process ends.  In host code the array's NAME is only a stand-in for the symbol
// (passing it to a kernel compiles and then faults on the GPU): its device address is resolved once per device with
// hipGetSymbolAddress and only that pointer is used afterwards.
namespace
{
constexpr size_t DECODE_SCRATCH_BYTES = size_t(8) << 20;
static __device__ __align__(256) unsigned char scratch[DECODE_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
std::atomic<void*> g_scratch_ptr[MAX_DEVICES];
std::mutex g_scratch_mutex;

void* decode_scratch_of_current_device()
{
    int dev = -1;
    MQ_HIP(hipGetDevice(&dev));
    MQ_CHECK(dev >= 0 && dev < MAX_DEVICES, "device ordinal %d outside 0..%d", dev, MAX_DEVICES - 1);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p)
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
        if (!p)
        {
            MQ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch))); // this device's copy (the device is current)
            MQ_CHECK(p != nullptr, "hipGetSymbolAddress returned a null address");
            g_scratch_ptr[dev].store(p, std::memory_order_release);
        }
    }
    return p;
}
} // namespace

size_t decode_attention_static_scratch_bytes()
{
    return DECODE_SCRATCH_BYTES;
}

void masked_multihead_attention_static(const mi355q_decode_attention_params& q, hipStream_t stream)
{
    const Plan pl = make_plan(q);
    MQ_CHECK(!q.position_shift_enabled, "position_shift_enabled needs the workspace entry point (the step's rotated keys)");
    mi355q_decode_attention_params c = q;
    c.workspace = nullptr;
    c.workspace_bytes = 0;
    if (pl.splits > 1)
    {
        // what THIS call's partials take (not the max-shape bound of the workspace query)
        const size_t need = partials_bytes(q, pl);
        MQ_CHECK(need <= DECODE_SCRATCH_BYTES,
            "split-KV partials of this call (%zu bytes: batch %d x %d heads x %d splits) exceed the %zu-byte static scratch; "
            "use mi355q_masked_multihead_attention with a workspace",
            need, q.batch, q.num_heads, pl.splits, DECODE_SCRATCH_BYTES);
        c.workspace = decode_scratch_of_current_device();
        c.workspace_bytes = DECODE_SCRATCH_BYTES;
    }
    // nothing has to be cleared: every partial the merge reads is written by this call first, and the arrival words
    // (arrival_ring.hip, not part of this scratch) put themselves back to zero
    masked_multihead_attention(c, stream);
}

} // namespace mi355q
