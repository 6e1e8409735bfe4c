// moe.hip -- the mixture-of-experts layer around the grouped weight-only GEMM (woq_prefill.hip, moe_weight_only_gemm):
// CutlassMoeFCRunner<T, uint8_t | uint4b_t>::runMoe (kernels/mixtureOfExperts/moe_kernels.cu:922-1035) behind the
// MixtureOfExperts plugin (plugins/mixtureOfExperts/mixtureOfExpertsPlugin.cpp:388-431).  Steps and what each replaces:
//   1. moe_route_kernel      softmax over the experts + top-k (moeSoftmax / moeTopK / topkGatingSoftmax, :62-178,
//                            :194-393): one wave per token, the row in registers, k rounds of a wave arg-max (ties go
//                            to the lower expert index, cub::ArgMax's rule)
//   2. moe_sort_*            the stable sort of the (token, k) pairs by expert (CubKeyValueSorter, :499-558) as a
//                            counting sort: experts are few, so histogram -> scan -> stable scatter; one workgroup does
//                            all three when the pairs fit one chunk (decode), else three launches.  Also the inclusive
//                            per-expert row counts (computeTotalRowsBeforeExpertKernel, :566-578) and both
//                            permutation maps
//   3. (the gather of the token rows into expert order, expandInputRowsKernel :593-623, is not a pass of its own: the
//      grouped GEMM 1 reads its activation rows through the dest_to_src map while it stages them)
//   4. grouped GEMM 1 (+ bias + activation in its epilogue; identity for a gated activation)
//   5. moe_gated_act_kernel  fc1 * act(gate) (doGatedActivationKernel, :795-817)
//   6. grouped GEMM 2
//   7. moe_finalize_kernel   un-permute, + bias 2, x routing scale, k-way sum in T, renormalise
//                            (finalizeMoeRoutingKernel, :646-717)
// Everything but the two GEMMs is HBM-bound byte / index work: 16-byte accesses, one workgroup per row.
#include "kernels.h"
#include <algorithm>
#include <cfloat>

namespace mi355q
{

namespace
{

constexpr int MOE_CHUNK = 2048;     // (token, k) pairs one sort workgroup owns: 256 threads x 8
constexpr int MOE_MAX_EXPERTS = 256;

struct MoeArgs
{
    const uint16_t* input;
    const float* gating;
    const uint8_t* finished;
    uint16_t* output;
    const uint16_t* bias2;
    float* scales;        // [n] routing probability of pair (token, j) at token * k + j
    int32_t* experts;     // [n] local expert of the pair, or the sentinel `buckets - 1`
    int32_t* src_to_dest; // [n] expanded source row (j * num_rows + token) -> row in expert order
    int32_t* dest_to_src; // [n]
    int32_t* blk_counts;  // [buckets, nblocks] histogram, then exclusive offsets
    int64_t* total_rows;  // [experts_per_node] inclusive
    uint16_t* fc1;        // [n, inter]
    uint16_t* glu;        // [n, 2 inter] (gated activations)
    uint16_t* fc2;        // [n, hidden]
    const float* fc2_slabs; // split-K w2 GEMM: fp32 [fc2_splits][n][hidden] instead of fc2, summed by the finalize step
    int fc2_splits;
    int num_rows, hidden, inter, num_experts, k, n;
    int start_expert, experts_per_node, buckets, nblocks;
    int act_fn, scale_mode, add_bias; // scale_mode: 0 none, 1 routing scale, 2 renormalised
};

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    return v;
}

// ---- 1. routing.  grid ceil(num_rows / 4), block 256: one wave per token; lane l holds experts l, l + 64, ...
template <int VPL>
__global__ __launch_bounds__(256) void moe_route_kernel(MoeArgs a)
{
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.num_rows)
        return;
    float v[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i)
    {
        const int e = lane + 64 * i;
        v[i] = e < a.num_experts ? a.gating[(size_t) row * a.num_experts + e] : -FLT_MAX;
    }
    float mx = v[0];
#pragma unroll
    for (int i = 1; i < VPL; ++i)
        mx = fmaxf(mx, v[i]);
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i)
    {
        const int e = lane + 64 * i;
        v[i] = e < a.num_experts ? expf(v[i] - mx) : 0.f;
        s += v[i];
    }
    const float inv = 1.f / wave_sum(s);
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        v[i] = lane + 64 * i < a.num_experts ? v[i] * inv : -1.f; // probabilities; dead slots below any of them
    const bool active = a.finished ? !a.finished[row] : true;
    for (int j = 0; j < a.k; ++j)
    {
        // arg-max over the row: larger value wins, equal values go to the lower index
        float bv = v[0];
        int be = lane;
#pragma unroll
        for (int i = 1; i < VPL; ++i)
            if (v[i] > bv)
            {
                bv = v[i];
                be = lane + 64 * i;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
        {
            const float ov = __shfl_xor(bv, o);
            const int oe = __shfl_xor(be, o);
            if (ov > bv || (ov == bv && oe < be))
            {
                bv = ov;
                be = oe;
            }
        }
#pragma unroll
        for (int i = 0; i < VPL; ++i) // the winner leaves the row
            if (lane + 64 * i == be)
                v[i] = -1.f;
        if (lane == 0)
        {
            // experts another rank owns (expert parallelism) and finished rows go to the sentinel bucket (:165-172)
            const bool mine = active && be >= a.start_expert && be < a.start_expert + a.experts_per_node;
            a.scales[(size_t) row * a.k + j] = bv;
            a.experts[(size_t) row * a.k + j] = mine ? be - a.start_expert : a.buckets - 1;
        }
    }
}

// ---- 2. counting sort of the n = num_rows * k pairs by bucket, stable in pair order i = token * k + j.
// Workgroup b owns pairs [b * MOE_CHUNK, +MOE_CHUNK), thread t of it 8 consecutive ones.
__device__ __forceinline__ void moe_load_keys(const MoeArgs& a, int base, int (&key)[8])
{
#pragma unroll
    for (int i = 0; i < 8; ++i)
    {
        const int idx = base + i;
        key[i] = idx < a.n ? a.experts[idx] : -1;
    }
}

__global__ __launch_bounds__(256) void moe_sort_count_kernel(MoeArgs a)
{
    __shared__ int hist[MOE_MAX_EXPERTS + 1];
    for (int e = threadIdx.x; e < a.buckets; e += 256)
        hist[e] = 0;
    __syncthreads();
    int key[8];
    moe_load_keys(a, blockIdx.x * MOE_CHUNK + threadIdx.x * 8, key);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (key[i] >= 0)
            atomicAdd(&hist[key[i]], 1);
    __syncthreads();
    for (int e = threadIdx.x; e < a.buckets; e += 256)
        a.blk_counts[e * a.nblocks + blockIdx.x] = hist[e];
}

// one workgroup: exclusive scan of blk_counts in (bucket, block) order, and the inclusive per-expert totals
__global__ __launch_bounds__(256) void moe_sort_scan_kernel(MoeArgs a)
{
    __shared__ int part[256];
    const int total = a.buckets * a.nblocks;
    const int per = (total + 255) / 256;
    const int lo = threadIdx.x * per, hi = min(lo + per, total);
    int s = 0;
    for (int i = lo; i < hi; ++i)
        s += a.blk_counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int acc = 0;
        for (int i = 0; i < 256; ++i)
        {
            const int c = part[i];
            part[i] = acc;
            acc += c;
        }
    }
    __syncthreads();
    int acc = part[threadIdx.x];
    for (int i = lo; i < hi; ++i)
    {
        const int c = a.blk_counts[i];
        a.blk_counts[i] = acc;
        acc += c;
        // the last block of an expert's row: acc is the count of pairs of experts <= e
        if (i % a.nblocks == a.nblocks - 1 && i / a.nblocks < a.experts_per_node)
            a.total_rows[i / a.nblocks] = acc;
    }
}

// stable placement.  `offset_of(e)` = first destination row of this workgroup's pairs of bucket e.
template <typename OffsetFn>
__device__ __forceinline__ void moe_place(const MoeArgs& a, int blk, const int (&key)[8], int* wave_tot, OffsetFn offset_of)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int base = blk * MOE_CHUNK + threadIdx.x * 8;
    for (int e = 0; e < a.buckets; ++e)
    {
        int c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            c += key[i] == e;
        // exclusive scan of c over the 256 threads
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1)
        {
            const int up = __shfl_up(incl, o);
            if (lane >= o)
                incl += up;
        }
        __syncthreads(); // wave_tot of the previous bucket has been read
        if (lane == 63)
            wave_tot[wave] = incl;
        __syncthreads();
        int before = incl - c;
        for (int w = 0; w < wave; ++w)
            before += wave_tot[w];
        // (a one-wave launch -- the small sort of a decode step -- has wave == 0: the barriers cost nothing)
        if (c)
        {
            int dest = offset_of(e) + before;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (key[i] == e)
                {
                    const int idx = base + i;                                // pair (token, j), token-major
                    const int src = (idx % a.k) * a.num_rows + idx / a.k;    // expanded source row (:174)
                    a.src_to_dest[src] = dest;
                    a.dest_to_src[dest] = src;
                    ++dest;
                }
        }
    }
}

__global__ __launch_bounds__(256) void moe_sort_scatter_kernel(MoeArgs a)
{
    __shared__ int wave_tot[4];
    int key[8];
    moe_load_keys(a, blockIdx.x * MOE_CHUNK + threadIdx.x * 8, key);
    moe_place(a, blockIdx.x, key, wave_tot, [&](int e) { return a.blk_counts[e * a.nblocks + blockIdx.x]; });
}

// n <= MOE_CHUNK (decode): histogram, scan and placement in one workgroup
// (launched with 64 threads when the pairs fit 64 x 8: no cross-wave step at all)
__global__ __launch_bounds__(256) void moe_sort_small_kernel(MoeArgs a)
{
    __shared__ int hist[MOE_MAX_EXPERTS + 1];
    __shared__ int wave_tot[4];
    for (int e = threadIdx.x; e < a.buckets; e += blockDim.x)
        hist[e] = 0;
    __syncthreads();
    int key[8];
    moe_load_keys(a, threadIdx.x * 8, key);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (key[i] >= 0)
            atomicAdd(&hist[key[i]], 1);
    __syncthreads();
    if (threadIdx.x == 0)
    {
        int acc = 0;
        for (int e = 0; e < a.buckets; ++e)
        {
            const int c = hist[e];
            hist[e] = acc;
            acc += c;
            if (e < a.experts_per_node)
                a.total_rows[e] = acc;
        }
    }
    __syncthreads();
    moe_place(a, 0, key, wave_tot, [&](int e) { return hist[e]; });
}

__device__ __forceinline__ float moe_act(float x, int fn) // Swiglu -> SiLU, Geglu -> GELU_taylor (:828-831)
{
    return fn == 3 ? x / (1.f + __expf(-x)) : 0.5f * x * (1.f + tanhf(0.7978845608028654f * x * (1.f + 0.044715f * x * x)));
}

// ---- 5. gated activation.  grid (n, column chunks of 2048), block 256: out[i] = fc1[i] * T(act(float(gate[i]))),
// the product in T.  One 16-byte vector per thread: a decode step has few rows, the columns fill the chip.
template <typename T>
__global__ __launch_bounds__(256) void moe_gated_act_kernel(MoeArgs a)
{
    const int row = blockIdx.x;
    if (row >= a.total_rows[a.experts_per_node - 1])
        return;
    const uint4* x = reinterpret_cast<const uint4*>(a.glu + (size_t) row * 2 * a.inter);
    const uint4* g = x + a.inter / 8;
    uint4* out = reinterpret_cast<uint4*>(a.fc1 + (size_t) row * a.inter);
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i < a.inter / 8)
    {
        const uint4 xv = x[i], gv = g[i];
        const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, gw[4] = {gv.x, gv.y, gv.z, gv.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            const float g0 = T16<T>::to_float(T16<T>::from_float(moe_act(T16<T>::lo(gw[j]), a.act_fn)));
            const float g1 = T16<T>::to_float(T16<T>::from_float(moe_act(T16<T>::hi(gw[j]), a.act_fn)));
            o[j] = pack2<T>(T16<T>::lo(xw[j]) * g0, T16<T>::hi(xw[j]) * g1);
        }
        out[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---- 7. finalize.  grid (num_rows, column chunks of 2048), block 256:
// out[token] = sum_j scale_j * T(fc2[dest(token, j)] + bias2[expert_j]), the running sum rounded to T after every term,
// as the reference's `T thread_output` does.  A thread owns 8 columns; the k rows are read in the order of j.
template <typename T>
__global__ __launch_bounds__(256) void moe_finalize_kernel(MoeArgs a)
{
    const int row = blockIdx.x;
    const int c8 = blockIdx.y * 256 + threadIdx.x; // 16-byte vector of the row
    if (c8 >= a.hidden / 8)
        return;
    const int64_t num_valid = a.total_rows[a.experts_per_node - 1];
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float rescale = 0.f;
    for (int j = 0; j < a.k; ++j)
    {
        const int dest = a.src_to_dest[j * a.num_rows + row];           // wave-uniform: scalar loads
        const float sc = a.scale_mode == 0 ? 1.f : a.scales[(size_t) row * a.k + j];
        rescale += sc;
        if (dest >= num_valid)
            continue;
        uint32_t w[4];
        if (a.fc2_splits > 1)
        {
            // the w2 GEMM's split-K slabs, summed in split order and rounded to T exactly as its reduce kernel would
            // have (woq_splitk_reduce_kernel): one launch and one pass over fc2 less
            float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const float* sl = a.fc2_slabs + (size_t) dest * a.hidden + 8 * c8;
            for (int s = 0; s < a.fc2_splits; ++s, sl += (size_t) a.n * a.hidden)
            {
                const float4 lo = *reinterpret_cast<const float4*>(sl), hi = *reinterpret_cast<const float4*>(sl + 4);
                f[0] += lo.x, f[1] += lo.y, f[2] += lo.z, f[3] += lo.w;
                f[4] += hi.x, f[5] += hi.y, f[6] += hi.z, f[7] += hi.w;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
                w[q] = pack2<T>(f[2 * q], f[2 * q + 1]);
        }
        else
        {
            const uint4 v = *reinterpret_cast<const uint4*>(a.fc2 + (size_t) dest * a.hidden + 8 * c8);
            w[0] = v.x, w[1] = v.y, w[2] = v.z, w[3] = v.w;
        }
        if (a.add_bias)
        {
            const int e = a.experts[(size_t) row * a.k + j];
            const uint4 b = *reinterpret_cast<const uint4*>(a.bias2 + (size_t) e * a.hidden + 8 * c8);
            const uint32_t bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) // the add in T
                w[q] = pack2<T>(T16<T>::lo(w[q]) + T16<T>::lo(bw[q]), T16<T>::hi(w[q]) + T16<T>::hi(bw[q]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
            const uint32_t acc = pack2<T>(o[2 * q] + sc * T16<T>::lo(w[q]), o[2 * q + 1] + sc * T16<T>::hi(w[q]));
            o[2 * q] = T16<T>::lo(acc);
            o[2 * q + 1] = T16<T>::hi(acc);
        }
    }
    if (a.scale_mode == 2)
    {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            o[q] /= rescale;
    }
    *reinterpret_cast<uint4*>(a.output + (size_t) row * a.hidden + 8 * c8)
        = make_uint4(pack2<T>(o[0], o[1]), pack2<T>(o[2], o[3]), pack2<T>(o[4], o[5]), pack2<T>(o[6], o[7]));
}

size_t align256(size_t x)
{
    return (x + 255) / 256 * 256;
}

struct MoeLayout
{
    size_t scales, experts, src_to_dest, dest_to_src, blk_counts, total_rows, fc1, glu, fc2, slabs, slab_bytes, total;
    int n, nblocks, buckets, experts_per_node;
    bool gated;
};

MoeLayout moe_layout(const mi355q_moe_params& p)
{
    MQ_CHECK(p.num_rows >= 0 && p.hidden_size > 0 && p.inter_size > 0 && p.k > 0, "bad sizes");
    MQ_CHECK(p.num_experts > 0 && p.num_experts <= MOE_MAX_EXPERTS, "num_experts %d (1 .. %d)", p.num_experts, MOE_MAX_EXPERTS);
    MQ_CHECK(p.k <= p.num_experts, "top_k %d > num_experts %d", p.k, p.num_experts);
    const int ep = p.ep_size > 0 ? p.ep_size : 1;
    MQ_CHECK(p.num_experts % ep == 0 && p.ep_rank >= 0 && p.ep_rank < ep, "num_experts %d over ep_size %d (rank %d)", p.num_experts,
        ep, p.ep_rank);
    MoeLayout l{};
    l.gated = p.activation_type == 3 || p.activation_type == 4;
    l.experts_per_node = p.num_experts / ep;
    l.buckets = l.experts_per_node + 1;
    const size_t n = (size_t) p.num_rows * p.k;
    MQ_CHECK(n < (1u << 30), "num_rows * k = %zu", n);
    l.n = (int) n;
    l.nblocks = std::max(1, (int) ((n + MOE_CHUNK - 1) / MOE_CHUNK));
    size_t off = 0;
    auto take = [&](size_t bytes)
    {
        const size_t at = off;
        off += align256(bytes);
        return at;
    };
    l.scales = take(n * 4);
    l.experts = take(n * 4);
    l.src_to_dest = take(n * 4);
    l.dest_to_src = take(n * 4);
    l.blk_counts = take((size_t) l.buckets * l.nblocks * 4);
    l.total_rows = take((size_t) l.experts_per_node * 8);
    l.fc1 = take(n * p.inter_size * 2);
    l.glu = take(l.gated ? n * p.inter_size * 4 : 0);
    l.fc2 = take(n * p.hidden_size * 2);
    // fp32 slabs for a split-K w2 GEMM (few n-tiles, long k) while its tiles leave CUs idle: up to 8 splits for
    // decode-sized calls, 4 / 2 up to 1024 / 2048 expanded rows (the same bytes at every size)
    const size_t splits = n <= 512 ? 8 : (n <= 1024 ? 4 : (n <= 2048 ? 2 : 0));
    l.slab_bytes = splits * n * p.hidden_size * sizeof(float);
    l.slabs = take(l.slab_bytes);
    l.total = off;
    return l;
}

} // namespace

size_t moe_workspace_size(const mi355q_moe_params& p)
{
    return moe_layout(p).total;
}

void run_moe(const mi355q_moe_params& p, hipStream_t stream)
{
    const MoeLayout l = moe_layout(p);
    MQ_CHECK(p.act_type == MI355Q_HALF || p.act_type == MI355Q_BF16, "act_type %d", p.act_type);
    MQ_CHECK(p.weight_type == MI355Q_WEIGHT_INT8 || p.weight_type == MI355Q_WEIGHT_INT4,
        "MoE: weight-only quantised experts only (int8 / int4), weight_type %d", p.weight_type);
    MQ_CHECK(p.activation_type >= 0 && p.activation_type <= 5, "activation_type %d", p.activation_type);
    MQ_CHECK(p.normalization_mode == 0 || p.normalization_mode == 1, "normalization_mode %d", p.normalization_mode);
    MQ_CHECK(p.hidden_size % 128 == 0 && p.inter_size % 128 == 0, "hidden_size %d / inter_size %d must be multiples of 128",
        p.hidden_size, p.inter_size);
    if (p.num_rows == 0)
        return;
    MQ_CHECK(p.input && p.gating_output && p.output, "null tensor");
    MQ_CHECK(p.fc1_expert_weights && p.fc2_expert_weights && p.fc1_scales && p.fc2_scales,
        "Scales expected but scale for a matmul is a null pointer"); // moe_kernels.cu:957-961
    MQ_CHECK(p.workspace && p.workspace_bytes >= l.total, "workspace too small: %zu < %zu", p.workspace_bytes, l.total);
    char* ws = static_cast<char*>(p.workspace);
    MoeArgs a{};
    a.input = static_cast<const uint16_t*>(p.input);
    a.gating = p.gating_output;
    a.finished = p.finished;
    a.output = static_cast<uint16_t*>(p.output);
    a.bias2 = static_cast<const uint16_t*>(p.fc2_expert_biases);
    a.scales = reinterpret_cast<float*>(ws + l.scales);
    a.experts = reinterpret_cast<int32_t*>(ws + l.experts);
    a.src_to_dest = reinterpret_cast<int32_t*>(ws + l.src_to_dest);
    a.dest_to_src = reinterpret_cast<int32_t*>(ws + l.dest_to_src);
    a.blk_counts = reinterpret_cast<int32_t*>(ws + l.blk_counts);
    a.total_rows = reinterpret_cast<int64_t*>(ws + l.total_rows);
    a.fc1 = reinterpret_cast<uint16_t*>(ws + l.fc1);
    a.glu = reinterpret_cast<uint16_t*>(ws + l.glu);
    a.fc2 = reinterpret_cast<uint16_t*>(ws + l.fc2);
    a.num_rows = p.num_rows;
    a.hidden = p.hidden_size;
    a.inter = p.inter_size;
    a.num_experts = p.num_experts;
    a.k = p.k;
    a.n = l.n;
    a.experts_per_node = l.experts_per_node;
    a.start_expert = l.experts_per_node * p.ep_rank;
    a.buckets = l.buckets;
    a.nblocks = l.nblocks;
    a.act_fn = p.activation_type;
    // finalizeMoeRoutingKernelLauncherSelectBias (:729-739): renormalise only for k > 1; bias only on tp rank 0
    a.scale_mode = p.normalization_mode == 1 ? (p.k == 1 ? 0 : 2) : 1;
    a.add_bias = p.fc2_expert_biases != nullptr && p.tp_rank == 0;
    const bool bf = p.act_type == MI355Q_BF16;

    const dim3 rgrid((p.num_rows + 3) / 4);
    if (p.num_experts <= 64)
        hipLaunchKernelGGL(moe_route_kernel<1>, rgrid, dim3(256), 0, stream, a);
    else if (p.num_experts <= 128)
        hipLaunchKernelGGL(moe_route_kernel<2>, rgrid, dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(moe_route_kernel<4>, rgrid, dim3(256), 0, stream, a);
    if (l.nblocks == 1)
        hipLaunchKernelGGL(moe_sort_small_kernel, dim3(1), dim3(l.n <= 512 ? 64 : 256), 0, stream, a);
    else
    {
        hipLaunchKernelGGL(moe_sort_count_kernel, dim3(l.nblocks), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(moe_sort_scan_kernel, dim3(1), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(moe_sort_scatter_kernel, dim3(l.nblocks), dim3(256), 0, stream, a);
    }
    MQ_HIP(hipGetLastError());

    // (no expandInputRows pass: the w1 GEMM gathers its activation rows through dest_to_src while it stages them)
    mi355q_moe_gemm_params g{};
    g.A = a.input;
    g.B = p.fc1_expert_weights;
    g.weight_scales = p.fc1_scales;
    g.biases = p.fc1_expert_biases;
    // gated activations: fused into the w1 GEMM's epilogue when the tiles divide the halves (inter % 256: every tile
    // form), else the [n, 2 inter] buffer and the gated-activation pass
    const bool fuse_glu = l.gated && p.inter_size % 256 == 0;
    g.C = (l.gated && !fuse_glu) ? a.glu : a.fc1;
    g.total_rows_before_expert = a.total_rows;
    g.total_rows = l.n;
    g.gemm_n = l.gated ? 2 * (int64_t) p.inter_size : p.inter_size;
    g.gemm_k = p.hidden_size;
    g.num_experts = l.experts_per_node;
    g.activation_type = l.gated ? 5 : p.activation_type;
    g.weight_type = p.weight_type;
    g.act_type = p.act_type;
    moe_weight_only_gemm(g, stream, true, a.dest_to_src, p.num_rows, fuse_glu ? (p.activation_type == 3 ? 2 : 0) : -1);
    if (l.gated && !fuse_glu)
    {
        if (bf)
            hipLaunchKernelGGL(moe_gated_act_kernel<bf16_t>, dim3(l.n, (p.inter_size / 8 + 255) / 256), dim3(256), 0, stream, a);
        else
            hipLaunchKernelGGL(moe_gated_act_kernel<f16_t>, dim3(l.n, (p.inter_size / 8 + 255) / 256), dim3(256), 0, stream, a);
    }
    g.A = a.fc1;
    g.B = p.fc2_expert_weights;
    g.weight_scales = p.fc2_scales;
    g.biases = nullptr; // added per (token, expert) by the finalize step
    g.C = a.fc2;
    g.gemm_n = p.hidden_size;
    g.gemm_k = p.inter_size;
    g.activation_type = 5;
    g.splitk_workspace = l.slab_bytes ? ws + l.slabs : nullptr;
    g.splitk_workspace_bytes = l.slab_bytes;
    a.fc2_splits = moe_weight_only_gemm(g, stream, false); // split-K: the slabs go to the finalize step, no reduce launch
    a.fc2_slabs = reinterpret_cast<const float*>(g.splitk_workspace);
    if (bf)
        hipLaunchKernelGGL(moe_finalize_kernel<bf16_t>, dim3(p.num_rows, (p.hidden_size / 8 + 255) / 256), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(moe_finalize_kernel<f16_t>, dim3(p.num_rows, (p.hidden_size / 8 + 255) / 256), dim3(256), 0, stream, a);
    MQ_HIP(hipGetLastError());
}

} // namespace mi355q
