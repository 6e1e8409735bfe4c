// allreduce.hip -- the custom all-reduce of the TP row-parallel GEMM outputs:
// kernels/customAllReduceKernels.cu:108-291 (multi_gpu_barrier, oneShotAllReduceKernel, twoShotAllReduceKernel),
// :295-352 kernelLaunchConfig, :398-454 AllReduceParams::deserialize / customAllReduce; the IPC buffer set-up of
// tensorrt_llm/_ipc_utils.py:66-133 (IpcMemory) is exposed as mi355q_ipc_*.
// Every rank owns a communication buffer and two barrier-flag arrays that its peers map through HIP IPC (xGMI peer
// access between GPUs of one node).  One-shot: after a flag barrier every rank reads all peers' buffers and sums them
// in T arithmetic, starting from its own rank (the reference's order, so results match it bit for bit).  Two-shot:
// reduce-scatter into the owner's buffer, per-block release/acquire flag barrier, all-gather.
// Differences from the reference, on purpose: flags are read / written with system-scope atomics (and the buffers are
// allocated fine-grained) because a spinning kernel on CDNA only observes a peer's store through them; spins are
// bounded (about a second) and report through `timeout_flag` instead of hanging the GPU.
#include "kernels.h"
#include <algorithm>
#include <cstring>

namespace mi355q
{
namespace
{

constexpr int AR_MAX_BLOCKS = 24;   // MAX_ALL_REDUCE_BLOCKS, customAllReduceKernels.h:32
constexpr int AR_BLOCK_SIZE = 1024; // DEFAULT_BLOCK_SIZE
constexpr int AR_WAVE = 64;
constexpr uint32_t AR_SPIN_LIMIT = 1u << 24; // x s_sleep: about a second

__device__ __forceinline__ void st_flag_release(uint32_t flag, uint32_t* addr)
{
    __hip_atomic_store(addr, flag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t ld_flag_acquire(uint32_t* addr)
{
    return __hip_atomic_load(addr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void st_flag_relaxed(uint32_t flag, uint32_t* addr)
{
    __hip_atomic_store(addr, flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ uint32_t ld_flag_relaxed(uint32_t* addr)
{
    return __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// spin until *addr == flag; false on timeout
template <bool ACQUIRE>
__device__ __forceinline__ bool wait_flag(uint32_t* addr, uint32_t flag)
{
    for (uint32_t i = 0; i < AR_SPIN_LIMIT; ++i)
    {
        if ((ACQUIRE ? ld_flag_acquire(addr) : ld_flag_relaxed(addr)) == flag)
            return true;
        __builtin_amdgcn_s_sleep(2);
    }
    return false;
}

// customAllReduceKernels.cu:108-129: on return at least block 0 of every rank has reached this point
__device__ __forceinline__ void multi_gpu_barrier(
    uint32_t* const* signals, uint32_t flag, int rank, int world, int tidx, int bidx, uint32_t* timeout_flag)
{
    if (tidx < world)
    {
        if (bidx == 0)
            st_flag_relaxed(flag, signals[tidx] + rank);
        if (!wait_flag<false>(signals[rank] + tidx, flag) && timeout_flag)
            *timeout_flag = 1;
    }
    __syncthreads();
}

typedef __attribute__((ext_vector_type(2))) _Float16 ar_f16x2;
typedef __attribute__((ext_vector_type(4))) uint32_t ar_u32x4;

// add128b (:98-106): element-wise add of two 16-byte packs in T arithmetic (each add rounds to T)
template <typename T>
__device__ __forceinline__ uint4 add_pack(const uint4& a, const uint4& b);
template <>
__device__ __forceinline__ uint4 add_pack<float>(const uint4& a, const uint4& b)
{
    return make_uint4(__float_as_uint(__uint_as_float(a.x) + __uint_as_float(b.x)),
        __float_as_uint(__uint_as_float(a.y) + __uint_as_float(b.y)),
        __float_as_uint(__uint_as_float(a.z) + __uint_as_float(b.z)),
        __float_as_uint(__uint_as_float(a.w) + __uint_as_float(b.w)));
}
__device__ __forceinline__ uint32_t add_h2(uint32_t a, uint32_t b)
{
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(ar_f16x2, a) + __builtin_bit_cast(ar_f16x2, b));
}
template <>
__device__ __forceinline__ uint4 add_pack<f16_t>(const uint4& a, const uint4& b)
{
    return make_uint4(add_h2(a.x, b.x), add_h2(a.y, b.y), add_h2(a.z, b.z), add_h2(a.w, b.w));
}
__device__ __forceinline__ uint32_t add_bf2(uint32_t a, uint32_t b)
{
    // gfx950 has no packed bf16 add: widen, add in fp32, round to nearest even
    return pack2<bf16_t>(T16<bf16_t>::lo(a) + T16<bf16_t>::lo(b), T16<bf16_t>::hi(a) + T16<bf16_t>::hi(b));
}
template <>
__device__ __forceinline__ uint4 add_pack<bf16_t>(const uint4& a, const uint4& b)
{
    return make_uint4(add_bf2(a.x, b.x), add_bf2(a.y, b.y), add_bf2(a.z, b.z), add_bf2(a.w, b.w));
}

template <typename T>
struct ArElem
{
    static constexpr int SIZE = sizeof(float) == sizeof(T) ? 4 : 2;
};

struct ArArgs
{
    size_t elts_total, elts_per_rank, elts_per_block, rank_offset;
    int ranks, rank;
    uint32_t barrier_flag;
    uint32_t* barrier_in[MI355Q_MAX_RANKS_PER_NODE];
    uint32_t* barrier_out[MI355Q_MAX_RANKS_PER_NODE];
    uint8_t* comm[MI355Q_MAX_RANKS_PER_NODE];
    uint8_t* out;
    uint32_t* timeout_flag;
};

__global__ void multi_gpu_barrier_kernel(ArArgs p)
{
    multi_gpu_barrier(p.barrier_out, p.barrier_flag, p.rank, p.ranks, threadIdx.x, blockIdx.x, p.timeout_flag);
}

// :137-189
template <typename T, int RANKS>
__global__ __launch_bounds__(AR_BLOCK_SIZE) void one_shot_kernel(ArArgs p)
{
    constexpr int ES = ArElem<T>::SIZE, NUM_ELTS = 16 / ES;
    const int bidx = blockIdx.x, tidx = threadIdx.x;
    multi_gpu_barrier(p.barrier_in, p.barrier_flag, p.rank, RANKS, tidx, bidx, p.timeout_flag);
    const uint8_t* src[RANKS];
#pragma unroll
    for (int ii = 0; ii < RANKS; ++ii)
        src[ii] = p.comm[(p.rank + ii) % RANKS];
    const size_t offset = (size_t) bidx * p.elts_per_block + (size_t) tidx * NUM_ELTS;
    const size_t max_offset = min((size_t) (bidx + 1) * p.elts_per_block, p.elts_per_rank);
    for (size_t it = offset; it < max_offset; it += (size_t) blockDim.x * NUM_ELTS)
    {
        uint4 vals[RANKS];
#pragma unroll
        for (int ii = 0; ii < RANKS; ++ii)
            vals[ii] = *reinterpret_cast<const uint4*>(src[ii] + it * ES);
        uint4 sums = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int ii = 0; ii < RANKS; ++ii)
            sums = add_pack<T>(sums, vals[ii]);
        *reinterpret_cast<uint4*>(p.out + it * ES) = sums;
    }
}

// :191-291
template <typename T, int RANKS>
__global__ __launch_bounds__(AR_BLOCK_SIZE) void two_shot_kernel(ArArgs p)
{
    constexpr int ES = ArElem<T>::SIZE, NUM_ELTS = 16 / ES;
    const int bidx = blockIdx.x, tidx = threadIdx.x;
    const size_t block_offset = (size_t) bidx * p.elts_per_block + (size_t) tidx * NUM_ELTS;
    const size_t block_start = p.rank_offset + block_offset;
    multi_gpu_barrier(p.barrier_in, p.barrier_flag, p.rank, RANKS, tidx, bidx, p.timeout_flag);
    uint8_t* src[RANKS];
    int dst_rank[RANKS];
#pragma unroll
    for (int ii = 0; ii < RANKS; ++ii)
    {
        dst_rank[ii] = (p.rank + ii) % RANKS;
        src[ii] = p.comm[dst_rank[ii]];
    }
    // reduce-scatter: this rank's slice, summed into its own buffer
    const size_t seg_end = min(p.rank_offset + (size_t) (bidx + 1) * p.elts_per_block, p.rank_offset + p.elts_per_rank);
    for (size_t off = block_start; off < seg_end; off += (size_t) blockDim.x * NUM_ELTS)
    {
        uint4 vals[RANKS];
#pragma unroll
        for (int ii = 0; ii < RANKS; ++ii)
            vals[ii] = *reinterpret_cast<const uint4*>(src[ii] + off * ES);
        uint4 sums = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int ii = 0; ii < RANKS; ++ii)
            sums = add_pack<T>(sums, vals[ii]);
        *reinterpret_cast<uint4*>(src[0] + off * ES) = sums;
    }
    __syncthreads();
    // barrier among the blocks of the same index (release / acquire): flags [RANKS + bidx * RANKS + rank]
    if (tidx < RANKS)
    {
        const uint32_t flag_block_offset = RANKS + bidx * RANKS;
        __threadfence_system();
        st_flag_release(p.barrier_flag, p.barrier_in[tidx] + flag_block_offset + p.rank);
        if (!wait_flag<true>(p.barrier_in[p.rank] + flag_block_offset + tidx, p.barrier_flag) && p.timeout_flag)
            *p.timeout_flag = 1;
    }
    __syncthreads();
    // all-gather: every rank's reduced slice
    const size_t max_block_offset = min((size_t) (bidx + 1) * p.elts_per_block, p.elts_per_rank);
    for (size_t off = block_offset; off < max_block_offset; off += (size_t) blockDim.x * NUM_ELTS)
    {
#pragma unroll
        for (int ii = 0; ii < RANKS; ++ii)
        {
            const size_t offset_rank = (size_t) dst_rank[ii] * p.elts_per_rank + off;
            if (offset_rank >= p.elts_total)
                continue;
            // the slice was written by a peer during this launch: read it past the caches
            const ar_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ar_u32x4*>(src[ii] + offset_rank * ES));
            *reinterpret_cast<ar_u32x4*>(p.out + offset_rank * ES) = v;
        }
    }
}

size_t div_up(size_t a, size_t b)
{
    return (a + b - 1) / b;
}

// kernelLaunchConfig, :295-352 (WARP_SIZE -> the 64-lane wave)
void launch_config(int strat, mi355q_allreduce_params& param, size_t elts_per_thread, int& blocks, int& threads)
{
    MQ_CHECK(param.elts_total % elts_per_thread == 0, "elts_total %zu must be a multiple of %zu", param.elts_total,
        elts_per_thread);
    blocks = 1;
    threads = AR_BLOCK_SIZE;
    if (strat == MI355Q_AR_ONESHOT)
    {
        const size_t total_threads = param.elts_total / elts_per_thread;
        if (total_threads <= AR_BLOCK_SIZE)
        {
            threads = (int) (AR_WAVE * div_up(total_threads, AR_WAVE));
            blocks = 1;
        }
        else
        {
            threads = AR_BLOCK_SIZE;
            blocks = (int) std::min<size_t>(AR_MAX_BLOCKS, div_up(total_threads, AR_BLOCK_SIZE));
        }
        param.elts_per_rank = param.elts_total;
        param.elts_per_block = elts_per_thread * div_up(param.elts_per_rank, elts_per_thread * blocks);
        param.rank_offset = 0;
    }
    else
    {
        MQ_CHECK(param.elts_total % param.ranks_per_node == 0, "elts_total %zu not divisible by %zu ranks",
            param.elts_total, param.ranks_per_node);
        const size_t elts_per_rank = param.elts_total / param.ranks_per_node;
        MQ_CHECK(elts_per_rank % elts_per_thread == 0, "elts_per_rank %zu must be a multiple of %zu", elts_per_rank,
            elts_per_thread);
        size_t total_threads = elts_per_rank / elts_per_thread;
        total_threads = AR_WAVE * div_up(total_threads, AR_WAVE);
        size_t b = 1;
        while (total_threads % b != 0 || total_threads / b > AR_BLOCK_SIZE)
            b += 1;
        threads = (int) (total_threads / b);
        if (b > AR_MAX_BLOCKS)
        {
            size_t iter_factor = 1;
            while (b / iter_factor > AR_MAX_BLOCKS || b % iter_factor)
                iter_factor += 1;
            b /= iter_factor;
        }
        blocks = (int) b;
        param.elts_per_rank = elts_per_rank;
        param.elts_per_block = param.elts_per_rank / blocks;
        param.elts_per_block = elts_per_thread * div_up(param.elts_per_block, elts_per_thread);
        param.rank_offset = param.rank * param.elts_per_rank;
    }
}

ArArgs to_args(const mi355q_allreduce_params& q)
{
    ArArgs a{};
    a.elts_total = q.elts_total;
    a.elts_per_rank = q.elts_per_rank;
    a.elts_per_block = q.elts_per_block;
    a.rank_offset = q.rank_offset;
    a.ranks = (int) q.ranks_per_node;
    a.rank = (int) q.local_rank;
    a.barrier_flag = q.barrier_flag;
    for (size_t i = 0; i < q.ranks_per_node; ++i)
    {
        a.barrier_in[i] = q.peer_barrier_ptrs_in[i];
        a.barrier_out[i] = q.peer_barrier_ptrs_out[i];
        a.comm[i] = static_cast<uint8_t*>(q.peer_comm_buffer_ptrs[i]);
    }
    a.out = static_cast<uint8_t*>(q.local_output_buffer_ptr);
    a.timeout_flag = q.timeout_flag;
    return a;
}

void check_params(const mi355q_allreduce_params& q)
{
    MQ_CHECK(q.ranks_per_node >= 2 && q.ranks_per_node <= MI355Q_MAX_RANKS_PER_NODE, "ranks_per_node %zu", q.ranks_per_node);
    MQ_CHECK(q.local_rank < q.ranks_per_node && q.rank < q.ranks_per_node, "rank %zu / local_rank %zu", q.rank, q.local_rank);
    for (size_t i = 0; i < q.ranks_per_node; ++i)
        MQ_CHECK(q.peer_barrier_ptrs_in[i] && q.peer_barrier_ptrs_out[i] && q.peer_comm_buffer_ptrs[i], "null peer pointer %zu", i);
}

template <typename T, int RANKS>
void dispatch_kernels(int strat, const ArArgs& a, int blocks, int threads, hipStream_t stream)
{
    if (strat == MI355Q_AR_ONESHOT)
        hipLaunchKernelGGL((one_shot_kernel<T, RANKS>), dim3(blocks), dim3(threads), 0, stream, a);
    else
        hipLaunchKernelGGL((two_shot_kernel<T, RANKS>), dim3(blocks), dim3(threads), 0, stream, a);
}

template <typename T>
void invoke(mi355q_allreduce_params& q, int strat, hipStream_t stream)
{
    const size_t elts_per_thread = 16 / ArElem<T>::SIZE;
    int blocks, threads;
    launch_config(strat, q, elts_per_thread, blocks, threads);
    const ArArgs a = to_args(q);
    switch (q.ranks_per_node)
    {
    case 2: dispatch_kernels<T, 2>(strat, a, blocks, threads, stream); break;
    case 4: dispatch_kernels<T, 4>(strat, a, blocks, threads, stream); break;
    case 6: dispatch_kernels<T, 6>(strat, a, blocks, threads, stream); break;
    case 8: dispatch_kernels<T, 8>(strat, a, blocks, threads, stream); break;
    default: MQ_CHECK(false, "customAllReduce: ranks_per_node %zu (2, 4, 6, 8)", q.ranks_per_node);
    }
    MQ_HIP(hipGetLastError());
}

} // namespace

// AllReduceParams::deserialize, :398-425: buffer = [ping ptrs | pong ptrs | barrier-in ptrs | barrier-out ptrs] x tpSize
void allreduce_params_deserialize(
    mi355q_allreduce_params& out, const void* const* buffer_ptrs, size_t tpSize, size_t tpRank, uint32_t flag_value)
{
    MQ_CHECK(buffer_ptrs, "null workspace");
    MQ_CHECK(tpSize >= 2 && tpSize <= MI355Q_MAX_RANKS_PER_NODE && tpRank < tpSize, "tpSize %zu tpRank %zu", tpSize, tpRank);
    out = mi355q_allreduce_params{};
    // even calls use the ping buffers, odd ones the pong buffers: no need to wait for the peers to be done with the
    // previous call before the input is copied in
    const size_t buffer_offset = (flag_value % 2 == 0) ? 0 : tpSize;
    for (size_t i = 0; i < tpSize; ++i)
    {
        out.peer_comm_buffer_ptrs[i] = const_cast<void*>(buffer_ptrs[buffer_offset + i]);
        out.peer_barrier_ptrs_in[i] = static_cast<uint32_t*>(const_cast<void*>(buffer_ptrs[2 * tpSize + i]));
        out.peer_barrier_ptrs_out[i] = static_cast<uint32_t*>(const_cast<void*>(buffer_ptrs[3 * tpSize + i]));
    }
    out.barrier_flag = flag_value;
    out.ranks_per_node = tpSize;
    out.rank = tpRank;
    out.local_rank = tpRank;
}

// customAllReduce, :427-454
void custom_all_reduce(mi355q_allreduce_params& params, void* data, size_t elts, int dtype, int strat, hipStream_t stream)
{
    MQ_CHECK(strat == MI355Q_AR_ONESHOT || strat == MI355Q_AR_TWOSHOT, "Algorithm not supported here. (strategy %d)", strat);
    MQ_CHECK(data, "null output");
    check_params(params);
    params.local_output_buffer_ptr = data;
    params.elts_total = elts;
    if (elts == 0)
        return;
    switch (dtype)
    {
    case MI355Q_FLOAT: invoke<float>(params, strat, stream); break;
    case MI355Q_HALF: invoke<f16_t>(params, strat, stream); break;
    case MI355Q_BF16: invoke<bf16_t>(params, strat, stream); break;
    default: MQ_CHECK(false, "Unsupported dataType for customAllReduce (%d)", dtype);
    }
}

// invokeMultiGpuBarrier, :391-394
void multi_gpu_barrier_launch(const mi355q_allreduce_params& params, hipStream_t stream)
{
    check_params(params);
    hipLaunchKernelGGL(multi_gpu_barrier_kernel, dim3(1), dim3(AR_WAVE), 0, stream, to_args(params));
    MQ_HIP(hipGetLastError());
}

// ---- IpcMemory (tensorrt_llm/_ipc_utils.py:66-133, cpp/tensorrt_llm/runtime/ipcUtils.cpp)
void ipc_alloc(size_t bytes, void** ptr)
{
    MQ_CHECK(ptr && bytes > 0, "ipc_alloc(%zu)", bytes);
    // fine-grained: a peer's flag store / slice store is observable by a kernel that is already running
    MQ_HIP(hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained));
    MQ_HIP(hipMemset(*ptr, 0, bytes));
    MQ_HIP(hipDeviceSynchronize());
}
void ipc_free(void* ptr)
{
    MQ_HIP(hipFree(ptr));
}
void ipc_get_handle(void* ptr, uint8_t* handle)
{
    static_assert(sizeof(hipIpcMemHandle_t) == MI355Q_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    hipIpcMemHandle_t h;
    MQ_HIP(hipIpcGetMemHandle(&h, ptr));
    memcpy(handle, &h, sizeof(h));
}
void ipc_open_handle(const uint8_t* handle, void** ptr)
{
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    MQ_HIP(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
}
void ipc_close_handle(void* ptr)
{
    MQ_HIP(hipIpcCloseMemHandle(ptr));
}

} // namespace mi355q
