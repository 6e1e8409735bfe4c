// arrival_ring.hip -- the arrival words of in-launch reductions (split-KV decode attention, split-K decode GEMV).
//
// A kernel whose workgroups hand partial results to "the workgroup that arrives last" counts arrivals on a word in
// device memory.  That word must be zero when the launch starts, and it cannot live in the caller's workspace: TensorRT
// hands every layer the same scratch area, so whatever the previous layer left there is what the counter would start
// from (the reference clears its block_counter with a memset node in front of every launch,
// gptAttentionCommon.cpp enqueueGeneration; a memset is a ~1.6 us launch boundary on MI355X).  The words therefore live
// in a zero-initialised array of this code object, one per 128-byte line (words of different groups that share a line
// serialise in the memory-side atomic unit: measured +9 us for 512 workgroups on 64 words in two lines), handed out as
// a ring.  The counting instruction is global_atomic_inc with wrap value (arrivals - 1): the last arrival puts the
// word back to zero by itself, so a slice is clean again when its launch has finished, and a hipGraph that replays a
// launch reuses its slice.  Two launches may share words only if they overlap in time on one GPU AND the ring has gone
// round in between (ARRIVAL_LINES lines handed out): not reachable by one serving stream.
#include "kernels.h"
#include <atomic>
#include <mutex>

namespace mi355q
{
namespace
{
constexpr size_t ARRIVAL_LINES = size_t(1) << 15; // x 128 B = 4 MiB per GPU
static __device__ __align__(256) uint32_t arrival_ring[ARRIVAL_LINES * 32];
constexpr int MAX_DEVICES = 64;
std::atomic<uint32_t*> g_ring_ptr[MAX_DEVICES];
std::mutex g_ring_mutex;
std::atomic<uint64_t> g_ring_pos{0};

uint32_t* ring_of_current_device()
{
    int dev = -1;
    MQ_HIP(hipGetDevice(&dev));
    MQ_CHECK(dev >= 0 && dev < MAX_DEVICES, "device ordinal %d outside 0..%d", dev, MAX_DEVICES - 1);
    uint32_t* p = g_ring_ptr[dev].load(std::memory_order_acquire);
    if (!p)
    {
        std::lock_guard<std::mutex> lock(g_ring_mutex);
        p = g_ring_ptr[dev].load(std::memory_order_relaxed);
        if (!p)
        {
            void* sym = nullptr; // the array's NAME is a host-side stand-in: resolve this device's copy
            MQ_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(arrival_ring)));
            MQ_CHECK(sym != nullptr, "hipGetSymbolAddress returned a null address");
            p = static_cast<uint32_t*>(sym);
            g_ring_ptr[dev].store(p, std::memory_order_release);
        }
    }
    return p;
}
} // namespace

size_t arrival_lines_max()
{
    return ARRIVAL_LINES;
}

uint32_t* arrival_lines(size_t count)
{
    MQ_CHECK(count >= 1 && count <= ARRIVAL_LINES, "%zu arrival words in one launch (at most %zu)", count, ARRIVAL_LINES);
    uint32_t* base = ring_of_current_device();
    uint64_t pos = g_ring_pos.load(std::memory_order_relaxed), start;
    do
        start = (pos % ARRIVAL_LINES) + count <= ARRIVAL_LINES ? pos : (pos / ARRIVAL_LINES + 1) * ARRIVAL_LINES;
    while (!g_ring_pos.compare_exchange_weak(pos, start + count, std::memory_order_relaxed));
    return base + (start % ARRIVAL_LINES) * ARRIVAL_WORD_STRIDE;
}

} // namespace mi355q
