// arrival_ring.hip -- the arrival words of in-launch reductions (split-KV decode attention, split-K decode GEMV).
//
// A kernel whose workgroups hand partial results to "the workgroup that arrives last" counts arrivals on a word in
// device memory.  That word must be zero when the launch starts, and it cannot live in the caller's workspace: TensorRT
// hands every layer the same scratch area, so whatever the previous layer left there is what the counter would start
// from (the reference clears its block_counter with a memset node in front of every launch,
// gptAttentionCommon.cpp enqueueGeneration; a memset is a ~1.6 us launch boundary on MI355X).  The words therefore live
// in a zero-initialised array of this code object, one per 128-byte line (words of different groups that share a line
// serialise in the memory-side atomic unit: measured +9 us for 512 workgroups on 64 words in two lines).  The counting
// instruction is global_atomic_inc with wrap value (arrivals - 1): the last arrival puts the word back to zero by
// itself, so a launch leaves its words clean for the next one.
//
// Who shares words (round 3; the reference keeps its counters in the execution context's own workspace and so allows
// two contexts on two streams).  Two launches may count on the same words only if they cannot overlap on the GPU:
//   * launches issued to ONE stream run in order, so a stream owns one region and every launch on it starts at the
//     region's first word (no ring: a finished launch has zeroed its words);
//   * a launch recorded under stream capture belongs to a graph that may later be replayed on any stream, next to
//     eager launches or to other graphs: it takes its words from a region keyed by the capture sequence
//     (hipStreamGetCaptureInfo's id) and the capturing stream, drawn from a pool that is NEVER recycled -- the nodes of
//     one capture on one stream are a chain, two captures never share, fork / join branches of a capture sit on
//     different streams and so in different regions.
// Regions grow by powers of two (a key holds at most twice its largest launch).  Eager regions come from the first quarter
// of the array; when that is used up (many distinct stream handles) the device is synchronised once and the quarter is
// handed out again.  The capture part only fills: when it is exhausted the call fails loudly
// (a process that captures without end must not).  What stays the caller's duty: the same graph, or two graphs of the
// same capture, must not run concurrently with themselves (HIP serialises launches of one hipGraphExec_t anyway), and
// a launch killed half-way (GPU fault) leaves words behind -- as it leaves everything else of that context.
#include "kernels.h"
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

namespace mi355q
{
namespace
{
constexpr size_t ARRIVAL_LINES = size_t(1) << 17; // x 128 B = 16 MiB per GPU
constexpr size_t EAGER_LINES = ARRIVAL_LINES / 4; // [0, EAGER_LINES): streams; [EAGER_LINES, ARRIVAL_LINES): captures
constexpr size_t MAX_LAUNCH_LINES = size_t(1) << 15;
static __device__ __align__(256) uint32_t arrival_ring[ARRIVAL_LINES * 32];
constexpr int MAX_DEVICES = 64;

struct Region
{
    size_t start, lines;
};
using Key = std::tuple<uintptr_t, unsigned long long>; // stream handle, capture id (0 = not capturing)

struct DeviceState
{
    uint32_t* base = nullptr; // this device's copy of arrival_ring
    size_t eager_next = 0, capture_next = EAGER_LINES;
    std::map<Key, Region> regions;
};
DeviceState g_dev[MAX_DEVICES];
std::mutex g_mutex;

size_t pow2_at_least(size_t v, size_t lo)
{
    size_t p = lo;
    while (p < v)
        p <<= 1;
    return p;
}

DeviceState& state_of_current_device() // g_mutex held
{
    int dev = -1;
    MQ_HIP(hipGetDevice(&dev));
    MQ_CHECK(dev >= 0 && dev < MAX_DEVICES, "device ordinal %d outside 0..%d", dev, MAX_DEVICES - 1);
    DeviceState& s = g_dev[dev];
    if (!s.base)
    {
        void* sym = nullptr; // the array's NAME is a host-side stand-in: resolve this device's copy
        MQ_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(arrival_ring)));
        MQ_CHECK(sym != nullptr, "hipGetSymbolAddress returned a null address");
        s.base = static_cast<uint32_t*>(sym);
    }
    return s;
}
} // namespace

size_t arrival_lines_max()
{
    return MAX_LAUNCH_LINES;
}

void arrival_words_prepare()
{
    std::lock_guard<std::mutex> lock(g_mutex);
    state_of_current_device();
}

uint32_t* arrival_lines(size_t count, hipStream_t stream)
{
    MQ_CHECK(count >= 1 && count <= MAX_LAUNCH_LINES, "%zu arrival words in one launch (at most %zu)", count, MAX_LAUNCH_LINES);
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long capture = 0;
    if (hipStreamGetCaptureInfo(stream, &status, &capture) != hipSuccess)
    {
        // the legacy default stream while another stream captures in global mode: the launch itself will be refused
        (void) hipGetLastError();
        status = hipStreamCaptureStatusNone;
    }
    const bool capturing = status == hipStreamCaptureStatusActive;
    uintptr_t handle = reinterpret_cast<uintptr_t>(stream);
    if (stream == hipStreamPerThread) // one handle value, one real stream per host thread
        handle = (std::hash<std::thread::id>{}(std::this_thread::get_id()) << 3) | 5; // no stream handle is odd
    const Key key{handle, capturing ? capture + 1 : 0ull};
    std::lock_guard<std::mutex> lock(g_mutex);
    DeviceState& s = state_of_current_device();
    auto it = s.regions.find(key);
    if (it != s.regions.end() && it->second.lines >= count)
        return s.base + it->second.start * ARRIVAL_WORD_STRIDE;
    // a first or a larger launch of this key: a new region (the key's earlier launches keep the old one; they are
    // ahead of this launch in the same stream / the same captured chain)
    const size_t lines = pow2_at_least(count, capturing ? 64 : 256);
    size_t start;
    if (capturing)
    {
        MQ_CHECK(s.capture_next + lines <= ARRIVAL_LINES,
            "arrival words for captured launches are used up (%zu of %zu lines; every stream capture that records a "
            "split-KV attention or split-K GEMV launch keeps its words for the life of the process)",
            s.capture_next - EAGER_LINES, ARRIVAL_LINES - EAGER_LINES);
        start = s.capture_next;
        s.capture_next += lines;
    }
    else
    {
        if (s.eager_next + lines > EAGER_LINES)
        {
            // every eager region is idle once the device has drained: start the eager part again
            MQ_HIP(hipDeviceSynchronize());
            for (auto r = s.regions.begin(); r != s.regions.end();)
                r = std::get<1>(r->first) == 0 ? s.regions.erase(r) : std::next(r);
            s.eager_next = 0;
        }
        start = s.eager_next;
        s.eager_next += lines;
    }
    s.regions[key] = Region{start, lines};
    return s.base + start * ARRIVAL_WORD_STRIDE;
}

} // namespace mi355q
