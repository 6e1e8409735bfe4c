// rccl_comm.cpp -- the RING leg of the AllReduce / AllGather plugins: RCCL communicators inside the boundary.
//
// The reference keeps a process-wide map from a rank group to an NCCL communicator (getCommMap / initCommMap,
// plugins/common/plugin.cpp:44-87): rank 0 of the group calls ncclGetUniqueId, ships the id to the others over MPI and
// everyone calls ncclCommInitRank; AllreducePlugin::enqueue then issues ncclAllReduce(in, out, size, dtype, ncclSum,
// comm, stream) itself (plugins/ncclPlugin/allreducePlugin.cpp:181-185) and AllgatherPlugin::enqueue ncclAllGather
// (allgatherPlugin.cpp:90-110).  There is no MPI here (one process per GPU under torch.distributed or any launcher),
// so the id exchange is the host's: mi355q_comm_get_unique_id on one rank, any transport, mi355q_comm_init_rank on all.
//
// librccl is loaded on first use (dlopen), not linked: the library must load on a machine without a GPU or without
// RCCL (the CPU-side ABI tests, single-GPU serving), and a process that already carries a librccl (PyTorch-ROCm ships
// one) gets that same copy by its soname.
#include "common.h"
#include "kernels.h"
#include <cstring>
#include <dlfcn.h>
#include <map>
#include <mutex>
#include <rccl/rccl.h>
#include <set>

namespace mi355q
{
namespace
{

struct Rccl
{
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) getUniqueId = nullptr;
    decltype(&ncclCommInitRank) commInitRank = nullptr;
    decltype(&ncclCommDestroy) commDestroy = nullptr;
    decltype(&ncclAllReduce) allReduce = nullptr;
    decltype(&ncclAllGather) allGather = nullptr;
    decltype(&ncclGetErrorString) getErrorString = nullptr;
};

std::mutex g_mutex;
Rccl g_rccl;
std::map<std::set<int>, ncclComm_t> g_comm_map; // getCommMap, plugin.cpp:44-48

template <typename F>
void sym(void* h, F& f, const char* name)
{
    f = reinterpret_cast<F>(dlsym(h, name));
    MQ_CHECK(f != nullptr, "librccl: symbol %s not found (%s)", name, dlerror());
}

Rccl& rccl() // g_mutex held
{
    if (g_rccl.handle)
        return g_rccl;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) != nullptr)
            break;
    MQ_CHECK(h != nullptr, "the RING strategy needs RCCL and librccl.so.1 could not be loaded (%s)", dlerror());
    Rccl r;
    sym(h, r.getUniqueId, "ncclGetUniqueId");
    sym(h, r.commInitRank, "ncclCommInitRank");
    sym(h, r.commDestroy, "ncclCommDestroy");
    sym(h, r.allReduce, "ncclAllReduce");
    sym(h, r.allGather, "ncclAllGather");
    sym(h, r.getErrorString, "ncclGetErrorString");
    r.handle = h;
    g_rccl = r;
    return g_rccl;
}

#define MQ_NCCL(call)                                                                                                  \
    do                                                                                                                 \
    {                                                                                                                  \
        const ncclResult_t r_ = (call);                                                                                \
        MQ_CHECK(r_ == ncclSuccess, "RCCL: %s failed: %s", #call, g_rccl.getErrorString(r_));                          \
    } while (0) /* NCCLCHECK, plugins/common/plugin.h:163 */

ncclDataType_t nccl_dtype(int dtype) // getDtypeMap, plugin.cpp:37-42 (+ the byte types an all-gather may carry)
{
    switch (dtype)
    {
    case MI355Q_FLOAT: return ncclFloat32;
    case MI355Q_HALF: return ncclFloat16;
    case MI355Q_BF16: return ncclBfloat16;
    case MI355Q_INT8: return ncclInt8;
    case MI355Q_INT32: return ncclInt32;
    case MI355Q_INT64: return ncclInt64;
    default: MQ_CHECK(false, "RCCL: unsupported data type %d", dtype);
    }
    return ncclInt8;
}

std::set<int> group_set(const int32_t* group, int n)
{
    MQ_CHECK(group && n >= 1, "empty rank group");
    std::set<int> s(group, group + n);
    MQ_CHECK((int) s.size() == n, "rank group holds a rank twice");
    return s;
}

ncclComm_t comm_of(const std::set<int>& g) // g_mutex held
{
    auto it = g_comm_map.find(g);
    MQ_CHECK(it != g_comm_map.end() && it->second != nullptr,
        "no RCCL communicator for this rank group: call mi355q_comm_init_rank on every rank of the group first (the "
        "reference does it in AllreducePlugin::initialize over MPI, plugins/common/plugin.cpp:50-87)");
    return it->second;
}

} // namespace

void comm_get_unique_id(uint8_t* id)
{
    static_assert(sizeof(ncclUniqueId) == MI355Q_COMM_UNIQUE_ID_BYTES, "ncclUniqueId size");
    MQ_CHECK(id, "null argument");
    std::lock_guard<std::mutex> lock(g_mutex);
    ncclUniqueId u;
    MQ_NCCL(rccl().getUniqueId(&u));
    std::memcpy(id, &u, sizeof(u));
}

// initCommMap(group), plugin.cpp:50-87, with the id handed in instead of sent over MPI.  group_rank = the position of
// this process's rank in the sorted group (the reference counts the same way, :61-69).  A group that already has a
// communicator is left alone (:54).
void comm_init_rank(const int32_t* group, int n, const uint8_t* id, int group_rank)
{
    MQ_CHECK(id, "null unique id");
    const std::set<int> g = group_set(group, n);
    MQ_CHECK(group_rank >= 0 && group_rank < n, "group_rank %d outside a group of %d", group_rank, n);
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_comm_map.find(g);
    if (it != g_comm_map.end() && it->second != nullptr)
        return;
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof(u));
    ncclComm_t c = nullptr;
    MQ_NCCL(rccl().commInitRank(&c, n, u, group_rank));
    g_comm_map[g] = c;
}

bool comm_exists(const int32_t* group, int n)
{
    const std::set<int> g = group_set(group, n);
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_comm_map.find(g);
    return it != g_comm_map.end() && it->second != nullptr;
}

// AllreducePlugin::terminate, allreducePlugin.cpp:254-267
void comm_destroy(const int32_t* group, int n)
{
    const std::set<int> g = group_set(group, n);
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_comm_map.find(g);
    if (it == g_comm_map.end() || it->second == nullptr)
        return;
    MQ_NCCL(g_rccl.commDestroy(it->second));
    g_comm_map.erase(it);
}

// allreducePlugin.cpp:181-185
void comm_all_reduce(const int32_t* group, int n, const void* in, void* out, size_t count, int dtype, hipStream_t stream)
{
    const std::set<int> g = group_set(group, n);
    std::lock_guard<std::mutex> lock(g_mutex);
    ncclComm_t c = comm_of(g);
    MQ_NCCL(g_rccl.allReduce(in, out, count, nccl_dtype(dtype), ncclSum, c, stream));
}

// allgatherPlugin.cpp:90-110
void comm_all_gather(const int32_t* group, int n, const void* in, void* out, size_t count, int dtype, hipStream_t stream)
{
    const std::set<int> g = group_set(group, n);
    std::lock_guard<std::mutex> lock(g_mutex);
    ncclComm_t c = comm_of(g);
    MQ_NCCL(g_rccl.allGather(in, out, count, nccl_dtype(dtype), c, stream));
}

} // namespace mi355q
