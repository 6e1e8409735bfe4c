"""IPC buffers and the custom all-reduce host side: the mirror of tensorrt_llm/_ipc_utils.py:66-133 (IpcMemory),
tensorrt_llm/plugin/plugin.py:355-403 (CustomAllReduceHelper) and tensorrt_llm/functional.py `allreduce`.

One process per GPU.  The handles travel through torch.distributed (the reference uses MPI allgather); any backend
works for that exchange (gloo on the CPU is enough), the data path is the kernels' own peer loads over xGMI."""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib as L
from .plugin import Plugin, torch_dtype_to_trt, current_stream_ptr


class IpcMemory:
    """A zeroed device buffer on every rank of `group`, mapped into every other rank (_ipc_utils.py:66-133)."""

    @staticmethod
    def barrier_bytes():
        """One barrier-flag array per rank: flags per (phase, source rank, workgroup) (replaces IPC_BARRIERS_SIZE_PER_GPU)."""
        return L.lib().mi355q_allreduce_barrier_bytes()

    def __init__(self, size, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        p = C.c_void_p()
        L.check(L.lib().mi355q_ipc_alloc(size, C.byref(p)))
        self.local_ptr = p.value
        handle = (C.c_uint8 * L.IPC_HANDLE_BYTES)()
        L.check(L.lib().mi355q_ipc_get_handle(self.local_ptr, handle))
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle), group=group)
        self.peer_ptrs = []
        for r, h in enumerate(handles):
            if r == self.rank:
                self.peer_ptrs.append(self.local_ptr)
            else:
                q = C.c_void_p()
                buf = (C.c_uint8 * L.IPC_HANDLE_BYTES).from_buffer_copy(h)
                L.check(L.lib().mi355q_ipc_open_handle(buf, C.byref(q)))
                self.peer_ptrs.append(q.value)
        self._open = True

    def serialize(self):
        return list(self.peer_ptrs)

    def close(self):
        if not self._open:
            return
        self._open = False
        if torch.cuda.is_available():
            torch.cuda.synchronize()          # no kernel of this rank still polls or writes these buffers ...
        dist.barrier(group=self.group)        # ... and none of a peer's either
        for r, ptr in enumerate(self.peer_ptrs):
            if r != self.rank:
                L.lib().mi355q_ipc_close_handle(ptr)
        dist.barrier(group=self.group)  # nobody frees while a peer still has the buffer mapped
        L.lib().mi355q_ipc_free(self.local_ptr)


def init_comm(group=None, ranks=None):
    """initCommMap (plugins/common/plugin.cpp:50-87) without MPI: the group's first rank draws the RCCL unique id, it
    travels through torch.distributed, every rank calls mi355q_comm_init_rank with the GPU that is current.  After this
    the AllReduce plugin's RING strategy and the one-input AllGather plugin run inside the library.  `ranks`: the
    group's global ranks as the plugins' `group` field names them (default 0 .. world-1)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    ranks = list(range(world)) if ranks is None else sorted(ranks)
    arr = (C.c_int32 * world)(*ranks)
    if L.lib().mi355q_comm_exists(arr, world) == 1:
        return
    uid = (C.c_uint8 * L.COMM_UNIQUE_ID_BYTES)()
    box = [None]
    if rank == 0:
        # a failure here (librccl not loadable) must reach every rank: the others are about to wait for the id
        rc = L.lib().mi355q_comm_get_unique_id(uid)
        box = [bytes(uid) if rc == 0 else "error: " + L.lib().mi355q_last_error().decode()]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if isinstance(box[0], str):
        raise RuntimeError("init_comm: rank 0 could not draw the RCCL unique id (" + box[0] + ")")
    uid = (C.c_uint8 * L.COMM_UNIQUE_ID_BYTES).from_buffer_copy(box[0])
    L.check(L.lib().mi355q_comm_init_rank(arr, world, uid, rank))


def destroy_comm(group=None, ranks=None):
    world = dist.get_world_size(group)
    ranks = list(range(world)) if ranks is None else sorted(ranks)
    L.check(L.lib().mi355q_comm_destroy((C.c_int32 * world)(*ranks), world))


class CustomAllReduceHelper:
    """allocate_workspace (plugin.py:386-403): ping / pong comm buffers + barrier-in / barrier-out flag arrays; the
    workspace tensor is the HOST int64 [4 * tp_size] pointer table the AllReduce plugin takes as its second input.
    The kernels number the collectives themselves (a device-resident epoch in the barrier arrays), so the `counter`
    plugin field below only mirrors the reference's gen_id() and may repeat (engine / hipGraph replay)."""

    POINTERS_PER_RANK = 4

    @staticmethod
    def max_workspace_size_auto(tp_size):
        return L.lib().mi355q_allreduce_max_workspace_size(tp_size)

    def __init__(self, size=None, group=None, timeout_ms=0, max_workgroups=0):
        """`max_workgroups`: 0 = one per CU.  Ranks that share ONE GPU (rehearsals, the one-GPU tests) must bound it so
        that all ranks' workgroups are resident at once -- a workgroup waiting for its peers keeps its slot."""
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        size = size or self.max_workspace_size_auto(self.world)
        self.size = size
        self.timeout_ms = timeout_ms
        self.buffers = [IpcMemory(size, group), IpcMemory(size, group),
                        IpcMemory(IpcMemory.barrier_bytes(), group), IpcMemory(IpcMemory.barrier_bytes(), group)]
        flag = C.c_void_p()
        L.check(L.lib().mi355q_allreduce_alloc_timeout_flag(C.byref(flag)))
        self._timeout_flag = flag.value                      # host-visible word the kernels set when a peer never arrives
        ptrs = []
        for b in self.buffers:
            ptrs += b.serialize()
        # the reference's table (4 pointers per rank) + the comm-buffer byte size + the timeout word
        self.max_workgroups = max_workgroups
        self.workspace = torch.tensor(ptrs + [size, self._timeout_flag, max_workgroups], dtype=torch.int64, device="cpu")
        self.counter = 0
        self._ranks = (C.c_int32 * self.world)(*range(self.world))

    def has_rccl(self):
        """True once init_comm() has given this group a communicator inside the library."""
        return L.lib().mi355q_comm_exists(self._ranks, self.world) == 1

    def timed_out(self):
        return C.cast(self._timeout_flag, C.POINTER(C.c_uint32))[0] != 0

    def timeout_info(self):
        """Which wait ran out of time: (phase, peer rank, workgroup & 255, flag waited for, value found)."""
        w = C.cast(self._timeout_flag, C.POINTER(C.c_uint32))
        return (w[1] >> 16, (w[1] >> 8) & 255, w[1] & 255, w[2], w[3])

    def close(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()   # a collective still in flight may write the host-mapped timeout word
        dist.barrier(group=self.group)
        timed_out, info = self.timed_out(), self.timeout_info()
        for b in self.buffers:
            b.close()
        L.lib().mi355q_allreduce_free_timeout_flag(self._timeout_flag)
        if timed_out:
            raise RuntimeError("custom all-reduce: a wait for a peer rank timed out during this session "
                               "(phase %d, peer %d, workgroup %d, flag %d, found %d)" % info)

    def _params(self, tensor, out):
        """AllReduceParams for the direct launcher (mi355q_custom_all_reduce), same state as the plugin path."""
        p = L.AllReduceParams()
        table = (C.c_void_p * (4 * self.world))(*[int(v) for v in self.workspace[:4 * self.world]])
        L.check(L.lib().mi355q_allreduce_params_deserialize(C.byref(p), table, self.world, self.rank, self.counter))
        p.local_input_buffer_ptr = tensor.data_ptr()
        p.comm_buffer_bytes = self.size
        p.timeout_flag = self._timeout_flag
        p.timeout_ms = self.timeout_ms
        p.max_workgroups = self.max_workgroups
        return p

    def allreduce(self, tensor, strategy=L.AR_AUTO, out=None, direct=False):
        """functional.py `allreduce(tensor, group, workspace, strategy)` through the AllReduce plugin: one-shot /
        two-shot push kernels, RING (or AUTO / an explicit strategy beyond the comm buffers) = ncclAllReduce inside the
        plugin once init_comm() has run; `direct`: the launcher below the plugin, which also takes `timeout_ms`.
        Without a communicator in the library (ranks sharing one GPU in the tests, a gloo rehearsal on the CPU -- RCCL
        takes one rank per GPU) the RING leg goes through torch.distributed instead."""
        nbytes = tensor.numel() * tensor.element_size()
        strat = strategy
        if strat == L.AR_AUTO:
            strat = L.lib().mi355q_allreduce_select_implementation(nbytes, self.world, self.size)
        elif strat == L.AR_ONESHOT and self.world * ((nbytes + 255) // 256 * 256) > self.size:
            strat = L.AR_RING          # an explicit strategy whose message does not fit the comm buffers
        elif strat == L.AR_TWOSHOT and 2 * nbytes > self.size:
            strat = L.AR_RING
        tensor = tensor.contiguous()
        if strat == L.AR_RING:
            if not (tensor.is_cuda and self.has_rccl()):
                out = tensor.clone() if out is None else out.copy_(tensor)
                dist.all_reduce(out, group=self.group)
                return out
            out = torch.empty_like(tensor) if out is None else out
            plg = Plugin("AllReduce", [("group", list(range(self.world)), L.FIELD_INT32),
                                       ("type_id", torch_dtype_to_trt(tensor.dtype), L.FIELD_INT32),
                                       ("strategy", L.AR_RING, L.FIELD_INT8), ("counter", 0, L.FIELD_INT32)])
            plg.enqueue([tensor], [out])
            return out
        self.counter += 1
        out = torch.empty_like(tensor) if out is None else out
        if direct:
            if self.timed_out():
                raise RuntimeError("custom all-reduce: an earlier all-reduce timed out waiting for a peer rank")
            p = self._params(tensor, out)
            L.check(L.lib().mi355q_custom_all_reduce(C.byref(p), out.data_ptr(), tensor.numel(),
                                                     torch_dtype_to_trt(tensor.dtype), strat, current_stream_ptr()))
            return out
        plg = Plugin("AllReduce", [("group", list(range(self.world)), L.FIELD_INT32),
                                   ("type_id", torch_dtype_to_trt(tensor.dtype), L.FIELD_INT32),
                                   ("strategy", strat, L.FIELD_INT8), ("counter", self.counter, L.FIELD_INT32),
                                   ("rank", self.rank, L.FIELD_INT32)])
        plg.enqueue([tensor, self.workspace], [out])
        return out

    def allgather(self, tensor, out=None, direct=False):
        """functional.py `allgather(tensor, group)` (AllGather plugin, allgatherPlugin.cpp:90-110): the ranks' tensors
        back to back along dim 0.  Messages that fit the comm buffer take the push kernel over the peer mappings;
        larger ones (or sizes that are not a multiple of 16 bytes) go to RCCL."""
        tensor = tensor.contiguous()
        nbytes = tensor.numel() * tensor.element_size()
        shape = (tensor.shape[0] * self.world, ) + tuple(tensor.shape[1:])
        out = torch.empty(shape, dtype=tensor.dtype, device=tensor.device) if out is None else out
        slot = (nbytes + 255) // 256 * 256
        if nbytes % 16 or slot * self.world > self.size:
            if tensor.is_cuda and self.has_rccl():   # the reference's form: one input, ncclAllGather inside the plugin
                plg = Plugin("AllGather", [("group", list(range(self.world)), L.FIELD_INT32),
                                           ("type_id", torch_dtype_to_trt(tensor.dtype), L.FIELD_INT32)])
                plg.enqueue([tensor], [out])
            else:
                dist.all_gather(list(out.chunk(self.world, dim=0)), tensor, group=self.group)
            return out
        self.counter += 1
        if direct:
            if self.timed_out():
                raise RuntimeError("custom all-gather: an earlier collective timed out waiting for a peer rank")
            p = self._params(tensor, out)
            L.check(L.lib().mi355q_custom_all_gather(C.byref(p), tensor.data_ptr(), out.data_ptr(), nbytes,
                                                     current_stream_ptr()))
            return out
        plg = Plugin("AllGather", [("group", list(range(self.world)), L.FIELD_INT32),
                                   ("type_id", torch_dtype_to_trt(tensor.dtype), L.FIELD_INT32),
                                   ("strategy", L.AR_ONESHOT, L.FIELD_INT8), ("rank", self.rank, L.FIELD_INT32)])
        plg.enqueue([tensor, self.workspace], [out])
        return out
