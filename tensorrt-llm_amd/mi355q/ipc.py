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

    IPC_BARRIERS_SIZE_PER_GPU = 25 * 4  # (MAX_ALL_REDUCE_BLOCKS + 1) * sizeof(uint32)

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
        for r, ptr in enumerate(self.peer_ptrs):
            if r != self.rank:
                L.lib().mi355q_ipc_close_handle(ptr)
        dist.barrier(group=self.group)  # nobody frees while a peer still has the buffer mapped
        L.lib().mi355q_ipc_free(self.local_ptr)


class CustomAllReduceHelper:
    """allocate_workspace (plugin.py:386-403): ping / pong comm buffers + barrier-in / barrier-out flag arrays; the
    workspace tensor is the HOST int64 [4 * tp_size] pointer table the AllReduce plugin takes as its second input."""

    POINTERS_PER_RANK = 4

    @staticmethod
    def max_workspace_size_auto(tp_size):
        return L.lib().mi355q_allreduce_max_workspace_size(tp_size)

    def __init__(self, size=None, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        size = size or self.max_workspace_size_auto(self.world)
        self.size = size
        self.buffers = [IpcMemory(size, group), IpcMemory(size, group),
                        IpcMemory(IpcMemory.IPC_BARRIERS_SIZE_PER_GPU * self.world, group),
                        IpcMemory(IpcMemory.IPC_BARRIERS_SIZE_PER_GPU * self.world, group)]
        ptrs = []
        for b in self.buffers:
            ptrs += b.serialize()
        self.workspace = torch.tensor(ptrs, dtype=torch.int64, device="cpu")
        self.counter = 0
        self._plugins = {}

    def close(self):
        for b in self.buffers:
            b.close()

    def allreduce(self, tensor, strategy=L.AR_AUTO):
        """functional.py `allreduce(tensor, group, workspace, strategy)`: RING -> RCCL (torch.distributed), else the
        AllReduce plugin (one-shot / two-shot).  Every call advances the barrier flag / ping-pong counter."""
        nbytes = tensor.numel() * tensor.element_size()
        strat = strategy
        if strat == L.AR_AUTO:
            strat = L.lib().mi355q_allreduce_select_implementation(nbytes, self.world, self.size)
        if strat == L.AR_RING or nbytes > self.size:
            out = tensor.clone()
            dist.all_reduce(out, group=self.group)
            return out
        self.counter += 1
        plg = Plugin("AllReduce", [("group", list(range(self.world)), L.FIELD_INT32),
                                   ("type_id", torch_dtype_to_trt(tensor.dtype), L.FIELD_INT32),
                                   ("strategy", strat, L.FIELD_INT8), ("counter", self.counter, L.FIELD_INT32),
                                   ("rank", self.rank, L.FIELD_INT32)])
        out = torch.empty_like(tensor)
        plg.enqueue([tensor.contiguous(), self.workspace], [out])
        return out
