"""GPU parity: the custom one-shot / two-shot all-reduce (row a16) through the AllReduce plugin and the launcher below
it, with the ranks as separate PROCESSES that map each other's buffers through HIP IPC.  The box has one GPU, so all
ranks sit on device 0: this exercises the kernels' indexing, per-workgroup flag protocol, ping-pong buffers, the
T-arithmetic summation order bit for bit against the oracle, and the bounded wait (a late rank -> NaN output + the
host-visible timeout word), but NOT the cross-GPU (xGMI) visibility of the flags -- DESIGN.md says so.  The AllGather plugin (allgatherPlugin.cpp) runs on the same buffers in the same workers."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _replay_cases(rank, world, dev, helper, oracle, L):
    """An engine replays the SAME build-time counters on every pass (allreducePlugin.cpp:194, functional.py gen_id()):
    plugins with counters 1..3, serialized and deserialized like an engine load, enqueued as one sequence twice with
    fresh data; then the same sequence captured in a hipGraph and replayed.  The kernels' device-resident epoch keeps
    the ranks synchronised whatever the counters say."""
    from mi355q.plugin import Plugin, torch_dtype_to_trt
    os.environ["RANK"] = str(rank)  # a deserialized plugin has no 'rank' field: the reference asks MPI, here $RANK
    n = 8 * 8192
    plugins = []
    for c, strat in ((1, L.AR_ONESHOT), (2, L.AR_TWOSHOT), (3, L.AR_ONESHOT)):
        made = Plugin("AllReduce", [("group", list(range(world)), L.FIELD_INT32),
                                    ("type_id", torch_dtype_to_trt(torch.bfloat16), L.FIELD_INT32),
                                    ("strategy", strat, L.FIELD_INT8), ("counter", c, L.FIELD_INT32)])
        plugins.append((Plugin.deserialize("AllReduce", made.serialize()), "oneshot" if strat == L.AR_ONESHOT else "twoshot"))
    gather = Plugin("AllGather", [("group", list(range(world)), L.FIELD_INT32), ("type_id", torch_dtype_to_trt(torch.bfloat16), L.FIELD_INT32),
                                  ("strategy", L.AR_ONESHOT, L.FIELD_INT8), ("rank", rank, L.FIELD_INT32)])

    def data(seed):
        g = torch.Generator().manual_seed(seed)
        return [[(torch.randn(n, generator=g) * 3).to(torch.bfloat16) for _ in range(world)] for _ in range(len(plugins))]

    def verify(tag, ins, outs, gout):
        for i, (_, sname) in enumerate(plugins):
            ref = oracle.custom_all_reduce(ins[i], rank, sname)
            if not torch.equal(outs[i].cpu().view(torch.uint8), ref.view(torch.uint8)):
                return False, f"rank {rank} {tag} call {i} ({sname}): max diff {(outs[i].cpu().float() - ref.float()).abs().max().item()}"
        if not torch.equal(gout.cpu().view(torch.uint8), torch.cat(ins[0]).view(torch.uint8)):
            return False, f"rank {rank} {tag}: all-gather differs"
        return True, ""

    x = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in plugins]
    y = [torch.empty(n, dtype=torch.bfloat16, device=dev) for _ in plugins]
    gy = torch.empty(n * world, dtype=torch.bfloat16, device=dev)

    def sequence():
        for i, (plg, _) in enumerate(plugins):
            plg.enqueue([x[i], helper.workspace], [y[i]])
        gather.enqueue([x[0], helper.workspace], [gy])

    for p in range(2):  # the same plugin objects, pass after pass
        ins = data(500 + p)
        for i in range(len(plugins)):
            x[i].copy_(ins[i][rank])
        sequence()
        torch.cuda.synchronize()
        ok, msg = verify(f"pass {p}", ins, y, gy)
        if not ok:
            return ok, msg
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    ins = data(600)
    for i in range(len(plugins)):
        x[i].copy_(ins[i][rank])
    torch.cuda.synchronize()
    dist.barrier()
    with torch.cuda.graph(graph, stream=side):
        sequence()
    for p in range(3):
        ins = data(601 + p)
        for i in range(len(plugins)):
            x[i].copy_(ins[i][rank])
        for t in y:
            t.zero_()
        graph.replay()
        torch.cuda.synchronize()
        ok, msg = verify(f"graph replay {p}", ins, y, gy)
        if not ok:
            return ok, msg
    return True, ""


def _worker(rank, world, port, results):
    sys.path.insert(0, os.path.join(HERE, ".."))
    sys.path.insert(0, os.path.join(HERE, "..", "tensorrt-llm_amd"))
    import oracle
    from mi355q import _lib as L
    from mi355q.ipc import CustomAllReduceHelper
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, msg = True, ""
    try:
        # timeout_ms: the direct launcher's bound (the plugin's: 30 s).  max_workgroups: the ranks share ONE GPU here, and
        # a workgroup that waits for its peers keeps its slot -- 8 ranks x 256 workgroups do not fit the chip together
        # (measured: some ranks' workgroups never start and the waits run out), so the count per rank is bounded
        helper = CustomAllReduceHelper(size=8 << 20, timeout_ms=10000, max_workgroups=max(32, 512 // world))
        say = (lambda m: print(f"[allreduce world {world}] {m}", flush=True)) if rank == 0 else (lambda m: None)
        # the last two: 1 MiB (256 workgroups of 4 KiB, the cfg4 M = 64 message) and 2 MiB per rank
        # incl. the cfg4 messages [M, 8192] bf16 for M = 1, 8, 64 (16 KB, 128 KB, 1 MB)
        cases = [(torch.float16, 8 * 64), (torch.float16, 8 * 1024 * 3), (torch.bfloat16, 8192 * 4),
                 (torch.float32, 4 * 1024 * 5), (torch.float16, 8 * world * 999), (torch.bfloat16, 64 * world),
                 (torch.bfloat16, 8192), (torch.bfloat16, 8 * 8192), (torch.bfloat16, 64 * 8192), (torch.float16, 1 << 20)]
        for strat, sname in ((L.AR_ONESHOT, "oneshot"), (L.AR_TWOSHOT, "twoshot")):
            for ci, (dtype, n) in enumerate(cases):
                nbytes = n * dtype.itemsize
                if (world * nbytes if strat == L.AR_ONESHOT else 2 * nbytes) > helper.size:
                    continue  # beyond the comm buffers an explicit strategy goes to RING: another summation order
                say(f"{sname} case {ci}")
                g = torch.Generator().manual_seed(1234 + n)
                inputs = [(torch.randn(n, generator=g) * 3).to(dtype) for _ in range(world)]  # every rank builds all
                x = inputs[rank].to(dev)
                # alternate: plugin / direct launcher, out of place / in place
                out = helper.allreduce(x, strategy=strat, direct=bool(ci & 1), out=x if ci & 2 else None)
                torch.cuda.synchronize()
                ref = oracle.custom_all_reduce(inputs, rank, sname)
                if not torch.equal(out.cpu().view(torch.uint8), ref.view(torch.uint8)):
                    d = (out.cpu().float() - ref.float()).abs().max().item()
                    ok, msg = False, (f"rank {rank} {sname} {dtype} n={n}: max abs diff {d}; out {out[:4].tolist()} ref "
                                     f"{ref[:4].tolist()} inputs {[t[:4].tolist() for t in inputs]}")
                    break
            if not ok:
                break
        if ok:  # AUTO: small message -> one-shot, larger than the workspace -> RING through torch.distributed (gloo here)
            x = torch.full((256,), float(rank + 1), dtype=torch.float16, device=dev)
            y = helper.allreduce(x)
            torch.cuda.synchronize()
            if not torch.equal(y.cpu(), torch.full((256,), world * (world + 1) / 2, dtype=torch.float16)):
                ok, msg = False, f"rank {rank} AUTO small: {y[:4].tolist()}"
        if ok:
            # AllGather (allgatherPlugin.cpp:90-110) on the same buffers, interleaved with all-reduces so the ping-pong
            # counter and flags are shared: out = the ranks' tensors back to back along dim 0, byte for byte
            gcases = [(torch.float16, (1, 8)), (torch.bfloat16, (3, 1000)), (torch.float32, (64, 1024)),
                      (torch.int8, (16, 4096)), (torch.float16, (512, 1024)), (torch.float16, (5, 3))]  # last: RCCL/gloo
            for ci, (dtype, shape) in enumerate(gcases):
                g = torch.Generator().manual_seed(77 + ci)
                n = shape[0] * shape[1]
                parts = [(torch.randn(n, generator=g) * 50).to(dtype).reshape(shape) for _ in range(world)]
                if dtype == torch.float16 and shape == (5, 3):
                    x = parts[rank].clone()                 # 30 bytes: not a multiple of 16 -> torch.distributed (gloo: CPU)
                    y = helper.allgather(x)
                else:
                    y = helper.allgather(parts[rank].to(dev), direct=bool(ci & 1))
                    torch.cuda.synchronize()
                    helper.allreduce(torch.ones(64, dtype=torch.float16, device=dev), strategy=L.AR_ONESHOT)
                ref = torch.cat(parts, dim=0)
                if y.shape != ref.shape or not torch.equal(y.cpu().view(torch.uint8), ref.view(torch.uint8)):
                    ok, msg = False, f"rank {rank} allgather {dtype} {shape}: {y.flatten()[:4].tolist()} vs {ref.flatten()[:4].tolist()}"
                    break
            if ok:  # a shard that does not fit ranks * slot in the comm buffer goes to torch.distributed too
                big = torch.zeros((8 << 20) // world + 256, dtype=torch.uint8)
                y = helper.allgather(big)
                ok = y.shape[0] == big.shape[0] * world
                msg = "" if ok else f"rank {rank} allgather fallback shape {tuple(y.shape)}"
        if ok:
            say("replay")
            ok, msg = _replay_cases(rank, world, dev, helper, oracle, L)
        say(f"done ok={ok} {msg}")
        dist.barrier()
        helper.close()
        if ok and world == 2:
            # a late rank: rank 1 arrives 1.5 s after rank 0, whose waits are bounded to 0.3 s -> rank 0 gets NaN and
            # the timeout word, and refuses further all-reduces; rank 1 finds rank 0's data and flags in place
            late = CustomAllReduceHelper(size=1 << 20, timeout_ms=300 if rank == 0 else 0, max_workgroups=128)
            x = torch.full((4096,), float(rank + 1), dtype=torch.float16, device=dev)
            dist.barrier()
            if rank == 1:
                import time
                time.sleep(1.5)
            y = late.allreduce(x, strategy=L.AR_ONESHOT, direct=True)
            torch.cuda.synchronize()
            if rank == 0:
                if not (torch.isnan(y).all().item() and late.timed_out()):
                    ok, msg = False, f"rank 0: expected NaN output and the timeout word, got {y[:4].tolist()} / {late.timed_out()}"
                try:
                    late.allreduce(x, strategy=L.AR_ONESHOT)
                    ok, msg = False, "rank 0: the plugin accepted work after a timeout"
                except Exception as e:  # noqa: BLE001
                    if "timed out" not in str(e):
                        ok, msg = False, f"rank 0: unexpected error {e}"
            elif not torch.equal(y.cpu(), torch.full((4096,), 3.0, dtype=torch.float16)) or late.timed_out():
                ok, msg = False, f"rank 1: {y[:4].tolist()} timed_out={late.timed_out()}"
            dist.barrier()
            try:
                late.close()
                if rank == 0:
                    ok, msg = False, "rank 0: close() did not report the timeout"
            except RuntimeError as e:
                if rank != 0 or "timed out" not in str(e):
                    ok, msg = False, f"rank {rank}: close raised {e}"
    except Exception as e:  # noqa: BLE001 -- report through the queue, the parent asserts
        ok, msg = False, f"rank {rank}: {type(e).__name__}: {e}"
    results.put((rank, ok, msg))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_custom_allreduce_multi_process_single_gpu(cuda, world):
    ctx = mp.get_context("spawn")
    results = ctx.Queue()
    port = 29650 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    for _ in range(world):
        got.append(results.get(timeout=240))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok, msg in sorted(got):
        assert ok, msg


def test_ring_strategy_runs_rccl_inside_the_plugin(cuda):
    """RING = ncclAllReduce issued by AllreducePlugin::enqueue itself (allreducePlugin.cpp:181-185) on the communicator
    of initCommMap (plugins/common/plugin.cpp:50-87); the reference's one-input AllGather likewise (allgatherPlugin.cpp:
    90-110).  RCCL takes one rank per GPU and the box has one GPU, so the group here is {0}: this runs the whole C path
    (librccl loaded on demand, unique id, ncclCommInitRank, the plugin's dispatch, the collective on torch's stream) --
    the cross-GPU run is the TP leg of `bench.py --gpus N`."""
    import ctypes as C
    from mi355q import _lib as L
    from mi355q.plugin import Plugin, torch_dtype_to_trt
    grp = (C.c_int32 * 1)(0)
    uid = (C.c_uint8 * L.COMM_UNIQUE_ID_BYTES)()
    L.check(L.lib().mi355q_comm_get_unique_id(uid))
    L.check(L.lib().mi355q_comm_init_rank(grp, 1, uid, 0))
    assert L.lib().mi355q_comm_exists(grp, 1) == 1
    L.check(L.lib().mi355q_comm_init_rank(grp, 1, uid, 0))  # a group that has a communicator is left alone (:54)
    try:
        for dtype in (torch.float16, torch.bfloat16, torch.float32):
            x = torch.randn(4096 * 33, device=cuda).to(dtype)
            for strat in (L.AR_RING, L.AR_AUTO):
                plg = Plugin("AllReduce", [("group", [0], L.FIELD_INT32), ("type_id", torch_dtype_to_trt(dtype), L.FIELD_INT32),
                                           ("strategy", strat, L.FIELD_INT8), ("counter", 0, L.FIELD_INT32)])
                plg = Plugin.deserialize("AllReduce", plg.serialize())
                y = torch.zeros_like(x)
                if strat == L.AR_RING:
                    plg.enqueue([x], [y])
                else:  # AUTO with comm buffers of 64 KiB: the 270 KB / 540 KB message goes to RING (:115-118)
                    table = torch.tensor([1, 2, 3, 4, 5, 6, 7, 8, 65536, 0], dtype=torch.int64)
                    plg = Plugin("AllReduce", [("group", [0, 1], L.FIELD_INT32), ("type_id", torch_dtype_to_trt(dtype), L.FIELD_INT32),
                                               ("strategy", strat, L.FIELD_INT8), ("counter", 0, L.FIELD_INT32), ("rank", 0, L.FIELD_INT32)])
                    with pytest.raises(L.Mi355qError, match="mi355q_comm_init_rank"):   # group {0, 1} has no communicator
                        plg.enqueue([x, table], [y])
                    continue
                torch.cuda.synchronize()
                assert torch.equal(x, y)
        x = torch.randint(-100, 100, (5, 3), dtype=torch.int8, device=cuda)   # 15 bytes: any size goes through RCCL
        ag = Plugin.deserialize("AllGather", Plugin("AllGather", [("group", [0], L.FIELD_INT32), ("type_id", L.INT8, L.FIELD_INT32)]).serialize())
        y = torch.zeros_like(x)
        ag.enqueue([x], [y])
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    finally:
        L.check(L.lib().mi355q_comm_destroy(grp, 1))
    assert L.lib().mi355q_comm_exists(grp, 1) == 0
