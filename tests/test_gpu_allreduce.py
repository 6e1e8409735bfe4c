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
        helper = CustomAllReduceHelper(size=8 << 20)
        # the last two: 1 MiB (256 workgroups of 4 KiB, the cfg4 M = 64 message) and 2 MiB per rank
        cases = [(torch.float16, 8 * 64), (torch.float16, 8 * 1024 * 3), (torch.bfloat16, 8192 * 4),
                 (torch.float32, 4 * 1024 * 5), (torch.float16, 8 * world * 999), (torch.bfloat16, 64 * world),
                 (torch.bfloat16, 64 * 8192), (torch.float16, 1 << 20)]
        for strat, sname in ((L.AR_ONESHOT, "oneshot"), (L.AR_TWOSHOT, "twoshot")):
            for ci, (dtype, n) in enumerate(cases):
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
        dist.barrier()
        helper.close()
        if ok and world == 2:
            # a late rank: rank 1 arrives 1.5 s after rank 0, whose waits are bounded to 0.3 s -> rank 0 gets NaN and
            # the timeout word, and refuses further all-reduces; rank 1 finds rank 0's data and flags in place
            late = CustomAllReduceHelper(size=1 << 20, timeout_ms=300 if rank == 0 else 0)
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


@pytest.mark.parametrize("world", [2, 4])
def test_custom_allreduce_multi_process_single_gpu(cuda, world):
    ctx = mp.get_context("spawn")
    results = ctx.Queue()
    port = 29650 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, results)) for r in range(world)]
    for p in procs:
        p.start()
    got = []
    for _ in range(world):
        got.append(results.get(timeout=100))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, ok, msg in sorted(got):
        assert ok, msg
