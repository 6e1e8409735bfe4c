"""Lab: two processes on device 0, one small one-shot all-reduce, print what came out."""
import os, sys
import torch, torch.distributed as dist, torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def worker(rank, world):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
    from mi355q import _lib as L
    from mi355q.ipc import CustomAllReduceHelper
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29711"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    h = CustomAllReduceHelper(size=int(os.environ.get("AR_SIZE", 1 << 20)))
    for strat in (L.AR_ONESHOT, L.AR_TWOSHOT):
        for direct in (True, False):
            x = (torch.arange(512, dtype=torch.float32) + 1000 * (rank + 1)).to(torch.float16).cuda()
            y = h.allreduce(x, strategy=strat, direct=direct)
            torch.cuda.synchronize()
            print(f"rank {rank} strat {strat} direct {direct}: in {x[:3].tolist()} out {y[:3].tolist()} .. {y[-2:].tolist()} timed_out {h.timed_out()}", flush=True)
            dist.barrier()
    h.close()
    dist.destroy_process_group()

if __name__ == "__main__":
    mp.spawn(worker, args=(2,), nprocs=2)
