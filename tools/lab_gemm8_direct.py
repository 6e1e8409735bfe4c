"""Lab: the 'weights straight from L2' int8 GEMM (MI355Q_GEMM8_DIRECT, LAB build) -- exactness against the shipped phased
kernel, then an interleaved A/B in one process."""
import sys, os, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mi355q import functional as F
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
VARIANTS = {"phased": {}, "direct": {"MI355Q_GEMM8_DIRECT": "1"}, "direct-no-touch": {"MI355Q_GEMM8_DIRECT": "1", "MI355Q_GEMM8_ABL": "1"},
            "direct-w-tile0": {"MI355Q_GEMM8_DIRECT": "1", "MI355Q_GEMM8_ABL": "3"},
            "direct-pretiled": {"MI355Q_GEMM8_DIRECT": "1", "MI355Q_GEMM8_ABL": "5"}}
def pretile(b):
    """[N, K] int8 -> fragment order: block (n-tile of 16 rows, k-step of 64 B) = 1 KiB, lane l = (k-chunk l >> 4, row l & 15)."""
    n, k = b.shape
    return b.view(n // 16, 16, k // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(n, k)
KEYS = ("MI355Q_GEMM8_W128", "MI355Q_GEMM8_ABL", "MI355Q_GEMM8_DIRECT")
def setenv(env):
    for kk in KEYS:
        os.environ.pop(kk, None)
    os.environ.update(env)
plg = F.Plugin("SmoothQuantGemm", [("has_per_channel_scaling", 1, 5), ("has_per_token_scaling", 1, 5), ("type_id", 1, 5)])
ok = True
for (m, n, k) in [(256, 256, 384), (512, 768, 1024), (1024, 1024, 640), (4096, 13824, 5120)]:
    torch.manual_seed(m + n + k)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sa = (torch.randint(1, 10, (m, 1), device=dev).float() * 1e-2)
    sb = (torch.randint(1, 10, (1, n), device=dev).float() * 1e-2)
    outs = {}
    bt = pretile(b)
    for v in ("phased", "direct", "direct-pretiled"):
        setenv(VARIANTS[v])
        os.environ["MI355Q_GEMM8_TILE"] = "0"
        o = torch.empty(m, n, dtype=torch.float16, device=dev)
        for _ in range(3):
            plg.enqueue([a, bt if v == "direct-pretiled" else b, sa, sb], [o])
        torch.cuda.synchronize()
        outs[v] = o.clone()
    e = torch.equal(outs["phased"], outs["direct"])
    e2 = torch.equal(outs["phased"], outs["direct-pretiled"])
    ok &= e and e2
    print((m, n, k), "direct == phased:", e, " pretiled == phased:", e2, flush=True)
os.environ.pop("MI355Q_GEMM8_TILE", None)
if not ok:
    sys.exit(1)
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("SHAPES", "4096x12288x5120,8192x8192x8192,4096x13824x5120").split(",")]
for (m, n, k) in shapes:
    torch.manual_seed(0)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sa = (torch.randint(1, 10, (m, 1), device=dev).float() * 1e-2)
    sb = (torch.randint(1, 10, (1, n), device=dev).float() * 1e-2)
    out = torch.empty(m, n, dtype=torch.float16, device=dev)
    res = {v: [] for v in VARIANTS}
    bt = pretile(b)
    for rnd in range(int(os.environ.get("ROUNDS", 5))):
        for v, env in VARIANTS.items():
            setenv(env)
            bb = bt if v == "direct-pretiled" else b
            _, dev_s = bench.time_steps(lambda: plg.enqueue([a, bb, sa, sb], [out]), 20, 2, lambda: None)
            res[v].append(2.0 * m * n * k / (dev_s / 20) / 1e12)
    for v, xs in res.items():
        print((m, n, k), f"{v:18s} median {statistics.median(xs):7.1f}  max {max(xs):7.1f}  min {min(xs):7.1f} TOPS", flush=True)
