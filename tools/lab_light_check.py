"""Lab: which (dtype, m, k, group) cases of the light GEMV match the oracle (prints max |err| per case)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import oracle
from mi355q import functional as F
dev = torch.device("cuda:0")
DT = {"float16": torch.float16, "bfloat16": torch.bfloat16}
for dts in ("float16", "bfloat16"):
    for (m, n, k, pre, zero, bias, gs) in ((2, 1024, 128, False, False, False, 128), (2, 1024, 128, False, True, False, 128),
                                           (2, 1024, 128, False, True, True, 128), (2, 1024, 256, False, True, True, 64),
                                           (1, 1024, 512, False, True, False, 128), (6, 1024, 512, False, True, False, 128),
                                           (12, 1024, 512, False, True, False, 128), (4, 1024, 128, True, True, True, 128)):
        torch.manual_seed(0)
        dt = DT[dts]
        g = (k + gs - 1) // gs
        act = torch.randn(m, k, dtype=dt)
        b = torch.randn(1, n, dtype=dt) if bias else None
        z = torch.randn(g, n, dtype=dt) if zero else None
        s = torch.rand(g, n, dtype=dt)
        pq = torch.rand(1, k, dtype=dt)
        w = torch.randint(-2 ** 31, 2 ** 31, (k, n // 8), dtype=torch.int32).view(torch.int8)
        q = F.unpack_int4_packed_tensor_to_int8(w)
        cw = F.preprocess_weights_for_mixed_gemm(w, torch.quint4x2).view(dt)
        algo = pre * F.PRE_QUANT_SCALE + zero * F.ZERO + bias * F.BIAS
        d = lambda t: None if t is None else t.to(dev)
        out = F.weight_only_groupwise_quant_matmul(d(act), d(pq), d(cw), d(s), d(z), d(b), None, algo, gs, dts)
        torch.cuda.synchronize()
        ref16, ref32 = oracle.woq_groupwise_matmul(act, pq if pre else None, q.numpy(), s, z, b, gs)
        err = (out.cpu().float() - ref32).abs()
        print(dts, (m, n, k, pre, zero, bias, gs), "max err %.4g (ref max %.4g), rows: %s" % (
            err.max().item(), ref32.abs().max().item(), [round(e, 3) for e in err.max(dim=1).values.tolist()]), flush=True)
