"""Lab: decode-attention kernel time for the BASELINE configs 2/4/5 shapes (run under rocprofv3 --kernel-trace)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
import torch
from mi355q import functional as F

dev = torch.device("cuda")
def case(B, Hq, Hkv, D, ctx, tpb, kvdt, reps=6, ragged=False):
    nb = ctx // tpb
    nblk = B * nb
    pool = (torch.randn(nblk, 2, Hkv, tpb, D, device=dev) * 0.5)
    pool = pool.to(torch.bfloat16) if kvdt == torch.bfloat16 else (pool * 20).clamp(-127, 127).to(torch.int8) if kvdt == torch.int8 else pool.to(kvdt)
    perm = torch.randperm(nblk).reshape(B, nb)
    be = Hkv * tpb * D * pool.element_size()
    ptrs = (torch.stack([perm * 2 * be, perm * 2 * be + be], 1) + pool.data_ptr()).reshape(1, B, 2, -1)
    qkv = torch.randn(B, 1, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
    seq = torch.randint(ctx // 2, ctx, (B,), dtype=torch.int32) if ragged else torch.full((B,), ctx, dtype=torch.int32)
    qm = 0 if kvdt == torch.bfloat16 else (1 << 6 if kvdt == torch.int8 else 1 << 7)
    plg = F.make_gpt_attention_plugin(num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="bfloat16", position_embedding_type=2,
                                      rotary_embedding_dim=D, tokens_per_block=tpb, kv_cache_quant_mode=qm)
    kw = dict(qkv=qkv, sequence_length=seq.to(dev), host_past_key_value_lengths=seq - 1,
              host_max_attention_window_sizes=torch.tensor([ctx], dtype=torch.int32), host_sink_token_length=torch.zeros(1, dtype=torch.int32),
              context_lengths=(seq - 1).to(dev), cache_indirection=torch.zeros(B, 1, ctx, dtype=torch.int32, device=dev),
              host_request_types=torch.ones(B, dtype=torch.int32), kv_cache_block_pointers=ptrs.to(dev), host_kv_cache_block_pointers=ptrs)
    if qm:
        kw["kv_orig_quant_scale"] = torch.tensor([20.0], device=dev); kw["kv_quant_orig_scale"] = torch.tensor([0.05], device=dev)
    for _ in range(reps):
        F.gpt_attention(plg, **kw)
    torch.cuda.synchronize()
    nbytes = int(seq.sum()) * 2 * Hkv * D * pool.element_size()
    print(f"CASE B={B} Hq={Hq} Hkv={Hkv} ctx={ctx} kv={kvdt} ragged={ragged} bytes={nbytes}", flush=True)

which = sys.argv[1]
{"cfg2": lambda: case(64, 32, 32, 128, 2048, 64, torch.bfloat16),
 "cfg2r": lambda: case(64, 32, 32, 128, 2048, 64, torch.bfloat16, ragged=True),
 "cfg4": lambda: case(64, 8, 1, 128, 2048, 64, torch.int8),
 "cfg4b": lambda: case(64, 64, 8, 128, 2048, 64, torch.int8),
 "cfg5": lambda: case(64, 32, 8, 128, 2048, 64, torch.float8_e4m3fn),
 "gqa16": lambda: case(64, 64, 8, 128, 2048, 64, torch.bfloat16),
 "b1": lambda: case(1, 32, 32, 128, 4096, 64, torch.bfloat16),
 "b8": lambda: case(8, 32, 32, 128, 2048, 64, torch.bfloat16)}[which]()
