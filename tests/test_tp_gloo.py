"""World-size-2 CPU (gloo) tests of the tensor-parallel split: column-parallel needs no
collective, row-parallel all-reduces the [M, N] outputs and adds the bias once, attention shards by
kv-head.  The local GEMM is the CPU checker (oracle), so what is under test is mi355q/tp.py's split
and reduce logic together with the host-side weight preprocessing of every shard."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _layer(seed, m, n, k, g, dt, bits=4):
    torch.manual_seed(seed)
    x = torch.randn(m, k, dtype=dt)
    groups = k // g if g else 1
    scale = (torch.rand(groups, n, dtype=dt) * 0.05) if g else torch.rand(n, dtype=dt) * 0.05
    zero = torch.randn(groups, n, dtype=dt) * 0.05 if g else None
    bias = torch.randn(n, dtype=dt)
    pq = torch.rand(1, k, dtype=dt) + 0.5 if g else None
    qw = torch.randint(-128, 128, (k, n * bits // 8), dtype=torch.int8)
    return x, qw, scale, zero, bias, pq


def _oracle_matmul(g):
    import oracle
    from mi355q import functional as F
    from mi355q import _lib as L

    def run(x, shard, bias):
        # every shard goes through the real host preprocessing and back: the device layout of a
        # shard must describe the same integers as the slice it was cut from
        qt = torch.int8 if shard.bits == 8 else torch.quint4x2
        back = F.unprocess_weights_for_mixed_gemm(shard.processed(), qt, L.LAYOUT_GFX950)
        assert torch.equal(back, shard.qweight)
        q = oracle.unpack_int4_packed_tensor_to_int8(shard.qweight.numpy()) if shard.bits == 4 else shard.qweight.numpy()
        if shard.group_size:
            ref16, _ = oracle.woq_groupwise_matmul(x, shard.pre_quant_scale, q, shard.scales, shard.zeros,
                                                   None if bias is None else bias.reshape(1, -1), shard.group_size)
        else:
            ref16, _ = oracle.woq_matmul(x, q, shard.scales)
            if bias is not None:
                ref16 = (ref16.float() + bias.float().reshape(1, -1)).to(x.dtype)
        return ref16
    return run


def _worker(rank, world, port, dtname, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mi355q import tp
        import oracle
        dt = getattr(torch, dtname)
        m, n, k, g = 3, 64, 256, 64
        x, qw, scale, zero, bias, pq = _layer(0, m, n, k, g, dt)
        mm = _oracle_matmul(g)
        q_full = oracle.unpack_int4_packed_tensor_to_int8(qw.numpy())
        full16, full32 = oracle.woq_groupwise_matmul(x, pq, q_full, scale, zero, bias.reshape(1, -1), g)

        # column parallel: disjoint N slices, gathered only to compare
        cs = tp.column_split(qw, scale, zero, bias, pq, g, 4, world, rank)
        y_col = tp.column_parallel_linear(x, cs, mm, gather_output=True)
        assert y_col.shape == (m, n)
        assert torch.equal(y_col, full16), "column-parallel slices must equal the unsplit result bit for bit"

        # row parallel: K slices, all-reduce, bias once
        rs = tp.row_split(qw, scale, zero, bias, pq, g, 4, world, rank)
        k0, k1 = tp.shard_range(k, world, rank)
        y_row = tp.row_parallel_linear(x[:, k0:k1].contiguous(), rs, mm)
        tol = 4 * torch.finfo(dt).eps * full32.abs().max().item()
        assert (y_row.float() - full32).abs().max().item() <= tol
        # every rank holds the same reduced tensor
        both = [torch.empty_like(y_row) for _ in range(world)]
        dist.all_gather(both, y_row)
        assert torch.equal(both[0], both[1])

        # per-channel int8 weights, row parallel (scales replicated)
        x8, qw8, s8, _, b8, _ = _layer(1, 2, 32, 128, 0, dt, bits=8)
        rs8 = tp.row_split(qw8, s8, None, b8, None, 0, 8, world, rank)
        k0, k1 = tp.shard_range(128, world, rank)
        y8 = tp.row_parallel_linear(x8[:, k0:k1].contiguous(), rs8, _oracle_matmul(0))
        ref8, ref8_32 = oracle.woq_matmul(x8, qw8.numpy(), s8)
        want = ref8_32 + b8.float().reshape(1, -1)
        assert (y8.float() - want).abs().max().item() <= 4 * torch.finfo(dt).eps * want.abs().max().item()

        # attention heads: Llama-2-70B (64 q / 8 kv) over 2 ranks; then more ranks than kv-heads
        q0, q1, kv0, kv1 = tp.attention_head_split(64, 8, world, rank)
        assert (q1 - q0, kv1 - kv0) == (32, 4) and q0 == rank * 32 and kv0 == rank * 4
        assert tp.attention_head_split(64, 8, 16, 5) == (20, 24, 2, 3)
        results[rank] = "ok"
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtname", ["float16", "bfloat16"])
def test_tp2_split_and_allreduce_gloo(dtname):
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(world, port, dtname, results), nprocs=world, join=True)
        assert dict(results) == {0: "ok", 1: "ok"}


def test_shard_range_and_strategy_policy():
    from mi355q import tp
    assert tp.shard_range(4096, 8, 3) == (1536, 2048)
    with pytest.raises(ValueError):
        tp.shard_range(100, 8, 0)
    # allreducePlugin.cpp:111-139: small messages one-shot, mid two-shot, large (or no workspace) ring
    assert tp.select_allreduce_strategy(16 * 1024, 8) == 1
    assert tp.select_allreduce_strategy(64 * 1024 * 1024, 8) == 0
    # sq splits keep shapes consistent
    w = torch.randint(-128, 128, (64, 128), dtype=torch.int8)
    wc, sc = tp.sq_column_split(w, torch.rand(1, 64), 2, 1)
    assert wc.shape == (32, 128) and sc.shape == (1, 32)
    assert tp.sq_row_split(w, 2, 0).shape == (64, 64)
