"""Two execution contexts at once: the reference keeps the multi-block counters of
the generation kernel in each context's own workspace (gptAttentionCommon.cpp:1253-1269), so two contexts may run on two
streams.  Here the arrival words of the in-launch reductions live in the library (csrc/arrival_ring.hip) and are handed
out per stream and per stream capture.  This test records split-KV generation steps of two different requests into two
hipGraphs captured at different times (with enough eager launches in between that a process-wide ring -- the round-2
design -- would have wrapped onto the first graph's words), replays both graphs concurrently on two streams next to
eager split-KV launches on a third, and checks every output against the oracle (generation tolerance) and bit for
bit against the same launch run alone."""
import ctypes as C

import pytest
import torch

import oracle
from mi355q import _lib as L

from test_gpu_attention import ATOL, build_case

pytestmark = pytest.mark.gpu


class _Step:
    """One generation step through mi355q_masked_multihead_attention with every buffer preallocated."""

    def __init__(self, dev, B, Hq, Hkv, D, seq_lens, tpb, kv_dtype, seed, splits=0):
        dtype = torch.bfloat16
        cache, qkv, kv_scale = build_case(B, Hq, Hkv, D, seq_lens, tpb, dtype, kv_dtype, seed=seed)
        self.pool = cache.pool.to(dev)                       # before the oracle appends the new token to its copy
        self.ref = oracle.decode_attention(qkv, cache, seq_lens, Hq, Hkv, D, rope="neox", rot_dim=D,
                                           kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale)
        offs = cache.offsets()
        self.ptrs = (offs * self.pool.element_size() + self.pool.data_ptr()).to(torch.int64).to(dev)
        self.qkv = qkv.to(dev)
        self.lens = torch.tensor(seq_lens, dtype=torch.int32, device=dev)
        self.out = torch.zeros(B, Hq * D, dtype=dtype, device=dev)
        self.scales = torch.tensor([1.0 / kv_scale, kv_scale], dtype=torch.float32, device=dev)
        p = L.DecodeAttentionParams()
        p.qkv, p.qkv_bias, p.out = self.qkv.data_ptr(), None, self.out.data_ptr()
        p.sequence_lengths, p.block_pointers = self.lens.data_ptr(), self.ptrs.data_ptr()
        p.batch, p.num_heads, p.num_kv_heads, p.head_size = B, Hq, Hkv, D
        p.max_blocks_per_seq, p.tokens_per_block, p.max_seq_len = cache.max_blocks, tpb, cache.max_blocks * tpb
        p.act_type = L.BF16
        p.kv_type = {torch.bfloat16: L.BF16, torch.int8: L.INT8}[kv_dtype]
        if kv_dtype == torch.int8:
            p.kv_scale_orig_quant, p.kv_scale_quant_orig = self.scales.data_ptr(), self.scales.data_ptr() + 4
        p.position_embedding_type, p.rotary_embedding_dim, p.rotary_embedding_base = 2, D, 10000.0
        p.rotary_embedding_scale_type, p.rotary_embedding_scale, p.rotary_embedding_max_positions = 0, 1.0, 8192
        p.q_scaling, p.multi_block_mode = 1.0, splits
        need = L.lib().mi355q_decode_attention_workspace_size(C.byref(p))
        assert need > 0, "the case must split the KV range (otherwise no arrival words are used)"
        self.ws = torch.empty(need, dtype=torch.uint8, device=dev)
        p.workspace, p.workspace_bytes = self.ws.data_ptr(), need
        self.p, self.dtype = p, dtype

    def launch(self):
        L.check(L.lib().mi355q_masked_multihead_attention(C.byref(self.p), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def verify(self, alone, tag):
        got = self.out.cpu()
        err = (got.float() - self.ref.float()).abs().max().item()
        assert err <= ATOL[self.dtype], f"{tag}: max abs err {err} against the oracle"
        assert torch.equal(got.view(torch.int16), alone.view(torch.int16)), f"{tag}: differs from the same launch run alone"


def test_two_graphs_and_an_eager_stream_run_concurrently(cuda):
    a = _Step(cuda, 6, 8, 1, 128, [3000, 2500, 1800, 2999, 700, 2048], 64, torch.int8, seed=5)      # GQA MFMA kernel, split
    b = _Step(cuda, 4, 16, 4, 128, [2100, 1500, 2049, 900], 64, torch.bfloat16, seed=6)             # another shape
    c = _Step(cuda, 3, 4, 4, 64, [2500, 1700, 2300], 32, torch.bfloat16, seed=7)                    # the dot-product kernel's split path
    filler = _Step(cuda, 64, 8, 8, 128, [300] * 64, 64, torch.bfloat16, seed=8, splits=4)           # 512 arrival words per launch
    alone = []
    for s in (a, b, c):
        s.launch()
        torch.cuda.synchronize()
        alone.append(s.out.cpu().clone())
        s.verify(alone[-1], "alone")
        s.out.zero_()
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(ga, stream=sa):
        for _ in range(12):
            a.launch()
    for _ in range(80):          # 80 x 512 lines: more than the 32768 lines of the old process-wide ring
        filler.launch()
    torch.cuda.synchronize()
    with torch.cuda.graph(gb, stream=sb):
        for _ in range(12):
            b.launch()
    torch.cuda.synchronize()
    for rnd in range(6):
        for s in (a, b, c):
            s.out.zero_()
        torch.cuda.synchronize()
        for _ in range(5):
            with torch.cuda.stream(sa):
                ga.replay()
            with torch.cuda.stream(sb):
                gb.replay()
            with torch.cuda.stream(sc):
                for _ in range(12):
                    c.launch()
            for _ in range(4):   # and the default stream
                filler.launch()
        torch.cuda.synchronize()
        for s, ref, name in ((a, alone[0], "graph A"), (b, alone[1], "graph B"), (c, alone[2], "eager stream")):
            s.verify(ref, f"round {rnd}, {name}")


class _Gemm:
    """One W4A16 GEMM through mi355q_weight_only_gemm with every buffer preallocated (its own workspace: the partial
    tiles of the in-launch split-K are the caller's; the arrival words are the library's)."""

    def __init__(self, dev, m, n, k, seed):
        g = torch.Generator().manual_seed(seed)
        dt = torch.bfloat16
        self.w = torch.randint(-128, 128, (k, n // 2), dtype=torch.int8, generator=g).to(dev)   # any bytes are int4 tiles
        self.scale = (torch.rand(k // 128, n, generator=g) * 0.02 + 0.001).to(dt).to(dev)
        self.zero = (torch.randn(k // 128, n, generator=g) * 0.05).to(dt).to(dev)
        self.act = torch.randn(m, k, generator=g).to(dt).to(dev)
        self.out = torch.zeros(m, n, dtype=dt, device=dev)
        need = L.lib().mi355q_weight_only_gemm_workspace_size(m, n, k, L.WEIGHT_INT4)
        self.ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev)
        p = L.WeightOnlyParams()
        p.qweight, p.scales, p.zeros = self.w.data_ptr(), self.scale.data_ptr(), self.zero.data_ptr()
        p.in_, p.out = self.act.data_ptr(), self.out.data_ptr()
        p.m, p.n, p.k, p.group_size = m, n, k, 128
        p.weight_type, p.act_type = L.WEIGHT_INT4, L.BF16
        p.workspace, p.workspace_bytes = self.ws.data_ptr(), need
        self.p = p

    def launch(self):
        L.check(L.lib().mi355q_weight_only_gemm(C.byref(self.p), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def alone(self):
        self.out.zero_()
        self.launch()
        torch.cuda.synchronize()
        res = self.out.clone()
        self.out.zero_()
        return res


def test_split_k_gemms_in_two_graphs_replayed_concurrently(cuda):
    # the GEMMs that close their split-K inside the launch -- the light GEMV (m = 1, a cfg4 slice: 8 k-segments per n-block)
    # and the skinny GEMM (m = 64: 8 pieces per tile) -- count arrivals on library words as well: captured into two graphs
    # at different times, replayed together on two streams next to eager launches of both on a third and on the default
    # stream; every result bit for bit what the same launch gives alone
    dev = cuda
    ga = [_Gemm(dev, 1, 1280, 8192, 1), _Gemm(dev, 64, 3584, 8192, 2)]
    gb = [_Gemm(dev, 4, 3584, 8192, 3), _Gemm(dev, 40, 1280, 8192, 4)]
    ge = [_Gemm(dev, 1, 1280, 8192, 5), _Gemm(dev, 64, 3584, 8192, 6)]
    want = [[g.alone() for g in grp] for grp in (ga, gb, ge)]

    def capture(grp):
        side, graph = torch.cuda.Stream(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                for g in grp:
                    g.launch()
        return graph

    graph_a = capture(ga)
    for _ in range(40):          # eager traffic between the two captures
        for g in ge:
            g.launch()
    graph_b = capture(gb)
    torch.cuda.synchronize()
    for g in ge:
        g.out.zero_()
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(6):
        with torch.cuda.stream(sa):
            graph_a.replay()
        with torch.cuda.stream(sb):
            graph_b.replay()
        with torch.cuda.stream(sc):
            ge[0].launch()
        ge[1].launch()
        torch.cuda.synchronize()
        for grp, exp in zip((ga, gb, ge), want):
            for g, e in zip(grp, exp):
                assert torch.equal(g.out.view(torch.int16), e.view(torch.int16)), f"repetition {rep}: m={g.p.m} n={g.p.n}"
                g.out.zero_()
