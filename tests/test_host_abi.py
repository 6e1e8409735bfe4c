"""CPU-only tests of the C-ABI library: it loads, exports every symbol include/mi355q.h declares,
its host-side weight tooling is bit-exact against the oracle, and the plugin objects reproduce the
reference's shape / format / serialisation logic.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F
from mi355q.plugin import Plugin, get_plugin_names

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi355q.h")).read()
    declared = set(re.findall(r"\b(mi355q_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = C.CDLL(L.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"symbols declared in mi355q.h but not exported: {missing}"
    # and the Python binding table covers exactly the header
    assert set(L.SYMBOLS) == declared


def test_version_and_plugin_names():
    assert b"gfx950" in L.lib().mi355q_version()
    assert get_plugin_names() == ["WeightOnlyGroupwiseQuantMatmul", "WeightOnlyQuantMatmul", "SmoothQuantGemm",
                                  "QuantizePerToken", "GPTAttention", "Fp8Gemm", "QuantizeTensor",
                                  "RmsnormQuantization", "LayernormQuantization", "AllReduce", "AllGather",
                                  "MixtureOfExperts"]


@pytest.mark.parametrize("qt,tq", [(oracle.INT8_W, torch.int8), (oracle.INT4_W, torch.quint4x2)])
@pytest.mark.parametrize("K,N", [(64, 64), (128, 192), (384, 128)])
def test_reference_layouts_bit_exact_vs_oracle(qt, tq, K, N):
    rng = np.random.default_rng(K * 1000 + N)
    bits = 8 if qt == oracle.INT8_W else 4
    w = rng.integers(-128, 128, size=(K, N * bits // 8), dtype=np.int8)
    wt = torch.from_numpy(w)
    for layout, arch in ((L.LAYOUT_REF_SM80, 80), (L.LAYOUT_REF_SM90, 90)):
        got = F.preprocess_weights_for_mixed_gemm(wt, tq, layout).numpy()
        want = oracle.preprocess_weights_for_mixed_gemm(w, qt, arch=arch)
        assert (got == want).all(), f"layout {layout} differs from the oracle"
        back = F.unprocess_weights_for_mixed_gemm(torch.from_numpy(got), tq, layout).numpy()
        assert (back == w).all()


def test_reference_layout_batched_experts():
    rng = np.random.default_rng(7)
    w = rng.integers(-128, 128, size=(3, 64, 32), dtype=np.int8)  # 3 experts, int4 [64, 64]
    got = F.preprocess_weights_for_mixed_gemm(torch.from_numpy(w), torch.quint4x2, L.LAYOUT_REF_SM80).numpy()
    want = oracle.preprocess_weights_for_mixed_gemm(w, oracle.INT4_W, arch=80)
    assert (got == want).all()


@pytest.mark.parametrize("K", [128, 192, 64])
@pytest.mark.parametrize("tq,bits", [(torch.int8, 8), (torch.quint4x2, 4)])
def test_gfx950_layout_roundtrip_and_fragment_order(tq, bits, K):
    N = 64
    rng = np.random.default_rng(3)
    w = rng.integers(-128, 128, size=(K, N * bits // 8), dtype=np.int8)
    p = F.preprocess_weights_for_mixed_gemm(torch.from_numpy(w), tq, L.LAYOUT_GFX950)
    assert p.shape == (K, N * bits // 8)  # same tensor shape / byte count as the reference's layouts
    back = F.unprocess_weights_for_mixed_gemm(p, tq, L.LAYOUT_GFX950).numpy()
    assert (back == w).all()
    # spot-check the documented map: n-block of 16 columns, lane l = h*16 + c
    un = oracle.unpack_int4_packed_tensor_to_int8(w) if bits == 4 else w
    raw = p.numpy().view(np.uint8).ravel()
    ktile = 128 if bits == 4 else 64
    run = K * 16 * bits // 8
    for (k, n) in [(0, 0), (5, 17), (37, 40), (K - 28, 63), (63, 31), (K - 1, N - 1), (K - 64, 3)]:
        nb, c = n // 16, n % 16
        kt, r = k // ktile, k % ktile
        s, h, j = r // 32, (r % 32) // 8, r % 8
        lane = h * 16 + c
        if bits == 4:
            half = kt == K // 128
            byte = nb * run + kt * 1024 + (lane * 8 if half else lane * 16) + s * 4
            pos = 2 * j if j < 4 else 2 * (j - 4) + 1  # rows 0..3 -> low nibbles, rows 4..7 -> high nibbles
            b = int(raw[byte + pos // 2])
            v = ((b >> 4) if pos & 1 else (b & 0xF)) - 8  # stored biased: u = q + 8
        else:
            v = int(raw[nb * run + kt * 1024 + lane * 16 + (s * 2 + j // 4) * 4 + j % 4])
            v = v - 256 if v >= 128 else v
        assert v == un[k, n]


def test_layout_shape_checks_raise():
    w = torch.zeros(48, 32, dtype=torch.int8)  # K = 48 not a multiple of 64
    with pytest.raises(L.Mi355qError):
        F.preprocess_weights_for_mixed_gemm(w, torch.int8, L.LAYOUT_REF_SM80)
    with pytest.raises(L.Mi355qError):
        F.preprocess_weights_for_mixed_gemm(w, torch.int8, L.LAYOUT_GFX950)


def test_pack_unpack_vs_oracle_and_errors():
    rng = np.random.default_rng(0)
    packed = rng.integers(-128, 128, size=(7, 33), dtype=np.int8)
    un = F.unpack_int4_packed_tensor_to_int8(torch.from_numpy(packed))
    assert (un.numpy() == oracle.unpack_int4_packed_tensor_to_int8(packed)).all()
    assert (F.pack_int8_tensor_to_packed_int4(un).numpy() == packed).all()
    with pytest.raises(L.Mi355qError, match="not in int4 range"):
        F.pack_int8_tensor_to_packed_int4(torch.tensor([[9, 0]], dtype=torch.int8))


@pytest.mark.parametrize("tq,qt", [(torch.int8, oracle.INT8_W), (torch.quint4x2, oracle.INT4_W)])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_symmetric_quantize_vs_oracle(tq, qt, dtype):
    torch.manual_seed(0)
    w = (torch.rand(128, 64) * 2 - 1).to(dtype)  # woq_gen_weights, _utils.py:44-48
    un, proc, scales = F._symmetric_quantize_last_axis_of_batched_matrix(w, tq, L.LAYOUT_REF_SM80)
    q_ref, s_ref = oracle.symmetric_quantize(w, qt)
    assert (un.numpy() == q_ref).all()
    assert torch.equal(scales, s_ref) and scales.dtype == dtype
    assert (proc.numpy() == oracle.preprocess_weights_for_mixed_gemm(q_ref, qt, arch=80)).all()
    proc2, scales2 = F.symmetric_quantize_last_axis_of_batched_matrix(w, tq)
    assert (F.unprocess_weights_for_mixed_gemm(proc2, tq, L.LAYOUT_GFX950).numpy() == q_ref).all()


def test_groupwise_plugin_input_compaction_and_shapes():
    # weightOnlyGroupwiseQuantMatmulPlugin.cpp:138-144, 243-281
    for algo in range(8):
        p = Plugin("WeightOnlyGroupwiseQuantMatmul", [("type_id", L.BF16, L.FIELD_INT32),
                                                       ("quant_algo", algo, L.FIELD_INT32),
                                                       ("group_size", 128, L.FIELD_INT32)])
        shapes = [[3, 5, 256]]
        if algo & 4:
            shapes.append([1, 256])
        shapes += [[256, 128], [2, 512]]
        if algo & 2:
            shapes.append([2, 512])
        if algo & 1:
            shapes.append([1, 512])
        assert p.get_output_dimensions(0, shapes) == [3, 5, 512]  # N = weight.d[1] * 4
        with pytest.raises(L.Mi355qError):
            p.get_output_dimensions(0, shapes + [[1]])
        assert p.get_output_data_type(0, [L.BF16] * len(shapes)) == L.BF16
        blob = p.serialize()
        q = Plugin.deserialize("WeightOnlyGroupwiseQuantMatmul", blob)
        assert q.serialize() == blob and q.plugin_type == "WeightOnlyGroupwiseQuantMatmul"
        with pytest.raises(L.Mi355qError, match="Expected length"):
            Plugin.deserialize("WeightOnlyGroupwiseQuantMatmul", blob[:-1])
        assert p.clone().serialize() == blob


def test_groupwise_plugin_accepts_w4a8_and_rejects_fp32():
    # FP8_ALPHA (W4A8-AWQ) is built: one more input (alpha, float32) after the optional bias (:138-146, :293-296)
    p = Plugin("WeightOnlyGroupwiseQuantMatmul", [("type_id", L.HALF, L.FIELD_INT32), ("quant_algo", 8 + 4 + 1, L.FIELD_INT32),
                                                   ("group_size", 128, L.FIELD_INT32)])
    descs = [L.TensorDesc() for _ in range(7)]
    for d_ in descs:
        d_.type, d_.format = L.HALF, 0
    descs[5].type = L.FLOAT  # activation, pre_quant_scale, weights, scales, bias, ALPHA, output
    arr = (L.TensorDesc * 7)(*descs)
    assert p.supports_format_combination(5, arr, 6, 1)
    descs[5].type = L.HALF
    arr = (L.TensorDesc * 7)(*descs)
    assert not p.supports_format_combination(5, arr, 6, 1)
    with pytest.raises(L.Mi355qError, match="Unsupported data type"):
        Plugin("WeightOnlyGroupwiseQuantMatmul", [("type_id", L.FLOAT, L.FIELD_INT32), ("quant_algo", 0, L.FIELD_INT32),
                                                   ("group_size", 128, L.FIELD_INT32)])
    with pytest.raises(L.Mi355qError, match="missing"):
        Plugin("WeightOnlyGroupwiseQuantMatmul", [("type_id", L.HALF, L.FIELD_INT32)])
    with pytest.raises(L.Mi355qError, match="unknown plugin"):
        Plugin("NoSuchPlugin")


def test_woq_and_sq_plugin_shapes_and_formats():
    p = Plugin("WeightOnlyQuantMatmul", [("type_id", L.HALF, L.FIELD_INT32), ("weight_type_id", 2, L.FIELD_INT32)])
    assert p.get_output_dimensions(0, [[4, 256], [256, 64], [128]]) == [4, 128]  # int4: n = d[1] * 2
    p8 = Plugin("WeightOnlyQuantMatmul", [("type_id", L.HALF, L.FIELD_INT32), ("weight_type_id", 1, L.FIELD_INT32)])
    assert p8.get_output_dimensions(0, [[4, 256], [256, 64], [64]]) == [4, 64]
    descs = Plugin._descs([torch.empty(4, 256, dtype=torch.float16), torch.empty(256, 64, dtype=torch.int8),
                           torch.empty(128, dtype=torch.float16), torch.empty(4, 128, dtype=torch.float16)])
    assert all(p.supports_format_combination(i, descs, 3, 1) for i in range(4))
    descs[1].type = L.HALF
    assert not p.supports_format_combination(1, descs, 3, 1)

    sq = Plugin("SmoothQuantGemm", [("has_per_channel_scaling", 1, L.FIELD_INT32),
                                    ("has_per_token_scaling", 1, L.FIELD_INT32), ("type_id", L.HALF, L.FIELD_INT32)])
    assert sq.get_output_dimensions(0, [[2, 16, 768], [2304, 768], [32, 1], [1, 2304]]) == [2, 16, 2304]
    assert sq.get_output_data_type(0, [L.INT8, L.INT8, L.FLOAT, L.FLOAT]) == L.HALF
    blob = sq.serialize()
    assert np.frombuffer(blob[:8], np.int32).tolist() == [(1 << 3) | (1 << 4), L.HALF]  # QuantMode bits


def test_attention_plugin_entry_indexing_and_errors():
    plg = F.make_gpt_attention_plugin(num_heads=8, num_kv_heads=2, head_size=64, dtype="float16",
                                      kv_cache_quant_mode=1 << 6, tokens_per_block=16, qkv_bias_enabled=True)
    assert plg.nb_outputs == 1
    B = 3
    shapes = [[B, 1, (8 + 4) * 64], [B], [B], [1], [1], [B], [B, 1, 64], [B], [1, B, 2, 4], [1, B, 2, 4], [1], [1],
              [(8 + 4) * 64]]
    assert plg.get_output_dimensions(0, shapes) == [B, 1, 8 * 64]
    blob = plg.serialize()
    assert Plugin.deserialize("GPTAttention", blob).serialize() == blob
    # context + generation requests in one padded batch are refused like gptAttentionPlugin.cpp:363-367
    # (host-only check: fails before anything is launched)
    qkv = torch.zeros(B, 1, 12 * 64, dtype=torch.float16)
    i32 = lambda *v: torch.tensor(v, dtype=torch.int32)
    with pytest.raises(L.Mi355qError, match="mixed requests require remove_input_padding"):
        plg.enqueue([qkv, i32(5, 5, 5), i32(4, 4, 4), i32(64), i32(0), i32(5, 5, 5),
                     torch.zeros(B, 1, 64, dtype=torch.int32), i32(0, 1, 1),
                     torch.zeros(1, B, 2, 4, dtype=torch.int64), torch.zeros(1, B, 2, 4, dtype=torch.int64),
                     torch.ones(1), torch.ones(1), torch.zeros(12 * 64, dtype=torch.float16)],
                    [torch.zeros(B, 1, 8 * 64, dtype=torch.float16)])


def test_feeder_plugins_shapes_formats_and_serialization():
    # RmsnormQuantization / LayernormQuantization / QuantizeTensor (host-only: no launch)
    rms = Plugin("RmsnormQuantization", [("eps", 1e-6, L.FIELD_FLOAT32), ("dyn_act_scaling", 1, L.FIELD_INT32),
                                         ("type_id", L.HALF, L.FIELD_INT32)])
    assert rms.nb_outputs == 2
    shapes = [[2, 5, 10, 10], [10], [10], [1]]
    assert rms.get_output_dimensions(0, shapes) == [2, 5, 10, 10]
    assert rms.get_output_dimensions(1, shapes) == [2, 5, 10, 1]
    assert rms.get_output_data_type(0, [L.HALF, L.HALF, L.HALF, L.FLOAT]) == L.INT8
    assert rms.get_output_data_type(1, [L.HALF, L.HALF, L.HALF, L.FLOAT]) == L.FLOAT
    blob = rms.serialize()  # float eps, bool dyn, int32 type (rmsnormQuantizationPlugin.cpp:205-217)
    assert len(blob) == 9 and np.frombuffer(blob[:4], np.float32)[0] == np.float32(1e-6) and blob[4] == 1
    assert Plugin.deserialize("RmsnormQuantization", blob).serialize() == blob
    ln = Plugin("LayernormQuantization", [("eps", 1e-5, L.FIELD_FLOAT32), ("use_diff_of_squares", 0, L.FIELD_INT32),
                                          ("dyn_act_scaling", 0, L.FIELD_INT32), ("type_id", L.FLOAT, L.FIELD_INT32)])
    assert ln.nb_outputs == 1 and len(ln.serialize()) == 10
    assert Plugin.deserialize("LayernormQuantization", ln.serialize()).serialize() == ln.serialize()
    descs = Plugin._descs([torch.empty(4, 64), torch.empty(64), torch.empty(64), torch.empty(1),
                           torch.empty(4, 64, dtype=torch.int8)])
    assert all(ln.supports_format_combination(i, descs, 4, 1) for i in range(5))
    descs[1].type = L.HALF
    assert not ln.supports_format_combination(1, descs, 4, 1)
    with pytest.raises(L.Mi355qError, match="unsupported data type"):
        Plugin("RmsnormQuantization", [("dyn_act_scaling", 0, L.FIELD_INT32), ("type_id", L.INT8, L.FIELD_INT32)])
    qt = Plugin("QuantizeTensor")
    assert qt.nb_outputs == 1 and qt.get_output_dimensions(0, [[3, 8], [1, 1]]) == [3, 8]
    assert qt.get_output_data_type(0, [L.HALF, L.FLOAT]) == L.INT8 and qt.serialize() == b""


def test_allreduce_plugin_fields_serialization_and_deserialize_params():
    plg = Plugin("AllReduce", [("group", [0, 1, 2, 3], L.FIELD_INT32), ("type_id", L.HALF, L.FIELD_INT32),
                               ("strategy", L.AR_ONESHOT, L.FIELD_INT8), ("counter", 7, L.FIELD_INT32)])
    blob = plg.serialize()  # int32 type, int8 strategy, int32 counter, int32 group[...] (allreducePlugin.cpp:268-284)
    assert len(blob) == 4 + 1 + 4 + 16 and blob[4] == L.AR_ONESHOT
    assert np.frombuffer(blob[9:], np.int32).tolist() == [0, 1, 2, 3]
    assert Plugin.deserialize("AllReduce", blob).serialize() == blob
    assert plg.get_output_dimensions(0, [[5, 8192], [16]]) == [5, 8192]
    # AllGather (allgatherPlugin.cpp:36-54, :66-72, :147-160): type then the group; dim 0 times the group size
    ag = Plugin("AllGather", [("group", [0, 1, 2, 3], L.FIELD_INT32), ("type_id", L.HALF, L.FIELD_INT32)])
    assert ag.get_output_dimensions(0, [[6, 5, 7], [18]]) == [24, 5, 7]
    agblob = ag.serialize()
    assert len(agblob) == 4 + 4 * 4 and Plugin.deserialize("AllGather", agblob).serialize() == agblob
    with pytest.raises(Exception):
        Plugin("AllGather", [("type_id", L.HALF, L.FIELD_INT32)])
    ring = Plugin("AllReduce", [("group", [0, 1], L.FIELD_INT32), ("type_id", L.HALF, L.FIELD_INT32),
                                ("strategy", L.AR_RING, L.FIELD_INT8), ("counter", 0, L.FIELD_INT32)])
    # RING = ncclAllReduce on the group's communicator (allreducePlugin.cpp:181-185); without one the plugin says which
    # call is missing instead of running something else.  Same for the reference's one-input AllGather.
    with pytest.raises(L.Mi355qError, match="mi355q_comm_init_rank"):
        ring.enqueue([torch.zeros(8, dtype=torch.float16)], [torch.zeros(8, dtype=torch.float16)])
    with pytest.raises(L.Mi355qError, match="mi355q_comm_init_rank"):
        Plugin.deserialize("AllGather", agblob).enqueue([torch.zeros(8, dtype=torch.float16)],
                                                        [torch.zeros(32, dtype=torch.float16)])
    grp = (C.c_int32 * 2)(0, 1)
    assert L.lib().mi355q_comm_exists(grp, 2) == 0
    L.check(L.lib().mi355q_comm_destroy(grp, 2))          # nothing to destroy: not an error (terminate, :254-267)
    assert L.lib().mi355q_comm_exists((C.c_int32 * 2)(1, 1), 2) < 0   # a rank twice
    # AllReduceParams::deserialize: even flags name the ping buffers, odd the pong buffers; both sets travel
    tp = 4
    table = (C.c_void_p * (4 * tp))(*[0x1000 * (i + 1) for i in range(4 * tp)])
    prm = L.AllReduceParams()
    L.check(L.lib().mi355q_allreduce_params_deserialize(C.byref(prm), table, tp, 2, 6))
    assert [prm.peer_comm_buffer_ptrs[i] for i in range(tp)] == [0x1000 * (i + 1) for i in range(tp)]
    assert prm.peer_barrier_ptrs_in[0] == 0x1000 * (2 * tp + 1) and prm.peer_barrier_ptrs_out[3] == 0x1000 * (4 * tp)
    assert (prm.ranks_per_node, prm.rank, prm.local_rank, prm.barrier_flag) == (4, 2, 2, 6)
    L.check(L.lib().mi355q_allreduce_params_deserialize(C.byref(prm), table, tp, 1, 7))
    assert prm.peer_comm_buffer_ptrs[0] == 0x1000 * (tp + 1)
    assert [prm.peer_comm_buffer_ptrs_ping[i] for i in range(tp)] == [0x1000 * (i + 1) for i in range(tp)]
    assert [prm.peer_comm_buffer_ptrs_pong[i] for i in range(tp)] == [0x1000 * (tp + i + 1) for i in range(tp)]
    # flags per (phase, rank, workgroup) + the rank's epoch and arrival lines
    assert L.lib().mi355q_allreduce_barrier_bytes() == (2 * 8 * 256 + 64) * 4
    assert L.lib().mi355q_allreduce_max_workspace_size(2) == 16_000_000
    assert L.lib().mi355q_allreduce_max_workspace_size(8) == 8_000_000


def test_allreduce_strategy_selection_matches_reference():
    # plugins/ncclPlugin/allreducePlugin.cpp:111-139
    sel = L.lib().mi355q_allreduce_select_implementation
    ws = 16 * 1000 * 1000
    assert sel(ws + 1, 8, ws) == 0                      # RING above the workspace
    assert sel(10, 2, ws) == 1                          # <= 2 ranks: ONESHOT
    assert sel(999_999, 4, ws) == 1 and sel(1_000_000, 4, ws) == 2
    assert sel(499_999, 8, ws) == 1 and sel(500_000, 8, ws) == 2


def test_workspace_size_queries_are_pure_host():
    n = L.lib().mi355q_weight_only_gemm_workspace_size(16, 4096, 4096, 2)
    assert n >= 16 * 4096 * 2
    # the answer covers every m' <= m (an engine is built for a maximum m): monotonic in m, also between 16-steps
    ws = [L.lib().mi355q_weight_only_gemm_workspace_size(m, 2048, 8192, 2) for m in range(1, 70)]
    assert all(b >= a for a, b in zip(ws, ws[1:])) and ws[23] > ws[15]
    # 32 < m <= 64 at a cfg4 slice: the skinny GEMM's partial tiles (2 m-tiles x 28 n-tiles x 8 pieces of 32 x 128 fp32)
    # behind the smoothed activations
    act = lambda m, k: (m * k * 2 + 255) // 256 * 256
    assert L.lib().mi355q_weight_only_gemm_workspace_size(64, 3584, 8192, 2) >= act(64, 8192) + 2 * 28 * 8 * 32 * 128 * 4
    ws = [L.lib().mi355q_weight_only_gemm_workspace_size(m, 3584, 8192, 2) for m in range(30, 70)]
    assert all(b >= a for a, b in zip(ws, ws[1:]))
    p = L.DecodeAttentionParams()
    p.batch, p.num_heads, p.num_kv_heads, p.head_size = 64, 8, 1, 128
    p.max_blocks_per_seq, p.tokens_per_block, p.max_seq_len = 32, 64, 2048
    p.act_type, p.kv_type = L.BF16, L.INT8
    assert L.lib().mi355q_decode_attention_workspace_size(C.byref(p)) > 0
    p.tokens_per_block = 48  # not a power of two (kvCacheUtils.h:80-82)
    assert L.lib().mi355q_decode_attention_workspace_size(C.byref(p)) == 0
    assert b"power of 2" in L.lib().mi355q_last_error()
