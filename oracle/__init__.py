"""CPU oracle for the quantized-inference hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  The product (``tensorrt-llm_amd/``) never imports it.

Two halves:
  * ``oracle.c`` (compiled to ``liboracle.so`` by ``oracle/Makefile``): the integer / bit-level
    restatement (int4 pack/unpack, SM80 layout pipeline, symmetric quantizer, single-rounding
    dequantisation, int32 GEMM accumulators, per-token quantisation).
  * this file: numpy/torch restatements of the reference's *test references* for floating-point
    outputs (tests/quantization/_utils.py, tests/attention/test_gpt_attention.py) and of the MMHA
    kernel's softmax conventions.

Pinning: see the header of oracle.c.  "parity unpinned" items are flagged where they are defined.
All reference paths are relative to /root/reference/.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# nvinfer1::DataType codes (also used by include/mi355q.h)
kFLOAT, kHALF, kINT8, kINT32, kFP8, kBF16 = 0, 1, 2, 3, 6, 7
INT8_W, INT4_W = 1, 2  # WeightTypeId, weightOnlyQuantMatmulPlugin.h:38-42

_c = ctypes
_vp, _sz, _i = _c.c_void_p, _c.c_size_t, _c.c_int


def build():
    """Compile liboracle.so (gcc); building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.orc_pack_int4.restype = _i
        _LIB.orc_preprocess_weights_for_mixed_gemm.restype = _i
    return _LIB


def _p(a):
    return _vp(a.ctypes.data)


def torch_dtype_code(dt):
    return {torch.float16: kHALF, torch.bfloat16: kBF16}[dt]


def _bits16(t):
    """torch fp16/bf16 tensor -> numpy uint16 view of its bits."""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def _from_bits16(a, dt):
    return torch.from_numpy(a.view(np.int16).copy()).view(dt)


# ----------------------------------------------------------------------------- int4 pack / unpack
def unpack_int4_packed_tensor_to_int8(packed):
    """thop/weightOnlyQuantOp.cpp:274-306.  numpy int8 [..., n] -> int8 [..., 2n]."""
    packed = np.ascontiguousarray(packed, dtype=np.int8)
    out = np.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), np.int8)
    lib().orc_unpack_int4(_p(packed), _p(out), _sz(packed.size))
    return out


def pack_int8_tensor_to_packed_int4(unpacked):
    """thop/weightOnlyQuantOp.cpp:308-343."""
    unpacked = np.ascontiguousarray(unpacked, dtype=np.int8)
    assert unpacked.shape[-1] % 2 == 0
    out = np.empty(unpacked.shape[:-1] + (unpacked.shape[-1] // 2,), np.int8)
    if lib().orc_pack_int4(_p(unpacked), _p(out), _sz(out.size)) != 0:
        raise ValueError("Value in unpacked tensor not in int4 range")
    return out


# ----------------------------------------------------------------------------- layout pipeline
def _bits(quant_type):
    return 8 if quant_type == INT8_W else 4


def permute_B_rows_for_mixed_gemm(w, rows, cols, quant_type):
    w = np.ascontiguousarray(w).view(np.uint8)
    out = np.empty_like(w)
    lib().orc_permute_B_rows(_p(w), _p(out), _sz(rows), _sz(cols), _i(_bits(quant_type)))
    return out


def subbyte_transpose(w, rows, cols, quant_type):
    w = np.ascontiguousarray(w).view(np.uint8)
    out = np.empty_like(w).reshape(cols, -1)
    lib().orc_subbyte_transpose(_p(w), _p(out), _sz(rows), _sz(cols), _i(_bits(quant_type)))
    return out


def interleave_column_major_tensor(w, rows, cols, quant_type, rows_per_tile=64, interleave=None):
    interleave = interleave or (2 if quant_type == INT8_W else 4)
    w = np.ascontiguousarray(w).view(np.uint8)
    out = np.empty_like(w)
    lib().orc_interleave_column_major(
        _p(w), _p(out), _sz(rows), _sz(cols), _i(_bits(quant_type)), _i(rows_per_tile), _i(interleave))
    return out


def add_bias_and_interleave_int8s(w):
    out = np.ascontiguousarray(w).view(np.uint8).copy()
    lib().orc_add_bias_and_interleave_int8(_p(out), _sz(out.size))
    return out


def add_bias_and_interleave_int4s(w):
    out = np.ascontiguousarray(w).view(np.uint8).copy()
    lib().orc_add_bias_and_interleave_int4(_p(out), _sz(out.size * 2))
    return out


def preprocess_weights_for_mixed_gemm(w, quant_type, arch=80):
    """cutlass_preprocessors.cpp:535-584.  w: int8 [K, N] (int8) or [K, N/2] (packed int4), or with
    a leading expert dim.  Returns an int8 array of the same shape (bytes re-laid out)."""
    w = np.ascontiguousarray(w, dtype=np.int8)
    experts = 1 if w.ndim == 2 else w.shape[0]
    rows = w.shape[-2]
    cols = w.shape[-1] * (8 // _bits(quant_type))
    out = np.empty_like(w)
    rc = lib().orc_preprocess_weights_for_mixed_gemm(
        _p(w), _p(out), _sz(experts), _sz(rows), _sz(cols), _i(quant_type), _i(arch))
    if rc != 0:
        raise ValueError("unsupported shape/arch for preprocess_weights_for_mixed_gemm")
    return out


def symmetric_quantize(weight, quant_type):
    """cutlass_preprocessors.cpp:621-729 / thop `_symmetric_quantize_last_axis_of_batched_matrix`.
    weight: torch [K, N] (or [E, K, N]) fp32/fp16/bf16.  Returns (unprocessed int8 numpy,
    scales torch in weight.dtype).  The processed layout is `preprocess_weights_for_mixed_gemm`."""
    wf = weight.float().contiguous().numpy()
    experts = 1 if wf.ndim == 2 else wf.shape[0]
    rows, cols = wf.shape[-2], wf.shape[-1]
    bits = _bits(quant_type)
    q = np.zeros(wf.shape[:-1] + (cols * bits // 8,), np.int8)
    scales = np.empty(wf.shape[:-2] + (cols,), np.float32)
    lib().orc_symmetric_quantize(_p(wf), _p(q), _p(scales), _sz(experts), _sz(rows), _sz(cols), _i(quant_type))
    return q, torch.from_numpy(scales).to(weight.dtype)


# ----------------------------------------------------------------------------- dequant + WOQ matmul
def dequant_fma(q, scale, zero, group_size):
    """Single-rounding dequantised weights (the bit-exact gate; oracle.c orc_dequant_fma).
    q: numpy int8 [K, N]; scale/zero: torch 16-bit [K/g, N] (or [N] per-channel).  -> torch [K, N]."""
    return _dequant("orc_dequant_fma", q, scale, zero, group_size)


def dequant_torch(q, scale, zero, group_size):
    """Two-rounding form used by the torch test reference
    (test_weight_only_groupwise_quant_matmul.py:190-195)."""
    return _dequant("orc_dequant_torch", q, scale, zero, group_size)


def _dequant(fn, q, scale, zero, group_size):
    q = np.ascontiguousarray(q, dtype=np.int8)
    K, N = q.shape
    dt = scale.dtype
    sc = _bits16(scale.reshape(-1, N))
    zr = _bits16(zero.reshape(-1, N)) if zero is not None and zero.numel() else None
    out = np.empty((K, N), np.uint16)
    getattr(lib(), fn)(_p(q), _p(sc), _p(zr) if zr is not None else None, _p(out), _sz(K), _sz(N),
                       _sz(group_size), _i(torch_dtype_code(dt)))
    return _from_bits16(out, dt).reshape(K, N)


def apply_per_channel_scale(act, pre_quant_scale):
    """kernels/preQuantScaleKernel.cu:26-90 == torch.mul in the activation dtype."""
    return (act * pre_quant_scale.reshape(1, -1).to(act.dtype))


def matmul_16(a, w):
    """fp64-accumulated A @ W on exactly widened 16-bit inputs -> torch fp32 [M, N]."""
    M, K = a.shape
    N = w.shape[1]
    out = np.empty((M, N), np.float32)
    lib().orc_matmul_16(_p(_bits16(a)), _p(_bits16(w)), _p(out), _sz(M), _sz(N), _sz(K),
                        _i(torch_dtype_code(a.dtype)))
    return torch.from_numpy(out)


def woq_groupwise_matmul(act, pre_quant_scale, q_unpacked, scale, zero, bias, group_size, dequant="fma", alpha=None,
                         act_fp8=False):
    """Reference result of WeightOnlyGroupwiseQuantMatmul (test_weight_only_groupwise_quant_matmul.py
    :186-210, _utils.py:63-67): (act * pre) @ (q*scale + zero) + bias, output in the act dtype.
    W4A8-AWQ (act_fp8, alpha; weightOnlyGroupwiseQuantMatmulPlugin.cpp:376-418, the test's :202-210): the smoothed
    activation is rounded through OCP e4m3 (apply_per_channel_scale<T, __nv_fp8_e4m3>: static_cast of the T-rounded
    product, saturating) and the accumulator is multiplied by alpha before the bias is added.
    Returns (out16, out_f32_before_cast)."""
    w = (dequant_fma if dequant == "fma" else dequant_torch)(q_unpacked, scale, zero, group_size)
    a = act if pre_quant_scale is None else apply_per_channel_scale(act, pre_quant_scale)
    if act_fp8:
        a = a.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(act.dtype)
    ref = matmul_16(a.reshape(-1, a.shape[-1]), w)
    if alpha is not None:
        ref = ref * np.float32(float(alpha))
    if bias is not None and bias.numel():
        ref = ref + bias.float().reshape(1, -1)
    return ref.to(act.dtype), ref


def woq_matmul(act, q_unpacked, scale):
    """WeightOnlyQuantMatmul per-channel reference (test_weight_only_quant_matmul.py:99-119,
    _utils.py:70-96): (act.float() @ q.float()) * scale, cast to the act dtype."""
    K, N = q_unpacked.shape
    qf = torch.from_numpy(np.ascontiguousarray(q_unpacked)).double()
    ref = (act.reshape(-1, K).double() @ qf).float() * scale.float().reshape(1, -1)
    return ref.to(act.dtype), ref


def moe_gemm_bias_act(a, q_unpacked, scales, biases, total_rows_before_expert, activation="identity"):
    """MoeGemmRunner::moeGemmBiasAct with weight-only experts, restated (moe_gemm_kernels.h:51-58,
    moe_gemm_kernels_template.h:401-470): rows of `a` are sorted by expert, expert e owns rows
    [total_rows_before_expert[e-1], total_rows_before_expert[e]); out = act(a_e @ dequant(q_e, scales_e) + bias_e) with
    the per-column dequantisation of the weight-only GEMM (dequant_fma), fp32 epilogue rounded once to T.
    activation: identity / relu / silu / gelu (cutlass GELU_taylor: 0.5 x (1 + tanh(0.79788456 x (1 + 0.044715 x^2)))).
    q_unpacked: numpy int8 [E, k, n]; scales [E, n]; biases [E, n] or None.  Returns (out16, out_f32)."""
    E, K, N = q_unpacked.shape
    out = torch.zeros(a.shape[0], N, dtype=torch.float32)
    r0 = 0
    for e in range(E):
        r1 = int(total_rows_before_expert[e])
        if r1 > r0:
            w = dequant_fma(q_unpacked[e], scales[e].reshape(1, N), None, K)
            ref = matmul_16(a[r0:r1], w)
            if biases is not None:
                ref = ref + biases[e].float().reshape(1, -1)
            out[r0:r1] = ref
        r0 = r1
    if activation == "relu":
        out = torch.relu(out)
    elif activation == "silu":
        out = out * torch.sigmoid(out)
    elif activation == "gelu":
        out = 0.5 * out * (1.0 + torch.tanh(0.7978845608028654 * out * (1.0 + 0.044715 * out * out)))
    return out.to(a.dtype), out


def moe_route(gating, k, finished=None, start_expert=0, experts_per_node=None):
    """topkGatingSoftmax / moeSoftmax + moeTopK (kernels/mixtureOfExperts/moe_kernels.cu:62-178): fp32 softmax over the
    experts, then k rounds of arg-max with the winner removed; equal values go to the lower index (cub::ArgMax).  Experts
    outside [start_expert, +experts_per_node) and finished rows get the sentinel index `experts_per_node`.
    Returns (scales fp32 [rows, k], local expert index int [rows, k])."""
    rows, E = gating.shape
    epn = E if experts_per_node is None else experts_per_node
    g = gating.float()
    p = torch.exp(g - g.max(dim=1, keepdim=True).values)
    p = p / p.sum(dim=1, keepdim=True)
    scales = torch.zeros(rows, k, dtype=torch.float32)
    idx = torch.zeros(rows, k, dtype=torch.int64)
    work = p.clone()
    for j in range(k):
        v, e = work.max(dim=1)  # torch.max returns the first maximal index
        scales[:, j] = v
        idx[:, j] = e
        work[torch.arange(rows), e] = -1.0
    local = idx - start_expert
    mine = (idx >= start_expert) & (idx < start_expert + epn)
    if finished is not None:
        mine &= ~finished.bool().reshape(-1, 1)
    local = torch.where(mine, local, torch.full_like(local, epn))
    return scales, local


def run_moe(x, gating, q1, s1, b1, q2, s2, b2, k, activation="identity", renormalize=False, finished=None,
            ep_size=1, ep_rank=0, tp_rank=0):
    """CutlassMoeFCRunner<T, uint8 | uint4>::runMoe (kernels/mixtureOfExperts/moe_kernels.cu:922-1035), restated:
      * routing (moe_route); the (token, j) pairs sorted by local expert, stably in pair order token * k + j (cub radix
        sort is stable, :499-558); total_rows_before_expert = inclusive counts (:566-578);
      * gather of the token rows (:593-623), expert GEMM 1 with bias and activation (identity + doGatedActivation for
        swiglu / geglu: fc1[:, :inter] * T(act(float(fc1[:, inter:]))), the product in T, :795-817), expert GEMM 2;
      * finalize (:646-717): out = sum_j scale_j * float(T(fc2_row + bias2[expert])), the running sum held in T;
        renormalize (k > 1): / sum_j scale_j; k == 1 with renormalize: scale 1; bias 2 only on tp rank 0 (:729-739).
    q1: numpy int8 [E_local, hidden, inter (x2 if gated)], s1 [E_local, .], q2: [E_local, inter, hidden].
    Returns (out in T, dict of intermediates)."""
    T = x.dtype
    rows, hidden = x.shape
    E = gating.shape[1]
    epn = E // ep_size
    scales, local = moe_route(gating, k, finished, epn * ep_rank, epn)
    keys = local.reshape(-1)                                  # pair i = token * k + j
    order = torch.sort(keys, stable=True).indices             # dest -> pair
    counts = torch.bincount(keys, minlength=epn + 1)[:epn]
    cum = torch.cumsum(counts, 0)
    num_valid = int(cum[-1])
    pair_of_dest = order[:num_valid]
    permuted = x[pair_of_dest // k]
    gated = activation in ("swiglu", "geglu")
    fc1, _ = moe_gemm_bias_act(permuted, q1, s1, b1, cum.tolist(), "identity" if gated else activation)
    if gated:
        inter = fc1.shape[1] // 2
        g = fc1[:, inter:].float()
        act = g * torch.sigmoid(g) if activation == "swiglu" else \
            0.5 * g * (1.0 + torch.tanh(0.7978845608028654 * g * (1.0 + 0.044715 * g * g)))
        fc1 = (fc1[:, :inter].float() * act.to(T).float()).to(T)
    fc2, _ = moe_gemm_bias_act(fc1, q2, s2, None, cum.tolist(), "identity")
    dest_of_pair = torch.empty_like(order)
    dest_of_pair[order] = torch.arange(order.numel())
    out = torch.zeros(rows, hidden, dtype=T)
    rescale = torch.zeros(rows, 1)
    mode = ("none" if k == 1 else "renorm") if renormalize else "scale"
    for j in range(k):
        pair = torch.arange(rows) * k + j
        dest = dest_of_pair[pair]
        sc = scales[:, j:j + 1] if mode != "none" else torch.ones(rows, 1)
        rescale += sc
        valid = dest < num_valid
        v = torch.zeros(rows, hidden, dtype=T)
        v[valid] = fc2[dest[valid]]
        if b2 is not None and tp_rank == 0:
            e = local[:, j].clamp(max=epn - 1)
            v = torch.where(valid.reshape(-1, 1), (v + b2[e].to(T)), v)
        term = (out.float() + sc * v.float()).to(T)
        out = torch.where(valid.reshape(-1, 1), term, out)
    if mode == "renorm":
        out = (out.float() / rescale).to(T)
    return out, dict(scales=scales, experts=local, total_rows=cum, dest_of_pair=dest_of_pair, fc2=fc2)


def woq_tolerance(ref, wtype, dtype=None):
    """_utils.py:99-109: atol = max|ref| * 2^-(bits-1) * 1.5 (x3 for bf16 in
    cpp/tests/kernels/weightOnly/weightOnlyKernelTest.cpp:316-328)."""
    bits = 8 if wtype == INT8_W else 4
    atol = ref.float().abs().max().item() / (1 << (bits - 1)) * 1.5
    return atol * (3.0 if dtype == torch.bfloat16 else 1.0)


# ----------------------------------------------------------------------------- SmoothQuant
def gemm_s8s8s32(a, b):
    """int32 accumulators of A[M,K] x B[N,K]^T (exact).  numpy int8 in, int32 out."""
    a = np.ascontiguousarray(a, dtype=np.int8)
    b = np.ascontiguousarray(b, dtype=np.int8)
    M, K = a.shape
    N = b.shape[0]
    out = np.empty((M, N), np.int32)
    lib().orc_gemm_s8s8s32(_p(a), _p(b), _p(out), _sz(M), _sz(N), _sz(K))
    return out


def smooth_quant_gemm(a, b, scale_tokens, scale_channels, per_token, per_channel, out_dtype):
    """tests/quantization/_utils.py:112-142 with the kernel's epilogue order acc * (s_col * s_row)
    (epilogue_per_row_per_col_scale.h:320-347).  out_dtype: torch.float16/float32/int32.
    Returns (out, int32 accumulators)."""
    acc = gemm_s8s8s32(a, b)
    M, N = acc.shape
    st = np.ascontiguousarray(scale_tokens.reshape(-1), np.float32)
    sc = np.ascontiguousarray(scale_channels.reshape(-1), np.float32)
    f = np.empty((M, N), np.float32)
    lib().orc_sq_epilogue(_p(acc), _p(st), _p(sc), _p(f), _sz(M), _sz(N), _i(int(per_token)), _i(int(per_channel)))
    t = torch.from_numpy(f)
    if out_dtype == torch.int32:
        t = torch.round(t).to(torch.int32)  # _utils.py:132-134 (rint == torch.round: half-to-even)
    else:
        t = t.to(out_dtype)
    return t, acc


def quantize_per_token(x):
    """kernels/quantization.cu:93-127, _utils.py:145-150.  x torch [M, K] fp16/bf16/fp32."""
    floor = torch.tensor(1e-6, dtype=x.dtype).float().item()
    xf = x.float().contiguous().numpy()
    M, K = xf.shape
    q = np.empty((M, K), np.int8)
    s = np.empty((M,), np.float32)
    lib().orc_quantize_per_token(_p(xf), _p(q), _p(s), _sz(M), _sz(K), _c.c_float(floor))
    return q, s.reshape(M, 1)


# ----------------------------------------------------------------------------- FP8 e4m3 scales
FP8_E4M3_MAX = 448.0


def fp8_scale(amax):
    """cpp/tensorrt_llm/common/cudaFp8Utils.cu:215-341: scale = max(amax / 448, 1 / (448 * 512))."""
    return np.maximum(np.asarray(amax, np.float32) / np.float32(FP8_E4M3_MAX),
                      np.float32(1.0 / (FP8_E4M3_MAX * 512.0)))


def quantize_e4m3(x, mode):
    """tests/quantization/test_fp8_quantization.py:37-120.  mode in {"activation" (per row),
    "weight" (per column), "per_tensor"}.  Returns (fp8 tensor, fp32 scale)."""
    xf = x.float()
    if mode == "activation":
        amax = xf.abs().amax(dim=-1, keepdim=True)
    elif mode == "weight":
        amax = xf.abs().amax(dim=0, keepdim=True)
    else:
        amax = xf.abs().amax().reshape(1, 1)
    scale = torch.from_numpy(fp8_scale(amax.numpy()))
    q = (xf / scale).to(torch.float8_e4m3fn)
    return q, scale


def fp8_gemm(a_fp8, b_fp8, scale_a, scale_b, out_dtype):
    """Semantic restated from quantization/layers.py:777-879 (Q/DQ + matmul; the arithmetic itself
    lives in closed libraries => PARITY UNPINNED): out = (a . b^T) * (s_a * s_b), fp32 accumulate.
    a_fp8 [M,K], b_fp8 [N,K]."""
    acc = a_fp8.float().double() @ b_fp8.float().double().t()
    return (acc.float() * (float(scale_a) * float(scale_b))).to(out_dtype), acc.float()


# ----------------------------------------------------------------------------- decode attention
def rope_coefficients(pos, rot_dim, base=10000.0, scale=1.0):
    """decoderMaskedMultiheadAttentionUtils.h:2581-2586: angle_i = pos*scale / base^(2i/rot_dim)."""
    i = np.arange(0, rot_dim, 2, dtype=np.float32)
    inv = np.float32(pos * scale) / np.power(np.float32(base), i / np.float32(rot_dim))
    return np.cos(inv).astype(np.float32), np.sin(inv).astype(np.float32)


def apply_rope(x, pos, rot_dim, kind, base=10000.0, scale=1.0):
    """x: float32 numpy [Dh] already widened from T; returns float32 (caller rounds to T like
    rotary_embedding_transform does, Utils.h:2596-2610).  kind: 'neox' pairs (i, i+rot/2)
    (Template.h:1651-1711), 'gptj' pairs (2i, 2i+1) (:1637-1650)."""
    c, s = rope_coefficients(pos, rot_dim, base, scale)
    y = x.copy()
    h = rot_dim // 2
    if kind == "neox":
        a, b = x[:h], x[h:rot_dim]
        y[:h] = c * a - s * b
        y[h:rot_dim] = c * b + s * a
    elif kind == "gptj":
        a, b = x[0:rot_dim:2], x[1:rot_dim:2]
        y[0:rot_dim:2] = c * a - s * b
        y[1:rot_dim:2] = c * b + s * a
    return y


class PagedKVCache:
    """KVBlockArray layout (kernels/kvCacheUtils.h:36-153): pool of blocks, each
    [2(K,V)][Hkv][tokens_per_block][Dh] like the test pool (test_gpt_attention.py:685-690); the
    table is [B, 2, max_blocks] of block *addresses* (here: element offsets into `pool`; the GPU
    harness turns them into device pointers)."""

    def __init__(self, batch, num_kv_heads, head_dim, max_seq, tokens_per_block, dtype, seed=42):
        assert tokens_per_block & (tokens_per_block - 1) == 0
        self.B, self.H, self.D, self.tpb = batch, num_kv_heads, head_dim, tokens_per_block
        self.max_blocks = (max_seq + tokens_per_block - 1) // tokens_per_block
        nblk = batch * self.max_blocks
        self.dtype = dtype
        g = torch.Generator().manual_seed(seed)
        perm = torch.randperm(nblk, generator=g)
        self.block_ids = perm.reshape(batch, self.max_blocks)  # same block id for K and V halves
        self.pool = torch.zeros(nblk, 2, num_kv_heads, tokens_per_block, head_dim, dtype=dtype)

    def block_elems(self):
        return self.H * self.tpb * self.D

    def offsets(self):
        """[B, 2, max_blocks] int64 element offsets of each K / V block inside pool.view(-1)."""
        be = self.block_elems()
        k = self.block_ids * (2 * be)
        return torch.stack([k, k + be], dim=1).contiguous()

    def write(self, b, kv, head, tok, vec):
        self.pool[self.block_ids[b, tok // self.tpb], kv, head, tok % self.tpb] = vec

    def gather(self, b, kv, head, length):
        nb = (length + self.tpb - 1) // self.tpb
        x = self.pool[self.block_ids[b, :nb], kv, head]  # [nb, tpb, D]
        return x.reshape(nb * self.tpb, self.D)[:length]


def quantize_kv(x, kv_dtype, scale_orig_quant):
    """Cache-write quantisation.  int8: cvt.rni.sat.s8.f32(x * s)
    (decoderMaskedMultiheadAttentionUtils.h:3629-3639, 3742-3749); fp8: e4m3 RNE of x * s."""
    xf = x.float() * scale_orig_quant
    if kv_dtype == torch.int8:
        return torch.clamp(torch.round(xf), -128, 127).to(torch.int8)
    if kv_dtype == torch.float8_e4m3fn:
        return xf.to(torch.float8_e4m3fn)
    return x.to(kv_dtype)


def relative_bucket_mmha(relative_position, num_buckets, max_distance):
    """The bucket of a relative position as the masked-MHA kernel computes it on the fly
    (decoderMaskedMultiheadAttentionTemplate.h:1960-1977, IMPLICIT_REL_ATTN_BIAS), fp32 like the kernel."""
    rb = 0
    nb = num_buckets // 2
    if relative_position > 0:
        rb += nb
    rp = abs(relative_position)
    max_exact = nb // 2
    if rp < max_exact:
        return rb + rp
    large = max_exact + int(np.float32(np.log(np.float32(rp) / np.float32(max_exact)))
                            / np.float32(np.log(np.float32(max_distance) / np.float32(max_exact))) * np.float32(nb - max_exact))
    return rb + min(large, nb - 1)


def relative_bucket_unidirectional(relative_position, num_buckets, max_distance):
    """The bucket of key position - query position as the context phase computes it (addRelativeAttentionBiasUnaligned
    with bidirectional = false, unfusedAttentionKernels.cu:1703-1727): keys after the query share bucket 0, num_buckets
    is not halved; fp32 like the kernel."""
    rp = max(-relative_position, 0)
    max_exact = num_buckets // 2
    if rp < max_exact:
        return rp
    large = max_exact + int(np.float32(np.log(np.float32(rp) / np.float32(max_exact)))
                            / np.float32(np.log(np.float32(max_distance) / np.float32(max_exact)))
                            * np.float32(num_buckets - max_exact))
    return min(large, num_buckets - 1)


def decode_attention(qkv, cache, seq_lens, num_heads, num_kv_heads, head_dim, q_scaling=1.0, rope=None,
                     rot_dim=0, rope_base=10000.0, rope_scale=1.0, qkv_bias=None, kv_scale_orig_quant=1.0,
                     kv_scale_quant_orig=1.0, update_cache=True, alibi_slopes=None, relative_attention_bias=None,
                     max_distance=0):
    """One generation step of GPTAttention over a paged cache, restating the MMHA math
    (decoderMaskedMultiheadAttentionTemplate.h; conventions pinned by test_gpt_attention.py:1018,1154):
      * seq_lens[b] INCLUDES the current token; tlength = seq_lens[b]-1 cached tokens (:1480-1482)
      * q = qkv[:, :Hq*Dh], k = next Hkv*Dh, v = last Hkv*Dh (gptAttentionCommon.cpp:271-306)
      * bias add, then RoPE at position tlength, results rounded to T (:1602-1711)
      * logits = (q.k) * 1/(sqrt(Dh)*q_scaling) in fp32 (:1797, :2026); ALiBi (alibi_slopes [Hq] in T, :1891-1895):
        + slope[h] * (t - tlength) (:2026)
      * relative attention bias (position embedding kRELATIVE; T values): logit += bias, after the scaling (:1797,
        :2026).  max_distance == 0: the explicit tensor [Hq, S, S], bias[h][tlength][t] (:1775-1790, :1982-1985);
        max_distance > 0: the table [Hq, num_buckets] and the bucket of (t - tlength) computed as the kernel does
        (:1960-1977: num_buckets halved, max_exact = half of that, logarithmic beyond)
      * p = exp(l - max); out = sum p v / (sum p + 1e-6)  (:2259-2290)
      * the current token uses its un-quantised k/v (registers), cached tokens are dequantised as
        float(k8) * kv_scale_quant_orig (:1999-2003); new k/v are appended at position tlength.
    fp32 everywhere else (the reference rounds P to T before P.V and reduces through T -- tolerance
    parity only, test atol 2e-3, test_gpt_attention.py:1290-1293).
    qkv: torch [B, (Hq+2Hkv)*Dh] in T.  Returns out torch [B, Hq*Dh] in T."""
    T = qkv.dtype
    B = qkv.shape[0]
    Hq, Hkv, D = num_heads, num_kv_heads, head_dim
    inv_sqrt = 1.0 / (math.sqrt(D) * q_scaling)
    x = qkv.float()
    if qkv_bias is not None:
        x = (qkv + qkv_bias.reshape(1, -1).to(T)).float()  # add in T (Template.h:1602-1606)
    out = torch.zeros(B, Hq * D, dtype=torch.float32)
    grp = Hq // Hkv
    for b in range(B):
        tl = int(seq_lens[b]) - 1
        q = x[b, :Hq * D].reshape(Hq, D).numpy().copy()
        k = x[b, Hq * D:(Hq + Hkv) * D].reshape(Hkv, D).numpy().copy()
        v = x[b, (Hq + Hkv) * D:].reshape(Hkv, D).numpy().copy()
        if rope is not None:
            for h in range(Hq):
                q[h] = apply_rope(q[h], tl, rot_dim, rope, rope_base, rope_scale)
            for h in range(Hkv):
                k[h] = apply_rope(k[h], tl, rot_dim, rope, rope_base, rope_scale)
            q = torch.from_numpy(q).to(T).float().numpy()
            k = torch.from_numpy(k).to(T).float().numpy()
        for hk in range(Hkv):
            kc = cache.gather(b, 0, hk, tl).float() * kv_scale_quant_orig
            vc = cache.gather(b, 1, hk, tl).float() * kv_scale_quant_orig
            kk = torch.cat([kc, torch.from_numpy(k[hk:hk + 1])], 0).double()
            vv = torch.cat([vc, torch.from_numpy(v[hk:hk + 1])], 0).double()
            for g in range(grp):
                h = hk * grp + g
                logits = (kk @ torch.from_numpy(q[h]).double()).float() * np.float32(inv_sqrt)
                if alibi_slopes is not None:
                    logits = logits + alibi_slopes[h].float() * (torch.arange(tl + 1) - tl).float()
                if relative_attention_bias is not None:
                    if max_distance > 0:
                        idx = torch.tensor([relative_bucket_mmha(t - tl, relative_attention_bias.shape[1], max_distance)
                                            for t in range(tl + 1)], dtype=torch.long)
                        logits = logits + relative_attention_bias[h, idx].float()
                    else:
                        logits = logits + relative_attention_bias[h, tl, :tl + 1].float()
                p = torch.exp(logits - logits.max())
                o = (p.double() @ vv).float() / (p.sum() + 1e-6)
                out[b, h * D:(h + 1) * D] = o
            if update_cache:
                cache.write(b, 0, hk, tl, quantize_kv(torch.from_numpy(k[hk]).to(T), cache.dtype, kv_scale_orig_quant))
                cache.write(b, 1, hk, tl, quantize_kv(torch.from_numpy(v[hk]).to(T), cache.dtype, kv_scale_orig_quant))
    return out.to(T)


class CyclicMap:
    """The slot map of KVBlockArray / KVLinearBuffer, restated from their constructors and getKVTokenIdx
    (kernels/kvCacheUtils.h:71-94, 109-117 paged; :277-295, 308-316 linear)."""

    def __init__(self, window, sink, tokens_per_block=None, max_blocks_per_seq=None, linear=False):
        self.window, self.sink = window, sink
        if linear:
            self.bubble = 0                                   # :292
            self.one_more = False                             # :294
            self.cyc_len = window - sink                      # :291
        else:
            in_last = sink % tokens_per_block                 # :89
            self.bubble = 0 if in_last == 0 else tokens_per_block - in_last                 # :90
            self.one_more = (max_blocks_per_seq - 1) * tokens_per_block >= window + self.bubble  # :91
            self.cyc_len = window + tokens_per_block - sink if self.one_more else window - sink  # :92-93

    def slot(self, t):
        if t < self.sink:                                     # isSinkToken, :96-99
            return t
        return self.sink + self.bubble + (t - self.sink) % self.cyc_len  # :114


def decode_attention_slots(qkv, slots, seq_lens, num_heads, num_kv_heads, head_dim, cmap, q_scaling=1.0, rope=None,
                           rot_dim=0, rope_base=10000.0, rope_scale=1.0, kv_scale_orig_quant=1.0,
                           kv_scale_quant_orig=1.0, beam_width=1, cache_indir=None, input_lengths=None, pos_shift=False):
    """decode_attention over an explicit slot space, with the reference's cyclic / sink addressing and beams
    (decoderMaskedMultiheadAttentionTemplate.h, POS_SHIFT = false):
      * slots: torch [batch_beam, 2, Hkv, capacity, Dh] in the cache dtype -- slot s of a sequence, however the
        cache stores it (paged or linear); updated in place with the new token;
      * tlength = seq_lens[b] - 1 (:1480-1482); the step reads slots [0, kv_loop_length), kv_loop_length =
        min(tlength, cyclic_attention_window_size) (:1493, the K loop :1900-1935 and :2056-2080 without POS_SHIFT
        index the cache by time_now itself);
      * the new K / V go to slot getKVTokenIdx(tlength) (:1485, :2221-2223, :2478-2480), AFTER the reads (:2207, :2469):
        a wrapped cyclic cache still shows this step the token it is about to overwrite;
      * RoPE position is tlength (:1500, POS_SHIFT false);
      * beams: batch_idx = b // beam_width (:1611); slot t < beam0_context_length is read from row
        batch_idx * beam_width (:1929), later slots from row batch_idx * beam_width + cache_indir[b][t] (:2076, :2441);
        beam0_context_length = 0 once tlength > window, else input_lengths[b] (:1497-1498).
    pos_shift (POS_SHIFT = true; StreamingLLM): the K cache holds keys WITHOUT position embedding.
      * invokeShiftKCache (unfusedAttentionKernels.cu:1776-1947) rotates, before the step, the key at window position x
        -- token_idx = x for the sink tokens, else cyclic_k_cache_start_idx + x - sink with the start = sink
        (tlength <= window) or sink + tlength - window (:1790-1795) -- to position x (token_pos_idx, :1797) and stores
        it in T (:1938-1941); the attention reads those keys (the same tokens: Template.h:1918-1927) and V from the
        tokens' cache slots (:2386-2395);
      * q and the new k are rotated to current_pos_idx = kv_loop_length (:1500); the cache receives k_wo_pos, the new
        key with bias and without rotation (:1745-1752, 2216-2233).
    Same arithmetic as decode_attention otherwise.  Returns out [batch_beam, Hq * Dh] in T."""
    assert not (pos_shift and (beam_width > 1 or rope is None))
    T = qkv.dtype
    B = qkv.shape[0]
    Hq, Hkv, D = num_heads, num_kv_heads, head_dim
    inv_sqrt = 1.0 / (math.sqrt(D) * q_scaling)
    x = qkv.float()
    out = torch.zeros(B, Hq * D, dtype=torch.float32)
    grp = Hq // Hkv
    writes = []
    for b in range(B):
        tl = int(seq_lens[b]) - 1
        kv_len = min(tl, cmap.window)
        q = x[b, :Hq * D].reshape(Hq, D).numpy().copy()
        k = x[b, Hq * D:(Hq + Hkv) * D].reshape(Hkv, D).numpy().copy()
        v = x[b, (Hq + Hkv) * D:].reshape(Hkv, D).numpy().copy()
        k_raw = k.copy()
        if rope is not None:
            pos = kv_len if pos_shift else tl
            for h in range(Hq):
                q[h] = apply_rope(q[h], pos, rot_dim, rope, rope_base, rope_scale)
            for h in range(Hkv):
                k[h] = apply_rope(k[h], pos, rot_dim, rope, rope_base, rope_scale)
            q = torch.from_numpy(q).to(T).float().numpy()
            k = torch.from_numpy(k).to(T).float().numpy()
        rows = torch.full((kv_len,), b, dtype=torch.long)
        if beam_width > 1:
            base = (b // beam_width) * beam_width
            ctx0 = 0 if tl > cmap.window else int(input_lengths[b])
            t = torch.arange(kv_len)
            rows = base + torch.where(t < ctx0, torch.zeros_like(t), cache_indir[b, :kv_len].long())
        for hk in range(Hkv):
            idx = torch.arange(kv_len)
            if pos_shift:
                start = cmap.sink if tl <= cmap.window else cmap.sink + tl - cmap.window
                idx = torch.tensor([cmap.slot(xx if xx < cmap.sink else start + xx - cmap.sink) for xx in range(kv_len)],
                                   dtype=torch.long)
            kc = slots[rows, 0, hk, idx].float() * kv_scale_quant_orig
            vc = slots[rows, 1, hk, idx].float() * kv_scale_quant_orig
            if pos_shift and kv_len:
                kc = torch.from_numpy(np.stack([apply_rope(kc[xx].numpy(), xx, rot_dim, rope, rope_base, rope_scale)
                                                for xx in range(kv_len)])).to(T).float()
            kk = torch.cat([kc, torch.from_numpy(k[hk:hk + 1])], 0).double()
            vv = torch.cat([vc, torch.from_numpy(v[hk:hk + 1])], 0).double()
            for g in range(grp):
                h = hk * grp + g
                logits = (kk @ torch.from_numpy(q[h]).double()).float() * np.float32(inv_sqrt)
                p = torch.exp(logits - logits.max())
                o = (p.double() @ vv).float() / (p.sum() + 1e-6)
                out[b, h * D:(h + 1) * D] = o
            writes.append((b, hk, cmap.slot(tl), quantize_kv(torch.from_numpy((k_raw if pos_shift else k)[hk]).to(T), slots.dtype, kv_scale_orig_quant),
                           quantize_kv(torch.from_numpy(v[hk]).to(T), slots.dtype, kv_scale_orig_quant)))
    for b, hk, s_, kq, vq in writes:  # all rows read before any is written (other beams read this row's slots)
        slots[b, 0, hk, s_] = kq
        slots[b, 1, hk, s_] = vq
    return out.to(T)


def decode_attention_c(qkv, cache, seq_lens, num_heads, num_kv_heads, head_dim, q_scaling=1.0, rope=None):
    """The same generation step through oracle.c's plain C / OpenMP loops (orc_decode_attention_16): 16-bit caches,
    NeoX RoPE over the whole head or none, no bias, cache not updated.  What bench.py's cpu_baseline leg times."""
    assert cache.dtype == qkv.dtype and rope in (None, "neox")
    B = qkv.shape[0]
    out = np.empty((B, num_heads * head_dim), np.float32)
    q16 = np.ascontiguousarray(_bits16(qkv.contiguous()))
    pool16 = np.ascontiguousarray(_bits16(cache.pool.contiguous().view(-1)))
    offs = np.ascontiguousarray(cache.offsets().numpy().astype(np.int64))
    lens = np.ascontiguousarray(np.asarray(seq_lens, np.int32))
    lib().orc_decode_attention_16(_p(q16), _p(pool16), _p(offs), _p(lens), _p(out), _i(B), _i(num_heads), _i(num_kv_heads),
                                  _i(head_dim), _i(cache.max_blocks), _i(cache.tpb), _i(torch_dtype_code(qkv.dtype)),
                                  _i(1 if rope == "neox" else 0), _c.c_float(q_scaling))
    return torch.from_numpy(out).to(qkv.dtype)


# ----------------------------------------------------------------------------- context attention
def apply_rope_rows(x, positions, rot_dim, kind, base=10000.0, scale=1.0):
    """apply_rope over rows: x float32 numpy [S, H, Dh], positions int [S] (same pairing / angle rule)."""
    i = np.arange(0, rot_dim, 2, dtype=np.float32)
    freq = np.power(np.float32(base), i / np.float32(rot_dim))
    ang = (positions.astype(np.float32) * np.float32(scale))[:, None] / freq[None, :]  # [S, rot/2]
    c, s = np.cos(ang).astype(np.float32)[:, None, :], np.sin(ang).astype(np.float32)[:, None, :]
    y = x.copy()
    h = rot_dim // 2
    if kind == "neox":
        a, b = x[..., :h], x[..., h:rot_dim]
        y[..., :h] = c * a - s * b
        y[..., h:rot_dim] = c * b + s * a
    elif kind == "gptj":
        a, b = x[..., 0:rot_dim:2], x[..., 1:rot_dim:2]
        y[..., 0:rot_dim:2] = c * a - s * b
        y[..., 1:rot_dim:2] = c * b + s * a
    return y


def context_attention(qkv_rows, context_lengths, num_heads, num_kv_heads, head_dim, cache=None, q_scaling=1.0,
                      rope=None, rot_dim=0, rope_base=10000.0, rope_scale=1.0, qkv_bias=None,
                      kv_scale_orig_quant=1.0, past_lens=None, window=None, cmap=None, alibi_slopes=None, pos_shift=False,
                      relative_attention_bias=None, max_distance=0):
    """The context phase of GPTAttention (gptAttentionCommon.cpp:641-1133 enqueueContext), restated:
      * per input token: bias add in T, RoPE at the token's position in its sequence, rounded to T
        (invokeApplyBiasRopeUpdateKVCache / invokeAddFusedQKVBiasTranspose, unfusedAttentionKernels.cu);
      * K / V of every token appended to the paged cache at that position, quantised for an 8-bit cache
        (same conversion as the generation kernel's, quantize_kv);
      * causal attention among the input tokens in fp32 over the T-rounded q / k / v -- the reference's
        unfused path (:1000-1120: cuBLAS QK^T, invokeMaskedSoftmax, cuBLAS PV); the un-quantised K / V are
        used, as in the reference's context FMHA.  Scale 1/(sqrt(Dh) q_scaling), normaliser sum + 1e-6
        like the MMHA kernel (the 1e-6 is below T's resolution).
      * past_lens (chunked context, use_paged_context_fmha, :825-864): the sequence already holds past tokens in the
        cache; the input tokens take positions past .. (RoPE and cache position, unfusedAttentionKernels_2_template.h:
        252-262) and attend to every earlier token through the (16-bit) cache;
      * window (cyclic_attention_window_size): key t is visible to the query at position a iff a - window <= t <= a
        (gptKernels.cu:174-177); cmap (CyclicMap): the append keeps the sink tokens and the last window - sink tokens, at
        their cyclic slots (unfusedAttentionKernels_2_template.h:336-340);
      * alibi_slopes [Hq] in T: score += slope[h] * (t - a) (unfusedAttentionKernels.cu:212-218);
      * pos_shift (position_shift_enabled): the attention is unchanged, the cache receives K without the rotation
        (k_to_cache = POS_SHIFT ? k_wo_pos : k, unfusedAttentionKernels_2_template.h:284-347);
      * relative_attention_bias (kRELATIVE; invokeAddRelativeAttentionBiasUnaligned, unfusedAttentionKernels.cu:1688-1751,
        on the scaled scores, gptAttentionCommon.cpp:816-817, 1006-1016): max_distance == 0: [Hq, S', S'] in T, entry
        [h][a][t]; max_distance > 0: the table [Hq, num_buckets] and relative_bucket_unidirectional(t - a).
    qkv_rows: list of torch [len_b, (Hq+2Hkv)*Dh] in T, one per sequence.  Returns a list of [len_b, Hq*Dh] in T.
    Tolerance parity like decode_attention (the reference rounds P to T before P.V).  No test of the reference covers
    chunked context, the sliding window or ALiBi in the context phase: those three are "parity unpinned"."""
    Hq, Hkv, D = num_heads, num_kv_heads, head_dim
    inv_sqrt = np.float32(1.0 / (math.sqrt(D) * q_scaling))
    grp = Hq // Hkv
    outs = []
    for b, rows in enumerate(qkv_rows):
        T = rows.dtype
        S = int(context_lengths[b])
        assert rows.shape[0] == S
        x = rows
        if qkv_bias is not None:
            x = x + qkv_bias.reshape(1, -1).to(T)
        x = x.float().numpy().reshape(S, Hq + 2 * Hkv, D).copy()
        past = 0 if past_lens is None else int(past_lens[b])
        kv_len = past + S
        pos = past + np.arange(S)
        k_wo_pos = torch.from_numpy(x[:, Hq:Hq + Hkv].copy())   # bias added, rounded to T, not rotated
        if rope is not None:
            x[:, :Hq + Hkv] = apply_rope_rows(x[:, :Hq + Hkv], pos, rot_dim, rope, rope_base, rope_scale)
            x = torch.from_numpy(x).to(T).float().numpy()
        q = torch.from_numpy(x[:, :Hq]).double()                 # [S, Hq, D]
        k = torch.from_numpy(x[:, Hq:Hq + Hkv]).double()         # [S, Hkv, D]
        v = torch.from_numpy(x[:, Hq + Hkv:]).double()
        if cache is not None:
            for hk in range(Hkv):
                for t in range(S):
                    a, slot = int(pos[t]), int(pos[t])
                    if cmap is not None:
                        if not (a < cmap.sink or a >= max(kv_len - cmap.window + cmap.sink, cmap.sink)):
                            continue
                        slot = cmap.slot(a)
                    k_to_cache = k_wo_pos[t, hk] if pos_shift else k[t, hk].float()
                    cache.write(b, 0, hk, slot, quantize_kv(k_to_cache.to(T), cache.dtype, kv_scale_orig_quant))
                    cache.write(b, 1, hk, slot, quantize_kv(v[t, hk].float().to(T), cache.dtype, kv_scale_orig_quant))
        if past > 0:  # every key of the sequence, as the cache holds it
            assert cache is not None and cache.dtype == T
            k = torch.stack([cache.gather(b, 0, hk, kv_len) for hk in range(Hkv)], 1).double()
            v = torch.stack([cache.gather(b, 1, hk, kv_len) for hk in range(Hkv)], 1).double()
        out = torch.zeros(S, Hq, D, dtype=torch.float32)
        tq = torch.from_numpy(pos)[:, None]
        tk = torch.arange(kv_len)[None, :]
        mask = tk <= tq
        if window is not None:
            mask &= tk >= tq - window
        for h in range(Hq):
            hk = h // grp
            logits = (q[:, h] @ k[:, hk].t()).float() * inv_sqrt
            if alibi_slopes is not None:
                logits = logits + alibi_slopes[h].float() * (tk - tq).float()
            if relative_attention_bias is not None:
                if max_distance > 0:
                    nb = relative_attention_bias.shape[1]
                    idx = torch.tensor([[relative_bucket_unidirectional(t - int(a), nb, max_distance) for t in range(kv_len)]
                                        for a in pos], dtype=torch.long)
                    logits = logits + relative_attention_bias[h][idx].float()
                else:
                    logits = logits + relative_attention_bias[h, past:kv_len, :kv_len].float()
            logits = logits.masked_fill(~mask, float("-inf"))
            p = torch.exp(logits - logits.max(dim=1, keepdim=True).values)
            out[:, h] = (p.double() @ v[:, hk]).float() / (p.sum(dim=1, keepdim=True) + 1e-6)
        outs.append(out.reshape(S, Hq * D).to(T))
    return outs


# ----------------------------------------------------------------------------- norm + quantisation feeders
def cast_int8_rni_sat(x):
    """cuda_cast<int8_t, float>: cvt.rni.sat.s8.f32 (common/cudaTypeUtils.cuh)."""
    return torch.clamp(torch.round(x.float()), -128, 127).to(torch.int8)


def quantize_tensor(x, scale):
    """kernels/quantization.cu:30-67 quantizedKernel: int8 = cvt.rni.sat(float(x) * scale)."""
    return cast_int8_rni_sat(x.float() * np.float32(scale))


def norm_quant(x, gamma, beta, eps, kind="rms", scale=None, dynamic=False, use_diff_of_squares=True):
    """generalRmsNorm (kernels/rmsnormKernels.cu:52-150) / generalLayerNorm (layernormKernels.cu:60-192), restated:
      statistics in fp32 (here: float64 sums rounded to fp32 -- the kernel's tree order is not part of the contract),
      val = T(((x [- mean]) * rstd) * gamma + beta), every product / sum in fp32, one rounding to T;
      dynamic: amax = max(|val|, T(1e-6)) per row, int8 = rni_sat(float(val) * (127 / amax)), scales = amax / 127;
      static:  int8 = rni_sat(float(val) * scale);  neither: returns val.
    x: torch [..., H] in T (float32 / float16 / bfloat16).  Returns (out, scales-or-None)."""
    T = x.dtype
    H = x.shape[-1]
    xf = x.float().reshape(-1, H)
    xd = xf.double()
    if kind == "rms":
        mean = None
        rstd = torch.rsqrt(((xd * xd).sum(-1, keepdim=True) / H).float() + np.float32(eps))
        cen = xf
    else:
        mean = (xd.sum(-1, keepdim=True) / H).float()
        if use_diff_of_squares:
            var = ((xd * xd).sum(-1, keepdim=True) / H).float() - mean * mean
        else:
            d = xd - mean.double()
            var = ((d * d).sum(-1, keepdim=True) / H).float()
        rstd = torch.rsqrt(var + np.float32(eps))
        cen = xf - mean
    val = (cen * rstd) * gamma.float().reshape(1, H)
    if beta is not None:
        val = val + beta.float().reshape(1, H)
    val = val.to(T)
    if dynamic:
        floor = torch.tensor(1e-6, dtype=torch.float32).to(T).float()
        amax = torch.maximum(val.float().abs().amax(-1, keepdim=True), floor)
        q = cast_int8_rni_sat(val.float() * (np.float32(127.0) / amax))
        return q.reshape(x.shape), (amax / np.float32(127.0)).reshape(*x.shape[:-1], 1)
    if scale is not None:
        return cast_int8_rni_sat(val.float() * np.float32(scale)).reshape(x.shape), None
    return val.reshape(x.shape), None


def gated_silu(act_in, gate_in, smoother=None, scale=None, dynamic=False):
    """SmoothQuantGatedMLP.forward (tensorrt_llm/quantization/layers.py:922-941) between fc / gate and proj, restated
    with one rounding to T per layer of the reference's graph: inter = T(silu(act_in)) (silu evaluated in fp32),
    y = T(inter * gate_in), y = T(y / smoother); then quantize_tensor (static scale: rni_sat(y * scale),
    quantization.cu:30-86), quantize_per_token (dynamic: amax = max(|y|, T(1e-6)), rni_sat(y * (127 / amax)), scales =
    amax / 127, quantization.cu:93-127), or y.  functional.swiglu (functional.py:736-752) is the same product with
    act_in / gate_in = the two halves of one tensor and no smoother.  Returns (out, scales-or-None)."""
    T = act_in.dtype
    a = act_in.float()
    y = (a / (1.0 + torch.exp(-a))).to(T)
    y = (y.float() * gate_in.float()).to(T)
    if smoother is not None:
        y = (y.float() / smoother.float()).to(T)
    if dynamic:
        floor = torch.tensor(1e-6, dtype=torch.float32).to(T).float()
        amax = torch.maximum(y.float().abs().amax(-1, keepdim=True), floor)
        return cast_int8_rni_sat(y.float() * (np.float32(127.0) / amax)), amax / np.float32(127.0)
    if scale is not None:
        return cast_int8_rni_sat(y.float() * np.float32(scale)), None
    return y, None


# ----------------------------------------------------------------------------- custom all-reduce
def _add_T(a, b):
    """One add in T arithmetic: fp32 add of the widened operands, one rounding to T (half2 / bfloat162 add)."""
    return (a.float() + b.float()).to(a.dtype)


def custom_all_reduce(inputs, rank, strategy="oneshot"):
    """What rank `rank` receives from the one-shot / two-shot kernels (customAllReduceKernels.cu:137-291):
    sums start from +0 and add the ranks' values in the order (owner + i) % N, every add rounded to T, where the
    owner is `rank` itself for one-shot and, for two-shot, the rank that owns the element's slice
    (slice r = elements [r * n/N, (r+1) * n/N))."""
    N = len(inputs)
    n = inputs[0].numel()
    flat = [x.reshape(-1) for x in inputs]

    def ordered_sum(owner, lo, hi):
        acc = torch.zeros(hi - lo, dtype=inputs[0].dtype)
        for i in range(N):
            acc = _add_T(acc, flat[(owner + i) % N][lo:hi])
        return acc
    if strategy == "oneshot":
        return ordered_sum(rank, 0, n).reshape(inputs[0].shape)
    per = n // N
    return torch.cat([ordered_sum(r, r * per, (r + 1) * per) for r in range(N)]).reshape(inputs[0].shape)
