"""Pins the attention oracle to the reference's OWN oracle: the Hugging Face attention modules
(/root/reference/tests/attention/test_gpt_attention.py:25-33 builds its expected values from transformers'
LlamaAttention / GPT-J rotary helpers).  Run in the build container (transformers is importable there):

    python tests/golden/make_attention_hf.py        # writes tests/golden/attention_hf.npz

For every case the packed qkv rows (values of the activation dtype, stored as their 16-bit patterns) go through
transformers' LlamaAttention in fp32 with selection matrices as q/k/v projections and an identity output projection
(so the module's attention math -- rotary embedding, GQA repeat, causal softmax(QK^T / sqrt(d)) V -- is all that is
left), and the module's output is stored.  Row S-1 of a causal pass over S tokens is also the expected output of one
generation step over a cache of S-1 tokens.  tests/test_attention_hf_pin.py compares oracle.context_attention /
oracle.decode_attention (CPU) and the HIP kernels (GPU) with these arrays.  The GPT-J (interleaved-pair) rotary
convention is pinned by transformers.models.gptj.modeling_gptj.apply_rotary_pos_emb on a small tensor.
"""
import os

import numpy as np
import torch

CASES = [  # name, dtype, B, S, Hq, Hkv, Dh
    ("mha_f16", "float16", 2, 37, 4, 4, 128),
    ("gqa4_bf16", "bfloat16", 2, 70, 8, 2, 128),
    ("mqa8_f16", "float16", 1, 133, 8, 1, 128),
    ("mha_d64_bf16", "bfloat16", 2, 50, 4, 4, 64),
]


def hf_llama_attention(x, Hq, Hkv, D):
    """x: fp32 [B, S, (Hq + 2 Hkv) D] -> fp32 [B, S, Hq D] through transformers' LlamaAttention."""
    from transformers.models.llama.configuration_llama import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaAttention, LlamaRotaryEmbedding
    hidden = (Hq + 2 * Hkv) * D
    cfg = LlamaConfig(hidden_size=hidden, num_attention_heads=Hq, num_key_value_heads=Hkv, head_dim=D,
                      max_position_embeddings=4096, attention_bias=False, attention_dropout=0.0)
    cfg._attn_implementation = "eager"
    attn = LlamaAttention(cfg, layer_idx=0).float().eval()
    eye = torch.eye(hidden)
    with torch.no_grad():
        attn.q_proj.weight.copy_(eye[:Hq * D])
        attn.k_proj.weight.copy_(eye[Hq * D:(Hq + Hkv) * D])
        attn.v_proj.weight.copy_(eye[(Hq + Hkv) * D:])
        attn.o_proj.weight.zero_()
        attn.o_proj.weight[:Hq * D, :Hq * D] = torch.eye(Hq * D)   # output lands in the first Hq*D channels
    B, S, _ = x.shape
    pos = torch.arange(S).unsqueeze(0).expand(B, S)
    cos, sin = LlamaRotaryEmbedding(cfg)(x, pos)
    mask = torch.full((S, S), float("-inf")).triu(1)[None, None].expand(B, 1, S, S)
    with torch.no_grad():
        out = attn(x, position_embeddings=(cos, sin), attention_mask=mask)[0]
    return out[..., :Hq * D].contiguous()


def main():
    import transformers
    out = {"transformers_version": np.array(transformers.__version__)}
    for idx, (name, dts, B, S, Hq, Hkv, D) in enumerate(CASES):
        g = torch.Generator().manual_seed(1000 + idx)
        dt = getattr(torch, dts)
        x = (torch.randn(B, S, (Hq + 2 * Hkv) * D, generator=g) * 0.7).to(dt).float()
        out[name + "/qkv_bits"] = x.to(dt).view(torch.int16).numpy()   # the 16-bit patterns: half the bytes, exact
        out[name + "/out"] = hf_llama_attention(x, Hq, Hkv, D).numpy()
        out[name + "/shape"] = np.array([B, S, Hq, Hkv, D])
        out[name + "/dtype"] = np.array(dts)
    # GPT-J rotary (pairs 2i, 2i+1): transformers' helper on [B, S, H, rot]
    from transformers.models.gptj.modeling_gptj import apply_rotary_pos_emb, create_sinusoidal_positions
    rot, S = 64, 9
    g = torch.Generator().manual_seed(5)
    t = torch.randn(1, S, 3, rot, generator=g)
    sincos = create_sinusoidal_positions(64, rot)[:S].unsqueeze(0)          # [1, S, rot] = sin | cos halves
    sin, cos = torch.split(sincos, rot // 2, dim=-1)
    out["gptj_rope/x"] = t.numpy()
    out["gptj_rope/y"] = apply_rotary_pos_emb(t, sin, cos).numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attention_hf.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if k.endswith("/out")})


if __name__ == "__main__":
    main()
