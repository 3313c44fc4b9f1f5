"""CPU restatement of the reference's text condition encoder.  TEST INFRASTRUCTURE ONLY.

Stage A (``text_pre_proj`` -> ``nn.TransformerEncoder`` -> ``text_ln``; diffusion_transformer.py:109-141,153-158) is
pinned against the reference's own ``encode_text(text, clip_feat)`` by tests/golden/make_golden.py ->
tests/golden/text_encoder.npz.

Stage B (the CLIP ViT-B/32 text transformer the reference calls at diffusion_transformer.py:144-151) lives in the
un-vendored ``clip`` package (openai/CLIP, no version pin in the reference's requirements), so it cannot be pinned to the
reference's own dependency; it IS pinned (bit-equal, tests/golden/clip_tower_hf.npz) to the installed independent
implementation of the same published architecture, ``transformers.CLIPTextModel``.  The published architecture is restated here (clip/model.py: ``ResidualAttentionBlock`` = x + attn(ln_1(x), causal
mask); x + c_proj(QuickGELU(c_fc(ln_2(x)))); ``QuickGELU`` = x * sigmoid(1.702 x); ``ln_final``), built from the same
torch primitives (``F.multi_head_attention_forward``, ``F.layer_norm``) that stage A pins.
"""
import torch
import torch.nn.functional as F


def _mha(x, p, pre, heads, mask=None, in_w='in_proj_weight', in_b='in_proj_bias'):
    """nn.MultiheadAttention forward on seq-first x [S, B, d]."""
    d = x.shape[-1]
    out, _ = F.multi_head_attention_forward(
        x, x, x, d, heads, p[pre + in_w], p[pre + in_b], None, None, False, 0.0,
        p[pre + 'out_proj.weight'], p[pre + 'out_proj.bias'], training=False, need_weights=False, attn_mask=mask)
    return out


def finetune_encoder(p, clip_feat, num_layers, heads=4):
    """encode_text with clip_feat given (:152-165): [B, 77, 512] -> xf_out [B, 77, Dt]."""
    x = clip_feat.permute(1, 0, 2)
    if 'text_pre_proj.weight' in p:
        x = F.linear(x, p['text_pre_proj.weight'], p['text_pre_proj.bias'])
    d = x.shape[-1]
    for i in range(num_layers):                 # nn.TransformerEncoderLayer, norm_first=False, activation gelu
        pre = f'textTransEncoder.layers.{i}.'
        x = F.layer_norm(x + _mha(x, p, pre + 'self_attn.', heads), (d,), p[pre + 'norm1.weight'], p[pre + 'norm1.bias'])
        ff = F.linear(F.gelu(F.linear(x, p[pre + 'linear1.weight'], p[pre + 'linear1.bias'])),
                      p[pre + 'linear2.weight'], p[pre + 'linear2.bias'])
        x = F.layer_norm(x + ff, (d,), p[pre + 'norm2.weight'], p[pre + 'norm2.bias'])
    x = F.layer_norm(x, (d,), p['text_ln.weight'], p['text_ln.bias'])
    return x.permute(1, 0, 2)


def clip_text_features(p, tokens, layers, heads=8):
    """token ids [B, 77] -> ln_final(transformer(token_embedding + positional_embedding)) [B, 77, width]."""
    x = p['clip.token_embedding.weight'][tokens] + p['clip.positional_embedding']
    x = x.permute(1, 0, 2)
    S, d = x.shape[0], x.shape[-1]
    mask = torch.full((S, S), float('-inf')).triu_(1)
    for i in range(layers):
        pre = f'clip.transformer.resblocks.{i}.'
        h = F.layer_norm(x, (d,), p[pre + 'ln_1.weight'], p[pre + 'ln_1.bias'])
        x = x + _mha(h, p, pre + 'attn.', heads, mask)
        h = F.layer_norm(x, (d,), p[pre + 'ln_2.weight'], p[pre + 'ln_2.bias'])
        h = F.linear(h, p[pre + 'mlp.c_fc.weight'], p[pre + 'mlp.c_fc.bias'])
        x = x + F.linear(h * torch.sigmoid(1.702 * h), p[pre + 'mlp.c_proj.weight'], p[pre + 'mlp.c_proj.bias'])
    x = F.layer_norm(x, (d,), p['clip.ln_final.weight'], p['clip.ln_final.bias'])
    return x.permute(1, 0, 2)
