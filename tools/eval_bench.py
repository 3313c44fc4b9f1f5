#!/usr/bin/env python
"""Evaluation embedding model timing at the shipped sizes (latent 256, 4 layers; DistilBERT-base widths): motions/s and
sentences/s of T2MContrastiveModel_SMPLX.encode_motion / encode_text on the device."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from motioncraft_amd.evaluation import NativeEvalEncoder
from motioncraft_amd.synthetic import eval_encoder_param_shapes, make_eval_encoder_state

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32          # BaseEvaluator.encode_motion batches 32 (base_evaluator.py:127)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 196
S = int(sys.argv[3]) if len(sys.argv) > 3 else 32
bert = dict(dim=768, n_layers=6, n_heads=12, hidden_dim=3072, max_position_embeddings=512, vocab_size=30522)
enc = NativeEvalEncoder(make_eval_encoder_state(eval_encoder_param_shapes(bert=bert), seed=0), bert=bert)
motion = torch.randn(B, T, 322, device='cuda')
lengths = torch.randint(40, T + 1, (B,), device='cuda', dtype=torch.int32)
ids = torch.randint(0, 30522, (B, S), device='cuda', dtype=torch.int32)
mask = torch.ones(B, S, device='cuda', dtype=torch.uint8)


def layer_flops(rows, seq, d, ff):
    return 2 * rows * (4 * d * d + 2 * d * ff) + 4 * rows * seq * d


fm = 2 * B * T * 322 * 256 + 4 * layer_flops(B * (T + 2), T + 2, 256, 1024)
ft = 6 * layer_flops(B * S, S, 768, 3072) + 2 * B * S * 768 * 256 + 4 * layer_flops(B * (S + 2), S + 2, 256, 1024)
for name, fn, fl in (('encode_motion', lambda: enc.encode_motion(motion, lengths), fm), ('encode_text', lambda: enc.encode_tokens(ids, mask), ft)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{name} B={B} (T={T} / S={S}): {ms:.3f} ms/batch -> {B / ms * 1e3:.0f} samples/s, {fl / ms / 1e9:.1f} TFLOP/s')

# HumanML3D / KIT evaluator (T2MContrastiveModel) at the shipped widths
from motioncraft_amd.evaluation import NativeT2MEvaluator
from motioncraft_amd.synthetic import make_t2m_eval_state, t2m_eval_param_shapes
t2m = NativeT2MEvaluator(make_t2m_eval_state(t2m_eval_param_shapes(), seed=0))
m263 = torch.randn(B, T, 263, device='cuda')
word = torch.randn(B, 22, 300, device='cuda')
pos = torch.nn.functional.one_hot(torch.randint(0, 15, (B, 22), device='cuda'), 15).float()
sent = torch.randint(3, 23, (B,), device='cuda', dtype=torch.int32)
for name, fn in (('t2m encode_motion', lambda: t2m.encode_motion(m263, lengths)), ('t2m encode_text', lambda: t2m.encode_word_vectors(word, pos, sent))):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{name} B={B}: {ms:.3f} ms/batch -> {B / ms * 1e3:.0f} samples/s')
