"""CPU restatement of the reference WavEncoder (mogen/models/utils/blocks.py:11-71).  TEST INFRASTRUCTURE ONLY.

Pinned against the reference class itself (blocks.py imports nothing but torch/numpy) by
tests/golden/make_golden.py -> tests/golden/wav_encoder.npz.
"""
import torch
import torch.nn.functional as F

# (stride, padding of conv1 / downsample -- the reference passes it as ``first_dilation``, blocks.py:18-19,27-30,58-63)
BLOCKS = [(5, 1600, True), (6, 0, True), (1, 7, False), (6, 0, True), (1, 7, False), (3, 0, True)]


def _bn(x, p, pre):
    return F.batch_norm(x, p[pre + 'running_mean'], p[pre + 'running_var'], p[pre + 'weight'], p[pre + 'bias'],
                        training=False, eps=1e-5)


def basic_block(p, i, x):
    """BasicBlock.forward (blocks.py:42-54), eval mode."""
    stride, pad, down = BLOCKS[i]
    h = F.conv1d(x, p[f'{i}.conv1.weight'], p[f'{i}.conv1.bias'], stride=stride, padding=pad)
    h = F.leaky_relu(_bn(h, p, f'{i}.bn1.'), 0.01)
    h = _bn(F.conv1d(h, p[f'{i}.conv2.weight'], p[f'{i}.conv2.bias'], padding=7), p, f'{i}.bn2.')
    sc = x
    if down:
        sc = _bn(F.conv1d(x, p[f'{i}.downsample.0.weight'], p[f'{i}.downsample.0.bias'], stride=stride, padding=pad),
                 p, f'{i}.downsample.1.')
    return F.leaky_relu(h + sc, 0.01)


def wav_encoder(p, wav):
    """WavEncoder.forward (blocks.py:65-71): wav [B, samples] or [B, samples, audio_in] -> [B, frames, out_dim]."""
    x = wav.unsqueeze(1) if wav.dim() == 2 else wav.transpose(1, 2)
    for i in range(6):
        x = basic_block(p, i, x)
    return x.transpose(1, 2)
