"""Shared synthetic-input recipe for tests; mirrors tests/golden/make_golden.py."""
import os

import numpy as np
import torch

from oracle import weights as W

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
SMALL = W.default_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8)
FULL = W.default_dims()
SMALL_SEED = 2
CTRL = W.default_dims(max_seq_len=24, L=32, NL=3, F=64, Te=64, Dt=32, Nt=8)
CTRL_COPY, CTRL_FEATS, CTRL_TC = 2, 35, 20
HML_SMALL = W.humanml3d_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8)
KIT_SMALL = W.humanml3d_dims(max_seq_len=24, L=32, NL=2, F=64, Te=64, Dt=32, Nt=8, input_feats=251, dataset='kit_ml')
HML_FULL = W.humanml3d_dims()          # reference configs/stmogen/T2M_humanml3d.py architecture


def synth_inputs(dims, B, T, seed, lengths=None):
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(B, T, dims['input_feats'], generator=g)
    xf = torch.nn.functional.layer_norm(torch.randn(B, dims['Nt'], dims['Dt'], generator=g), (dims['Dt'],))
    mask = torch.ones(B, T)
    if lengths is not None:
        for b, n in enumerate(lengths):
            mask[b, n:] = 0
    return x_T, xf, mask


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def step_noise_from_seed(seed, shape, num):
    """The per-step randn_like stream of the reference loop under torch.manual_seed(seed)."""
    g = torch.Generator().manual_seed(int(seed))
    return [torch.randn(shape, generator=g) for _ in range(num)]
