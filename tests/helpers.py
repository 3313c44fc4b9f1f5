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


# ---- evaluation-side fixtures (shared with tests/golden/make_golden.py) -----------------------------------------
EVAL_DIMS = dict(nfeats=322, latent_dim=128, ff_size=256, num_layers=2, num_heads=2)
EVAL_BERT = dict(dim=128, n_layers=2, n_heads=2, hidden_dim=256, max_position_embeddings=64)
T2M_DIMS = dict(input_size=263, movement_hidden_size=64, movement_latent_size=64, motion_hidden_size=128, motion_latent_size=32)
T2M_TEXT = dict(word_size=300, pos_size=15, hidden_size=64, output_size=32)


class StubEvalModel:
    """Deterministic stand-in for the embedding model so that the evaluators' host logic can be compared end to end."""
    device = 'cpu'

    def __init__(self, nfeats=12, d=16):
        g = torch.Generator().manual_seed(5)
        self.P = torch.randn(nfeats, d, generator=g)

    def encode_motion(self, motion, motion_length=None, motion_mask=None, **kw):
        m = motion_mask[..., None].to(motion.dtype)
        return torch.tanh(((motion * m).sum(1) / motion_length[:, None].to(motion.dtype)) @ self.P)

    def encode_text(self, text, token=None, device=None, **kw):
        v = torch.tensor([[float(ord(c) % 13) for c in (t + ' ' * 12)[:12]] for t in text])
        return torch.tanh((v - 6.0) / 4.0 @ self.P[:12])

    def to(self, device):
        return self

    def eval(self):
        return self


def _stub_results(n, nfeats=12, seed=9, T=14):
    """results as the test loop collects them: every motion padded to the dataset's fixed T with a length mask (the
    reference's own prepare_results cannot pad the ground-truth side: base_evaluator.py:75 calls type_as on a list)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n):
        n_i = int(torch.randint(6, T + 1, (1,), generator=g))
        msk = (torch.arange(T) < n_i).float()
        mot = torch.randn(T, nfeats, generator=g) * msk[:, None]
        out.append(dict(motion=mot, pred_motion=mot + 0.7 * torch.randn(T, nfeats, generator=g) * msk[:, None], motion_mask=msk,
                        pred_motion_mask=msk, motion_length=torch.tensor(n_i), pred_motion_length=torch.tensor(n_i),
                        text=''.join(chr(97 + int(c)) for c in torch.randint(0, 26, (10,), generator=g)), token=None))
    return out


def stub_eval_results(n, replications, append_indexes):
    """Result list in the layout the evaluators slice: per replication, n samples followed by the MultiModality repeats
    (fresh predictions for the re-drawn indexes)."""
    out = []
    for rep in range(replications):
        base = _stub_results(n, seed=9 + rep)
        g = torch.Generator().manual_seed(100 + rep)
        extra = []
        for i in append_indexes[rep]:
            r = dict(base[int(i)])
            r['pred_motion'] = r['motion'] + 0.7 * torch.randn(r['motion'].shape, generator=g) * r['motion_mask'][:, None]
            extra.append(r)
        out += base + extra
    return out
