#!/usr/bin/env python
"""Dump one call of tutel's ``moe_layer`` as the reference configures it (mogen/models/attentions/st_attention.py:28-45)
into the fixture format tests/test_oracle.py::test_tutel_moe_dump_fixture replays (the tutel boundary is
PARITY UNPINNED in this repository because tutel is neither vendored nor installed here).

Run this where microsoft/tutel IS installed (any device):

    python tools/dump_tutel_moe.py --out tests/golden/tutel_dump_real.npz [--ties] [--device cuda]

and commit the file: the existing test then pins oracle/tutel_restated.py (and, through the GPU parity tests, the HIP
routing kernels) to the real library, and reports which tie policy ('stable' / 'reverse', see mc_ctx_set_tie_policy)
the installed tutel + torch build implements.

    --source restated   writes the same format from oracle/tutel_restated.py instead (loader self-test only; this is
                        what `tutel_dump_restated.npz` in tests/golden/ is -- it does NOT pin anything)
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--source', default='tutel', choices=['tutel', 'restated'])
    ap.add_argument('--tokens', type=int, default=960)
    ap.add_argument('--dim', type=int, default=32)
    ap.add_argument('--experts', type=int, default=16)
    ap.add_argument('--ties', action='store_true', help='second half of the tokens = exact copies of the first (CFG twins)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--device', default='cpu')
    a = ap.parse_args()
    g = torch.Generator().manual_seed(a.seed)
    N, D, E = a.tokens, a.dim, a.experts
    x = torch.randn(N, D, generator=g)
    if a.ties:
        x = x[:N // 2].repeat(2, 1)
    P = dict(proj_w=torch.randn(256, D, generator=g) / D ** 0.5, proj_b=torch.randn(256, generator=g) * 0.1,
             sim_matrix=torch.randn(256, E, generator=g), temperature=torch.tensor([0.7]),
             fc1_w=torch.randn(E, 4 * D, D, generator=g) * 0.3, fc1_b=torch.randn(E, 4 * D, generator=g) * 0.1,
             fc2_w=torch.randn(E, 4 * D, D, generator=g) * 0.3, fc2_b=torch.randn(E, D, generator=g) * 0.1)
    cf, k = 1.5, 2
    if a.source == 'restated':
        from oracle import tutel_restated as TR
        y, r = TR.moe_forward(x, *[P[n] for n in ('proj_w', 'proj_b', 'sim_matrix', 'temperature', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b')],
                              top_k=k, capacity_factor=cf, batch_prioritized_routing=True, return_routing=True)
        scores, indices, locations, gates, capacity = r['scores'], r['indices'], r['locations'], r['gates'], r['capacity']
        source = 'oracle/tutel_restated.py (NOT tutel: loader self-test only)'
    else:
        import tutel
        from tutel import moe as tutel_moe
        from tutel.impls import fast_dispatch
        dev = torch.device(a.device)
        layer = tutel_moe.moe_layer(
            gate_type={'type': 'cosine_top', 'k': k, 'fp32_gate': True, 'gate_noise': 1.0, 'capacity_factor': cf},
            experts={'type': 'ffn', 'count_per_node': E, 'hidden_size_per_expert': 4 * D, 'activation_fn': lambda t: F.gelu(t)},
            model_dim=D, batch_prioritized_routing=True, is_gshard_loss=False).to(dev).eval()
        sd = layer.state_dict()
        new = {'gates.0.cosine_projector.weight': P['proj_w'], 'gates.0.cosine_projector.bias': P['proj_b'],
               'gates.0.sim_matrix': P['sim_matrix'], 'gates.0.temperature': P['temperature'],
               'experts.batched_fc1_w': P['fc1_w'], 'experts.batched_fc1_bias': P['fc1_b'],
               'experts.batched_fc2_w': P['fc2_w'], 'experts.batched_fc2_bias': P['fc2_b']}
        missing = set(new) - set(sd)
        assert not missing, f'this tutel version names its parameters differently: {sorted(missing)} vs {sorted(sd)}'
        layer.load_state_dict({**sd, **{n: v.to(sd[n].dtype).reshape(sd[n].shape) for n, v in new.items()}})
        with torch.no_grad():
            xd = x.to(dev)
            y = layer(xd).cpu()
            gate = layer.gates[0]
            logits = gate(xd.float())
            scores = F.softmax(logits, dim=1)
            crit = fast_dispatch.extract_critical(scores, k, cf, batch_prioritized_routing=True, normalize_gate=True)
            # tutel returns (num_global_experts, indices_s, locations_s, gates_s, capacity)
            _, indices, locations, gates, capacity = crit[:5]
        scores = scores.cpu()
        indices, locations, gates = [t.cpu() for t in indices], [t.cpu() for t in locations], [t.cpu() for t in gates]
        source = f'tutel {getattr(tutel, "__version__", "?")} / torch {torch.__version__} / {a.device}'
    np.savez_compressed(a.out, x=x.numpy(), **{n: v.numpy() for n, v in P.items()}, top_k=np.int64(k),
                        capacity_factor=np.float64(cf), bpr=np.bool_(True), capacity=np.int64(int(capacity)),
                        scores=scores.numpy(), indices=np.stack([t.numpy() for t in indices]),
                        locations=np.stack([t.numpy() for t in locations]), gates=np.stack([t.numpy() for t in gates]),
                        y=y.numpy(), source=np.str_(source))
    print('wrote', a.out, '| source:', source)


if __name__ == '__main__':
    main()
