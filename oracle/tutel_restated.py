"""Restatement of ``tutel.moe.moe_layer`` as the reference calls it.  TEST INFRASTRUCTURE ONLY.

**PARITY UNPINNED.**  tutel (microsoft/tutel) is an un-vendored, un-pinned
third-party dependency of the reference (absent from its requirements.txt; the
reference imports it at ``mogen/models/attentions/st_attention.py:10-14`` and
constructs it at ``:28-45``).  It is not installed in this image and nothing in
the reference tests its output, so this file restates tutel's *published*
algorithm (tutel/impls/moe_layer.py ``MOELayer.forward``, tutel/gates/cosine_top.py
``CosineTopKGate``, tutel/impls/fast_dispatch.py ``extract_critical`` /
``fast_encode`` / ``fast_decode``, tutel/experts/ffn.py ``FusedExpertsNetwork``)
for exactly the constructor arguments the reference passes:

    gate_type = {type:'cosine_top', k:2, fp32_gate:True, gate_noise:1.0, capacity_factor:1.5}
    experts   = {type:'ffn', count_per_node:16, hidden_size_per_expert:4*model_dim, activation_fn:gelu}
    batch_prioritized_routing=True, is_gshard_loss=False, group=None
    (defaults: normalize_gate=True, is_postscore=True, proj_dim=256, init_t=0.5)

Eval-mode semantics (``self.training == False``; no gate noise, l_aux ignored):

1. gate   : logits = normalize(Linear_{D->256}(x), dim=1) @ normalize(sim_matrix, dim=0)
            * exp(min(temperature, ln 100));  scores = softmax(logits, dim=1)
2. top-k  : topk_indices = topk(scores, k).indices; gate_k = scores[idx_k],
            renormalised by max(sum_k gate_k, eps)
3. slots  : with batch-prioritised routing tokens are ranked by DESCENDING
            max(scores) (stable: equal keys keep token order -- torch's sort
            is stable on CPU and its CUDA radix sort is too); location of
            (token, choice k) = number of earlier-ranked tokens whose k-th
            choice is the same expert + sum_{j<k} count_j[expert]
4. capacity = k * int(capacity_factor * ceil(N / E)); a (token, choice) whose
            location >= capacity is DROPPED (contributes 0, no renormalisation)
5. experts: slot buffer [E, capacity, D] (zeros in empty slots) ->
            gelu(x W1_e^T + b1_e) W2_e + b2_e with W1,W2 of shape [E, 4D, D]
6. decode : y[token] = sum_k gate_k * expert_out[idx_k, loc_k]  (post-score)

Risk stated per SURVEY.md section 8c: a divergence from the real tutel on the
capacity-overflow / tie order cannot be detected offline.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def gate_scores(x, proj_w, proj_b, sim_matrix, temperature):
    """cosine_top gate (tutel/gates/cosine_top.py CosineTopKGate.forward) + softmax
    (tutel/impls/moe_layer.py routing()).  fp32_gate=True -> computed in fp32
    (or the dtype of ``x`` when the oracle is run in fp64 for drift studies)."""
    dt = torch.float32 if x.dtype != torch.float64 else torch.float64
    x = x.to(dt)
    proj = F.linear(x, proj_w.to(dt), proj_b.to(dt))
    logits = torch.matmul(F.normalize(proj, dim=1), F.normalize(sim_matrix.to(dt), dim=0))
    logit_scale = torch.clamp(temperature.to(dt), max=math.log(1.0 / 0.01)).exp()
    return F.softmax(logits * logit_scale, dim=1)


def capacity_of(num_tokens, num_experts, top_k, capacity_factor):
    """tutel/impls/fast_dispatch.py extract_critical: capacity per expert."""
    return top_k * int(capacity_factor * ((num_tokens + num_experts - 1) // num_experts))


# Order of tokens with EXACTLY equal importance in the batch-prioritised ranking.  tutel calls
# ``importance_scores.argsort(dim=0)`` (not stable: the order of ties is implementation-defined).  'stable' = lower token
# index first (what torch's CPU sort and the CUDA radix sort produce in practice); 'reverse' = higher index first.
# The HIP path has the same switch (mc_ctx_set_tie_policy); tests run both.
TIE_POLICY = 'stable'


def _importance_order(importance):
    if TIE_POLICY == 'stable':
        return torch.argsort(importance, dim=0, stable=True)
    if TIE_POLICY == 'reverse':
        n = importance.shape[0]
        return (n - 1) - torch.argsort(importance.flip(0), dim=0, stable=True)
    raise ValueError(f'unknown TIE_POLICY {TIE_POLICY!r}')


def extract_critical(scores, top_k, capacity_factor, batch_prioritized_routing=True):
    """indices_s, locations_s, gates_s, capacity  (tutel fast_dispatch.extract_critical,
    normalize_gate=True)."""
    N, E = scores.shape
    topk_indices = torch.topk(scores, top_k, dim=1).indices
    indices_s = [topk_indices[:, k].contiguous() for k in range(top_k)]
    masks_se = [F.one_hot(idx, num_classes=E) for idx in indices_s]
    gates_s = [(scores * m).sum(dim=1) for m in masks_se]
    denom = torch.clamp(sum(gates_s), min=torch.finfo(scores.dtype).eps)
    gates_s = [g / denom for g in gates_s]

    if batch_prioritized_routing:
        importance = -1 * scores.max(dim=1)[0]
        order = _importance_order(importance)
        inv = torch.argsort(order, dim=0, stable=True)
        compute_location = lambda m: (torch.cumsum(m[order], dim=0) - 1)[inv]
    else:
        compute_location = lambda m: torch.cumsum(m, dim=0) - 1

    locations_s = []
    acc_base = None
    for k in range(top_k):
        loc = compute_location(masks_se[k])
        if k > 0:
            s = masks_se[k - 1].sum(dim=0, keepdim=True)
            acc_base = s if acc_base is None else acc_base + s
            loc = loc + acc_base
        locations_s.append((loc * masks_se[k]).sum(dim=1))
    capacity = capacity_of(N, E, top_k, capacity_factor)
    return indices_s, locations_s, gates_s, capacity


def moe_forward(x, proj_w, proj_b, sim_matrix, temperature, fc1_w, fc1_b, fc2_w, fc2_b,
                top_k=2, capacity_factor=1.5, batch_prioritized_routing=True,
                return_routing=False, forced=None):
    """Whole layer, x: [N, D] -> [N, D].

    ``forced = (indices [N, k] long, keep [N, k] bool)`` replaces the DISCRETE routing decisions
    (top-k expert ids and the capacity drop set) by externally supplied ones while every
    continuous quantity (scores, gates, expert FFNs) is still computed here.  Tests use it to
    separate arithmetic parity from the inherently discontinuous capacity-boundary decision
    (a 1-ulp change of a max score can swap which of two near-tied tokens is dropped)."""
    N, D = x.shape
    E = fc1_w.shape[0]
    scores = gate_scores(x, proj_w, proj_b, sim_matrix, temperature)
    indices_s, locations_s, gates_s, capacity = extract_critical(
        scores, top_k, capacity_factor, batch_prioritized_routing)
    free = dict(indices=indices_s, keeps=[l < capacity for l in locations_s])
    if forced is not None:
        f_idx, f_keep = forced
        indices_s = [f_idx[:, k].long().contiguous() for k in range(top_k)]
        g = [scores.gather(1, i.unsqueeze(1)).squeeze(1) for i in indices_s]
        denom = torch.clamp(sum(g), min=torch.finfo(scores.dtype).eps)
        gates_s = [v / denom for v in g]
        # give every kept pair its own slot (slot identity does not influence the result):
        # first choices of expert e occupy 0..c0[e]-1, kept second choices follow
        keep_s = [f_keep[:, k].bool() for k in range(top_k)]
        base = torch.zeros(E, dtype=torch.long)
        locations_s, big = [], 10 ** 9
        for k in range(top_k):
            oh = F.one_hot(indices_s[k], num_classes=E) * keep_s[k].unsqueeze(1).long()
            pos = (torch.cumsum(oh, dim=0) - 1).gather(1, indices_s[k].unsqueeze(1)).squeeze(1)
            locations_s.append(torch.where(keep_s[k], pos + base[indices_s[k]], torch.full_like(pos, big)))
            base = base + oh.sum(0)
        capacity = int(base.max()) + 1
    xg = x.to(scores.dtype)
    # fast_encode, is_postscore=True: slot buffer accumulates x (weight 1)
    disp = torch.zeros(E * capacity, D, dtype=xg.dtype)
    keeps = []
    for idx, loc in zip(indices_s, locations_s):
        keep = loc < capacity
        keeps.append(keep)
        disp.index_add_(0, (idx * capacity + loc)[keep], xg[keep])
    disp = disp.view(E, capacity, D).to(x.dtype)
    # FusedExpertsNetwork.forward
    h = torch.matmul(disp, fc1_w.permute(0, 2, 1)) + fc1_b.unsqueeze(1)
    h = F.gelu(h)
    out = torch.matmul(h, fc2_w) + fc2_b.unsqueeze(1)
    out = out.reshape(E * capacity, -1).to(scores.dtype)
    # fast_decode, is_postscore=True
    y = torch.zeros(N, out.shape[1], dtype=out.dtype)
    for idx, loc, g, keep in zip(indices_s, locations_s, gates_s, keeps):
        slot = (idx * capacity + loc)[keep]
        y[keep] += g[keep].unsqueeze(1) * out[slot]
    y = y.to(x.dtype)
    if return_routing:
        return y, dict(scores=scores, indices=indices_s, locations=locations_s,
                       gates=gates_s, capacity=capacity, keeps=keeps, free=free)
    return y


# --------------------------------------------------------------------------------------
# nn.Module form with tutel's state-dict key names; injected as ``tutel.moe.moe_layer``
# when the reference's own modules are imported by oracle/ref_shim.py.
# --------------------------------------------------------------------------------------
class _CosineTopKGate(nn.Module):
    def __init__(self, model_dim, num_experts, k, proj_dim=256, init_t=0.5):
        super().__init__()
        self.top_k = k
        self.temperature = nn.Parameter(torch.log(torch.full([1], 1.0 / init_t)))
        self.cosine_projector = nn.Linear(model_dim, proj_dim)
        self.sim_matrix = nn.Parameter(torch.randn(proj_dim, num_experts) * 0.01)


class _FusedExperts(nn.Module):
    def __init__(self, num_experts, model_dim, hidden):
        super().__init__()
        self.batched_fc1_w = nn.Parameter(torch.randn(num_experts, hidden, model_dim) * 0.02)
        self.batched_fc2_w = nn.Parameter(torch.randn(num_experts, hidden, model_dim) * 0.02)
        self.batched_fc1_bias = nn.Parameter(torch.zeros(num_experts, hidden))
        self.batched_fc2_bias = nn.Parameter(torch.zeros(num_experts, model_dim))


class RestatedMoELayer(nn.Module):
    """Drop-in for ``tutel.moe.moe_layer(gate_type=..., experts=..., model_dim=..., ...)``."""

    def __init__(self, gate_type, experts, model_dim, batch_prioritized_routing=False,
                 is_gshard_loss=True, group=None, **kwargs):
        super().__init__()
        assert gate_type['type'] == 'cosine_top'
        assert experts['type'] == 'ffn'
        self.num_experts = experts['count_per_node']
        self.top_k = min(int(gate_type['k']), self.num_experts)
        self.capacity_factor = float(gate_type.get('capacity_factor', 1.0))
        self.bpr = batch_prioritized_routing
        self.gates = nn.ModuleList([_CosineTopKGate(model_dim, self.num_experts, self.top_k)])
        self.experts = _FusedExperts(self.num_experts, model_dim, experts['hidden_size_per_expert'])
        self.l_aux = torch.zeros(())

    def forward(self, x):
        g, e = self.gates[0], self.experts
        return moe_forward(x, g.cosine_projector.weight, g.cosine_projector.bias, g.sim_matrix,
                           g.temperature, e.batched_fc1_w, e.batched_fc1_bias, e.batched_fc2_w,
                           e.batched_fc2_bias, self.top_k, self.capacity_factor, self.bpr)
