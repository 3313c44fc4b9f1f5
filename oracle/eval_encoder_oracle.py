"""CPU restatement of the evaluation embedding model and the quality metrics.  TEST INFRASTRUCTURE ONLY.

Evaluation side of SURVEY.md section 8f.4:
  * ``ActorAgnosticEncoder`` (motion -> mu)            mogen/models/rnns/t2m_bigru_smplx.py:66-195
  * ``DistilbertActorAgnosticEncoder`` (tokens -> mu)   mogen/models/rnns/t2m_bigru_smplx.py:198-394; the DistilBERT
    network itself is the ``transformers`` package (importable in this image, v4/5 ``DistilBertModel``: embeddings =
    LayerNorm(word + position, eps 1e-12); layer = LN(x + out_lin(softmax(q k^T / sqrt(dh), key mask) v));
    LN(x + lin2(gelu(lin1(x)))))
  * FID / R-precision / matching score / diversity / multimodality   mogen/core/evaluation/utils.py:12-140 and
    evaluators/{fid,precision,matching_score,diversity,multimodality}_evaluator.py

Pinned by tests/golden/make_golden.py (group ``evaluator``) against the reference's own classes (a reduced
DistilBERT saved to a temporary directory stands for the pretrained ``modelpath``) -> tests/golden/evaluator.npz.
Written with explicit matmul / softmax / layer_norm (no nn.TransformerEncoder) so that it is a restatement rather
than a second call of the same module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _attention(q, k, v, valid, heads):
    """q, k, v [B, S, d]; valid bool [B, S] (keys that may be attended) -> [B, S, d]."""
    B, S, d = q.shape
    dh = d // heads
    q = q.view(B, S, heads, dh).transpose(1, 2) / math.sqrt(dh)
    k = k.view(B, S, heads, dh).transpose(1, 2)
    v = v.view(B, S, heads, dh).transpose(1, 2)
    sc = q @ k.transpose(-1, -2)
    sc = sc.masked_fill(~valid[:, None, None, :], float('-inf'))
    return (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(B, S, d)


def _post_ln_layer(x, p, pre, valid, heads, names, eps):
    """x [B, S, d]: LN(x + SA(x)); LN(x + FF(x)) with GELU(erf)."""
    d = x.shape[-1]
    if names == 'torch':
        qkv = F.linear(x, p[pre + 'self_attn.in_proj_weight'], p[pre + 'self_attn.in_proj_bias'])
        q, k, v = qkv.split(d, dim=-1)
        ow, ob = p[pre + 'self_attn.out_proj.weight'], p[pre + 'self_attn.out_proj.bias']
        n1, n2, l1, l2 = 'norm1', 'norm2', 'linear1', 'linear2'
    else:
        q, k, v = (F.linear(x, p[pre + f'attention.{n}.weight'], p[pre + f'attention.{n}.bias']) for n in ('q_lin', 'k_lin', 'v_lin'))
        ow, ob = p[pre + 'attention.out_lin.weight'], p[pre + 'attention.out_lin.bias']
        n1, n2, l1, l2 = 'sa_layer_norm', 'output_layer_norm', 'ffn.lin1', 'ffn.lin2'
    a = F.linear(_attention(q, k, v, valid, heads), ow, ob)
    x = F.layer_norm(x + a, (d,), p[pre + n1 + '.weight'], p[pre + n1 + '.bias'], eps)
    h = F.linear(F.gelu(F.linear(x, p[pre + l1 + '.weight'], p[pre + l1 + '.bias'])), p[pre + l2 + '.weight'], p[pre + l2 + '.bias'])
    return F.layer_norm(x + h, (d,), p[pre + n2 + '.weight'], p[pre + n2 + '.bias'], eps)


def _vae_transformer(p, pre, x, valid, num_layers, heads):
    """shared tail of both encoders (:164-195, :366-394): [mu_token, logvar_token, x] + pe -> post-LN encoder -> row 0."""
    B, S, d = x.shape
    tok = torch.stack([p[pre + 'mu_token'], p[pre + 'logvar_token']])[None].expand(B, 2, d)
    xs = torch.cat([tok, x], dim=1) + p[pre + 'sequence_pos_encoding.pe'][:S + 2, 0][None]
    vm = torch.cat([torch.ones(B, 2, dtype=torch.bool), valid], dim=1)
    for i in range(num_layers):
        xs = _post_ln_layer(xs, p, f'{pre}seqTransEncoder.layers.{i}.', vm, heads, 'torch', 1e-5)
    return xs[:, 0], xs[:, 1]


def encode_motion(p, motion, lengths, num_layers=4, heads=4, pre='motionencoder.'):
    """T2MContrastiveModel_SMPLX.encode_motion (:404-410): motion [B, T, nfeats], lengths [B] -> mu [B, latent]."""
    B, T, _ = motion.shape
    x = F.linear(motion, p[pre + 'skel_embedding.weight'], p[pre + 'skel_embedding.bias'])
    valid = torch.arange(T)[None] < torch.as_tensor(lengths)[:, None]
    return _vae_transformer(p, pre, x, valid, num_layers, heads)[0]


def distilbert_hidden(p, ids, mask, n_layers, heads, pre='textencoder.text_model.'):
    """DistilBertModel(input_ids, attention_mask).last_hidden_state."""
    S = ids.shape[1]
    e = pre + 'embeddings.'
    x = p[e + 'word_embeddings.weight'][ids] + p[e + 'position_embeddings.weight'][:S][None]
    x = F.layer_norm(x, (x.shape[-1],), p[e + 'LayerNorm.weight'], p[e + 'LayerNorm.bias'], 1e-12)
    valid = mask.bool()
    for i in range(n_layers):
        x = _post_ln_layer(x, p, f'{pre}transformer.layer.{i}.', valid, heads, 'distilbert', 1e-12)
    return x


def encode_text_tokens(p, ids, mask, bert_layers, bert_heads, num_layers=4, heads=4, pre='textencoder.'):
    """T2MContrastiveModel_SMPLX.encode_text (:412-414) after tokenisation: ids/mask [B, S] -> mu [B, latent]."""
    h = distilbert_hidden(p, ids, mask, bert_layers, bert_heads, pre + 'text_model.')
    x = F.linear(F.relu(h), p[pre + 'projection.1.weight'], p[pre + 'projection.1.bias'])
    return _vae_transformer(p, pre, x, mask.bool(), num_layers, heads)[0]


# ---- metrics (mogen/core/evaluation/utils.py) ----------------------------------------------------------------------
def pairwise_l2(a, b):
    """utils.py:12-26: sqrt(-2 a b^T + |a|^2 + |b|^2) in the input dtype."""
    return np.sqrt(-2 * np.dot(a, b.T) + np.sum(np.square(a), axis=1, keepdims=True) + np.sum(np.square(b), axis=1))


def r_precision_counts(text_emb, motion_emb, top_k=3):
    """precision_evaluator.py:43-49 + utils.py:29-39: how many rows have their own index among the k nearest."""
    order = np.argsort(pairwise_l2(text_emb, motion_emb), axis=1)
    hit = order[:, :top_k] == np.arange(order.shape[0])[:, None]
    return np.cumsum(hit, axis=1).astype(bool).sum(axis=0)


def matching_score_sum(text_emb, motion_emb):
    """matching_score_evaluator.py:37-58: both sides z-scored, then the trace of the distance matrix."""
    return pairwise_l2(standardise(text_emb), standardise(motion_emb)).trace()


def diversity(emb, times, emb_scale=1, norm_scale=1):
    """utils.py:111-125; consumes np.random exactly like the reference (two choice() draws without replacement)."""
    e = emb * emb_scale
    i = np.random.choice(e.shape[0], times, replace=False)
    j = np.random.choice(e.shape[0], times, replace=False)
    return np.linalg.norm((e[i] - e[j]) * norm_scale, axis=1).mean()


def multimodality(emb, picks):
    """utils.py:128-140: emb [num_samples, num_repeats, d]."""
    i = np.random.choice(emb.shape[1], picks, replace=False)
    j = np.random.choice(emb.shape[1], picks, replace=False)
    return np.linalg.norm(emb[:, i] - emb[:, j], axis=2).mean()


def standardise(e):
    """fid_evaluator.py:48-56: per-dimension z-score, zero std replaced by 1e-8."""
    sd = np.std(e, axis=0)
    sd[sd == 0] = 1e-8
    return (e - np.mean(e, axis=0)) / sd


def frechet(mu1, c1, mu2, c2):
    """utils.py:59-108 (without the singular-product retry, which the fixtures do not reach)."""
    from scipy import linalg
    root, _ = linalg.sqrtm(c1.dot(c2), disp=False)
    if np.iscomplexobj(root):
        root = root.real
    diff = mu1 - mu2
    return diff.dot(diff) + np.trace(c1) + np.trace(c2) - 2 * np.trace(root)


def fid(pred_emb, gt_emb, emb_scale=1):
    """fid_evaluator.py:32-63."""
    a, b = standardise(gt_emb) * emb_scale, standardise(pred_emb) * emb_scale
    return frechet(np.mean(a, axis=0), np.cov(a, rowvar=False), np.mean(b, axis=0), np.cov(b, rowvar=False))
