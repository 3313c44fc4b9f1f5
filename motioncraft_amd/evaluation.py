"""Evaluation side of the sampling path: the embedding model on the device + the metrics.

  * ``T2MContrastiveModel_SMPLX`` (registered in ``SUBMODULES`` under the reference's name; same constructor arguments
    and ``encode_motion`` / ``encode_text`` methods as ``mogen/models/rnns/t2m_bigru_smplx.py:396-437``) runs both
    encoders through ``mc_evalenc_*`` of the HIP library: nothing here computes on the CPU, and construction fails when
    the library or the GPU is missing.
  * the metric functions and the five evaluators of ``mogen/core/evaluation/`` (``EVALUATORS`` registry with the
    reference's metric names 'R Precision', 'Matching Score', 'FID', 'Diversity', 'MultiModality'; ``build_evaluator``).
    Like the reference they are host numpy over the [N, 256] embedding tables (the reference calls
    ``.cpu().detach().numpy()`` before every metric; ``scipy.linalg.sqrtm`` for the Frechet distance).
"""
import copy
import ctypes
import os

import numpy as np
import torch

from . import lib as _lib
from .builder import SUBMODULES
from .registry import Registry
from .synthetic import sinusoid_table
from .wordpiece import WordPieceTokenizer

DISTILBERT_BASE = dict(dim=768, n_layers=6, n_heads=12, hidden_dim=3072, vocab_size=30522, max_position_embeddings=512)


class NativeEvalEncoder:
    """Device encoders over the evaluator checkpoint's own keys (``motionencoder.*``, ``textencoder.*``)."""

    def __init__(self, state_dict, nfeats=322, latent_dim=256, ff_size=1024, num_layers=4, num_heads=4, bert=None):
        self.lib = _lib.load(require_gpu=True)
        sd = dict(state_dict)
        has_text = bert is not None and 'textencoder.projection.1.weight' in sd
        cfg = _lib.EvalEncConfig()
        cfg.nfeats, cfg.latent_dim, cfg.ff_size, cfg.num_layers, cfg.num_heads = nfeats, latent_dim, ff_size, num_layers, num_heads
        for pre in ('motionencoder.', 'textencoder.'):            # the table is a buffer; older checkpoints may omit it
            if pre + 'sequence_pos_encoding.pe' not in sd and (pre == 'motionencoder.' or has_text):
                sd[pre + 'sequence_pos_encoding.pe'] = sinusoid_table(5000, latent_dim)
        cfg.pe_len = int(sd['motionencoder.sequence_pos_encoding.pe'].shape[0])
        if has_text:
            if int(sd['textencoder.sequence_pos_encoding.pe'].shape[0]) != cfg.pe_len:
                raise ValueError('motion and text positional tables differ in length')
            b = dict(DISTILBERT_BASE, **bert)
            cfg.bert_dim, cfg.bert_layers, cfg.bert_heads, cfg.bert_ff = b['dim'], b['n_layers'], b['n_heads'], b['hidden_dim']
            cfg.bert_vocab = int(sd['textencoder.text_model.embeddings.word_embeddings.weight'].shape[0])
            cfg.bert_max_pos = int(sd['textencoder.text_model.embeddings.position_embeddings.weight'].shape[0])
        self.cfg, self.has_text = cfg, has_text
        h = ctypes.c_void_p()
        _lib.check(self.lib.mc_evalenc_create(ctypes.byref(cfg), ctypes.byref(h)), 'mc_evalenc_create')
        self.handle = h
        for k, v in sd.items():
            if not (k.startswith('motionencoder.') or (has_text and k.startswith('textencoder.'))):
                continue
            if not torch.is_floating_point(torch.as_tensor(v)):
                continue                                            # e.g. embeddings.position_ids of older transformers
            a = np.ascontiguousarray(torch.as_tensor(v).detach().cpu().float().numpy())
            _lib.check(self.lib.mc_evalenc_set_param(self.handle, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                       f'mc_evalenc_set_param({k})')
        _lib.check(self.lib.mc_evalenc_finalize(self.handle), 'mc_evalenc_finalize')

    def encode_motion(self, motion, motion_length):
        """motion [B, T, nfeats] float32 device tensor, motion_length [B] -> mu [B, latent_dim] (device)."""
        if not (motion.is_cuda and motion.dim() == 3 and motion.shape[2] == self.cfg.nfeats):
            raise ValueError(f'motion must be a device tensor [B, T, {self.cfg.nfeats}]')
        m = motion.to(torch.float32).contiguous()
        n = torch.as_tensor(motion_length).to(device=m.device, dtype=torch.int32).contiguous()
        if n.numel() != m.shape[0]:
            raise ValueError('motion_length must hold one length per sample')
        out = torch.empty(m.shape[0], self.cfg.latent_dim, device=m.device, dtype=torch.float32)
        _lib.check(self.lib.mc_evalenc_encode_motion(self.handle, ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(n.data_ptr()),
                                                     m.shape[0], m.shape[1], ctypes.c_void_p(out.data_ptr()),
                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'mc_evalenc_encode_motion')
        return out

    def encode_tokens(self, input_ids, attention_mask):
        """input_ids / attention_mask [B, S] (device) -> mu [B, latent_dim]."""
        if not self.has_text:
            raise RuntimeError('the evaluator was built without textencoder.* weights')
        ids = input_ids.to(dtype=torch.int32).contiguous()
        mask = attention_mask.to(device=ids.device, dtype=torch.uint8).contiguous()
        if not ids.is_cuda or ids.dim() != 2 or mask.shape != ids.shape:
            raise ValueError('input_ids and attention_mask must be device tensors of one shape [B, S]')
        out = torch.empty(ids.shape[0], self.cfg.latent_dim, device=ids.device, dtype=torch.float32)
        _lib.check(self.lib.mc_evalenc_encode_text(self.handle, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(mask.data_ptr()),
                                                   ids.shape[0], ids.shape[1], ctypes.c_void_p(out.data_ptr()),
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'mc_evalenc_encode_text')
        return out

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.mc_evalenc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _bert_config(modelpath):
    import json
    with open(os.path.join(modelpath, 'config.json')) as f:
        c = json.load(f)
    return {k: c[k] for k in DISTILBERT_BASE if k in c}


@SUBMODULES.register_module()
class T2MContrastiveModel_SMPLX:
    """``evaluator_model=dict(type='T2MContrastiveModel_SMPLX', motion_encoder=dict(nfeats, vae, num_layers, ...),
    text_encoder=dict(modelpath, num_layers, ...), init_cfg=dict(type='Pretrained', checkpoint=...))``
    (configs/_base_/datasets/motionx_bs128.py:38-51).  ``state_dict=`` may be given instead of a checkpoint file."""

    def __init__(self, motion_encoder=None, text_encoder=None, init_cfg=None, state_dict=None):
        from .checkpoint import read_state_dict
        me, te = dict(motion_encoder or {}), dict(text_encoder or {})
        if not me.get('vae', True) or not te.get('vae', True):
            raise NotImplementedError('vae=False (emb_token) is not used by the shipped evaluator configs')
        for c in (me, te):
            if c.get('activation', 'gelu') != 'gelu':
                raise NotImplementedError('the evaluator encoders use GELU layers')
        if state_dict is None:
            assert init_cfg is not None and init_cfg['type'] == 'Pretrained'
            state_dict, _ = read_state_dict(init_cfg['checkpoint'])
        d, ff = me.get('latent_dim', 256), me.get('ff_size', 1024)
        layers, heads = me.get('num_layers', 4), me.get('num_heads', 4)
        if te and (te.get('latent_dim', 256), te.get('ff_size', 1024), te.get('num_layers', 4), te.get('num_heads', 4)) != (d, ff, layers, heads):
            raise NotImplementedError('motion and text encoders of different shapes')
        bert = None
        self.tokenizer = None
        if te.get('modelpath'):
            bert = _bert_config(te['modelpath'])
            self.tokenizer = WordPieceTokenizer(te['modelpath'])
        elif te:
            bert = dict(te.get('bert', {}))
        self.encoder = NativeEvalEncoder(state_dict, nfeats=me['nfeats'], latent_dim=d, ff_size=ff, num_layers=layers,
                                         num_heads=heads, bert=bert)

    def to(self, device):
        return self

    def eval(self):
        return self

    def encode_motion(self, motion, motion_length=None, motion_mask=None, **kwargs):
        if motion_length is None:
            motion_length = [motion.shape[1]] * motion.shape[0]
        return self.encoder.encode_motion(motion, motion_length)

    def encode_text(self, text, token=None, device=None, **kwargs):
        """text: list of sentences (``token`` is accepted and ignored, like the reference); or pass
        ``input_ids`` / ``attention_mask`` tensors when the vocabulary is not at hand."""
        if 'input_ids' in kwargs:
            ids, mask = kwargs['input_ids'], kwargs['attention_mask']
        else:
            if self.tokenizer is None:
                raise RuntimeError('no vocabulary: build the evaluator with text_encoder.modelpath or pass input_ids/attention_mask')
            ids, mask = (torch.from_numpy(a) for a in self.tokenizer(list(text)))
        device = device if device is not None else 'cuda'
        return self.encoder.encode_tokens(ids.to(device), mask.to(device))


# ---- metric functions (mogen/core/evaluation/utils.py) --------------------------------------------------------------
def get_metric_statistics(values, replication_times):
    """utils.py:5-9: mean and the 95 % interval half-width over replications."""
    return np.mean(values, axis=0), 1.96 * np.std(values, axis=0) / np.sqrt(replication_times)


def euclidean_distance_matrix(matrix1, matrix2):
    """utils.py:12-26: dist[i, j] = |matrix1[i] - matrix2[j]| through the expanded square."""
    assert matrix1.shape[1] == matrix2.shape[1]
    sq1 = np.sum(np.square(matrix1), axis=1, keepdims=True)
    sq2 = np.sum(np.square(matrix2), axis=1)
    return np.sqrt(-2 * np.dot(matrix1, matrix2.T) + sq1 + sq2)


def calculate_top_k(mat, top_k):
    """utils.py:29-39: mat = argsort of a square distance matrix; [i, k] is True when row i's own index is in its first k+1."""
    own = np.arange(mat.shape[0])[:, None]
    return np.logical_or.accumulate(mat[:, :top_k] == own, axis=1)


def calculate_activation_statistics(activations, emb_scale):
    """utils.py:42-54."""
    a = activations * emb_scale
    return np.mean(a, axis=0), np.cov(a, rowvar=False)


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """utils.py:57-108: |mu1 - mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)), with the diagonal-offset retry for a singular product."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape and sigma1.shape == sigma2.shape
    root, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(root).all():
        off = np.eye(sigma1.shape[0]) * eps
        root = linalg.sqrtm((sigma1 + off).dot(sigma2 + off))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError('Imaginary component {}'.format(np.max(np.abs(root.imag))))
        root = root.real
    delta = mu1 - mu2
    return delta.dot(delta) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(root)


def calculate_diversity(activation, diversity_times, emb_scale, norm_scale):
    """utils.py:111-125 (draws from the global numpy generator in the reference's order)."""
    assert activation.ndim == 2 and activation.shape[0] > diversity_times
    a = activation * emb_scale
    first = np.random.choice(a.shape[0], diversity_times, replace=False)
    second = np.random.choice(a.shape[0], diversity_times, replace=False)
    return np.linalg.norm((a[first] - a[second]) * norm_scale, axis=1).mean()


def calculate_multimodality(activation, multimodality_times):
    """utils.py:128-140: activation [num_sentences, num_repeats, d]."""
    assert activation.ndim == 3 and activation.shape[1] > multimodality_times
    first = np.random.choice(activation.shape[1], multimodality_times, replace=False)
    second = np.random.choice(activation.shape[1], multimodality_times, replace=False)
    return np.linalg.norm(activation[:, first] - activation[:, second], axis=2).mean()


def _zscore(e):
    sd = np.std(e, axis=0)
    sd[sd == 0] = 1e-8
    return (e - np.mean(e, axis=0)) / sd


# ---- evaluators (mogen/core/evaluation/evaluators/) ------------------------------------------------------------------
def _pad_time(x, T):
    return x if x.shape[0] >= T else torch.cat([x, x.new_zeros((T - x.shape[0],) + tuple(x.shape[1:]))], dim=0)


class BaseEvaluator:
    """base_evaluator.py:7-165: slices the result list per replication (and per batch), embeds, reduces."""
    ENC_BATCH = 32

    def __init__(self, batch_size=None, drop_last=False, replication_times=1, replication_reduction='statistics',
                 eval_begin_idx=None, eval_end_idx=None, evaluator_model=None):
        assert replication_reduction in ('statistics', 'mean', 'concat')
        self.batch_size, self.drop_last = batch_size, drop_last
        self.replication_times, self.replication_reduction = replication_times, replication_reduction
        self.eval_begin_idx, self.eval_end_idx = eval_begin_idx, eval_end_idx
        self.evaluator_model = evaluator_model
        self.append_indexes = None

    def evaluate(self, results):
        per_rep = len(results) // self.replication_times
        vals = []
        for rep in range(self.replication_times):
            part = results[rep * per_rep:(rep + 1) * per_rep]
            if self.batch_size is None:
                vals.append(self.single_evaluate(part[self.eval_begin_idx:self.eval_end_idx]))
                continue
            got = []
            for b0 in range(self.eval_begin_idx, self.eval_end_idx, self.batch_size):
                chunk = part[b0:b0 + self.batch_size]
                if len(chunk) < self.batch_size and self.drop_last:
                    continue
                got.append(self.single_evaluate(chunk))
            vals.append(self.concat_batch_metrics(got))
        vals = np.stack(vals, axis=0)
        if self.replication_reduction == 'statistics':
            vals = get_metric_statistics(vals, self.replication_times)
        elif self.replication_reduction == 'mean':
            vals = np.mean(vals, axis=0)
        return self.parse_values(vals)

    def prepare_results(self, results):
        T = max(r['motion'].shape[0] for r in results)
        out = {k: torch.stack([_pad_time(r[k], T) for r in results], dim=0)
               for k in ('pred_motion', 'pred_motion_mask', 'motion', 'motion_mask')}
        for k in ('motion_length', 'pred_motion_length'):
            out[k] = torch.tensor([int(r[k].item() if hasattr(r[k], 'item') else r[k]) for r in results], dtype=torch.long)
        out['text'] = [r['text'] for r in results if 'text' in r]
        out['token'] = [r['token'] for r in results if 'token' in r]
        return out

    def encode_motion(self, motion, motion_length, motion_mask, device):
        embs = []
        for i in range(0, motion.shape[0], self.ENC_BATCH):
            sl = slice(i, i + self.ENC_BATCH)
            embs.append(self.evaluator_model.encode_motion(motion=motion[sl].to(device), motion_length=motion_length[sl].to(device),
                                                           motion_mask=motion_mask[sl].to(device), device=device).cpu())
        return torch.cat(embs, dim=0)

    def encode_text(self, text, token, device):
        embs = []
        for i in range(0, len(text), self.ENC_BATCH):
            sl = slice(i, i + self.ENC_BATCH)
            embs.append(self.evaluator_model.encode_text(text=text[sl], token=None if token is None else token[sl], device=device).cpu())
        return torch.cat(embs, dim=0)

    def _device(self):
        # the embedding model decides where it runs; T2MContrastiveModel_SMPLX is device-only
        return torch.device(getattr(self.evaluator_model, 'device', 'cuda'))

    def _motion_emb(self, res, prefix='pred_'):
        dev = self._device()
        return self.encode_motion(res[prefix + 'motion'], res[prefix + 'motion_length'], res[prefix + 'motion_mask'], dev).numpy()

    def _text_emb(self, res):
        return self.encode_text(res['text'], res['token'] or None, self._device()).numpy()


class PrecisionEvaluator(BaseEvaluator):
    def __init__(self, data_len=0, evaluator_model=None, top_k=3, batch_size=32, drop_last=False, replication_times=1,
                 replication_reduction='statistics', **kwargs):
        super().__init__(batch_size, drop_last, replication_times, replication_reduction, 0, data_len, evaluator_model)
        self.top_k = top_k

    def single_evaluate(self, results):
        res = self.prepare_results(results)
        order = np.argsort(euclidean_distance_matrix(self._text_emb(res), self._motion_emb(res)), axis=1)
        return calculate_top_k(order, top_k=self.top_k).sum(axis=0), order.shape[0]

    def concat_batch_metrics(self, batch_metrics):
        return sum(c for c, _ in batch_metrics) / sum(n for _, n in batch_metrics)

    def parse_values(self, values):
        out = {}
        for k in range(self.top_k):
            out['R_precision Top %d (mean)' % (k + 1)] = values[0][k]
            out['R_precision Top %d (conf)' % (k + 1)] = values[1][k]
        return out


class MatchingScoreEvaluator(BaseEvaluator):
    def __init__(self, data_len=0, evaluator_model=None, batch_size=32, drop_last=False, replication_times=1,
                 replication_reduction='statistics', **kwargs):
        super().__init__(batch_size, drop_last, replication_times, replication_reduction, 0, data_len, evaluator_model)

    def single_evaluate(self, results):
        res = self.prepare_results(results)
        text, motion = _zscore(self._text_emb(res)), _zscore(self._motion_emb(res))
        return euclidean_distance_matrix(text, motion).trace(), text.shape[0]

    def concat_batch_metrics(self, batch_metrics):
        return sum(s for s, _ in batch_metrics) / sum(n for _, n in batch_metrics)

    def parse_values(self, values):
        return {'Matching Score (mean)': values[0], 'Matching Score (conf)': values[1]}


class FIDEvaluator(BaseEvaluator):
    def __init__(self, data_len=0, evaluator_model=None, batch_size=None, drop_last=False, replication_times=1, emb_scale=1,
                 replication_reduction='statistics', **kwargs):
        super().__init__(batch_size, drop_last, replication_times, replication_reduction, 0, data_len, evaluator_model)
        self.emb_scale = emb_scale

    def single_evaluate(self, results):
        res = self.prepare_results(results)
        pred, gt = _zscore(self._motion_emb(res)), _zscore(self._motion_emb(res, prefix=''))
        gt_mu, gt_cov = calculate_activation_statistics(gt, self.emb_scale)
        pr_mu, pr_cov = calculate_activation_statistics(pred, self.emb_scale)
        return calculate_frechet_distance(gt_mu, gt_cov, pr_mu, pr_cov)

    def parse_values(self, values):
        return {'FID (mean)': values[0], 'FID (conf)': values[1]}


class DiversityEvaluator(BaseEvaluator):
    def __init__(self, data_len=0, evaluator_model=None, num_samples=300, batch_size=None, drop_last=False, replication_times=1,
                 replication_reduction='statistics', emb_scale=1, norm_scale=1, **kwargs):
        super().__init__(batch_size, drop_last, replication_times, replication_reduction, 0, data_len, evaluator_model)
        self.num_samples, self.emb_scale, self.norm_scale = num_samples, emb_scale, norm_scale

    def single_evaluate(self, results):
        return calculate_diversity(self._motion_emb(self.prepare_results(results)), self.num_samples, self.emb_scale, self.norm_scale)

    def parse_values(self, values):
        return {'Diversity (mean)': values[0], 'Diversity (conf)': values[1]}


class MultiModalityEvaluator(BaseEvaluator):
    def __init__(self, data_len=0, evaluator_model=None, num_samples=100, num_repeats=30, num_picks=10, batch_size=None,
                 drop_last=False, replication_times=1, replication_reduction='statistics', **kwargs):
        super().__init__(batch_size, drop_last, replication_times, replication_reduction, data_len,
                         data_len + num_samples * num_repeats, evaluator_model)
        self.num_samples, self.num_repeats, self.num_picks = num_samples, num_repeats, num_picks
        self.append_indexes = [np.repeat(np.random.choice(data_len, num_samples), num_repeats) for _ in range(replication_times)]

    def single_evaluate(self, results):
        emb = self._motion_emb(self.prepare_results(results))
        return calculate_multimodality(emb.reshape(self.num_samples, self.num_repeats, -1), self.num_picks)

    def parse_values(self, values):
        return {'MultiModality (mean)': values[0], 'MultiModality (conf)': values[1]}


EVALUATORS = Registry('evaluators')
for _name, _cls in (('R Precision', PrecisionEvaluator), ('Matching Score', MatchingScoreEvaluator), ('FID', FIDEvaluator),
                    ('Diversity', DiversityEvaluator), ('MultiModality', MultiModalityEvaluator)):
    EVALUATORS.register_module(name=_name, module=_cls)


def build_evaluator(metric, eval_cfg, data_len, eval_indexes):
    """core/evaluation/builder.py:22-33: eval_cfg + one entry of its ``metrics`` list -> evaluator; evaluators that need
    extra samples (MultiModality) extend the per-replication index lists."""
    cfg = copy.copy(dict(eval_cfg))
    cfg.update(metric)
    cfg.pop('metrics', None)
    cfg['data_len'], cfg['eval_indexes'] = data_len, eval_indexes
    ev = EVALUATORS.build(cfg)
    if ev.append_indexes is not None:
        for i in range(eval_cfg['replication_times']):
            eval_indexes[i] = np.concatenate((eval_indexes[i], ev.append_indexes[i]), axis=0)
    return ev, eval_indexes


class EvaluationPlan:
    """The evaluation flow a test-mode dataset drives (``BaseMotionDataset.prepare_evaluation`` / ``evaluate``,
    mogen/datasets/base_dataset.py:99-139), detached from the dataset classes: build the embedding model once, draw the
    per-replication sample orders, build one evaluator per entry of ``eval_cfg['metrics']`` (MultiModality extends the
    orders), and reduce a result list to the metrics dict.  ``eval_indexes`` is the order in which the sampler has to
    produce results (``dataset[i]`` of the reference returns sample ``eval_indexes[i]``)."""

    def __init__(self, eval_cfg, data_len):
        from .builder import build_submodule
        cfg = dict(eval_cfg)
        model = cfg.get('evaluator_model', None)
        if isinstance(model, dict):
            model = build_submodule(model).to('cuda').eval()
        cfg['evaluator_model'] = self.evaluator_model = model
        orders = []
        for _ in range(cfg['replication_times']):
            order = np.arange(data_len)
            if cfg.get('shuffle_indexes', False):
                np.random.shuffle(order)
            orders.append(order)
        self.evaluators = []
        for metric in cfg['metrics']:
            ev, orders = build_evaluator(metric, cfg, data_len, orders)
            self.evaluators.append(ev)
        self.eval_indexes = np.concatenate(orders)

    def evaluate(self, results):
        if results[0]['pred_motion'].shape[-1] == 322:
            # the SMPL-X evaluators score body + hands only: face / expression channels are taken from the ground truth
            for r in results:
                r['pred_motion'][:, 156:309] = r['motion'][:, 156:309]
                r['pred_motion'][:, 312:] = r['motion'][:, 312:]
        metrics = {}
        for ev in self.evaluators:
            metrics.update(ev.evaluate(results))
        return metrics


# ---- HumanML3D / KIT evaluator (mogen/models/rnns/t2m_bigru.py) ------------------------------------------------------
class NativeT2MEvaluator:
    """Device encoders of ``T2MContrastiveModel`` over the checkpoint's three state dicts flattened with their names as
    prefixes (``movement_encoder.*``, ``motion_encoder.*``, ``text_encoder.*``)."""

    def __init__(self, state_dict, input_size=263, movement_hidden_size=512, movement_latent_size=512, motion_hidden_size=1024,
                 motion_latent_size=512, word_size=300, pos_size=15, hidden_size=512, output_size=512):
        self.lib = _lib.load(require_gpu=True)
        sd = dict(state_dict)
        self.has_text = 'text_encoder.pos_emb.weight' in sd
        cfg = _lib.T2MEvalConfig()
        cfg.input_size, cfg.movement_hidden, cfg.movement_latent = input_size, movement_hidden_size, movement_latent_size
        cfg.motion_hidden, cfg.motion_latent = motion_hidden_size, motion_latent_size
        cfg.word_size, cfg.pos_size, cfg.text_hidden, cfg.text_out = (word_size if self.has_text else 0), pos_size, hidden_size, output_size
        self.cfg = cfg
        h = ctypes.c_void_p()
        _lib.check(self.lib.mc_t2meval_create(ctypes.byref(cfg), ctypes.byref(h)), 'mc_t2meval_create')
        self.handle = h
        for k, v in sd.items():
            if not k.startswith(('movement_encoder.', 'motion_encoder.', 'text_encoder.')):
                continue
            a = np.ascontiguousarray(torch.as_tensor(v).detach().cpu().float().numpy())
            _lib.check(self.lib.mc_t2meval_set_param(self.handle, k.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                       f'mc_t2meval_set_param({k})')
        _lib.check(self.lib.mc_t2meval_finalize(self.handle), 'mc_t2meval_finalize')

    def encode_motion(self, motion, motion_length):
        if not (motion.is_cuda and motion.dim() == 3 and motion.shape[2] == self.cfg.input_size):
            raise ValueError(f'motion must be a device tensor [B, T, {self.cfg.input_size}]')
        m = motion.to(torch.float32).contiguous()
        n = torch.as_tensor(motion_length).reshape(-1).to(device=m.device, dtype=torch.int32).contiguous()
        if n.numel() != m.shape[0] or int(n.min()) < 4 or int(n.max()) > m.shape[1]:
            raise ValueError('motion_length: one length in [4, T] per sample (lengths // 4 must be >= 1, like pack_padded_sequence)')
        out = torch.empty(m.shape[0], self.cfg.motion_latent, device=m.device, dtype=torch.float32)
        _lib.check(self.lib.mc_t2meval_encode_motion(self.handle, ctypes.c_void_p(m.data_ptr()), ctypes.c_void_p(n.data_ptr()),
                                                     m.shape[0], m.shape[1], ctypes.c_void_p(out.data_ptr()),
                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'mc_t2meval_encode_motion')
        return out

    def encode_word_vectors(self, word_emb, pos_onehot, sent_len):
        """word_emb [B, S, word_size], pos_onehot [B, S, pos_size], sent_len [B] (device) -> [B, output_size]."""
        if not self.has_text:
            raise RuntimeError('the evaluator was built without text_encoder.* weights')
        w = word_emb.to(torch.float32).contiguous()
        p = pos_onehot.to(device=w.device, dtype=torch.float32).contiguous()
        n = torch.as_tensor(sent_len).reshape(-1).to(device=w.device, dtype=torch.int32).contiguous()
        if not w.is_cuda or w.dim() != 3 or w.shape[2] != self.cfg.word_size or tuple(p.shape) != (w.shape[0], w.shape[1], self.cfg.pos_size):
            raise ValueError('word_emb [B, S, word_size] / pos_onehot [B, S, pos_size] device tensors expected')
        if n.numel() != w.shape[0] or int(n.min()) < 1 or int(n.max()) > w.shape[1]:
            raise ValueError('sent_len: one length in [1, S] per sentence')
        out = torch.empty(w.shape[0], self.cfg.text_out, device=w.device, dtype=torch.float32)
        _lib.check(self.lib.mc_t2meval_encode_text(self.handle, ctypes.c_void_p(w.data_ptr()), ctypes.c_void_p(p.data_ptr()),
                                                   ctypes.c_void_p(n.data_ptr()), w.shape[0], w.shape[1], ctypes.c_void_p(out.data_ptr()),
                                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                   'mc_t2meval_encode_text')
        return out

    def close(self):
        if getattr(self, 'handle', None):
            self.lib.mc_t2meval_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def vectorize_tokens(token, w_vectorizer, max_text_len):
    """T2MTextEncoder.forward's host part (t2m_bigru.py:131-165): ``token`` strings of 'word/POS' items -> padded word
    vectors, part-of-speech one-hots and sentence lengths.  ``w_vectorizer['word/POS']`` returns (vector, one-hot) -- the
    GloVe-backed ``WordVectorizer`` of the dataset side."""
    words, poss, lens = [], [], []
    for sent in token:
        items = sent.split(' ')
        if len(items) < max_text_len:
            items = ['sos/OTHER'] + items + ['eos/OTHER']
            n = len(items)
            items += ['unk/OTHER'] * (max_text_len + 2 - n)
        else:
            items = ['sos/OTHER'] + items[:max_text_len] + ['eos/OTHER']
            n = len(items)
        pairs = [w_vectorizer[it] for it in items]
        words.append(np.stack([np.asarray(p[0], dtype=np.float32) for p in pairs]))
        poss.append(np.stack([np.asarray(p[1], dtype=np.float32) for p in pairs]))
        lens.append(n)
    return torch.from_numpy(np.stack(words)), torch.from_numpy(np.stack(poss)), torch.tensor(lens, dtype=torch.int32)


@SUBMODULES.register_module()
class T2MContrastiveModel:
    """``evaluator_model=dict(type='T2MContrastiveModel', motion_encoder=dict(input_size, movement_hidden_size, ...),
    text_encoder=dict(word_size, pos_size, hidden_size, output_size, max_text_len), init_cfg=dict(type='Pretrained',
    checkpoint=...))`` (configs/_base_/datasets/human_ml3d_bs128.py:42-58).  The checkpoint holds the three state dicts
    ``movement_encoder`` / ``motion_encoder`` / ``text_encoder`` (t2m_bigru.py:84-87,126-128).  ``w_vectorizer``: the
    dataset side's word-vector lookup (the reference builds ``WordVectorizer('./data/glove', 'our_vab')`` itself)."""

    def __init__(self, motion_encoder=None, text_encoder=None, init_cfg=None, state_dict=None, w_vectorizer=None):
        me, te = dict(motion_encoder or {}), dict(text_encoder or {})
        if state_dict is None:
            assert init_cfg is not None and init_cfg['type'] == 'Pretrained'
            ck = torch.load(init_cfg['checkpoint'], map_location='cpu', weights_only=False)
            state_dict = {f'{part}.{k}': v for part in ('movement_encoder', 'motion_encoder', 'text_encoder') if part in ck
                          for k, v in ck[part].items()}
        self.max_text_len = te.pop('max_text_len', 20)
        self.w_vectorizer = w_vectorizer
        self.encoder = NativeT2MEvaluator(state_dict, **me, **te)

    def to(self, device):
        return self

    def eval(self):
        return self

    def encode_motion(self, motion, motion_length=None, motion_mask=None, **kwargs):
        if motion_length is None:
            motion_length = [motion.shape[1]] * motion.shape[0]
        return self.encoder.encode_motion(motion, motion_length)

    def encode_text(self, text, token=None, device=None, **kwargs):
        if 'word_emb' in kwargs:
            w, p, n = kwargs['word_emb'], kwargs['pos_onehot'], kwargs['sent_len']
        else:
            if self.w_vectorizer is None:
                raise RuntimeError('encode_text needs the word-vector lookup: pass w_vectorizer= (GloVe WordVectorizer of the dataset '
                                   'side) at construction, or word_emb / pos_onehot / sent_len tensors')
            w, p, n = vectorize_tokens(token, self.w_vectorizer, self.max_text_len)
        device = device if device is not None else 'cuda'
        return self.encoder.encode_word_vectors(w.to(device), p.to(device), n.to(device))
