"""CPU: host logic of the drop-in boundary -- config loader, registry, schedule tables, weight
packing, and that the C-ABI library loads and exports every declared symbol (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import motioncraft_amd as mc
from motioncraft_amd import diffusion as D, lib, synthetic, weights
from helpers import FULL, SMALL, load

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF_CFG = '/root/reference/configs/stmogen'


def test_config_base_inheritance_and_access():
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    assert cfg.model.type == 'MotionDiffusion' and cfg.model['model']['type'] == 'STMoGenTransformer'
    assert cfg.data.samples_per_gpu == 4 and cfg.data.workers_per_gpu == 0          # recursive merge
    assert cfg.data.test.dataset_name == 'motionx' and cfg.dist_params.backend == 'nccl'
    assert cfg.get('copy_blocks_num', 7) == 7 and cfg.latent_dim == 32
    cfg.model['opt'] = {'same_overlap_noisy': False}
    assert cfg.model.opt.same_overlap_noisy is False
    cfg.merge_from_dict({'model.inference_type': 'ddpm', 'data.samples_per_gpu': 9})
    assert cfg.model.inference_type == 'ddpm' and cfg.data.samples_per_gpu == 9
    assert cfg.model.model.num_layers == 2
    with pytest.raises(AttributeError):
        cfg.no_such_key
    with pytest.raises(FileNotFoundError):
        mc.Config.fromfile(os.path.join(HERE, 'configs', 'missing.py'))


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason='reference tree only exists in the build container')
def test_reference_configs_load_and_build_unchanged():
    names = sorted(f for f in os.listdir(REF_CFG) if f.endswith('.py'))
    assert len(names) == 11
    for n in names:
        cfg = mc.Config.fromfile(os.path.join(REF_CFG, n))
        assert cfg.model.type == 'MotionDiffusion'
        arch = mc.build_architecture(cfg.model)
        if n == 'T2M_humanml3d.py':                     # 263-d HumanML3D, 8 parts x 64
            d = arch.model.dims
            assert (d['H'], d['L'], d['input_feats'], d['dataset']) == (8, 64, 263, 'human_ml3d')
        else:
            assert arch.model.dims['H'] == 12 and arch.model.dims['L'] in (64, 128)
        assert arch.diffusion_test.num_timesteps == 50 and arch.inference_type == 'ddim'


def test_registry_semantics():
    reg = mc.Registry('things')

    @reg.register_module()
    class A:
        def __init__(self, x, y=2):
            self.x, self.y = x, y
    assert reg.get('A') is A and 'A' in reg
    a = reg.build(dict(type='A', x=1))
    assert (a.x, a.y) == (1, 2)
    assert reg.build(None) is None
    with pytest.raises(KeyError):
        reg.build(dict(type='B'))
    with pytest.raises(KeyError):
        reg.build(dict(x=1))
    with pytest.raises(TypeError):
        reg.build([1, 2])
    with pytest.raises(KeyError):
        reg.register_module()(A)
    assert mc.MODELS is mc.ARCHITECTURES is mc.SUBMODULES is mc.ATTENTIONS is mc.LOSSES
    for n in ('MotionDiffusion', 'STMoGenTransformer', 'STMA', 'MSELoss'):
        assert mc.MODELS.get(n) is not None


def test_build_architecture_small_and_unsupported_options():
    cfg = mc.Config.fromfile(os.path.join(HERE, 'configs', 'stmogen_small.py'))
    arch = mc.build_architecture(cfg.model)
    assert arch.model.dims == dict(input_feats=322, max_seq_len=24, L=32, H=12, NL=2, F=64, Te=64, Dt=32, Nt=8,
                                   E=16, topk=2, dataset='motionx')
    assert arch.model.cfg_scale == 6.5
    with pytest.raises(NotImplementedError):
        arch.train()
    with pytest.raises(RuntimeError):               # text encoder: no weights loaded -> loud failure
        arch.model.get_precompute_condition(text=['a person walks'])
    assert arch.model.get_precompute_condition(xf_out='given')['xf_out'] == 'given'
    with pytest.raises(RuntimeError):               # no weights loaded -> loud failure, no fallback
        arch.model.native
    bad = cfg.model.model.ca_block_cfg
    bad['gate_type'] = 'top'
    with pytest.raises(NotImplementedError):
        mc.build_attention(bad)


def test_schedule_tables_against_reference_golden():
    g = load('schedules.npz')
    base = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large')
    for tag, respace in (('ddim50', '15,15,8,6,6'), ('ddpm1000', None)):
        d = D.build_diffusion(dict(base, respace=respace))
        assert list(g[f'{tag}.timestep_map']) == list(d.timestep_map)
        for name in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_recip_alphas_cumprod',
                     'sqrt_recipm1_alphas_cumprod', 'posterior_mean_coef1', 'posterior_mean_coef2'):
            assert np.array_equal(g[f'{tag}.{name}'], getattr(d, name)), (tag, name)
        lv = g[f'{tag}.model_log_variance']
        for i in (0, 1, d.num_timesteps - 1):
            c = d.step_coefs(i, 'ddpm', 6.5)
            assert abs(c.log_var - np.float32(lv[i])) <= 1e-6 * abs(lv[i])
            assert c.c1 == np.float32(d.posterior_mean_coef1[i]) and c.nonzero == (0.0 if i == 0 else 1.0)
            t = d.timestep_map[i]
            assert c.text_coef == np.float32(1 + 6.5 * t / 1000) or abs(c.text_coef - (1 + 6.5 * t / 1000)) < 1e-6
    assert D.space_timesteps(1000, 'ddim50') == set(range(0, 1000, 20))
    with pytest.raises(ValueError):
        D.space_timesteps(10, '20')


def test_sampler_rejects_options_outside_the_path():
    d = D.build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                               model_var_type='fixed_large'))
    with pytest.raises(NotImplementedError):
        d.p_sample_loop(None, (1, 24, 322), clip_denoised=True, model_kwargs={})
    keep = torch.ones(1, 24, 322, dtype=torch.bool)
    with pytest.raises(ValueError):                 # outpainting mode needs opt (overlap_len, addBlend, ...)
        d.ddim_sample_loop(None, (1, 24, 322), clip_denoised=False, device='cpu',
                           model_kwargs={'y': {'outpainting_mask': keep, 'gt': torch.zeros(1, 24, 322)}})
    with pytest.raises(KeyError):                   # ddim_sample reads y['gt'] whenever the mask has a True
        d.ddim_sample_loop(None, (1, 24, 322), clip_denoised=False, device='cpu',
                           model_kwargs={'y': {'outpainting_mask': keep}})
    import types
    d2 = D.build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x',
                                model_var_type='fixed_large', respace='15,15,8,6,6'),
                           opt=types.SimpleNamespace(same_overlap_noisy=True))
    with pytest.raises(NotImplementedError):
        d2.ddim_sample_loop(None, (1, 24, 322), clip_denoised=False, model_kwargs={})
    with pytest.raises(NotImplementedError):
        D.build_diffusion(dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='epsilon',
                               model_var_type='fixed_large')).p_sample_loop(None, (1, 2, 3), clip_denoised=False)


def test_resampling_jump_schedule():
    """scheduler.py:178-208: starts 60 % into the DDIM schedule, +-1 moves only, every level below the top visited
    jump_n_sample times on the way down."""
    from oracle import stmogen_oracle as O
    for a in ((50,), (50, 3, 5), (25,), (25, 3, 2), (50, 1, 1), (10, 2, 3), (100, 3, 5), (20, 5, 4)):
        assert D.get_schedule_jump_cjm_ddim(*a) == O.jump_schedule(*a)
    ts = D.get_schedule_jump_cjm_ddim(50, 3, 5)
    assert ts[0] == 29 and ts[-2:] == [0, -1] and len(ts) == 247
    assert sum(1 for a, b in zip(ts[:-1], ts[1:]) if b < a) == 30 + 9 * 4 * 3     # denoiser calls
    assert D.get_schedule_jump_cjm_ddim(50) == list(range(29, -2, -1))
    assert D.get_schedule_jump_cjm_ddim(25)[0] == 14


def test_gaussian_taps_match_scipy():
    """postprocess.gaussian_taps == the kernel scipy.ndimage.gaussian_filter correlates with (delta response)."""
    from scipy.ndimage import gaussian_filter
    from motioncraft_amd.postprocess import gaussian_taps
    for sigma in (1.0, 2.0, 2.5, 3.0, 3.5):
        r, w = gaussian_taps(sigma)
        d = np.zeros(4 * r + 1)
        d[2 * r] = 1.0
        resp = gaussian_filter(d, sigma=sigma, mode='nearest')
        assert r == int(4 * sigma + 0.5) and np.abs(resp[2 * r - r:2 * r + r + 1] - w).max() <= 1e-15
        assert resp[2 * r - r - 1] == 0.0


def test_wav_encoder_bn_folding_and_tap_major_layout():
    """The packed GEMM weight applied to the im2col VIEW of a channels-last, zero-padded signal == conv1d + eval BN."""
    import torch.nn.functional as F
    from motioncraft_amd import wav_encoder as WE
    sd = synthetic.make_wav_encoder_state(64, 2, seed=1)
    packed = WE.pack_wav_encoder({'feat_extractor.' + k: v for k, v in sd.items()}, prefix='feat_extractor.')
    assert set(packed) == {f'b{i}.{c}.{t}' for i in range(6) for c in ('conv1', 'conv2') for t in 'wb'} | \
        {f'b{i}.down.{t}' for i in WE.BLOCKS_WITH_DOWNSAMPLE for t in 'wb'}
    assert packed['b0.conv1.w'].shape == (16, 32) and packed['b1.conv1.w'].shape == (16, 240)     # 15*2 -> 32, 15*16
    x = torch.randn(1, 16, 50)                                   # block 1: Cin=16, stride 6, pad 0
    ref = F.batch_norm(F.conv1d(x, sd['1.conv1.weight'], sd['1.conv1.bias'], stride=6), sd['1.bn1.running_mean'],
                       sd['1.bn1.running_var'], sd['1.bn1.weight'], sd['1.bn1.bias'], training=False, eps=1e-5)
    xl = x[0].T.contiguous().reshape(-1)                         # channels-last [T, C] flattened
    T1 = (50 - 15) // 6 + 1
    rows = torch.stack([xl[t * 6 * 16:t * 6 * 16 + 240] for t in range(T1)])       # overlapping-row view, row stride 6*16
    out = rows @ packed['b1.conv1.w'].T + packed['b1.conv1.b']
    assert torch.allclose(out.T, ref[0], atol=1e-5)


def test_weight_packing_layouts():
    sd = synthetic.make_state_dict(SMALL, 0)
    p = weights.pack_state_dict({'model.' + k: v for k, v in sd.items()}, SMALL)   # checkpoint prefix stripped
    L, H, C = SMALL['L'], SMALL['H'], 322
    assert p['enc.w'].shape == (L * H, 352) and p['dec.w'].shape == (C, L * H)
    sl = synthetic.smplx_part_slices()
    x = torch.randn(3, C)
    # dense encoder == per-part linears (reference stmogen.py:336-353)
    dense = x @ torch.from_numpy(p['enc.w'][:, :C]).t() + torch.from_numpy(p['enc.b'])
    for i, n in enumerate(synthetic.PART_NAMES):
        ref = torch.nn.functional.linear(x[:, sl[n]], sd[f'joint_embed.{n}_embed.weight'], sd[f'joint_embed.{n}_embed.bias'])
        assert torch.allclose(dense[:, i * L:(i + 1) * L], ref, atol=1e-5)
    body = [c for n in synthetic.PART_NAMES for c in sl[n]]
    ref = torch.nn.functional.linear(x[:, body], sd['joint_embed.body_embed.weight'], sd['joint_embed.body_embed.bias'])
    assert torch.allclose(dense[:, (H - 1) * L:], ref, atol=1e-5)
    assert np.all(p['enc.w'][:, C:] == 0)
    # every 322 channel is owned by exactly one part
    assert sorted(body) == list(range(C))
    assert p['l0.mm.fc2_wt'].shape == (16, L, 4 * L) and p['l0.tm.fc1_w'].shape == (16, 4 * SMALL['Dt'], SMALL['Dt'])
    assert np.allclose(np.linalg.norm(p['l1.mm.sim_n'], axis=0), 1.0, atol=1e-6)
    assert np.allclose(p['l0.body_wsm'].sum(1), 1.0, atol=1e-6)
    assert p['l0.ffn.w1'].shape == (H, SMALL['F'], L) and p['l0.dyn.qkv_w'].shape == (3 * L, L)
    # full-size parameter budget of SURVEY.md: 127.9 M
    n = sum(int(np.prod(s)) for s in synthetic.param_shapes(FULL).values())
    assert abs(n / 1e6 - 127.9) < 0.1


def test_skeleton_part_layouts_and_packing():
    """8-part layouts: every channel of the 263-d / 251-d vector is owned by exactly one
    part, body_embed sees joints in index order, and the dense packing reproduces the part-wise encoder/decoder."""
    for dataset, C, nj in (('human_ml3d', 263, 22), ('kit_ml', 251, 21)):
        names, sl, body = synthetic.part_layout(dataset)
        assert names == ['head', 'stem', 'larm', 'rarm', 'lleg', 'rleg', 'root']
        owned = sorted(c for n in names for c in sl[n])
        assert owned == list(range(C)) and sorted(body) == list(range(C))
        assert body[:11] == [0, 1, 2, 3, 4 + 9 * (nj - 1), 5 + 9 * (nj - 1), 6 + 9 * (nj - 1), C - 4, C - 3, C - 2, C - 1]
        assert len(sl['root']) == 11 and all(len(sl[n]) % 12 == 0 for n in names[:-1])
        dims = synthetic.humanml3d_dims(max_seq_len=24, L=32, NL=1, F=64, Te=64, Dt=32, Nt=8, input_feats=C, dataset=dataset)
        sd = synthetic.make_state_dict(dims, 0)
        p = weights.pack_state_dict(sd, dims)
        L, H = 32, 8
        assert p['enc.w'].shape == (L * H, (C + 31) // 32 * 32) and p['dec.w'].shape == (C, L * H)
        x = torch.randn(3, C)
        feats = [x[:, sl[n]] @ sd[f'joint_embed.{n}_embed.weight'].T + sd[f'joint_embed.{n}_embed.bias'] for n in names]
        feats.append(x[:, body] @ sd['joint_embed.body_embed.weight'].T + sd['joint_embed.body_embed.bias'])
        assert torch.allclose(torch.cat(feats, 1), x @ p['enc.w'][:, :C].T + p['enc.b'], atol=1e-5)
        h = torch.randn(3, L * H)
        out = torch.zeros(3, C)
        for i, n in enumerate(names):
            out[:, sl[n]] = h[:, i * L:(i + 1) * L] @ sd[f'out.{n}_out.weight'].T + sd[f'out.{n}_out.bias']
        out = (out + h[:, 7 * L:] @ sd['out.body_out.weight'].T + sd['out.body_out.bias']) / 2
        assert torch.allclose(out, h @ p['dec.w'].T + p['dec.b'], atol=1e-5)
    with pytest.raises(NotImplementedError):
        synthetic.part_layout('openpose17')


def test_checkpoint_formats(tmp_path):
    """load_checkpoint: mmcv-style {'state_dict', 'meta'} .pth with a DataParallel 'module.' prefix, flat .pth, .npz."""
    sd = {'model.a.weight': torch.arange(6.).reshape(2, 3), 'model.b': torch.ones(2)}

    class Sink:
        def load_state_dict(self, state_dict, strict=True):
            self.sd = state_dict
    torch.save({'state_dict': {'module.' + k: v for k, v in sd.items()}, 'meta': {'epoch': 3}}, tmp_path / 'a.pth')
    torch.save(sd, tmp_path / 'b.pth')
    np.savez(tmp_path / 'c.npz', **{k: v.numpy() for k, v in sd.items()})
    for name in ('a.pth', 'b.pth', 'c.npz'):
        m = Sink()
        ck = mc.load_checkpoint(m, str(tmp_path / name), map_location='cpu')
        assert set(m.sd) == set(sd) and all(torch.equal(m.sd[k], sd[k]) for k in sd), name
        assert 'state_dict' in ck
    with pytest.raises(IOError):
        mc.load_checkpoint(Sink(), str(tmp_path / 'missing.pth'))


def test_c_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'motioncraft_amd.h')).read()
    declared = set(re.findall(r'\b(mc_[a-z_0-9]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(lib.EXPORTED_SYMBOLS), declared ^ set(lib.EXPORTED_SYMBOLS)
    l = lib.load(require_gpu=False)           # dlopen only, no compute
    for name in declared:
        assert hasattr(l, name), name
    assert ctypes.sizeof(lib.ModelConfig) == 17 * 4 and ctypes.sizeof(lib.StepCoefs) == 12 * 4
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):     # product path fails loudly without an MI355X
            lib.load(require_gpu=True)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'motioncraft_amd')
    for f in os.listdir(pkg):
        if f.endswith('.py'):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f


def test_wordpiece_tokenizer_matches_the_transformers_tokenizer_golden(tmp_path):
    """The evaluator's text side tokenises with the DistilBERT directory's vocab.txt (t2m_bigru_smplx.py:229,276)."""
    from motioncraft_amd.wordpiece import WordPieceTokenizer
    g = load('evaluator.npz')
    (tmp_path / 'vocab.txt').write_text('\n'.join(str(v) for v in g['vocab']) + '\n', encoding='utf-8')
    tok = WordPieceTokenizer(str(tmp_path))
    ids, mask = tok([str(t) for t in g['texts']])
    assert np.array_equal(ids, g['input_ids']) and np.array_equal(mask, g['attention_mask'])
    assert tok.encode('') == [tok.cls, tok.sep]
    assert tok.pieces('q' * 101) == [tok.unk] and tok.pieces('walks') == [tok.ids['walk'], tok.ids['##s']]


def test_evaluation_metric_functions_against_reference_golden():
    """mogen/core/evaluation/utils.py outputs on seeded float32 embeddings."""
    from motioncraft_amd import evaluation as E
    g = load('evaluator.npz')
    a, b = g['met_a'], g['met_b']
    dist = E.euclidean_distance_matrix(a, b)
    assert dist.dtype == np.float32 and np.array_equal(dist, g['met_dist'])
    assert np.array_equal(E.calculate_top_k(np.argsort(dist, axis=1), 3), g['met_topk'])
    mu1, c1 = E.calculate_activation_statistics(a, 1.0)
    mu2, c2 = E.calculate_activation_statistics(b + 0.3, 1.0)
    assert abs(E.calculate_frechet_distance(mu1, c1, mu2, c2) - float(g['met_fid'])) <= 1e-9
    np.random.seed(11)
    assert abs(E.calculate_diversity(a, 20, 1.0, 1.0) - float(g['met_div'])) <= 1e-12
    assert abs(E.calculate_multimodality(a.reshape(5, 8, 16), 4) - float(g['met_mm'])) <= 1e-12
    m, c = E.get_metric_statistics(np.array([1.0, 2.0, 4.0]), 3)
    assert abs(m - 7 / 3) < 1e-12 and abs(c - 1.96 * np.std([1.0, 2.0, 4.0]) / np.sqrt(3)) < 1e-12


def test_evaluators_host_logic_against_reference_golden():
    """The five evaluators (replication slicing, batching, z-scoring, statistics) driven by the same stub embedding
    model as the reference's own evaluators in tests/golden/make_golden.py; built through build_evaluator with the
    configs' eval_cfg layout (configs/_base_/datasets/motionx_bs128.py:34-57)."""
    from motioncraft_amd import evaluation as E
    from helpers import StubEvalModel, stub_eval_results
    g = load('evaluator.npz')
    N, REP = 48, 2
    eval_cfg = dict(shuffle_indexes=True, replication_times=REP, replication_reduction='statistics', evaluator_model=StubEvalModel(),
                    metrics=[dict(type='R Precision', batch_size=16, top_k=3), dict(type='Matching Score', batch_size=16),
                             dict(type='FID', emb_scale=1.0), dict(type='Diversity', num_samples=20),
                             dict(type='MultiModality', num_samples=4, num_repeats=5, num_picks=3)])
    idx = [np.arange(N) for _ in range(REP)]
    evs = []
    for metric in eval_cfg['metrics']:
        if metric['type'] == 'MultiModality':
            np.random.seed(23)
        ev, idx = E.build_evaluator(metric, eval_cfg, N, idx)
        evs.append(ev)
    assert np.array_equal(np.stack(evs[-1].append_indexes), g['ev_append']) and len(idx[0]) == N + 20
    results = stub_eval_results(N, REP, evs[-1].append_indexes)
    np.random.seed(29)
    metrics = {}
    for ev in evs:
        metrics.update(ev.evaluate(results))
    assert list(metrics) == [str(n) for n in g['ev_names']]
    got = np.array([float(v) for v in metrics.values()])
    assert np.allclose(got, g['ev_values'], rtol=1e-6, atol=1e-7), (got, g['ev_values'])


def test_evaluation_plan_orders_and_face_alignment():
    """BaseMotionDataset.prepare_evaluation / evaluate (base_dataset.py:99-139): per-replication orders (shuffled, MultiModality
    appends its repeats), one evaluator per metric, 322-d predictions get the ground-truth face channels before scoring."""
    from motioncraft_amd import evaluation as E
    from helpers import StubEvalModel
    N, REP = 40, 2
    cfg = dict(shuffle_indexes=True, replication_times=REP, replication_reduction='statistics', evaluator_model=StubEvalModel(nfeats=322),
               metrics=[dict(type='R Precision', batch_size=20, top_k=3), dict(type='FID'), dict(type='Diversity', num_samples=10),
                        dict(type='MultiModality', num_samples=3, num_repeats=4, num_picks=2)])
    np.random.seed(5)
    plan = E.EvaluationPlan(cfg, N)
    assert len(plan.evaluators) == 4 and plan.eval_indexes.shape == (REP * (N + 12),)
    first = plan.eval_indexes[:N]
    assert sorted(first.tolist()) == list(range(N)) and first.tolist() != list(range(N))
    g = torch.Generator().manual_seed(1)
    results = []
    for i in plan.eval_indexes:
        gt = torch.randn(10, 322, generator=g)
        results.append(dict(motion=gt, pred_motion=gt + torch.randn(10, 322, generator=g), motion_mask=torch.ones(10),
                            pred_motion_mask=torch.ones(10), motion_length=torch.tensor(10), pred_motion_length=torch.tensor(10),
                            text='sample %d' % int(i)))
    metrics = plan.evaluate(results)
    assert {'R_precision Top 3 (mean)', 'FID (mean)', 'FID (conf)', 'Diversity (mean)', 'MultiModality (mean)'} <= set(metrics)
    r = results[7]
    assert torch.equal(r['pred_motion'][:, 156:309], r['motion'][:, 156:309]) and torch.equal(r['pred_motion'][:, 312:], r['motion'][:, 312:])
    assert not torch.equal(r['pred_motion'][:, :156], r['motion'][:, :156])
    assert all(np.isfinite(float(v)) for v in metrics.values())


def test_clip_byte_pair_tokenizer_on_a_synthetic_merges_file(tmp_path):
    """clip_bpe.ClipBPE (restated clip.tokenize; the real vocabulary is un-vendored -> parity unpinned): id layout, merge
    order, byte fallback, clean-up, markers, padding and truncation on a hand-made merges file."""
    import gzip
    from motioncraft_amd.clip_bpe import ClipBPE, byte_symbols
    merges = ['#version: test', 'w a', 'l k</w>', 'wa lk</w>', 'r u', 'ru n</w>', 'a </w>']          # last line never applies
    path = tmp_path / 'bpe.txt.gz'
    with gzip.open(path, 'wb') as f:
        f.write('\n'.join(merges).encode('utf-8'))
    n = len(merges) - 1
    bpe = ClipBPE(str(path), vocab_size=256 + 256 + n + 2)
    sym = byte_symbols()
    assert len(set(sym.values())) == 256 and sym[ord('a')] == 'a' and sym[ord(' ')] == chr(256 + 32)
    assert bpe.ids['!'] == 0 and bpe.ids['!</w>'] == 256 and bpe.ids['wa'] == 512 and bpe.sot == 512 + n and bpe.eot == 513 + n
    # "walk" -> w a l k</w> -> wa l k</w> -> wa lk</w> -> walk</w>
    assert bpe.encode('walk') == [bpe.ids['walk</w>']]
    assert bpe.encode('  WALK \n run ') == [bpe.ids['walk</w>'], bpe.ids['run</w>']]
    # no merge for these symbols: one id per byte symbol, the last one tagged; punctuation splits off
    assert bpe.encode('ok!') == [bpe.ids['o'], bpe.ids['k</w>'], bpe.ids['!</w>']]
    e_acute = 'é'.encode('utf-8')
    assert bpe.encode('é') == [bpe.ids[sym[e_acute[0]]], bpe.ids[sym[e_acute[1]] + '</w>']]
    assert bpe.encode('a &amp;amp; 7') == [bpe.ids['a</w>'], bpe.ids['&</w>'], bpe.ids['7</w>']]
    t = bpe.tokenize(['walk', 'run ' * 100], context_length=8)
    assert t.dtype == np.int64 and t.shape == (2, 8)
    assert t[0].tolist() == [bpe.sot, bpe.ids['walk</w>'], bpe.eot, 0, 0, 0, 0, 0]
    assert t[1, 0] == bpe.sot and t[1, -1] == bpe.eot and (t[1, 1:-1] == bpe.ids['run</w>']).all()
    with pytest.raises(RuntimeError):
        bpe.tokenize(['run ' * 100], context_length=8, truncate=False)


def test_clip_bpe_vs_huggingface_tokenizer_on_a_learned_synthetic_vocabulary(tmp_path):
    """Second, independent implementation of the CLIP byte-pair scheme: transformers.CLIPTokenizer (the Rust `tokenizers`
    BPE model with the `</w>` suffix, ByteLevel alphabet and CLIP's split regex).  The real vocabulary file is
    un-vendored, so both are driven by the SAME synthetic vocabulary: 60 merges learned greedily on a toy corpus, ids in
    the layout clip/simple_tokenizer.py builds.  Ids must agree up to and including the end token (the pad id differs by
    design: clip.tokenize pads with 0, the HF tokenizer with its pad token)."""
    import gzip
    transformers = pytest.importorskip('transformers')
    from motioncraft_amd.clip_bpe import ClipBPE, byte_symbols
    corpus = ("a person walks forward then turns left and waves both hands . the dancer jumps , spins & lands softly ; "
              "it's running 12 times while they're clapping").split()
    sym = byte_symbols()
    words = [[sym[b] for b in w.encode()] for w in corpus]
    words = [w[:-1] + [w[-1] + '</w>'] for w in words]
    merges = []
    for _ in range(60):
        cnt = {}
        for w in words:
            for pr in zip(w[:-1], w[1:]):
                cnt[pr] = cnt.get(pr, 0) + 1
        if not cnt:
            break
        best = max(sorted(cnt), key=lambda pr: cnt[pr])
        merges.append(best)
        nw = []
        for w in words:
            o, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and (w[i], w[i + 1]) == best:
                    o.append(w[i] + w[i + 1])
                    i += 2
                else:
                    o.append(w[i])
                    i += 1
            nw.append(o)
        words = nw
    path = tmp_path / 'bpe.txt.gz'
    with gzip.open(path, 'wb') as f:
        f.write(('#version: test\n' + '\n'.join(' '.join(m) for m in merges)).encode('utf-8'))
    bpe = ClipBPE(str(path), vocab_size=256 + 256 + len(merges) + 2)
    hf = transformers.CLIPTokenizer(vocab=dict(bpe.ids), merges=[tuple(m) for m in merges])
    texts = ['A person walks forward, then turns LEFT!', "the dancer's hands... wave   softly\n12 times", 'naïve café 3x — ¿qué?',
             "they're running & it's   spinning;lands", 'x' * 300, 'walks ' * 90, '', '  ']
    for t in texts:
        a = bpe.tokenize([t], context_length=77)[0].tolist()
        b = hf(t, padding='max_length', max_length=77, truncation=True)['input_ids']
        n = a.index(bpe.eot) + 1
        assert a[:n] == b[:n], t
        assert all(v == 0 for v in a[n:])
    # what clip.tokenize does and the HF tokenizer does not: html entities are unescaped (twice) first
    assert bpe.encode('a &amp;amp; b') == bpe.encode('a & b') != hf('a &amp;amp; b')['input_ids'][1:-1]


def test_t2m_token_vectorisation_layout():
    """T2MTextEncoder.forward's host part (t2m_bigru.py:131-165): sos / eos / unk padding and truncation at max_text_len."""
    from motioncraft_amd.evaluation import vectorize_tokens

    class Lookup:
        def __getitem__(self, item):
            word, pos = item.split('/')
            return np.full(4, float(len(word))), np.eye(3)[0 if pos == 'OTHER' else 1]
    w, p, n = vectorize_tokens(['walk/VERB fast/ADV', ' '.join(['a/DET'] * 7)], Lookup(), max_text_len=5)
    assert w.shape == (2, 7, 4) and p.shape == (2, 7, 3) and n.tolist() == [4, 7]
    assert w[0, :, 0].tolist() == [3, 4, 4, 3, 3, 3, 3]            # sos walk fast eos unk unk unk
    assert w[1, :, 0].tolist() == [3, 1, 1, 1, 1, 1, 3]            # sos + 5 kept + eos
    assert p[0, 1].tolist() == [0, 1, 0] and p[0, 0].tolist() == [1, 0, 0]


def test_bench_self_launches_one_rank_per_gpu(monkeypatch):
    """`python bench.py --gpus N` (N > 1, no WORLD_SIZE in the env) re-executes itself under torch.distributed.run with one rank
    per GPU on 127.0.0.1 (the launcher role of tools/dist_train.sh:8-10 in front of mogen/apis/test.py:36-82)."""
    import importlib
    import sys
    sys.path.insert(0, ROOT)
    bench = importlib.import_module('bench')
    argv = bench.self_launch_argv(4, ['--gpus', '4', '--steps', '7'], port=29512)
    assert argv[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nproc-per-node=4' in argv and '--nnodes=1' in argv
    assert argv[argv.index('--master-addr') + 1] == '127.0.0.1' and argv[argv.index('--master-port') + 1] == '29512'
    assert argv[-5:] == [os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '7']
    seen = {}

    def fake_execv(exe, args):
        seen['exe'], seen['args'] = exe, args
        raise SystemExit(0)
    monkeypatch.setattr(os, 'execv', fake_execv)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--batch', '16'])
    with pytest.raises(SystemExit):
        bench.main()
    assert seen['exe'] == sys.executable and '--nproc-per-node=2' in seen['args'] and seen['args'][-4:] == ['--gpus', '2', '--batch', '16']
    # the workload string follows the real batch
    assert 'batch 16 per GPU' in bench.workload_name(16, 196) and bench.workload_name(16, 196).startswith('configs[1] at a NON-BASELINE')
    assert bench.workload_name(64, 196).startswith('configs[1]: ')
