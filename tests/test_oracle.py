"""CPU: the oracle restatement against the committed golden vectors (generated from the
reference's own modules by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import stmogen_oracle as O, tutel_restated as TR, weights as W
from helpers import CTRL, CTRL_COPY, CTRL_FEATS, SMALL, SMALL_SEED, FULL, load, synth_inputs


def T_(a):
    return torch.from_numpy(np.asarray(a))


def maxabs(a, b):
    return float((a.double() - b.double()).abs().max())


def test_schedule_tables_match_reference():
    g = load('schedules.npz')
    for tag, respace in (('ddim50', '15,15,8,6,6'), ('ddpm1000', None)):
        s = O.Schedule(1000, respace)
        assert list(g[f'{tag}.timestep_map']) == list(s.timestep_map)
        for name in ('betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_recip_alphas_cumprod',
                     'sqrt_recipm1_alphas_cumprod', 'posterior_mean_coef1', 'posterior_mean_coef2',
                     'model_log_variance'):
            assert np.array_equal(g[f'{tag}.{name}'], getattr(s, name)), (tag, name)
    s = O.Schedule(1000, '15,15,8,6,6')
    assert s.num_timesteps == 50 and s.timestep_map[:5] == [0, 14, 28, 43, 57]
    assert s.timestep_map[-5:] == [840, 880, 919, 959, 999]


def test_small_denoiser_modules():
    g = load('small_modules.npz')
    sd = W.make_state_dict(SMALL, SMALL_SEED)
    cap = {}
    x0 = O.denoise(sd, SMALL, T_(g['x_t']), int(g['t']), T_(g['xf_out']), T_(g['motion_mask']), cap=cap)
    assert float((x0 - T_(g['x0'])).abs().max()) <= 1e-5
    assert float((cap['emb'] - T_(g['emb'])).abs().max()) <= 1e-5
    assert float((cap['out2'] - T_(g['out2'])).abs().max()) <= 1e-5
    for i in range(SMALL['NL']):
        for k in ('motion_feat', 'text_feat', 'after_stma', 'after_ffn'):
            assert float((cap[f'layer{i}'][k] - T_(g[f'layer{i}.{k}'])).abs().max()) <= 1e-5, (i, k)
    # the fixture exercises capacity overflow (dropped second choices)
    assert int(g['layer0.dropped'][1]) > 0 and int(g['layer1.dropped'][1]) > 0


def test_text_hoist_is_identical():
    g = load('small_modules.npz')
    sd = W.make_state_dict(SMALL, SMALL_SEED)
    xf = T_(g['xf_out'])
    tf = O.precompute_text(sd, xf, SMALL)
    a = O.denoise(sd, SMALL, T_(g['x_t']), int(g['t']), xf, T_(g['motion_mask']), text_feats=tf)
    assert torch.equal(a, O.denoise(sd, SMALL, T_(g['x_t']), int(g['t']), xf, T_(g['motion_mask'])))


def test_small_ddim_trajectory():
    g = load('small_ddim.npz')
    sd = W.make_state_dict(SMALL, SMALL_SEED)
    traj = []
    torch.manual_seed(int(g['noise_seed']))
    out = O.sample_loop(sd, SMALL, O.Schedule(1000, '15,15,8,6,6'), 'ddim', T_(g['x_T']), T_(g['xf_out']),
                        T_(g['motion_mask']), trajectory=traj)
    assert float((out - T_(g['final'])).abs().max()) <= 1e-5
    for n, ref in zip(range(9, 50, 10), g['traj']):
        assert float((traj[n][1] - T_(ref)).abs().max()) <= 1e-5


def test_small_ddpm_truncated():
    g = load('small_ddpm.npz')
    sd = W.make_state_dict(SMALL, SMALL_SEED)
    traj = []
    torch.manual_seed(int(g['noise_seed']))
    O.sample_loop(sd, SMALL, O.Schedule(1000, None), 'ddpm', T_(g['x_T']), T_(g['xf_out']),
                  T_(g['motion_mask']), num_steps=20, trajectory=traj)
    for n, ref in zip(range(4, 20, 5), g['traj']):
        assert float((traj[n][1] - T_(ref)).abs().max()) <= 1e-5


def test_pre_seq_and_transl_req_seeding_against_reference_golden():
    """p_sample :664-674 / ddim_sample :816-820: the oracle's seeded loops vs the reference's own (preseq_small.npz)."""
    g = load('preseq_small.npz')
    sd = W.make_state_dict(SMALL, SMALL_SEED)
    x_T, xf, mask, pre = T_(g['x_T']), T_(g['xf_out']), T_(g['motion_mask']), T_(g['pre_seq'])
    transl = [[int(r[0]), float(r[1]), float(r[2])] for r in g['transl_req']]
    traj = []
    torch.manual_seed(int(g['ddpm_seed']))
    O.sample_loop(sd, SMALL, O.Schedule(1000, None), 'ddpm', x_T, xf, mask, num_steps=12, trajectory=traj, pre_seq=pre,
                  transl_req=transl)
    for n, ref in zip(range(3, 12, 4), g['ddpm_traj']):
        assert float((traj[n][1] - T_(ref)).abs().max()) <= 1e-5
    torch.manual_seed(int(g['ddim_seed']))
    out = O.sample_loop(sd, SMALL, O.Schedule(1000, '15,15,8,6,6'), 'ddim', x_T, xf, mask, pre_seq=pre)
    assert float((out - T_(g['ddim_final'])).abs().max()) <= 1e-5


def test_moe_tie_policy_switch():
    """a16: tutel ranks tokens with `importance_scores.argsort(dim=0)` (not stable).  Both orders of exactly tied tokens
    sit behind one switch; away from ties the policies agree."""
    torch.manual_seed(1)
    E, D, N = 4, 8, 64
    x = torch.randn(N // 2, D).repeat(2, 1)                       # exact duplicates: every token is tied with its twin
    pw, pb, sim, temp = torch.randn(256, D), torch.randn(256) * 0.1, torch.randn(256, E), torch.tensor([0.7])
    w1, b1 = torch.randn(E, 4 * D, D) * 0.3, torch.randn(E, 4 * D) * 0.1
    w2, b2 = torch.randn(E, 4 * D, D) * 0.3, torch.randn(E, D) * 0.1
    _, rs = TR.moe_forward(x, pw, pb, sim, temp, w1, b1, w2, b2, return_routing=True)
    TR.TIE_POLICY = 'reverse'
    try:
        _, rr = TR.moe_forward(x, pw, pb, sim, temp, w1, b1, w2, b2, return_routing=True)
        xu = torch.randn(N, D)                                    # no ties: same routing under both policies
        yu_r, ru_r = TR.moe_forward(xu, pw, pb, sim, temp, w1, b1, w2, b2, return_routing=True)
    finally:
        TR.TIE_POLICY = 'stable'
    yu_s, ru_s = TR.moe_forward(xu, pw, pb, sim, temp, w1, b1, w2, b2, return_routing=True)
    for i in range(N // 2):
        assert int(rs['locations'][0][i]) < int(rs['locations'][0][i + N // 2])
        assert int(rr['locations'][0][i]) > int(rr['locations'][0][i + N // 2])
    assert all(torch.equal(a, b) for a, b in zip(ru_s['locations'], ru_r['locations'])) and torch.equal(yu_s, yu_r)
    with pytest.raises(ValueError):
        TR.TIE_POLICY = 'nope'
        try:
            TR.moe_forward(xu, pw, pb, sim, temp, w1, b1, w2, b2)
        finally:
            TR.TIE_POLICY = 'stable'


def _tutel_dumps():
    import glob
    return sorted(glob.glob(os.path.join(os.path.dirname(__file__), 'golden', 'tutel_dump_*.npz')))


@pytest.mark.parametrize('path', _tutel_dumps(), ids=os.path.basename)
def test_tutel_moe_dump_fixture(path):
    """a16 pin, ready for a REAL tutel dump: tools/dump_tutel_moe.py (run where tutel is installed) writes
    {x, weights, scores, indices, locations, gates, y, capacity, source} of one moe_layer call into
    tests/golden/tutel_dump_<name>.npz; this test replays it through oracle/tutel_restated.py and reports which tie
    policy matches.  `tutel_dump_restated.npz` (source='oracle/tutel_restated.py', NOT tutel) only exercises the
    loader; with it alone the boundary stays PARITY UNPINNED."""
    g = np.load(path)
    t = lambda k: torch.from_numpy(np.asarray(g[k]))
    args = (t('x'), t('proj_w'), t('proj_b'), t('sim_matrix'), t('temperature'), t('fc1_w'), t('fc1_b'), t('fc2_w'), t('fc2_b'))
    kw = dict(top_k=int(g['top_k']), capacity_factor=float(g['capacity_factor']), batch_prioritized_routing=bool(g['bpr']))
    matched = []
    for pol in ('stable', 'reverse'):
        TR.TIE_POLICY = pol
        try:
            y, r = TR.moe_forward(*args, return_routing=True, **kw)
        finally:
            TR.TIE_POLICY = 'stable'
        assert int(r['capacity']) == int(g['capacity'])
        assert float((r['scores'] - t('scores')).abs().max()) <= 1e-5
        ok_idx = all(torch.equal(a, b.long()) for a, b in zip(r['indices'], t('indices')))
        keep_ref = t('locations') < int(g['capacity'])
        ok_keep = all(torch.equal(a, b) for a, b in zip(r['keeps'], keep_ref))
        if ok_idx and ok_keep and float((y - t('y')).abs().max()) <= 1e-4:
            matched.append(pol)
    print(f'{os.path.basename(path)} (source: {str(g["source"])}): matching tie policies {matched}')
    assert matched, 'the restated tutel semantics do not reproduce this dump under either tie policy'


def test_clip_text_tower_against_the_huggingface_implementation_golden():
    """Stage B of encode_text (the un-vendored `clip` package's text transformer): oracle/text_encoder_oracle.py vs the
    features of transformers.CLIPTextModel on the same (seed, key) weights (clip_tower_hf.npz, made by make_golden.py)."""
    from oracle import text_encoder_oracle as TO
    g = load('clip_tower_hf.npz')
    shapes = W.text_encoder_param_shapes(256, 2, 2048, clip_width=int(g['width']), clip_layers=int(g['layers']),
                                         clip_ff=int(g['ff']), vocab=int(g['vocab']))
    sd = W.make_text_encoder_state(shapes, seed=int(g['seed']))
    feat = TO.clip_text_features(sd, torch.from_numpy(g['tokens']), int(g['layers']), heads=int(g['heads']))
    assert float((feat - T_(g['feat'])).abs().max()) <= 1e-5


def test_full_size_denoise_against_golden():
    g = load('full_denoise.npz')
    sd = W.make_state_dict(FULL, 0)
    x_T, xf, mask = synth_inputs(FULL, 1, 196, int(g['input_seed']))
    x0 = O.denoise(sd, FULL, x_T, 999, xf, mask)
    assert float((x0 - T_(g['x0_t999'])).abs().max()) <= 1e-4
    _, _, mask2 = synth_inputs(FULL, 1, 196, int(g['input_seed']), lengths=[150])
    x0 = O.denoise(sd, FULL, x_T, 500, xf, mask2)
    assert float((x0 - T_(g['x0_t500_len150'])).abs().max()) <= 1e-4


def test_moe_capacity_and_ties():
    """tutel boundary (a16): capacity formula, BPR drop order, stable tie order for duplicated tokens."""
    assert TR.capacity_of(301056, 16, 2, 1.5) == 56448
    assert TR.capacity_of(4, 16, 2, 1.5) == 2
    torch.manual_seed(0)
    E, D, N = 4, 8, 64
    x = torch.randn(N // 2, D).repeat(2, 1)                       # CFG-style exact duplicates
    pw, pb = torch.randn(256, D), torch.randn(256) * 0.1
    sim, temp = torch.randn(256, E), torch.tensor([0.7])
    w1, b1 = torch.randn(E, 4 * D, D) * 0.3, torch.randn(E, 4 * D) * 0.1
    w2, b2 = torch.randn(E, 4 * D, D) * 0.3, torch.randn(E, D) * 0.1
    y, r = TR.moe_forward(x, pw, pb, sim, temp, w1, b1, w2, b2, return_routing=True)
    cap = r['capacity']
    assert cap == 2 * int(1.5 * 16)
    # brute-force the spec: rank by (-max score, index) among same expert, second choices offset
    imp = r['scores'].max(1)[0]
    order = sorted(range(N), key=lambda i: (-float(imp[i]), i))
    cnt0 = [0] * E
    loc = [[None, None] for _ in range(N)]
    for i in order:
        e = int(r['indices'][0][i]); loc[i][0] = cnt0[e]; cnt0[e] += 1
    cnt1 = list(cnt0)
    for i in order:
        e = int(r['indices'][1][i]); loc[i][1] = cnt1[e]; cnt1[e] += 1
    for k in range(2):
        assert [int(v) for v in r['locations'][k]] == [loc[i][k] for i in range(N)]
    # duplicates: the lower index ranks first
    for i in range(N // 2):
        assert int(r['locations'][0][i]) < int(r['locations'][0][i + N // 2])
    # direct evaluation
    yy = torch.zeros(N, D)
    for i in range(N):
        for k in range(2):
            if loc[i][k] < cap:
                e = int(r['indices'][k][i])
                h = torch.nn.functional.gelu(w1[e] @ x[i] + b1[e])
                yy[i] += r['gates'][k][i] * (h @ w2[e] + b2[e])
    assert float((y - yy).abs().max()) < 1e-4


def test_control_branch_against_reference_golden():
    g = load('control_small.npz')
    sd = W.make_state_dict(CTRL, SMALL_SEED, shapes=W.control_param_shapes(CTRL, CTRL_COPY, CTRL_FEATS))
    for t in (640, 3):
        out = O.denoise_control(sd, CTRL, T_(g['x_t']), t, T_(g['xf_out']), T_(g['motion_mask']), T_(g['c']), CTRL_COPY)
        assert float((out - T_(g[f'x0_t{t}'])).abs().max()) <= 1e-5
    # the condition must matter (otherwise the zero-init projections would make the test vacuous)
    assert float((T_(g['x0_t640']) - T_(g['x0_noc_t640'])).abs().max()) > 0.1


def test_skeleton_part_layouts_against_reference_golden():
    """human_ml3d (263-d) and kit_ml (251-d) 8-part PoseEncoder/PoseDecoder + the T2M_humanml3d.py architecture."""
    from helpers import HML_FULL, HML_SMALL, KIT_SMALL
    g = load('skeleton_parts.npz')
    for tag, dims in (('hml', HML_SMALL), ('kit', KIT_SMALL)):
        sd = W.make_state_dict(dims, SMALL_SEED)
        x, xf, mask = T_(g[f'{tag}_x_t']), T_(g[f'{tag}_xf_out']), T_(g[f'{tag}_motion_mask'])
        for t in (901, 12):
            assert maxabs(O.denoise(sd, dims, x, t, xf, mask), T_(g[f'{tag}_x0_t{t}'])) <= 1e-5
    sd = W.make_state_dict(HML_FULL, 0)
    x, xf, mask = synth_inputs(HML_FULL, 1, 196, seed=32, lengths=[163])
    assert maxabs(O.denoise(sd, HML_FULL, x, 500, xf, mask), T_(g['hmlfull_x0_t500_len163'])) <= 1e-5


def test_repaint_mode_against_reference_golden():
    """outpainting DDIM loop (resampling jumps 3 x 5) -- oracle vs the reference's ddim_sample_loop."""
    g = load('repaint_small.npz')
    sd = W.make_state_dict(SMALL, SMALL_SEED)
    x_T, xf, mask = T_(g['x_T']), T_(g['xf_out']), T_(g['motion_mask'])
    gt, keep, ov = T_(g['gt']), T_(g['keep']), int(g['overlap_len'])
    gen = torch.Generator().manual_seed(int(g['noise_seed']))
    draws = (torch.randn(x_T.shape, generator=gen) for _ in range(10 ** 6))
    out = O.sample_loop_repaint(sd, SMALL, O.Schedule(1000, '15,15,8,6,6'), x_T, xf, mask, keep, gt, draws, ov, 50,
                                no_resample=True)
    assert maxabs(out, T_(g['final_noresample'])) <= 1e-5
    assert maxabs(out[:, 0], gt[:, 0]) == 0.0        # frame 0 of the kept region: pure gt at alpha_bar_prev = 1


def test_wav_encoder_against_reference_golden():
    """oracle/wav_encoder_oracle.py vs the reference WavEncoder class (eval-mode BatchNorm)."""
    from oracle import wav_encoder_oracle as WO
    g = load('wav_encoder.npz')
    sd = W.make_wav_encoder_state(64, 2, seed=int(g['seed']))
    assert maxabs(WO.wav_encoder(sd, T_(g['wav'])), T_(g['out'])) <= 1e-6


def test_text_encoder_against_reference_golden():
    """oracle/text_encoder_oracle.py stage A vs the reference's encode_text(text, clip_feat)."""
    from oracle import text_encoder_oracle as TO
    g = load('text_encoder.npz')
    sd = W.make_text_encoder_state(W.text_encoder_param_shapes(256, 2, 2048), seed=int(g['seed']))
    assert maxabs(TO.finetune_encoder(sd, T_(g['clip_feat']), 2), T_(g['xf_out'])) <= 1e-5


def test_evaluation_encoders_against_reference_golden():
    """oracle restatement of ActorAgnosticEncoder / DistilbertActorAgnosticEncoder vs the reference outputs."""
    from oracle import eval_encoder_oracle as EO
    from helpers import EVAL_BERT, EVAL_DIMS
    g = load('evaluator.npz')
    shapes = W.eval_encoder_param_shapes(bert=dict(EVAL_BERT, vocab_size=len(g['vocab'])), **EVAL_DIMS)
    sd = W.make_eval_encoder_state(shapes, seed=int(g['seed']))
    om = EO.encode_motion(sd, T_(g['motion']), g['lengths'].tolist(), EVAL_DIMS['num_layers'], EVAL_DIMS['num_heads'])
    ot = EO.encode_text_tokens(sd, torch.from_numpy(g['input_ids']).long(), torch.from_numpy(g['attention_mask']),
                               EVAL_BERT['n_layers'], EVAL_BERT['n_heads'], EVAL_DIMS['num_layers'], EVAL_DIMS['num_heads'])
    assert maxabs(om, T_(g['motion_mu'])) <= 1e-5 and maxabs(ot, T_(g['text_mu'])) <= 1e-5
    # padded frames / padded tokens must not reach the embedding
    m2 = T_(g['motion']).clone()
    m2[2, 9:] = 123.0
    assert torch.equal(EO.encode_motion(sd, m2, g['lengths'].tolist(), EVAL_DIMS['num_layers'], EVAL_DIMS['num_heads'])[2], om[2])


def test_t2m_evaluator_against_reference_golden():
    """HumanML3D / KIT evaluator (t2m_bigru.py): explicit-loop GRU restatement vs the reference modules' outputs."""
    from oracle import t2m_eval_oracle as TO
    from helpers import T2M_DIMS, T2M_TEXT
    g = load('t2m_evaluator.npz')
    sd = W.make_t2m_eval_state(W.t2m_eval_param_shapes(**T2M_DIMS, **T2M_TEXT), seed=int(g['seed']))
    om = TO.encode_motion(sd, T_(g['motion']), torch.from_numpy(g['lengths']))
    ot = TO.encode_text(sd, T_(g['word_emb']), T_(g['pos_onehot']), torch.from_numpy(g['sent_len']))
    assert maxabs(om, T_(g['motion_emb'])) <= 1e-5 and maxabs(ot, T_(g['text_emb'])) <= 1e-5


def test_philox_restatement_against_the_published_known_answer_vectors():
    """oracle/philox_oracle.py (the CPU restatement of the library's device noise stream) vs the Random123 known-answer vectors
    of Philox4x32-10, and the moments of the Box-Muller normals built on it."""
    import numpy as np
    from oracle import philox_oracle as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = P.philox4x32_10(np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32))
        assert tuple(int(v) for v in got) == want
    # counter layout of a draw: block g of draw d under seed s = philox((g_lo, g_hi, d_lo, d_hi), (s_lo, s_hi))
    seed, d = 0x0123456789abcdef, (5 << 32) | 7
    b = P.draw_bits(12, seed, d)
    for g in range(3):
        want = P.philox4x32_10(np.array([g, 0, 7, 5], dtype=np.uint32), np.array([0x89abcdef, 0x01234567], dtype=np.uint32))
        assert (b[4 * g:4 * g + 4] == want).all()
    z = P.draw_normal(2_000_000, 99, 3, dtype=np.float64)
    assert abs(z.mean()) < 4e-3 and abs(z.var() - 1) < 6e-3 and abs((z ** 3).mean()) < 1.2e-2 and abs((z ** 4).mean() - 3) < 4e-2
    assert not np.array_equal(P.draw_bits(64, 99, 3), P.draw_bits(64, 99, 4)) and not np.array_equal(P.draw_bits(64, 99, 3), P.draw_bits(64, 98, 3))
