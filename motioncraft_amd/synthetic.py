"""Synthetic, deterministic weights: every parameter is a function of (seed, state-dict key).

Data generation only (no arithmetic of the path): used by bench.py / tools for the random-init
weights of the named architecture, and re-exported by oracle/weights.py so the CPU oracle, the
reference modules (tests/golden/make_golden.py) and the HIP path all see identical weights.
``param_shapes`` is checked key-by-key against the reference's own ``state_dict()`` there.

Released checkpoints are not available offline and most output layers of the
reference are zero-initialised (``zero_module``: reference
``mogen/models/utils/stylization_block.py:26``, ``mogen/models/transformers/stmogen.py:641``),
so a freshly constructed model outputs exactly 0 and parity tests would be
vacuous.  Instead both sides (reference modules, oracle, HIP path) regenerate the
same non-trivial weights from the key names of SURVEY.md Appendix B; nothing is
committed.
"""
import math
import zlib
from collections import OrderedDict

import torch

PART_NAMES = ['head', 'stem', 'larm', 'rarm', 'lleg', 'rleg', 'root', 'trans', 'face', 'lhand', 'rhand']


def smplx_part_slices():
    """Channel lists of the 11 body parts of the 322-d SMPL-X vector, in PoseEncoder
    concat order (reference stmogen.py:53-68 ``get_smplx_slice``)."""
    j = lambda *ids: [3 * i + c for i in ids for c in range(3)]
    d = OrderedDict()
    d['head'] = j(12, 15) + [156, 157, 158]
    d['stem'] = j(3, 6, 9)
    d['larm'] = j(14, 17, 19, 21)
    d['rarm'] = j(13, 16, 18, 20)
    d['lleg'] = j(2, 5, 8, 11)
    d['rleg'] = j(1, 4, 7, 10)
    d['root'] = [0, 1, 2] + list(range(312, 322))
    d['trans'] = [309, 310, 311]
    d['face'] = list(range(159, 309))
    d['lhand'] = list(range(66, 111))
    d['rhand'] = list(range(111, 156))
    return d


def _hml_joint_channels(j, nj):
    """Channels of joint ``j`` in a HumanML3D-style vector of ``nj`` joints:
    [root 4 | ric (nj-1)*3 | rot6d (nj-1)*6 | vel nj*3 | foot 4]; the root joint owns the
    4 root scalars, its velocity and the foot contacts (reference stmogen.py:12-52)."""
    ric, rot, vel = 4, 4 + 3 * (nj - 1), 4 + 9 * (nj - 1)
    foot = vel + 3 * nj
    if j == 0:
        return [0, 1, 2, 3, vel, vel + 1, vel + 2] + list(range(foot, foot + 4))
    return ([ric + 3 * (j - 1) + c for c in range(3)] + [rot + 6 * (j - 1) + c for c in range(6)]
            + [vel + 3 * j + c for c in range(3)])


_SKELETON_PARTS = {
    # dataset -> (num joints, part -> joint ids)   (reference stmogen.py:188-209 PoseEncoder.__init__)
    'human_ml3d': (22, OrderedDict(head=[12, 15], stem=[3, 6, 9], larm=[14, 17, 19, 21], rarm=[13, 16, 18, 20],
                                   lleg=[2, 5, 8, 11], rleg=[1, 4, 7, 10], root=[0])),
    'kit_ml': (21, OrderedDict(head=[4], stem=[1, 2, 3], larm=[8, 9, 10], rarm=[5, 6, 7],
                               lleg=[16, 17, 18, 19, 20], rleg=[11, 12, 13, 14, 15], root=[0])),
}


def part_layout(dataset='motionx'):
    """(part names, part -> channel list, channel list fed to ``body_embed``) of a dataset's pose vector.
    motionx: 11 parts + body over the 322-d SMPL-X vector (body = the parts concatenated);
    human_ml3d / kit_ml: 7 parts + body over the 263-d / 251-d vector (body = joints in index order)."""
    if dataset == 'motionx':
        sl = smplx_part_slices()
        return list(PART_NAMES), sl, [c for n in PART_NAMES for c in sl[n]]
    if dataset in _SKELETON_PARTS:
        nj, parts = _SKELETON_PARTS[dataset]
        sl = OrderedDict((n, [c for j in js for c in _hml_joint_channels(j, nj)]) for n, js in parts.items())
        return list(parts), sl, [c for j in range(nj) for c in _hml_joint_channels(j, nj)]
    raise NotImplementedError(f"pose layout of dataset_name={dataset!r} is not on the MI355X path "
                              "(motionx, human_ml3d, kit_ml)")


def default_dims(**over):
    """0.125b config (reference configs/stmogen/T2M_motionx_align_Finedance_Beats2_face_no_loss_0_125b.py:26-83)."""
    d = dict(input_feats=322, max_seq_len=196, L=128, H=12, NL=4, F=512, Te=2048, Dt=256, Nt=77,
             E=16, topk=2, scale=6.5, dyn_heads=8, dataset='motionx')
    d.update(over)
    return d


def humanml3d_dims(**over):
    """reference configs/stmogen/T2M_humanml3d.py:26-83 (263-d HumanML3D, 8 parts x 64)."""
    return default_dims(**{**dict(input_feats=263, L=64, H=8, F=256, dataset='human_ml3d'), **over})


def param_shapes(dims):
    """name -> shape for the denoiser's state dict."""
    L, H, NL, F, Te, Dt, Nt, E = (dims[k] for k in ('L', 'H', 'NL', 'F', 'Te', 'Dt', 'Nt', 'E'))
    D, C, Tm = L * H, dims['input_feats'], dims['max_seq_len']
    names, sl, body = part_layout(dims.get('dataset', 'motionx'))
    assert H == len(names) + 1 and len(body) == C, (H, len(names), len(body), C)
    s = OrderedDict()
    s['sequence_embedding'] = (Tm, D)
    for p in names:
        s[f'joint_embed.{p}_embed.weight'] = (L, len(sl[p]))
        s[f'joint_embed.{p}_embed.bias'] = (L,)
    s['joint_embed.body_embed.weight'] = (L, C)
    s['joint_embed.body_embed.bias'] = (L,)
    s['time_embed.0.weight'] = (Te, D)
    s['time_embed.0.bias'] = (Te,)
    s['time_embed.2.weight'] = (Te, Te)
    s['time_embed.2.bias'] = (Te,)

    def stylization(pre):
        s[pre + 'emb_layers.1.weight'] = (2 * D, Te)
        s[pre + 'emb_layers.1.bias'] = (2 * D,)
        s[pre + 'norm.weight'] = (D,)
        s[pre + 'norm.bias'] = (D,)
        s[pre + 'out_layers.2.weight'] = (D, D)
        s[pre + 'out_layers.2.bias'] = (D,)

    def moe(pre, seq, heads, din, dout):
        s[pre + 'proj.weight'] = (dout, din)
        s[pre + 'proj.bias'] = (dout,)
        s[pre + 'model.gates.0.temperature'] = (1,)
        s[pre + 'model.gates.0.cosine_projector.weight'] = (256, din)
        s[pre + 'model.gates.0.cosine_projector.bias'] = (256,)
        s[pre + 'model.gates.0.sim_matrix'] = (256, E)
        s[pre + 'model.experts.batched_fc1_w'] = (E, 4 * din, din)
        s[pre + 'model.experts.batched_fc2_w'] = (E, 4 * din, din)
        s[pre + 'model.experts.batched_fc1_bias'] = (E, 4 * din)
        s[pre + 'model.experts.batched_fc2_bias'] = (E, din)
        s[pre + 'embedding'] = (1, seq, heads, din)

    for i in range(NL):
        ca = f'temporal_decoder_blocks.{i}.ca_block.'
        s[ca + 'body_weight'] = (H, H)
        s[ca + 'norm.weight'] = (L,)
        s[ca + 'norm.bias'] = (L,)
        s[ca + 'text_norm.weight'] = (Dt,)
        s[ca + 'text_norm.bias'] = (Dt,)
        moe(ca + 'text_moe.', Nt, 1, Dt, 2 * L)
        moe(ca + 'motion_moe.', Tm, H, L, 4 * L)
        s[ca + 'body_d_attn.norm.weight'] = (L,)
        s[ca + 'body_d_attn.norm.bias'] = (L,)
        for n in ('query', 'key', 'value'):
            s[ca + f'body_d_attn.{n}.weight'] = (L, L)
            s[ca + f'body_d_attn.{n}.bias'] = (L,)
        stylization(ca + 'proj_out.')
        ff = f'temporal_decoder_blocks.{i}.ffn.'
        for p in range(H):
            s[ff + f'linear1_list.{p}.weight'] = (F, L)
            s[ff + f'linear1_list.{p}.bias'] = (F,)
            s[ff + f'linear2_list.{p}.weight'] = (L, F)
            s[ff + f'linear2_list.{p}.bias'] = (L,)
        stylization(ff + 'proj_out.')
    for p in names:
        s[f'out.{p}_out.weight'] = (len(sl[p]), L)
        s[f'out.{p}_out.bias'] = (len(sl[p]),)
    s['out.body_out.weight'] = (C, L)
    s['out.body_out.bias'] = (C,)
    return s


def control_wav_param_shapes(dims, copy_blocks_num, audio_in):
    """ControlT2MHalf with condition_pre_encode=True / 'wav' (S2G): control_cond_input takes the WavEncoder's D-wide
    output and the encoder's own parameters sit under condition_pre_encoder.pre_encoder.feat_extractor."""
    from .wav_encoder import wav_encoder_param_shapes
    D = dims['L'] * dims['H']
    s = control_param_shapes(dims, copy_blocks_num, D)
    for k, v in wav_encoder_param_shapes(D, audio_in).items():
        s['condition_pre_encoder.pre_encoder.feat_extractor.' + k] = v
    return s


def make_control_wav_state(dims, copy_blocks_num, audio_in, seed=0):
    shapes = control_wav_param_shapes(dims, copy_blocks_num, audio_in)
    pre = 'condition_pre_encoder.pre_encoder.feat_extractor.'
    wav = make_wav_encoder_state(dims['L'] * dims['H'], audio_in, seed)
    sd = OrderedDict()
    for k, v in shapes.items():
        sd[k] = wav[k[len(pre):]] if k.startswith(pre) else make_param(seed, k, tuple(v))
    return sd


def control_param_shapes(dims, copy_blocks_num, cond_feats):
    """Extra state-dict entries of ``ControlT2MHalf`` (reference controlnet.py:107-183) for the
    ``condition_pre_encode=False`` form (pre-encoded / raw feature condition of width ``cond_feats``).
    Key names are relative to the wrapper: base keys move under ``base_model.``."""
    base = param_shapes(dims)
    D = dims['L'] * dims['H']
    s = OrderedDict(('base_model.' + k, v) for k, v in base.items())
    for j in range(copy_blocks_num):
        for k, v in base.items():
            pre = 'temporal_decoder_blocks.0.'
            if k.startswith(pre):
                s[f'controlnet.{j}.copied_block.' + k[len(pre):]] = v
        if j == 0:
            s['controlnet.0.before_proj.weight'] = (D, D)
            s['controlnet.0.before_proj.bias'] = (D,)
        s[f'controlnet.{j}.after_proj.weight'] = (D, D)
        s[f'controlnet.{j}.after_proj.bias'] = (D,)
    s['control_cond_input.weight'] = (D, cond_feats)
    s['control_cond_input.bias'] = (D,)
    return s


def _randn(seed, name, shape):
    g = torch.Generator(device='cpu')
    g.manual_seed(zlib.crc32(f'{seed}:{name}'.encode()) & 0x7FFFFFFF)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


def make_param(seed, name, shape):
    r = _randn(seed, name, shape)
    if name.endswith('temperature'):
        return math.log(2.0) + 0.1 * r                      # tutel init: log(1/init_t), init_t=0.5
    if name.endswith('sim_matrix'):
        return 0.01 * r                                      # tutel init: normal(0, 0.01)
    if name.endswith('embedding') or name.endswith('body_weight'):
        return r                                             # reference init: torch.randn
    if name.endswith('batched_fc1_w'):
        return r / math.sqrt(shape[2])
    if name.endswith('batched_fc2_w'):
        return r / math.sqrt(shape[1])
    if name.endswith('_bias') or name.endswith('.bias'):
        if '.norm.' in name or 'text_norm' in name:
            return 0.1 * r
        return 0.02 * r
    if name.endswith('.weight'):
        if '.norm.' in name or 'text_norm' in name:
            return 1.0 + 0.1 * r
        scale = 1.0 / math.sqrt(shape[1])
        if 'out_layers.2' in name or 'after_proj' in name or 'before_proj' in name:   # zero-init in the reference
            scale *= 0.5
        return scale * r
    raise KeyError(name)


def text_encoder_param_shapes(Dt, num_layers, ff, clip_width=512, clip_layers=0, clip_ff=2048, vocab=49408, context=77):
    """reference keys (relative to the denoiser) of text_pre_proj / textTransEncoder / text_ln and, optionally, the
    parameters of the CLIP text tower (clip.* as registered by ``self.clip``)."""
    s = OrderedDict()
    if clip_width != Dt:
        s['text_pre_proj.weight'], s['text_pre_proj.bias'] = (Dt, clip_width), (Dt,)
    for i in range(num_layers):
        p = f'textTransEncoder.layers.{i}.'
        s[p + 'self_attn.in_proj_weight'], s[p + 'self_attn.in_proj_bias'] = (3 * Dt, Dt), (3 * Dt,)
        s[p + 'self_attn.out_proj.weight'], s[p + 'self_attn.out_proj.bias'] = (Dt, Dt), (Dt,)
        s[p + 'linear1.weight'], s[p + 'linear1.bias'] = (ff, Dt), (ff,)
        s[p + 'linear2.weight'], s[p + 'linear2.bias'] = (Dt, ff), (Dt,)
        for n in ('norm1', 'norm2'):
            s[p + n + '.weight'], s[p + n + '.bias'] = (Dt,), (Dt,)
    s['text_ln.weight'], s['text_ln.bias'] = (Dt,), (Dt,)
    if clip_layers:
        w = clip_width
        s['clip.token_embedding.weight'], s['clip.positional_embedding'] = (vocab, w), (context, w)
        for i in range(clip_layers):
            p = f'clip.transformer.resblocks.{i}.'
            s[p + 'attn.in_proj_weight'], s[p + 'attn.in_proj_bias'] = (3 * w, w), (3 * w,)
            s[p + 'attn.out_proj.weight'], s[p + 'attn.out_proj.bias'] = (w, w), (w,)
            s[p + 'mlp.c_fc.weight'], s[p + 'mlp.c_fc.bias'] = (clip_ff, w), (clip_ff,)
            s[p + 'mlp.c_proj.weight'], s[p + 'mlp.c_proj.bias'] = (w, clip_ff), (w,)
            for n in ('ln_1', 'ln_2'):
                s[p + n + '.weight'], s[p + n + '.bias'] = (w,), (w,)
        s['clip.ln_final.weight'], s['clip.ln_final.bias'] = (w,), (w,)
    return s


def make_text_encoder_state(shapes, seed=0):
    sd = OrderedDict()
    for k, shape in shapes.items():
        r = _randn(seed, 'text.' + k, shape)
        if k.endswith('.bias') or k.endswith('_bias'):
            sd[k] = (0.1 if ('norm' in k or 'ln' in k) else 0.02) * r
        elif 'norm' in k or '.ln_' in k or 'text_ln' in k or 'ln_final' in k:
            sd[k] = 1.0 + 0.1 * r
        elif k.endswith('embedding.weight') or k.endswith('positional_embedding'):
            sd[k] = 0.5 * r
        else:
            sd[k] = r / math.sqrt(shape[-1])
    return sd


def _vae_transformer_shapes(s, pre, d, ff, num_layers, max_len=5000):
    s[pre + 'mu_token'], s[pre + 'logvar_token'] = (d,), (d,)
    s[pre + 'sequence_pos_encoding.pe'] = (max_len, 1, d)
    for i in range(num_layers):
        p = f'{pre}seqTransEncoder.layers.{i}.'
        s[p + 'self_attn.in_proj_weight'], s[p + 'self_attn.in_proj_bias'] = (3 * d, d), (3 * d,)
        s[p + 'self_attn.out_proj.weight'], s[p + 'self_attn.out_proj.bias'] = (d, d), (d,)
        s[p + 'linear1.weight'], s[p + 'linear1.bias'] = (ff, d), (ff,)
        s[p + 'linear2.weight'], s[p + 'linear2.bias'] = (d, ff), (d,)
        for n in ('norm1', 'norm2'):
            s[p + n + '.weight'], s[p + n + '.bias'] = (d,), (d,)


def eval_encoder_param_shapes(nfeats=322, latent_dim=256, ff_size=1024, num_layers=4, bert=None, **unused):
    """Keys of the evaluation embedding model's checkpoint as T2MContrastiveModel_SMPLX.load_pretrained reads them
    (``motionencoder.*`` / ``textencoder.*``, t2m_bigru_smplx.py:417-435).  ``bert``: dict(dim, n_layers, hidden_dim,
    vocab_size, max_position_embeddings) of the DistilBERT text model, or None for the motion side only."""
    s = OrderedDict()
    m = 'motionencoder.'
    s[m + 'skel_embedding.weight'], s[m + 'skel_embedding.bias'] = (latent_dim, nfeats), (latent_dim,)
    _vae_transformer_shapes(s, m, latent_dim, ff_size, num_layers)
    if bert is not None:
        t, w = 'textencoder.', bert['dim']
        e = t + 'text_model.embeddings.'
        s[e + 'word_embeddings.weight'] = (bert['vocab_size'], w)
        s[e + 'position_embeddings.weight'] = (bert['max_position_embeddings'], w)
        s[e + 'LayerNorm.weight'], s[e + 'LayerNorm.bias'] = (w,), (w,)
        for i in range(bert['n_layers']):
            p = f'{t}text_model.transformer.layer.{i}.'
            for n in ('q_lin', 'k_lin', 'v_lin', 'out_lin'):
                s[p + f'attention.{n}.weight'], s[p + f'attention.{n}.bias'] = (w, w), (w,)
            s[p + 'sa_layer_norm.weight'], s[p + 'sa_layer_norm.bias'] = (w,), (w,)
            s[p + 'ffn.lin1.weight'], s[p + 'ffn.lin1.bias'] = (bert['hidden_dim'], w), (bert['hidden_dim'],)
            s[p + 'ffn.lin2.weight'], s[p + 'ffn.lin2.bias'] = (w, bert['hidden_dim']), (w,)
            s[p + 'output_layer_norm.weight'], s[p + 'output_layer_norm.bias'] = (w,), (w,)
        s[t + 'projection.1.weight'], s[t + 'projection.1.bias'] = (latent_dim, w), (latent_dim,)
        _vae_transformer_shapes(s, t, latent_dim, ff_size, num_layers)
    return s


def sinusoid_table(max_len, d):
    """PositionalEncoding.pe (t2m_bigru_smplx.py:24-32), fp32 torch arithmetic: [max_len, 1, d]."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(1)


def make_eval_encoder_state(shapes, seed=0):
    sd = OrderedDict()
    for k, shape in shapes.items():
        if k.endswith('sequence_pos_encoding.pe'):
            sd[k] = sinusoid_table(shape[0], shape[2])
            continue
        r = _randn(seed, 'eval.' + k, shape)
        if k.endswith('.bias') or k.endswith('_bias'):
            sd[k] = (0.1 if 'norm' in k.lower() else 0.02) * r
        elif 'norm' in k.lower():
            sd[k] = 1.0 + 0.1 * r
        elif k.endswith('_token'):
            sd[k] = r
        elif 'embeddings.' in k:
            sd[k] = 0.5 * r
        else:
            sd[k] = r / math.sqrt(shape[-1])
    return sd


def _bigru_head_shapes(s, pre, din, hid, dout):
    s[pre + 'hidden'] = (2, 1, hid)
    s[pre + 'input_emb.weight'], s[pre + 'input_emb.bias'] = (hid, din), (hid,)
    for sfx in ('', '_reverse'):
        s[pre + 'gru.weight_ih_l0' + sfx], s[pre + 'gru.weight_hh_l0' + sfx] = (3 * hid, hid), (3 * hid, hid)
        s[pre + 'gru.bias_ih_l0' + sfx], s[pre + 'gru.bias_hh_l0' + sfx] = (3 * hid,), (3 * hid,)
    s[pre + 'output_net.0.weight'], s[pre + 'output_net.0.bias'] = (hid, 2 * hid), (hid,)
    s[pre + 'output_net.1.weight'], s[pre + 'output_net.1.bias'] = (hid,), (hid,)
    s[pre + 'output_net.3.weight'], s[pre + 'output_net.3.bias'] = (dout, hid), (dout,)


def t2m_eval_param_shapes(input_size=263, movement_hidden_size=512, movement_latent_size=512, motion_hidden_size=1024,
                          motion_latent_size=512, word_size=300, pos_size=15, hidden_size=512, output_size=512, **unused):
    """The three state dicts of the T2M evaluator checkpoint (``movement_encoder`` / ``motion_encoder`` / ``text_encoder``,
    t2m_bigru.py:84-87,126-128) flattened with their names as prefixes."""
    s = OrderedDict()
    m = 'movement_encoder.'
    s[m + 'main.0.weight'], s[m + 'main.0.bias'] = (movement_hidden_size, input_size - 4, 4), (movement_hidden_size,)
    s[m + 'main.3.weight'], s[m + 'main.3.bias'] = (movement_latent_size, movement_hidden_size, 4), (movement_latent_size,)
    s[m + 'out_net.weight'], s[m + 'out_net.bias'] = (movement_latent_size, movement_latent_size), (movement_latent_size,)
    _bigru_head_shapes(s, 'motion_encoder.', movement_latent_size, motion_hidden_size, motion_latent_size)
    s['text_encoder.pos_emb.weight'], s['text_encoder.pos_emb.bias'] = (word_size, pos_size), (word_size,)
    _bigru_head_shapes(s, 'text_encoder.', word_size, hidden_size, output_size)
    return s


def make_t2m_eval_state(shapes, seed=0):
    sd = OrderedDict()
    for k, shape in shapes.items():
        r = _randn(seed, 't2m.' + k, shape)
        if k.endswith('hidden'):
            sd[k] = r
        elif 'bias' in k:
            sd[k] = (0.1 if 'output_net.1' in k else 0.05) * r
        elif 'output_net.1' in k:
            sd[k] = 1.0 + 0.1 * r
        else:
            fan = shape[1] * (shape[2] if len(shape) == 3 else 1)
            sd[k] = r / math.sqrt(fan)
    return sd


def make_wav_encoder_state(out_dim, audio_in, seed=0):
    """Deterministic non-trivial WavEncoder weights (BatchNorm running stats included) keyed like the reference."""
    from .wav_encoder import wav_encoder_param_shapes
    sd = OrderedDict()
    for k, shape in wav_encoder_param_shapes(out_dim, audio_in).items():
        r = _randn(seed, 'wav.' + k, shape) if shape else torch.zeros(())
        if k.endswith('num_batches_tracked'):
            sd[k] = torch.tensor(100, dtype=torch.long)
        elif k.endswith('running_var'):
            sd[k] = 0.5 + r.abs()
        elif k.endswith('running_mean') or k.endswith('.bias'):
            sd[k] = 0.1 * r
        elif 'bn' in k or 'downsample.1' in k:      # BatchNorm gamma
            sd[k] = 1.0 + 0.1 * r
        else:                                        # conv weight [Cout, Cin, 15]: keep activations O(1)
            sd[k] = r / math.sqrt(shape[1] * shape[2])
    return sd


def make_state_dict(dims, seed=0, shapes=None):
    shapes = param_shapes(dims) if shapes is None else shapes
    return OrderedDict((k, make_param(seed, k, tuple(v))) for k, v in shapes.items())


def reference_model_cfg(dims):
    """The ``model=dict(type='STMoGenTransformer', ...)`` dict of the reference configs for these dims."""
    L, H = dims['L'], dims['H']
    return dict(
        type='STMoGenTransformer', input_feats=dims['input_feats'], max_seq_len=dims['max_seq_len'],
        latent_dim=L * H, time_embed_dim=dims['Te'], num_layers=dims['NL'],
        ca_block_cfg=dict(type='STMA', latent_dim=L, text_latent_dim=dims['Dt'], num_heads=H,
                          num_text_heads=1, num_experts=dims['E'], topk=dims['topk'],
                          gate_type='cosine_top', gate_noise=1.0, ffn_dim=dims['F'],
                          time_embed_dim=dims['Te'], max_seq_len=dims['max_seq_len'],
                          max_text_seq_len=dims['Nt'], temporal_comb=False, dropout=0,
                          dynamic_body=True),
        ffn_cfg=dict(latent_dim=L, ffn_dim=dims['F'], dropout=0, time_embed_dim=dims['Te'], num_heads=H),
        text_encoder=None,
        pose_encoder_cfg=dict(dataset_name=dims.get('dataset', 'motionx'), latent_dim=L, input_dim=dims['input_feats']),
        pose_decoder_cfg=dict(dataset_name=dims.get('dataset', 'motionx'), latent_dim=L, output_dim=dims['input_feats']),
        scale_func_cfg=dict(scale=dims['scale']), moe_route_loss_weight=10.0,
        template_kl_loss_weight=0.0001, use_pos_embedding=True)
