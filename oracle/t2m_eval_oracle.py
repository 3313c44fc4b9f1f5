"""CPU restatement of the HumanML3D / KIT evaluation embedding model (``T2MContrastiveModel``).  TEST INFRASTRUCTURE ONLY.

  * ``MovementConvEncoder``   two Conv1d(k=4, s=2, p=1) + LeakyReLU(0.2), Linear     mogen/models/rnns/t2m_bigru.py:226-246
  * ``MotionEncoderBiGRUCo``  Linear -> bidirectional GRU over the length-packed sequence (learned initial state) ->
                              Linear, LayerNorm, LeakyReLU(0.2), Linear                  t2m_bigru.py:249-282
  * ``TextEncoderBiGRUCo``    word vectors + pos_emb(one-hot) -> Linear -> the same BiGRU head  t2m_bigru.py:186-223
  * ``T2MMotionEncoder``      drops the last 4 channels, lengths // 4                  t2m_bigru.py:72-110

Pinned by tests/golden/make_golden.py (group ``t2m_evaluator``) against the reference's own modules ->
tests/golden/t2m_evaluator.npz.  The GRU is written as an explicit per-step loop (torch.nn.GRU semantics: gate order
r | z | n, n = tanh(W_in x + b_in + r * (W_hn h + b_hn)), h' = (1 - z) n + z h); a packed sequence means sample b is
updated at steps t < len[b] in the forward direction and visits t = len[b]-1 ... 0 in the reverse direction.
"""
import torch
import torch.nn.functional as F


def _gru_final(x, lens, p, pre, reverse):
    """x [B, S, H] (already through input_emb), lens [B] -> final hidden [B, H] of one direction."""
    sfx = '_reverse' if reverse else ''
    w_ih, w_hh = p[pre + 'gru.weight_ih_l0' + sfx], p[pre + 'gru.weight_hh_l0' + sfx]
    b_ih, b_hh = p[pre + 'gru.bias_ih_l0' + sfx], p[pre + 'gru.bias_hh_l0' + sfx]
    B, S, H = x.shape
    h = p[pre + 'hidden'][1 if reverse else 0].expand(B, H).clone()
    lens = torch.as_tensor(lens)
    gi_all = F.linear(x, w_ih, b_ih)
    for s in range(int(lens.max())):
        t = (lens - 1 - s).clamp(min=0) if reverse else torch.full_like(lens, s)
        gi = gi_all[torch.arange(B), t]
        gh = F.linear(h, w_hh, b_hh)
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        hn = (1 - z) * n + z * h
        h = torch.where((s < lens)[:, None], hn, h)
    return h


def bigru_head(p, pre, x, lens):
    """input_emb -> BiGRU final states -> output_net."""
    e = F.linear(x, p[pre + 'input_emb.weight'], p[pre + 'input_emb.bias'])
    h = torch.cat([_gru_final(e, lens, p, pre, False), _gru_final(e, lens, p, pre, True)], dim=-1)
    y = F.linear(h, p[pre + 'output_net.0.weight'], p[pre + 'output_net.0.bias'])
    y = F.leaky_relu(F.layer_norm(y, (y.shape[-1],), p[pre + 'output_net.1.weight'], p[pre + 'output_net.1.bias']), 0.2)
    return F.linear(y, p[pre + 'output_net.3.weight'], p[pre + 'output_net.3.bias'])


def movement_encoder(p, x, pre='movement_encoder.'):
    """x [B, T, C] -> [B, T // 4, latent]."""
    y = x.permute(0, 2, 1)
    for k in ('main.0', 'main.3'):
        w, b = p[pre + k + '.weight'], p[pre + k + '.bias']
        cols = F.pad(y, (1, 1)).unfold(2, 4, 2)                           # [B, C, T_out, 4]
        y = F.leaky_relu(torch.einsum('bcts,ocs->bot', cols, w) + b[None, :, None], 0.2)
    return F.linear(y.permute(0, 2, 1), p[pre + 'out_net.weight'], p[pre + 'out_net.bias'])


def encode_motion(p, motion, motion_length):
    """T2MContrastiveModel.encode_motion: motion [B, T, input_size], motion_length [B] -> [B, motion_latent]."""
    mov = movement_encoder(p, motion[..., :-4].float())
    return bigru_head(p, 'motion_encoder.', mov, torch.as_tensor(motion_length) // 4)


def encode_text(p, word_emb, pos_onehot, sent_len):
    """TextEncoderBiGRUCo.forward: word_emb [B, S, word_size], pos_onehot [B, S, pos_size], sent_len [B]."""
    x = word_emb + F.linear(pos_onehot, p['text_encoder.pos_emb.weight'], p['text_encoder.pos_emb.bias'])
    return bigru_head(p, 'text_encoder.', x, sent_len)
