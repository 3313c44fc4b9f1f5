# Reduced stmogen config, written in the key/nesting scheme the reference's configs/stmogen/*.py use so that the same
# mmcv-style Config loader reads both; dims match tests/helpers.py SMALL so the golden fixtures apply.
_base_ = ['_base_/data_small.py']

input_feats, max_seq_len, dataset_name = 322, 24, "motionx"
latent_dim, text_latent_dim, time_embed_dim, ff_size = 32, 32, 64, 64
num_heads, num_layers, dropout = 12, 2, 0

_attn = dict(type='STMA', latent_dim=latent_dim, text_latent_dim=text_latent_dim, num_heads=num_heads, num_text_heads=1,
             num_experts=16, topk=2, gate_type='cosine_top', gate_noise=1.0, ffn_dim=ff_size,
             time_embed_dim=time_embed_dim, max_seq_len=max_seq_len, max_text_seq_len=8, temporal_comb=False,
             dropout=dropout, dynamic_body=True)
_sffn = dict(latent_dim=latent_dim, ffn_dim=ff_size, dropout=dropout, time_embed_dim=time_embed_dim, num_heads=num_heads)
_text = dict(pretrained_model='clip', latent_dim=text_latent_dim, num_layers=2, ff_size=2048, dropout=dropout,
             use_text_proj=False)
_pose = dict(dataset_name=dataset_name, latent_dim=latent_dim)
_schedule = dict(beta_scheduler='linear', diffusion_steps=1000, model_mean_type='start_x', model_var_type='fixed_large')

model = dict(
    type='MotionDiffusion',
    model=dict(type='STMoGenTransformer', input_feats=input_feats, max_seq_len=max_seq_len,
               latent_dim=latent_dim * num_heads, time_embed_dim=time_embed_dim, num_layers=num_layers,
               ca_block_cfg=_attn, ffn_cfg=_sffn, text_encoder=_text,
               pose_encoder_cfg=dict(_pose, input_dim=input_feats), pose_decoder_cfg=dict(_pose, output_dim=input_feats),
               scale_func_cfg=dict(scale=6.5), moe_route_loss_weight=10.0, template_kl_loss_weight=0.0001,
               use_pos_embedding=True),
    loss_recon=dict(type='MSELoss', loss_weight=1, reduction='none'),
    diffusion_train=_schedule, diffusion_test=dict(_schedule, respace='15,15,8,6,6'),
    inference_type='ddim', loss_reduction='batch')
data = dict(samples_per_gpu=4)
