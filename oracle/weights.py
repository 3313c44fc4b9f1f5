"""Re-export of the synthetic (seed, key) -> tensor weight recipe.  TEST INFRASTRUCTURE ONLY.

The recipe itself is plain data generation and lives in ``motioncraft_amd/synthetic.py`` so that
``bench.py`` can build random-init weights without importing anything under ``oracle/``."""
from motioncraft_amd.synthetic import *  # noqa: F401,F403
from motioncraft_amd.synthetic import (PART_NAMES, control_param_shapes, default_dims, make_param, make_state_dict, param_shapes, part_layout, humanml3d_dims, make_wav_encoder_state, text_encoder_param_shapes, make_text_encoder_state, control_wav_param_shapes, make_control_wav_state, eval_encoder_param_shapes, make_eval_encoder_state, sinusoid_table, t2m_eval_param_shapes, make_t2m_eval_state,  # noqa: F401
                                       reference_model_cfg, smplx_part_slices)
