"""ctypes binding of libmotioncraft_amd.so (C-ABI in include/motioncraft_amd.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  Importing this module
succeeds without the library (so configs/registries can be used on a CPU box), but any attempt
to run the path raises ``RuntimeError`` when the HIP library is missing or no MI355X is visible.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmotioncraft_amd.so')

MC_OK = 0
ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2


class ModelConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        'input_feats', 'max_seq_len', 'latent_dim', 'num_parts', 'num_layers', 'ffn_dim',
        'time_embed_dim', 'text_latent_dim', 'max_text_len', 'num_experts', 'topk', 'dyn_heads')] + [
        ('capacity_factor', ctypes.c_float), ('cfg_scale', ctypes.c_float),
        ('num_ctrl_layers', ctypes.c_int32), ('ctrl_cond_feats', ctypes.c_int32), ('ctrl_condition_cfg', ctypes.c_int32)]


class StepCoefs(ctypes.Structure):
    _fields_ = [('mode', ctypes.c_int32)] + [(n, ctypes.c_float) for n in (
        'text_coef', 'none_coef', 'c1', 'c2', 'log_var', 'sqrt_recip', 'sqrt_recipm1', 'ab', 'ab_prev',
        'eta', 'nonzero')]


_P = ctypes.c_void_p
class Inpaint(ctypes.Structure):
    """mc_inpaint (include/motioncraft_amd.h): model_kwargs['y'] operands of one RePaint step."""
    _fields_ = [('gt_dev', ctypes.c_void_p), ('keep_dev', ctypes.c_void_p), ('gt_noise_dev', ctypes.c_void_p),
                ('blend_w_dev', ctypes.c_void_p), ('blend_len', ctypes.c_int32)]


MAX_TRANSL = 8               # MC_MAX_TRANSL


class Seed(ctypes.Structure):
    """mc_seed (include/motioncraft_amd.h): pre_seq / transl_req operands of one step."""
    _fields_ = [('pre_seq_dev', ctypes.c_void_p), ('pre_noise_dev', ctypes.c_void_p), ('pre_len', ctypes.c_int32),
                ('sqrt_ab', ctypes.c_float), ('sqrt_1mab', ctypes.c_float), ('num_transl', ctypes.c_int32),
                ('transl_channel', ctypes.c_int32 * MAX_TRANSL), ('transl_value', (ctypes.c_float * 2) * MAX_TRANSL)]


class EvalEncConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('nfeats', 'latent_dim', 'ff_size', 'num_layers', 'num_heads', 'pe_len', 'bert_dim',
                                              'bert_layers', 'bert_heads', 'bert_ff', 'bert_vocab', 'bert_max_pos')]


class T2MEvalConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('input_size', 'movement_hidden', 'movement_latent', 'motion_hidden', 'motion_latent',
                                              'word_size', 'pos_size', 'text_hidden', 'text_out')]


class TextEncConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ('clip_dim', 'text_latent_dim', 'num_layers', 'ff_size', 'num_heads', 'max_len',
                                              'clip_layers', 'clip_heads', 'clip_ff', 'vocab')]


_SIGNATURES = {
    'mc_last_error': (ctypes.c_char_p, []),
    'mc_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'mc_set_device': (ctypes.c_int, [ctypes.c_int]),
    'mc_model_create': (ctypes.c_int, [ctypes.POINTER(ModelConfig), ctypes.POINTER(_P)]),
    'mc_model_destroy': (None, [_P]),
    'mc_model_set_param': (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.c_int64]),
    'mc_model_finalize': (ctypes.c_int, [_P]),
    'mc_ctx_create': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_P)]),
    'mc_ctx_destroy': (None, [_P]),
    'mc_ctx_workspace_bytes': (ctypes.c_int64, [_P]),
    'mc_ctx_check': (ctypes.c_int, [_P, _P]),
    'mc_ctx_uses_coop_routing': (ctypes.c_int, [_P]),
    'mc_ctx_effective_precision': (ctypes.c_int, [_P]),
    'mc_ctx_profile': (ctypes.c_int, [_P, ctypes.c_int32]),
    'mc_ctx_profile_read': (ctypes.c_int, [_P, ctypes.c_int64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32),
                                           ctypes.POINTER(ctypes.c_double)]),
    'mc_ctx_enable_capture': (ctypes.c_int, [_P]),
    'mc_ctx_set_tie_policy': (ctypes.c_int, [_P, ctypes.c_int32]),
    'mc_ctx_set_option': (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int64]),
    'mc_ctx_set_precision': (ctypes.c_int, [_P, ctypes.c_int32]),
    'mc_op_gemm_f16': (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P]),
    'mc_ctx_set_timesteps': (ctypes.c_int, [_P, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, _P]),
    'mc_ctx_set_condition': (ctypes.c_int, [_P, _P, _P, _P]),
    'mc_ctx_set_control': (ctypes.c_int, [_P, _P, ctypes.c_int32, _P]),
    'mc_denoise': (ctypes.c_int, [_P, _P, ctypes.c_int32, _P, ctypes.c_int32, _P]),
    'mc_sample_step': (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.POINTER(StepCoefs), _P, _P, _P, _P]),
    'mc_sample_loop': (ctypes.c_int, [_P, _P, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(StepCoefs), ctypes.c_int32, _P,
                                      ctypes.c_uint64, ctypes.c_uint64, _P, _P]),
    'mc_op_philox_normal': (ctypes.c_int, [_P, _P, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, _P]),
    'mc_ctx_graph_capture': (ctypes.c_int, [_P, _P, _P, ctypes.POINTER(StepCoefs), ctypes.c_int32, _P]),
    'mc_ctx_graph_step': (ctypes.c_int, [_P, ctypes.c_int32, _P]),
    'mc_ctx_graph_release': (ctypes.c_int, [_P]),
    'mc_sample_step_seeded': (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.POINTER(StepCoefs), _P, ctypes.POINTER(Seed), _P, _P, _P]),
    'mc_sample_step_inpaint': (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.POINTER(StepCoefs), _P,
                                              ctypes.POINTER(Inpaint), _P, _P, _P]),
    'mc_ctx_get_buffer': (ctypes.c_int, [_P, ctypes.c_char_p, ctypes.c_int32, ctypes.POINTER(_P),
                                         ctypes.POINTER(ctypes.c_int64)]),
    'mc_op_gemm': (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                  ctypes.c_int32, ctypes.c_int32, _P]),
    'mc_op_gemm_tail': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_float,
                                       ctypes.c_int32, _P]),
    'mc_op_ln_rows': (ctypes.c_int, [_P, ctypes.c_int64, _P, _P, _P, ctypes.c_int32, _P, ctypes.c_int64,
                                     ctypes.c_int32, _P]),
    'mc_op_sampler_update': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, ctypes.c_int64, ctypes.POINTER(StepCoefs), _P]),
    'mc_postprocess_smplx': (ctypes.c_int, [_P, _P, _P, _P, _P, ctypes.POINTER(ctypes.c_int32 * 4), ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P]),
    'mc_postprocess_smplx_stitched': (ctypes.c_int, [_P, _P, ctypes.c_int32, _P, _P, _P, ctypes.POINTER(ctypes.c_int32 * 4),
                                                     ctypes.c_int32, ctypes.c_int32, _P, _P, _P, _P]),
    'mc_textenc_create': (ctypes.c_int, [ctypes.POINTER(TextEncConfig), ctypes.POINTER(_P)]),
    'mc_textenc_destroy': (None, [_P]),
    'mc_textenc_set_param': (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.c_int64]),
    'mc_textenc_finalize': (ctypes.c_int, [_P]),
    'mc_textenc_forward_feat': (ctypes.c_int, [_P, _P, ctypes.c_int32, _P, _P]),
    'mc_textenc_forward_tokens': (ctypes.c_int, [_P, _P, ctypes.c_int32, _P, _P, _P]),
    'mc_evalenc_create': (ctypes.c_int, [ctypes.POINTER(EvalEncConfig), ctypes.POINTER(_P)]),
    'mc_evalenc_destroy': (None, [_P]),
    'mc_evalenc_set_param': (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.c_int64]),
    'mc_evalenc_finalize': (ctypes.c_int, [_P]),
    'mc_evalenc_encode_motion': (ctypes.c_int, [_P, _P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    'mc_evalenc_encode_text': (ctypes.c_int, [_P, _P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    'mc_t2meval_create': (ctypes.c_int, [ctypes.POINTER(T2MEvalConfig), ctypes.POINTER(_P)]),
    'mc_t2meval_destroy': (None, [_P]),
    'mc_t2meval_set_param': (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.c_int64]),
    'mc_t2meval_finalize': (ctypes.c_int, [_P]),
    'mc_t2meval_encode_motion': (ctypes.c_int, [_P, _P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    'mc_t2meval_encode_text': (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    'mc_wavenc_create': (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(_P)]),
    'mc_wavenc_destroy': (None, [_P]),
    'mc_wavenc_set_param': (ctypes.c_int, [_P, ctypes.c_char_p, _P, ctypes.c_int64]),
    'mc_wavenc_finalize': (ctypes.c_int, [_P]),
    'mc_wavenc_out_len': (ctypes.c_int, [_P, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    'mc_wavenc_forward': (ctypes.c_int, [_P, _P, ctypes.c_int32, ctypes.c_int32, _P, _P]),
    'mc_op_renoise': (ctypes.c_int, [_P, _P, ctypes.c_float, ctypes.c_float, _P, ctypes.c_int64, _P]),
    'mc_debug_flop_ledger': (ctypes.c_int, [ctypes.c_int32]),
    'mc_debug_flop_ledger_dump': (ctypes.c_int64, [ctypes.c_char_p, ctypes.c_int64]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
POST_MAXTAP = 129            # MC_POST_MAXTAP

_lib = None


def load(require_gpu=False):
    """Load the shared library (no compute call).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} not found: build it with `python -m motioncraft_amd.build` '
                '(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.')
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    if require_gpu:
        n = ctypes.c_int(0)
        rc = _lib.mc_device_count(ctypes.byref(n))
        if rc != MC_OK or n.value < 1:
            raise RuntimeError('motioncraft_amd: no HIP device visible (the hot path runs on MI355X only): '
                               + last_error())
    return _lib


def last_error():
    return (_lib.mc_last_error() or b'').decode() if _lib is not None else ''


def check(rc, what=''):
    if rc != MC_OK:
        raise RuntimeError(f'motioncraft_amd {what} failed (code {rc}): {last_error()}')
