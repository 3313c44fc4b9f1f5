"""motioncraft_amd: MI355X-native (gfx950) STMoGen sampling hot path behind the reference's
mogen registry / mmcv-Config API.  See DESIGN.md and INTEGRATION.md."""
from .builder import (ARCHITECTURES, ATTENTIONS, LOSSES, MODELS, SUBMODULES, build_architecture, build_attention,
                      build_loss, build_submodule)
from .config import Config, ConfigDict
from .registry import Registry, build_from_cfg
from . import models as _models  # registers MotionDiffusion / STMoGenTransformer / STMA / MSELoss
from .models import ControlT2MHalf, wrap_fp16_model
from .checkpoint import load_checkpoint

__all__ = ['ARCHITECTURES', 'ATTENTIONS', 'LOSSES', 'MODELS', 'SUBMODULES', 'build_architecture',
           'build_attention', 'build_loss', 'build_submodule', 'Config', 'ConfigDict', 'Registry', 'build_from_cfg', 'ControlT2MHalf',
           'load_checkpoint', 'wrap_fp16_model']
