"""Registry surface the reference's callers import (names as in mogen/models/builder.py:1-36): ONE registry visible
under four role names, and one ``build_<role>`` helper per role."""
from .registry import Registry, build_from_cfg

MODELS = Registry('models', build_func=build_from_cfg)
# the reference keeps losses, architectures, sub-modules and attention blocks in the same table
LOSSES = ARCHITECTURES = SUBMODULES = ATTENTIONS = MODELS


def _make_builder(role):
    def build(cfg):
        return MODELS.build(cfg)
    build.__name__ = build.__qualname__ = f'build_{role}'
    build.__doc__ = f"cfg dict with type='...' -> instance of the registered {role} class (None stays None)."
    return build


build_loss, build_architecture, build_submodule, build_attention = (
    _make_builder(r) for r in ('loss', 'architecture', 'submodule', 'attention'))
