"""The reference's model registry surface (mogen/models/builder.py:1-36): one registry aliased as
LOSSES / ARCHITECTURES / SUBMODULES / ATTENTIONS plus the four ``build_*`` helpers."""
from .registry import Registry, build_from_cfg

MODELS = Registry('models', build_func=build_from_cfg)

LOSSES = MODELS
ARCHITECTURES = MODELS
SUBMODULES = MODELS
ATTENTIONS = MODELS


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_architecture(cfg):
    return ARCHITECTURES.build(cfg)


def build_submodule(cfg):
    return SUBMODULES.build(cfg)


def build_attention(cfg):
    return ATTENTIONS.build(cfg)
