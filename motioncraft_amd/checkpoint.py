"""Checkpoint loading: the counterpart of ``mmcv.runner.load_checkpoint(model, filename, map_location='cpu')`` as the
reference tools call it (``tools/test.py:99``, ``tools/visualize.py:157``).  Accepts the formats a converted checkpoint
can come in: a torch ``.pth`` / ``.pt`` (flat state dict or mmcv-style ``{'state_dict': ..., 'meta': ...}``), a numpy
``.npz`` of arrays keyed by parameter name, or a ``.safetensors`` file.  Key prefixes (``module.`` of DataParallel
wrappers, ``model.`` / ``base_model.`` of the architecture / control wrapper) are normalised by the model's own
``load_state_dict``."""
import os

import numpy as np
import torch


def read_state_dict(filename, map_location='cpu'):
    if not os.path.isfile(filename):
        raise IOError(f'{filename} is not a checkpoint file')
    ext = os.path.splitext(filename)[1].lower()
    if ext == '.npz':
        with np.load(filename) as z:
            sd = {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}
        meta = {}
    elif ext == '.safetensors':
        from safetensors.torch import load_file
        sd, meta = load_file(filename), {}
    else:
        ck = torch.load(filename, map_location=map_location, weights_only=False)
        if not isinstance(ck, dict):
            raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
        meta = ck.get('meta', {}) if 'state_dict' in ck else {}
        sd = ck['state_dict'] if 'state_dict' in ck else ck
    out = {}
    for k, v in sd.items():
        if k.startswith('module.'):
            k = k[len('module.'):]
        out[k] = v
    return out, meta


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None):
    """Returns the checkpoint dict like mmcv does (``{'state_dict': ..., 'meta': ...}``)."""
    sd, meta = read_state_dict(filename, map_location)
    model.load_state_dict(sd, strict=strict)
    return {'state_dict': sd, 'meta': meta}
