"""Minimal ``Config.fromfile`` with the mmcv surface the reference's configs and tools use
(SURVEY.md section 5): python config files, ``_base_`` inheritance with recursive dict merge,
attribute + item access, ``.get``, item assignment, ``merge_from_dict`` for ``--cfg-options``.
No ``_delete_`` / ``{{ }}`` substitutions (none of the shipped configs use them).
"""
import ast
import copy
import os
import types


class ConfigDict(dict):
    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'{type(self).__name__}' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = _wrap(value)

    def __setitem__(self, name, value):
        super().__setitem__(name, _wrap(value))

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(v):
    if isinstance(v, ConfigDict):
        return v
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    if isinstance(v, tuple):
        return tuple(_wrap(x) for x in v)
    return v


def _merge(base, new):
    out = dict(base)
    for k, v in new.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict) and not v.get('_delete_', False):
            out[k] = _merge(out[k], v)
        else:
            out[k] = {kk: vv for kk, vv in v.items() if kk != '_delete_'} if isinstance(v, dict) else v
    return out


def _load_py(path):
    path = os.path.abspath(os.path.expanduser(path))
    if not os.path.isfile(path):
        raise FileNotFoundError(f'file "{path}" does not exist')
    if not path.endswith('.py'):
        raise IOError('Only py type are supported on this path')
    with open(path, 'r') as f:
        src = f.read()
    ast.parse(src)  # SyntaxError like mmcv's _validate_py_syntax
    ns = {'__file__': path, '__name__': '_mc_cfg_'}
    exec(compile(src, path, 'exec'), ns)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    base = cfg.pop('_base_', None)
    if base is not None:
        merged = {}
        for b in (base if isinstance(base, list) else [base]):
            bcfg = _load_py(os.path.join(os.path.dirname(path), b))
            dup = set(merged) & set(bcfg)
            if dup:
                raise KeyError(f'Duplicate key is not allowed among bases: {dup}')
            merged.update(bcfg)
        cfg = _merge(merged, cfg)
    return cfg


class Config:
    def __init__(self, cfg_dict=None, filename=None):
        if cfg_dict is None:
            cfg_dict = {}
        if not isinstance(cfg_dict, dict):
            raise TypeError(f'cfg_dict must be a dict, but got {type(cfg_dict)}')
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict))
        object.__setattr__(self, '_filename', filename)

    @staticmethod
    def fromfile(filename):
        return Config(_load_py(str(filename)), filename=str(filename))

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __setitem__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def __iter__(self):
        return iter(self._cfg_dict)

    def __len__(self):
        return len(self._cfg_dict)

    def __repr__(self):
        return f'Config (path: {self._filename}): {dict(self._cfg_dict)!r}'

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def keys(self):
        return self._cfg_dict.keys()

    def items(self):
        return self._cfg_dict.items()

    def merge_from_dict(self, options):
        """--cfg-options a.b.c=v style overrides (tools/test.py:30-35,67-68)."""
        nested = {}
        for full, v in options.items():
            d = nested
            parts = full.split('.')
            for p in parts[:-1]:
                d = d.setdefault(p, {})
            d[parts[-1]] = v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge(dict(self._cfg_dict), nested)))
