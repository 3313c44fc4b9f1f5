"""Minimal mmcv-compatible ``Registry`` (the surface the reference uses: mogen/models/builder.py:1-36).

``@REG.register_module()`` registers a class under its name; ``REG.build(cfg)`` instantiates
``cfg['type']`` with the remaining keys as kwargs.  ``build_from_cfg`` mirrors mmcv's error
behaviour: TypeError for a non-dict cfg, KeyError for a missing/unknown ``type``.
"""
import inspect


def build_from_cfg(cfg, registry, default_args=None):
    if cfg is None:
        return None
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg and not (default_args and 'type' in default_args):
        raise KeyError(f'`cfg` or `default_args` must contain the key "type", but got {cfg}\n{default_args}')
    args = dict(cfg)
    if default_args is not None:
        for k, v in default_args.items():
            args.setdefault(k, v)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError(f'{obj_type} is not in the {registry.name} registry')
    elif inspect.isclass(obj_type) or inspect.isfunction(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(obj_type)}')
    return obj_cls(**args)


class Registry:
    def __init__(self, name, build_func=None, parent=None, scope=None):
        self._name, self._module_dict, self.parent = name, {}, parent
        self.build_func = build_func or (parent.build_func if parent is not None else build_from_cfg)

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def __contains__(self, key):
        return self.get(key) is not None

    def __repr__(self):
        return f'{type(self).__name__}(name={self._name}, items={sorted(self._module_dict)})'

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        return self.parent.get(key) if self.parent is not None else None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, name=None, force=False):
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            if not force and n in self._module_dict:
                raise KeyError(f'{n} is already registered in {self._name}')
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco
