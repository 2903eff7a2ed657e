"""mmcv-style registries of the refiner hot path.

Mirrors the reference's four registries and their ``build_*`` helpers
(models/refiner/builder.py:3, models/encoder/builder.py:3,
models/decoder/builder.py:3, models/head/builder.py:3): a config is a dict with
a ``type`` key naming a registered class, the remaining keys are constructor
arguments -- so ``configs/refine_models/scflow.py``'s ``model`` dict builds the
HIP refiner unchanged.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional

__all__ = ['Registry', 'build_from_cfg', 'REFINERS', 'ENCODERS', 'DECODERS', 'HEAD',
           'build_refiner', 'build_encoder', 'build_decoder', 'build_head']


class Registry:
    def __init__(self, name: str) -> None:
        self.name = name
        self._modules: Dict[str, type] = {}

    @property
    def module_dict(self) -> Dict[str, type]:
        return self._modules

    def get(self, key: str) -> Optional[type]:
        return self._modules.get(key)

    def register_module(self, name: Optional[str] = None, force: bool = False,
                        module: Optional[type] = None) -> Callable:
        def _register(cls: type) -> type:
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f'{key} is already registered in {self.name}')
            self._modules[key] = cls
            return cls
        return _register(module) if module is not None else _register

    def __contains__(self, key: str) -> bool:
        return key in self._modules

    def __repr__(self) -> str:
        return f'Registry(name={self.name}, items={sorted(self._modules)})'


def build_from_cfg(cfg: Dict[str, Any], registry: Registry,
                   default_args: Optional[Dict[str, Any]] = None) -> Any:
    if not isinstance(cfg, dict):
        raise TypeError(f'cfg must be a dict, but got {type(cfg)}')
    if 'type' not in cfg:
        raise KeyError(f'`cfg` must contain the key "type", but got {cfg}')
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    kind = args.pop('type')
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f'{kind} is not in the {registry.name} registry')
    elif isinstance(kind, type):
        cls = kind
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(kind)}')
    return cls(**args)


REFINERS = Registry('refiner')
ENCODERS = Registry('encoder')
DECODERS = Registry('decoder')
HEAD = Registry('head')


def build_refiner(cfg):
    return build_from_cfg(cfg, REFINERS)


def build_encoder(cfg):
    return build_from_cfg(cfg, ENCODERS)


def build_decoder(cfg):
    return build_from_cfg(cfg, DECODERS)


def build_head(cfg):
    return build_from_cfg(cfg, HEAD)
