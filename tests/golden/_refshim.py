"""DEV-ONLY import shim used by ``make_golden.py`` in the build container.

The reference (YangHai-1218/SCFlow) cannot be imported as shipped: its
third-party dependencies (mmcv 1.3.16, kornia, pytorch3d, cv2, trimesh, ...)
are absent from the image and ``models/__init__.py`` imports a name that does
not exist.  This module lets the reference's own
hot-path source files execute UNMODIFIED from ``/root/reference`` by

  1. registering permissive *import-only* stub modules for packages the hot
     path imports but never calls (cv2, kornia, pytorch3d, ...);
  2. providing a hand-written mini-mmcv (Registry, build_from_cfg, BaseModule,
     Sequential, ConvModule, build_conv_layer, build_norm_layer,
     build_activation_layer).  Its semantics restate mmcv 1.3.16 from
     knowledge of that package -- UNVERIFIED against the real thing;
  3. installing a synthetic parent package ``models`` whose ``__path__``
     points at the reference directory, so ``models/__init__.py`` (broken) is
     never executed.

Nothing here is reference code, nothing here ships to the GPU box as a
dependency of anything that runs there, and nothing in ``scflow_amd`` or
``oracle`` imports it.
"""
import sys
import types

import torch.nn as nn

REFERENCE_ROOT = '/root/reference'


class _Permissive(types.ModuleType):
    """module whose unknown attributes resolve to dummy classes/submodules."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        if name[0].isupper():
            obj = type(name, (), {'__init__': lambda self, *a, **k: None})
        else:
            obj = _Permissive(self.__name__ + '.' + name)
            sys.modules[obj.__name__] = obj
        setattr(self, name, obj)
        return obj


def _stub(name):
    parts = name.split('.')
    for i in range(1, len(parts) + 1):
        n = '.'.join(parts[:i])
        if n not in sys.modules:
            m = _Permissive(n)
            m.__path__ = []
            sys.modules[n] = m
            if i > 1:
                setattr(sys.modules['.'.join(parts[:i - 1])], parts[i - 1], m)
    return sys.modules[name]


# ---------------------------------------------------------------- mini-mmcv
class Registry:
    def __init__(self, name):
        self.name = name
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self.module_dict[name or cls.__name__] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self.module_dict.get(key)


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    kind = args.pop('type')
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    cls = registry.get(kind) if isinstance(kind, str) else kind
    if cls is None:
        raise KeyError(f'{kind} is not in the {registry.name} registry')
    return cls(**args)


class BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class Sequential(BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get('type') in ('Conv2d', 'Conv')
    return nn.Conv2d(*args, **kwargs)


_NORMS = {'BN': ('bn', nn.BatchNorm2d), 'SyncBN': ('bn', nn.BatchNorm2d),
          'IN': ('in', nn.InstanceNorm2d), 'GN': ('gn', nn.GroupNorm)}


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    kind = cfg.pop('type')
    abbr, cls = _NORMS[kind]
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    layer = cls(num_channels=num_features, **cfg) if kind == 'GN' else cls(num_features, **cfg)
    for prm in layer.parameters():
        prm.requires_grad = requires_grad
    return abbr + str(postfix), layer


_ACTS = {'ReLU': nn.ReLU, 'LeakyReLU': nn.LeakyReLU, 'Sigmoid': nn.Sigmoid, 'Tanh': nn.Tanh}


def build_activation_layer(cfg):
    cfg = dict(cfg)
    return _ACTS[cfg.pop('type')](**cfg)


class ConvModule(nn.Module):
    """conv -> norm -> act; bias='auto' means bias iff there is no norm."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, **_):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = build_conv_layer(conv_cfg, in_channels, out_channels, kernel_size,
                                     stride=stride, padding=padding, dilation=dilation,
                                     groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            act = dict(act_cfg)
            if act['type'] not in ('Tanh', 'PReLU', 'Sigmoid', 'HSigmoid', 'Swish'):
                act.setdefault('inplace', inplace)
            self.activate = build_activation_layer(act)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def install():
    """register stubs + mini-mmcv + the synthetic ``models`` parent package."""
    for n in ['cv2', 'kornia', 'kornia.augmentation', 'kornia.geometry',
              'kornia.geometry.conversions', 'trimesh', 'turtle', 'pytorch3d', 'pytorch3d.ops',
              'pytorch3d.structures', 'pytorch3d.renderer', 'pytorch3d.renderer.mesh',
              'pytorch3d.renderer.mesh.renderer', 'pytorch3d.io', 'pytorch3d.io.ply_io',
              'iopath', 'iopath.common', 'iopath.common.file_io', 'torchvision',
              'torchvision.utils', 'pycocotools', 'pycocotools.mask', 'terminaltables',
              'tensorboardX', 'transforms3d', 'mmcv', 'mmcv.cnn', 'mmcv.runner', 'mmcv.utils',
              'mmcv.ops', 'mmcv.ops.roi_align', 'mmcv.runner.hooks', 'mmcv.runner.hooks.logger',
              'mmcv.runner.dist_utils', 'mmcv.parallel', 'mmcv.image']:
        _stub(n)
    sys.modules['turtle'].forward = None
    sys.modules['pytorch3d.ops'].knn_points = None
    sys.modules['torchvision.utils'].save_image = None
    sys.modules['pytorch3d.structures'].join_meshes_as_batch = None
    cnn = sys.modules['mmcv.cnn']
    cnn.ConvModule = ConvModule
    cnn.build_conv_layer = build_conv_layer
    cnn.build_norm_layer = build_norm_layer
    cnn.build_activation_layer = build_activation_layer
    cnn.build_plugin_layer = None
    run = sys.modules['mmcv.runner']
    run.BaseModule = BaseModule
    run.Sequential = Sequential
    utl = sys.modules['mmcv.utils']
    utl.Registry = Registry
    utl.build_from_cfg = build_from_cfg
    sys.modules['mmcv.ops'].Correlation = type('Correlation', (), {})
    sys.modules['mmcv.ops.roi_align'].roi_align = None

    class _Hooks:
        def register_module(self, *a, **k):
            return lambda c: c
    hooks = sys.modules['mmcv.runner.hooks']
    hooks.HOOKS = _Hooks()
    hooks.Hook = object
    sys.modules['mmcv.runner.hooks.logger'].TensorboardLoggerHook = object
    sys.modules['mmcv.runner.hooks.logger'].TextLoggerHook = object
    sys.modules['mmcv.runner.dist_utils'].master_only = lambda f: f

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    pkg = types.ModuleType('models')
    pkg.__path__ = [REFERENCE_ROOT + '/models']
    sys.modules['models'] = pkg
