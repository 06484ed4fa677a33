"""The det3d registry / builder / config interface the reference's configs and tools talk to -- same names,
argument meaning and error behaviour (det3d/utils/registry.py:6-78, det3d/models/registry.py:3-11,
det3d/models/builder.py:17-52, det3d/torchie/utils/config.py:12-100) -- plus a shim that makes
``configs/cruw_pose/*.py`` importable unchanged on a machine without det3d's dependency tree.

Quirks absorbed: HRNet3D is registered unconditionally (the reference registers it only when
spconv is importable, det3d/models/__init__.py:2-7); ``cfg.enable_amp`` missing from a config reads as False
(torchie/apis/train.py:296 raises AttributeError on hr3d.py).
"""
import importlib
import inspect
import os
import sys
import types


class Registry(object):
    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __repr__(self):
        return "%s(name=%s, items=%s)" % (self.__class__.__name__, self._name, list(self._module_dict.keys()))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register_module(self, module_class):
        if not inspect.isclass(module_class):
            raise TypeError("module must be a class, but got {}".format(type(module_class)))
        name = module_class.__name__
        if name in self._module_dict:
            raise KeyError("{} is already registered in {}".format(name, self.name))
        self._module_dict[name] = module_class

    def register_module(self, cls):  # used as a bare decorator: @REG.register_module
        self._register_module(cls)
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(cfg, dict) and "type" in cfg
    assert isinstance(default_args, dict) or default_args is None
    args = dict(cfg)
    obj_type = args.pop("type")
    if isinstance(obj_type, str):
        obj_cls = registry.get(obj_type)
        if obj_cls is None:
            raise KeyError("{} is not in the {} registry".format(obj_type, registry.name))
    elif inspect.isclass(obj_type):
        obj_cls = obj_type
    else:
        raise TypeError("type must be a str or valid type, but got {}".format(type(obj_type)))
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    return obj_cls(**args)


READERS = Registry("reader")
BACKBONES = Registry("backbone")
FEAT_TRANSFORMS = Registry("feat_transform")
NECKS = Registry("neck")
HEADS = Registry("head")
LOSSES = Registry("loss")
DETECTORS = Registry("detector")
SECOND_STAGE = Registry("second_stage")
ROI_HEAD = Registry("roi_head")


def build(cfg, registry, default_args=None):
    from . import modules  # noqa: F401  (registers RadarPoseNet / HRNet3D / CenterHead / RadarFeatureNet on first use)
    if isinstance(cfg, list):
        import torch.nn as nn
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_reader(cfg):
    return build(cfg, READERS)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_feat_transform(cfg):
    return build(cfg, FEAT_TRANSFORMS)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


# ---------------------------------------------------------------------------------------------- config
class ConfigDict(dict):
    """Attribute-style dict; a missing key raises AttributeError like the reference's addict subclass
    (torchie/utils/config.py:12-29)."""

    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = self._wrap(v)

    @classmethod
    def _wrap(cls, v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._wrap(x) for x in v)
        return v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'{}' object has no attribute '{}'".format(self.__class__.__name__, name))

    def __setattr__(self, name, value):
        self[name] = self._wrap(value)


class Config(object):
    """Config.fromfile(path.py): the file is imported as a module and its public names become the config
    (torchie/utils/config.py:78-100)."""

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, "_cfg_dict", ConfigDict(cfg_dict or {}))
        object.__setattr__(self, "_filename", filename)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        if not filename.endswith(".py"):
            raise IOError("Only py type are supported now!")
        install_det3d_shim()
        module_name = os.path.basename(filename)[:-3]
        if "." in module_name:
            raise ValueError("Dots are not allowed in config file path.")
        config_dir = os.path.dirname(filename)
        sys.path.insert(0, config_dir)
        try:
            sys.modules.pop(module_name, None)
            mod = importlib.import_module(module_name)
        finally:
            sys.path.pop(0)
        cfg = {k: v for k, v in mod.__dict__.items() if not k.startswith("__") and not inspect.ismodule(v)
               and not inspect.isfunction(v) and not inspect.isclass(v)}
        sys.modules.pop(module_name, None)
        return Config(cfg, filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        if name == "enable_amp" and name not in self._cfg_dict:
            return False
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def __repr__(self):
        return "Config (path: {}): {}".format(self._filename, dict.__repr__(self._cfg_dict))


def get_downsample_factor(model_config):
    """det3d/utils/config_tool.py:39-52 (without its protobuf import)."""
    import numpy as np
    try:
        neck_cfg = model_config["neck"]
    except Exception:
        model_config = model_config["first_stage_cfg"]
        neck_cfg = model_config["neck"]
    neck_cfg = neck_cfg or {}
    factor = np.prod(neck_cfg.get("ds_layer_strides", [1]))
    if len(neck_cfg.get("us_layer_strides", [])) > 0:
        factor /= neck_cfg.get("us_layer_strides", [])[-1]
    factor *= model_config["backbone"]["ds_factor"]
    return int(factor)


# ---------------------------------------------------------------------------------------------- det3d shim
_SHIM_DONE = False


def install_det3d_shim(force=False):
    """Make ``import det3d...`` / ``import munch`` in the reference config files resolve to this package.

    If a REAL det3d is importable, its registries receive our classes instead (same names; a name the real package
    already registered is replaced so `build_detector(cfg.model)` returns the MI355X implementation)."""
    global _SHIM_DONE
    if _SHIM_DONE and not force:
        return
    from . import modules  # registers the classes below in this module's registries  # noqa: F401
    real = None
    if "det3d" not in sys.modules or not getattr(sys.modules["det3d"], "__rtp_shim__", False):
        try:
            real = importlib.import_module("det3d.models.registry")
        except Exception:
            real = None
    if real is not None:
        from . import deform_conv_cuda as _dcc
        sys.modules.setdefault("det3d.ops.dcn.deform_conv_cuda", _dcc)   # picked up by `from . import deform_conv_cuda`
        for ours, theirs in ((READERS, real.READERS), (BACKBONES, real.BACKBONES), (HEADS, real.HEADS),
                             (DETECTORS, real.DETECTORS)):
            for name, cls in ours.module_dict.items():
                theirs.module_dict[name] = cls
    else:
        def mk(name):
            m = types.ModuleType(name)
            m.__rtp_shim__ = True
            m.__path__ = []
            sys.modules[name] = m
            if "." in name:
                parent, child = name.rsplit(".", 1)
                setattr(sys.modules[parent], child, m)
            return m

        me = sys.modules[__name__]
        det3d = mk("det3d")
        utils = mk("det3d.utils")
        utils.Registry, utils.build_from_cfg = Registry, build_from_cfg
        ct = mk("det3d.utils.config_tool")
        ct.get_downsample_factor = get_downsample_factor
        models = mk("det3d.models")
        reg = mk("det3d.models.registry")
        bld = mk("det3d.models.builder")
        for k in ("READERS", "BACKBONES", "FEAT_TRANSFORMS", "NECKS", "HEADS", "LOSSES", "DETECTORS", "SECOND_STAGE", "ROI_HEAD"):
            setattr(reg, k, getattr(me, k))
            setattr(models, k, getattr(me, k))
        for k in ("build", "build_reader", "build_backbone", "build_neck", "build_head", "build_loss",
                  "build_feat_transform", "build_detector"):
            setattr(bld, k, getattr(me, k))
            setattr(models, k, getattr(me, k))
        # native-op seam: `from . import deform_conv_cuda` in det3d/ops/dcn/deform_conv.py:11 resolves to the ctypes binding
        from . import dcn as _dcn, deform_conv_cuda as _dcc
        ops = mk("det3d.ops")
        opd = mk("det3d.ops.dcn")
        sys.modules["det3d.ops.dcn.deform_conv_cuda"] = _dcc
        opd.deform_conv_cuda = _dcc
        for k in ("DeformConv", "DeformConvPack", "ModulatedDeformConv", "ModulatedDeformConvPack", "DeformConvFunction",
                  "ModulatedDeformConvFunction", "deform_conv", "modulated_deform_conv"):
            setattr(opd, k, getattr(_dcn, k))
        sys.modules["det3d.ops.dcn.deform_conv"] = _dcn
        opd.deform_conv = _dcn.deform_conv
        torchie = mk("det3d.torchie")
        torchie.Config = Config
        torchie.is_str = lambda x: isinstance(x, str)
        tu = mk("det3d.torchie.utils")
        tu.Config, tu.ConfigDict = Config, ConfigDict
        det3d.__version__ = "rt_pose_amd-shim"
    try:
        importlib.import_module("munch")
    except Exception:
        munch = types.ModuleType("munch")

        class DefaultMunch(ConfigDict):
            @classmethod
            def fromDict(cls, d, default=None):
                return cls(d)

        munch.DefaultMunch = DefaultMunch
        munch.Munch = ConfigDict
        sys.modules["munch"] = munch
    _SHIM_DONE = True
