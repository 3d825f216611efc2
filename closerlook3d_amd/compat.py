"""Configuration and checkpoint compatibility with the reference tree.

  * `default_config()` / `load_config(yaml)`  the reference's option tree (utils/config.py:4-103) and its
    YAML merge rule (`update_config`, :106-117: top-level keys must exist, nested dicts are merged one level
    deep), so the 20 files under cfgs/ load unchanged;
  * `build_model(config, task)`               models/build.py's three wrappers (`backbone` + `classifier` /
    `segmentation_head` attribute names, hence the same state-dict keys) on this package's modules;
  * `load_reference_checkpoint(model, path)`  reads a `.pth` written by the reference's `save_checkpoint`
    (function/train_*_dist.py: a dict with 'model', 'optimizer', 'scheduler', 'epoch', 'config', ...),
    strips DistributedDataParallel's `module.` prefix and loads it with strict key checking.  The pickled
    'config' entry is an `easydict.EasyDict`; when that package is absent a minimal stand-in class is
    registered for the duration of the load (it only has to unpickle, never to behave).
"""
import copy
import pickle
import sys
import types

import torch
import torch.nn as nn

from .backbones import ClassifierResNet, MultiPartSegHeadResNet, ResNet, SceneSegHeadResNet


class Config(dict):
    """dict with attribute access, nested dicts converted on assignment (what the reference gets from easydict)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def __setitem__(self, k, v):
        super().__setitem__(k, Config(v) if isinstance(v, dict) and not isinstance(v, Config) else v)

    def __deepcopy__(self, memo):
        return Config({k: copy.deepcopy(v, memo) for k, v in self.items()})


def default_config():
    c = Config()
    # training
    c.update(epochs=600, start_epoch=1, base_learning_rate=0.01, lr_scheduler='step', optimizer='sgd',
             warmup_epoch=5, warmup_multiplier=100, lr_decay_steps=20, lr_decay_rate=0.7, weight_decay=0,
             momentum=0.9, grid_clip_norm=-1)
    # model
    c.update(backbone='resnet', head='resnet_cls', radius=0.05, sampleDl=0.02, density_parameter=5.0,
             nsamples=[], npoints=[], width=144, depth=2, bottleneck_ratio=2, bn_momentum=0.1)
    # data
    c.update(datasets='modelnet40', data_root='', num_classes=40, num_parts=0, input_features_dim=3,
             batch_size=32, num_points=5000, num_workers=4, x_angle_range=0.0, y_angle_range=0.0,
             z_angle_range=0.0, scale_low=2. / 3., scale_high=3. / 2., noise_std=0.01, noise_clip=0.05,
             translate_range=0.2, color_drop=0.2, augment_symmetries=[0, 0, 0], in_radius=2.0, num_steps=500)
    # io and misc
    c.update(load_path='', print_freq=10, save_freq=10, val_freq=10, log_dir='log', local_rank=0,
             amp_opt_level='', rng_seed=0)
    # local aggregation
    c.local_aggregation_type = 'pospool'
    c.pospool = dict(position_embedding='xyz', reduction='sum', output_conv=False)
    c.adaptive_weight = dict(weight_type='dp', num_mlps=1, shared_channels=1, weight_softmax=False,
                             reduction='avg', output_conv=False)
    c.pointwisemlp = dict(feature_type='dp_fj', num_mlps=1, reduction='max')
    c.pseudo_grid = dict(fixed_kernel_points='center', KP_influence='linear', KP_extent=1.0,
                         num_kernel_points=15, convolution_mode='sum', output_conv=False)
    for k, v in list(c.items()):  # dict.update() bypasses __setitem__: convert the nested dicts
        c[k] = v
    return c


def load_config(yaml_path, base=None):
    """The reference's `update_config`: unknown top-level keys raise, nested option groups merge key by key."""
    import yaml
    cfg = copy.deepcopy(base) if base is not None else default_config()
    with open(yaml_path) as fh:
        exp = yaml.safe_load(fh) or {}
    for k, v in exp.items():
        if k not in cfg:
            raise ValueError(f"{k} key must exist in config.py")
        if isinstance(v, dict):
            for vk, vv in v.items():
                cfg[k][vk] = vv
        else:
            cfg[k] = v
    return cfg


class _Model(nn.Module):
    def _backbone(self, config):
        if config.backbone != 'resnet':
            raise NotImplementedError(f"Backbone {config.backbone} not implemented")
        return ResNet(config, config.input_features_dim, config.radius, config.sampleDl, config.nsamples,
                      config.npoints, width=config.width, depth=config.depth,
                      bottleneck_ratio=config.bottleneck_ratio)

    def init_weights(self):
        """models/build.py:55-62: Kaiming-normal conv weights, zero conv biases."""
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Conv1d)):
                torch.nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    torch.nn.init.zeros_(m.bias)


class ClassificationModel(_Model):
    """models/build.py:37-62."""

    def __init__(self, config):
        super().__init__()
        self.backbone = self._backbone(config)
        if config.head != 'resnet_cls':
            raise NotImplementedError(f"Head {config.head} not implemented in Classification Model")
        self.classifier = ClassifierResNet(config.num_classes, config.width)

    def forward(self, xyz, mask, features):
        return self.classifier(self.backbone(xyz, mask, features))


class MultiPartSegmentationModel(_Model):
    """models/build.py:65-92."""

    def __init__(self, config):
        super().__init__()
        self.backbone = self._backbone(config)
        if config.head != 'resnet_part_seg':
            raise NotImplementedError(f"Head {config.head} not implemented in Multi-Part Segmentation Model")
        self.segmentation_head = MultiPartSegHeadResNet(config.num_classes, config.width, config.radius,
                                                        config.nsamples, config.num_parts, config=config)

    def forward(self, xyz, mask, features):
        return self.segmentation_head(self.backbone(xyz, mask, features))


class SceneSegmentationModel(_Model):
    """models/build.py:95-121."""

    def __init__(self, config):
        super().__init__()
        self.backbone = self._backbone(config)
        if config.head != 'resnet_scene_seg':
            raise NotImplementedError(f"Head {config.head} not implemented in Scene Segmentation Model")
        self.segmentation_head = SceneSegHeadResNet(config.num_classes, config.width, config.radius, config.nsamples,
                                                    config=config)

    def forward(self, xyz, mask, features):
        return self.segmentation_head(self.backbone(xyz, mask, features))


_TASKS = {'classification': ClassificationModel, 'multi_part_segmentation': MultiPartSegmentationModel,
          'scene_segmentation': SceneSegmentationModel}
_HEAD_TASK = {'resnet_cls': 'classification', 'resnet_part_seg': 'multi_part_segmentation',
              'resnet_scene_seg': 'scene_segmentation'}


def build_model(config, task=None):
    """The model the reference's `build_<task>(config)` returns (without the criterion); task defaults to the
    one `config.head` belongs to."""
    return _TASKS[task or _HEAD_TASK[config.head]](config)


def _strip_module(state):
    return {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in state.items()}


class _ConfigDict(dict):
    """Stand-in for easydict.EasyDict when a checkpoint's 'config' entry is unpickled (attribute access, data only)."""

    def __setstate__(self, st):
        self.update(st or {})

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


_SAFE_GLOBALS = {
    ('collections', 'OrderedDict'), ('torch._utils', '_rebuild_tensor_v2'), ('torch._utils', '_rebuild_parameter'),
    ('torch._utils', '_rebuild_tensor'), ('torch', 'Size'), ('torch', 'device'), ('torch', 'dtype'),
    ('numpy.core.multiarray', 'scalar'), ('numpy._core.multiarray', 'scalar'), ('numpy', 'dtype'),
    ('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'), ('numpy', 'ndarray'),
    ('builtins', 'set'), ('builtins', 'frozenset'), ('builtins', 'slice'), ('builtins', 'complex'),
    # scheduler.state_dict() is part of every checkpoint the reference writes (train_modelnet_dist.py:157-164): MultiStepLR
    # -- its default 'step' schedule, also nested under GradualWarmupScheduler's 'after_scheduler'
    # (utils/lr_scheduler.py:41-50) -- keeps its milestones in a collections.Counter
    ('collections', 'Counter'), ('collections', 'defaultdict'),
}


class _RestrictedUnpickler(pickle.Unpickler):
    """Unpickler for reference checkpoints: tensors, containers, numpy scalars and the config's EasyDict (mapped to a
    plain attribute dict) -- nothing that can run code.  torch.load subclasses whatever `pickle_module.Unpickler` it is
    given and defers unknown globals to it, so this is the gate every GLOBAL opcode of the file goes through."""

    def find_class(self, module, name):
        if (module, name) == ('easydict', 'EasyDict'):
            return _ConfigDict
        if (module, name) == (__name__, 'Config'):
            return Config
        if (module, name) in _SAFE_GLOBALS or (module == 'torch' and name.endswith('Storage')):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"checkpoint references {module}.{name}; pass trusted=True to "
                                     "load_reference_checkpoint if the file comes from a source you trust")


def _restricted_pickle_module():
    mod = types.ModuleType('cl3d_restricted_pickle')
    for k in ('load', 'loads', 'dump', 'dumps', 'Pickler', 'PickleError', 'PicklingError', 'UnpicklingError',
              'HIGHEST_PROTOCOL', 'DEFAULT_PROTOCOL'):
        setattr(mod, k, getattr(pickle, k))
    mod.Unpickler = _RestrictedUnpickler
    return mod


def load_reference_checkpoint(model, path, strict=True, map_location='cpu', trusted=False):
    """Load the 'model' entry of a checkpoint written by the reference's `save_checkpoint` (or a bare state dict).
    Returns the rest of the checkpoint (epoch, best_acc / best_miou, config, optimizer and scheduler state).

    Only the weights are needed, but the reference pickles its EasyDict config, optimizer and scheduler state into the
    same file.  The file is therefore read through a restricted unpickler (tensors, containers, numpy scalars, the
    config as a plain attribute dict): a checkpoint from an untrusted source cannot run code.  A file that holds other
    pickled objects raises, naming the first one; `trusted=True` then falls back to the unrestricted loader."""
    stub = None
    if not trusted:
        ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_restricted_pickle_module())
        state = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
        model.load_state_dict(_strip_module(state), strict=strict)
        return {k: v for k, v in ckpt.items() if k != 'model'} if isinstance(ckpt, dict) and 'model' in ckpt else {}
    if 'easydict' not in sys.modules:
        try:
            import easydict  # noqa: F401
        except ImportError:
            stub = types.ModuleType('easydict')
            stub.EasyDict = type('EasyDict', (dict,), {'__setstate__': lambda self, st: self.update(st or {})})
            sys.modules['easydict'] = stub
    try:
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
    finally:
        if stub is not None:
            sys.modules.pop('easydict', None)
    state = ckpt['model'] if isinstance(ckpt, dict) and 'model' in ckpt else ckpt
    model.load_state_dict(_strip_module(state), strict=strict)
    return {k: v for k, v in ckpt.items() if k != 'model'} if isinstance(ckpt, dict) and 'model' in ckpt else {}
