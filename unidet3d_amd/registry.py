"""Registry shim.  With mmdetection3d installed the classes register into its own
``MODELS`` / ``TASK_UTILS`` (so ``custom_imports=dict(imports=['unidet3d_amd'])`` makes the
reference configs build these modules); otherwise a minimal registry with the same
``register_module()`` / ``build(cfg)`` surface is used (mmengine is not in this image)."""
from __future__ import annotations


class _Registry:
    def __init__(self, name):
        self.name = name
        self._mods = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            key = name or cls.__name__
            if key in self._mods and not force:
                raise KeyError(f'{key} already registered in {self.name}')
            self._mods[key] = cls
            return cls
        if module is not None:
            return deco(module)
        return deco

    def get(self, key):
        return self._mods.get(key)

    def build(self, cfg, **default_args):
        if cfg is None:
            return None
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise TypeError(f'{self.name}.build needs a dict with a "type" key, got {cfg!r}')
        args = dict(cfg)
        typ = args.pop('type')
        cls = self._mods.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f'{typ} is not registered in {self.name}')
        for k, v in default_args.items():
            args.setdefault(k, v)
        return cls(**args)


try:  # pragma: no cover - mmdet3d is not installed in the build image
    from mmdet3d.registry import MODELS, TASK_UTILS, TRANSFORMS  # type: ignore
    HAVE_MMDET3D = True
except Exception:  # noqa: BLE001
    MODELS = _Registry('models')
    TASK_UTILS = _Registry('task util')
    TRANSFORMS = _Registry('transform')
    HAVE_MMDET3D = False
