"""Small host-side helpers of the training surface (reference ``utils/detection_utils.py:7-29``
and the OmegaConf usage in ``model.py:27-31, :76-90``)."""
import importlib
import os
from typing import Any

import yaml


def collate_fn(batch):
    "Detection batches are tuples of per-sample tuples (images, targets, ids): no stacking."
    return tuple(zip(*batch))


def load_obj(obj_path: str, default_obj_path: str = "") -> Any:
    """Import ``pkg.mod.Name`` and return ``Name`` (used for optimizers / schedulers named in hparams)."""
    parts = obj_path.rsplit(".", 1)
    module_path, name = (parts[0], parts[1]) if len(parts) > 1 else (default_obj_path, parts[0])
    module = importlib.import_module(module_path)
    if not hasattr(module, name):
        raise AttributeError(f"Object `{name}` cannot be loaded from `{module_path}`.")
    return getattr(module, name)


class AttrDict(dict):
    """dict with attribute access and ``**`` unpacking: stands in for OmegaConf's DictConfig
    (``conf.model.backbone_kind``, ``Retinanet(**conf.model)``), which is not installed here."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = _wrap(value)


def _wrap(v):
    if isinstance(v, dict) and not isinstance(v, AttrDict):
        return AttrDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return [_wrap(x) for x in v]
    return v


def load_hparams(path: str = None) -> AttrDict:
    """Load an ``hparams.yaml`` (default: the one shipped with the package; same keys as the reference's)."""
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "hparams.yaml")
    with open(path) as f:
        return _wrap(yaml.safe_load(f))
