"""Convenience constructors: config -> detector, synthetic batch -> batch_dict on a device."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from . import synthetic
from .pcdet.config import cfg_from_yaml_file
from .pcdet.models import build_network

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfgs")


def load_cfg(dataset="scannet"):
    return cfg_from_yaml_file(os.path.join(CFG_DIR, "%s_models" % dataset, "CAGroup3D.yaml"))


def _set_voxel_size(node, vs):
    """Every VOXEL_SIZE key of the (nested) model config (the yaml shares one anchor; BASELINE.json configs[4] asks for 0.01 m)."""
    if isinstance(node, dict):
        for k in list(node.keys()):
            if k == "VOXEL_SIZE":
                node[k] = vs
            else:
                _set_voxel_size(node[k], vs)
    elif isinstance(node, (list, tuple)):
        for v in node:
            _set_voxel_size(v, vs)


VOXEL_SIZE_OF_CONFIG = {"S200k": 0.01}      # synthetic configurations that do not use the yaml's 0.02 m


def build_cagroup3d(dataset="scannet", seed=0, cfg=None, voxel_size=None):
    """CAGroup3D with the reference initialisers under torch.manual_seed(seed)."""
    cfg = cfg or load_cfg(dataset)
    if voxel_size is not None:
        _set_voxel_size(cfg, float(voxel_size))
    torch.manual_seed(seed)
    ds = SimpleNamespace(class_names=cfg.CLASS_NAMES, num_point_features=3, grid_size=None,
                         point_cloud_range=None, voxel_size=None)
    return build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=ds), cfg


def batch_to_device(batch, device):
    out = {}
    for k, v in batch.items():
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(v).float().to(device)
        elif isinstance(v, list) and len(v) and isinstance(v[0], np.ndarray):
            out[k] = [torch.from_numpy(x).to(device) for x in v]
        else:
            out[k] = v
    return out


def synthetic_batch(config="S50k", batch_size=4, first_scene=0, device="cuda"):
    return batch_to_device(synthetic.make_batch(config, batch_size, first_scene), device)
