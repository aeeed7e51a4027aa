"""Model API boundary (mirror of pcdet/models/__init__.py:16-52)."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset)


def load_data_to_gpu(batch_dict, device="cuda"):
    """numpy -> device tensors; mask lists / metadata stay on the host (models/__init__.py:23-34)."""
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray) or key in ("frame_id", "metadata", "calib"):
            continue
        if key in ("image_shape",):
            batch_dict[key] = torch.from_numpy(val).int().to(device)
        else:
            batch_dict[key] = torch.from_numpy(val).float().to(device)


def model_fn_decorator():
    ModelReturn = namedtuple("ModelReturn", ["loss", "tb_dict", "disp_dict"])

    def model_func(model, batch_dict):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict["loss"].mean()
        (model.module if hasattr(model, "module") else model).update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)
    return model_func
