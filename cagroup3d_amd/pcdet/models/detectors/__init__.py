from .cagroup3d import CAGroup3D
from .detector3d_template import Detector3DTemplate

__all__ = {"Detector3DTemplate": Detector3DTemplate, "CAGroup3D": CAGroup3D}


def build_detector(model_cfg, num_class, dataset):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset)
