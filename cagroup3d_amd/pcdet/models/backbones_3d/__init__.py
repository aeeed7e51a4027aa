from .biresnet import BiResNet

__all__ = {"BiResNet": BiResNet}
