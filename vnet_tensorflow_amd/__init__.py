"""vnet_tensorflow_amd -- MI355X-native V-Net hot path behind the callable surface of
jackyko1991/vnet-tensorflow (main.py -> model.image2label -> networks.VNet / VNet.VNet, layers2,
dice_coe).  Compute lives in libvnet_hip.so (csrc/, C ABI in include/vnet_hip.h); Python + PyTorch
are host plumbing only (device memory, streams, autograd tape, torch.distributed/RCCL)."""
from . import _lib
from ._lib import VnetHipError, build

__all__ = ["_lib", "VnetHipError", "build", "layers2", "networks", "VNet", "model", "ops", "optim", "parallel", "data"]
