"""reference nr4seg/dataset/__init__.py.  Only the per-scene NGP dataset of
the hot path is mirrored; the ScanNet-25k datasets
(``ScanNet``, ``ScanNetCL*``, ``ScanNetNGP``) are out of scope."""
from ucsa_neural_rendering_amd.dataset import ngp_utils  # noqa: F401
from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import ScanNetNGPJoint  # noqa: F401

__all__ = ["ScanNetNGPJoint"]
