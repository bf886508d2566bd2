"""reference nr4seg/dataset/scannet_ngp_joint.py."""
from ucsa_neural_rendering_amd.dataset.scannet_ngp_joint import ScanNetNGPJoint  # noqa: F401
