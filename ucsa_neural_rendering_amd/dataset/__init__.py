from .synthetic_scene import SyntheticRoom, SyntheticSceneDataset  # noqa: F401
from .scannet_ngp_joint import ScanNetNGPJoint  # noqa: F401,E402
