from .synthetic_scene import SyntheticRoom, SyntheticSceneDataset  # noqa: F401
