from .joint_train_data_module import JointTrainDataModule  # noqa: F401
from .joint_train_lightning_net import JointTrainLightningNet  # noqa: F401
from .trainer import Trainer, seed_everything  # noqa: F401

__all__ = ["JointTrainDataModule", "JointTrainLightningNet", "Trainer",
           "seed_everything"]
