"""reference nr4seg/lightning/__init__.py: the joint-training module and data
module (the pretrain / fine-tune modules are other experiments)."""
from ucsa_neural_rendering_amd.lightning import (JointTrainDataModule,  # noqa: F401
                                                 JointTrainLightningNet)

__all__ = ["JointTrainDataModule", "JointTrainLightningNet"]
