"""Alias package: the reference's module paths (``nr4seg.*``) re-exported from
``ucsa_neural_rendering_amd`` so reference-side code such as
``from nr4seg.nerf.network_tcnn_semantics import SemanticNeRFNetwork``
(nr4seg/lightning/joint_train_lightning_net.py:15) runs unchanged on MI355X."""
from ucsa_neural_rendering_amd import ROOT_DIR  # noqa: F401
