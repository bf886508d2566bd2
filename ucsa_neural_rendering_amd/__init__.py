"""MI355X-native (gfx950) implementation of the Semantic-NeRF volume-rendering
hot path of ethz-asl/ucsa_neural_rendering, behind the reference's own module
API (``nr4seg.nerf.network_tcnn_semantics.SemanticNeRFNetwork`` etc.).

Arithmetic runs in hand-written HIP kernels reached through the C ABI of
``libucsa_hip.so`` (``include/ucsa_hip.h``); there is no CPU fallback.
"""
import os

ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))

if "ENV_WORKSTATION_NAME" not in os.environ:  # reference nr4seg/__init__.py:5-6
    os.environ["ENV_WORKSTATION_NAME"] = "env"

from ._miopen_db import use_shipped_db as _use_shipped_db  # noqa: E402

MIOPEN_DB_PATH = _use_shipped_db()
