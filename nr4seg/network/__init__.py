"""reference nr4seg/network/__init__.py (``from .deeplabv3 import *``)."""
from ucsa_neural_rendering_amd.network.deeplabv3 import *  # noqa: F401,F403
from ucsa_neural_rendering_amd.network import deeplabv3  # noqa: F401
