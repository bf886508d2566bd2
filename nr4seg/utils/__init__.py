"""reference nr4seg/utils/__init__.py (flatten_dict, loading; the logger
factories are observability)."""
from ucsa_neural_rendering_amd.utils.flatten_dict import *  # noqa: F401,F403
from ucsa_neural_rendering_amd.utils.loading import *  # noqa: F401,F403
from ucsa_neural_rendering_amd.utils import metrics  # noqa: F401
