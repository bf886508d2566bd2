"""reference nr4seg/utils/metrics.py."""
from ucsa_neural_rendering_amd.utils.metrics import SemanticsMeter  # noqa: F401
