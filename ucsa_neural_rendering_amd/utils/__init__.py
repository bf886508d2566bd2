from .metrics import SemanticsMeter  # noqa: F401
from .loading import load_yaml  # noqa: F401
from .flatten_dict import flatten_dict  # noqa: F401
