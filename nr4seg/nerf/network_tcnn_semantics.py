from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import *  # noqa: F401,F403
from ucsa_neural_rendering_amd.nerf.network_tcnn_semantics import SemanticNeRFNetwork  # noqa: F401
