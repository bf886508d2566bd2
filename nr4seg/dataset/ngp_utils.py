"""reference nr4seg/dataset/ngp_utils.py."""
from ucsa_neural_rendering_amd.dataset.ngp_utils import *  # noqa: F401,F403
from ucsa_neural_rendering_amd.dataset.ngp_utils import get_rays, nerf_matrix_to_ngp  # noqa: F401
