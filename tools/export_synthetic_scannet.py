"""CLI of ucsa_neural_rendering_amd.dataset.synthetic_export:
python tools/export_synthetic_scannet.py ROOT [scene_seed] [n_views]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ucsa_neural_rendering_amd.dataset.synthetic_export import (  # noqa: E402,F401
    export, ngp_to_nerf_matrix)

if __name__ == "__main__":
    export(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0,
           int(sys.argv[3]) if len(sys.argv) > 3 else 10)
