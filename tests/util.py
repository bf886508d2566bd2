"""Shared helpers for the parity tests (test infrastructure).  The builders
live in ``oracle/fixtures.py`` (so that ``smoke()`` does not import the test
tree); this module adds the golden-file loader."""
import os

import numpy as np
import torch

from oracle.fixtures import (AABB4, hip_network_from_oracle,  # noqa: F401
                             lively_oracle_field, make_rays, march_scene,
                             maxabs, slab_near_far)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item())
            for k in z.files}
