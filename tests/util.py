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


_BENCH_FIELD = {}


def bench_field(dev, train_steps=200):
    """The bench's parameter state for the parity tests: ``bench.build_field``
    trained in DETERMINISTIC mode, once per session; every
    caller gets a fresh network holding a copy of it.  The default (float-atomic)
    training gave every box -- and every run -- a different field, so a failure of
    the whole-view test could not be replayed; the sweep over many fields lives in
    tests/scripts/whole_view_seeds.py, outside ``-m gpu``."""
    import bench
    from tools.bench_legs.common import field_checksum
    seed = int(os.environ.get("UCSA_TEST_FIELD_SEED", "123"))     # (whole_view_seeds.py: other fields)
    key = (str(dev), train_steps, seed)
    if key not in _BENCH_FIELD:
        net, ds = bench.build_field(dev, seed=seed, train_steps=train_steps, deterministic=True)
        state = [p.detach().clone() for p in (net.encoder.params, net.sigma_net.params,
                                              net.color_net.params, net.semantics_net.params)]
        _BENCH_FIELD[key] = (state, ds, field_checksum(net))
    state, ds, chk = _BENCH_FIELD[key]
    net, _ = bench.build_field(dev, train_steps=0)
    with torch.no_grad():
        for p, v in zip((net.encoder.params, net.sigma_net.params, net.color_net.params,
                         net.semantics_net.params), state):
            p.copy_(v)
    print(f"bench field (seed {seed}, deterministic training, {train_steps} steps): checksum {chk}")
    return net, ds
