"""Drop-in for the reference package ``nr4seg.nerf.raymarching``
(``from .raymarching import raymarching`` in renderer_semantics.py:7)."""
from . import raymarching  # noqa: F401
