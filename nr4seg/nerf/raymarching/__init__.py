from ucsa_neural_rendering_amd.nerf.raymarching import raymarching  # noqa: F401
