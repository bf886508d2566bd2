from ucsa_neural_rendering_amd.nerf.activation import trunc_exp  # noqa: F401
