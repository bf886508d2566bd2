from ucsa_neural_rendering_amd.nerf.renderer_semantics import SemanticNeRFRenderer  # noqa: F401
