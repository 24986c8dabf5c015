from genie2_amd.diffusion import mse  # noqa: F401
