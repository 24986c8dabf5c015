from genie2_amd.diffusion import Genie  # noqa: F401
