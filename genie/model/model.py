from genie2_amd.model import Denoiser  # noqa: F401
