from genie2_amd.config import Config  # noqa: F401
