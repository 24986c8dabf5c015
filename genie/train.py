from genie2_amd.train import main  # noqa: F401
