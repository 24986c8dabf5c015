from genie2_amd.data import GenieDataModule  # noqa: F401
