from genie2_amd.data import GenieDataset  # noqa: F401
