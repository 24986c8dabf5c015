from genie2_amd.multiprocessor import MultiProcessor  # noqa: F401
