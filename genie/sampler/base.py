from genie2_amd.sampler import BaseSampler  # noqa: F401
