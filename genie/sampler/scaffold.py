from genie2_amd.sampler import ScaffoldSampler  # noqa: F401
