from genie2_amd.sampler import UnconditionalSampler  # noqa: F401
