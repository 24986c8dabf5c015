from genie2_amd.diffusion import load_pretrained_model  # noqa: F401
