from genie2_amd.affine import T  # noqa: F401
