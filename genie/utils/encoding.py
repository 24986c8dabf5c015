from genie2_amd.pack import sinusoidal_encoding  # noqa: F401
