from genie2_amd.geometry import compute_frenet_frames  # noqa: F401
