from genie2_amd.engine import compute_frenet_frames  # noqa: F401
