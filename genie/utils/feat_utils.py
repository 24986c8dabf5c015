from genie2_amd.features import *  # noqa: F401,F403
