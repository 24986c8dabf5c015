"""compute_frenet_frames (genie/utils/geo_utils.py:21-85) on the GPU."""
import ctypes as C

import torch

from . import capi


def compute_frenet_frames(coords, chains, mask, engine=None):
    """coords [B,N,3] on a GPU, chains/mask [B,N].  With `engine` (a bound
    GenieEngine) the frames come from the batch already bound; otherwise a
    throw-away binding is made through `engine_for` of the caller."""
    if engine is None:
        raise capi.GenieError('compute_frenet_frames needs a bound GenieEngine (Denoiser.bind(features))')
    return engine.frenet(coords)
