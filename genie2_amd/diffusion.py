"""`Genie`: what BaseSampler needs from genie.diffusion.genie.Genie
(genie/diffusion/ddpm.py:10-66, genie/diffusion/genie.py) without Lightning:
`.config`, `.model` (Denoiser), `.device`, `.setup_schedule()` and the schedule
tensors; plus the checkpoint loader of genie/utils/model_io.py:139-173."""
import os

import torch
from torch import nn

from . import pack
from .config import Config
from .model import Denoiser


class Genie(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.config = config
        self.model = Denoiser(**config.model, n_timestep=config.diffusion['n_timestep'],
                              max_n_res=config.io['max_n_res'], max_n_chain=config.io['max_n_chain'])

    @property
    def device(self):
        return next(self.model.parameters()).device

    def setup_schedule(self):
        """ddpm.py:36-66: the terms the samplers index by timestep."""
        if self.config.diffusion['schedule'] != 'cosine':
            raise ValueError('Invalid schedule: {}'.format(self.config.diffusion['schedule']))
        for k, v in pack.schedule_tensors(self.config.diffusion['n_timestep']).items():
            setattr(self, k, v.to(self.device))

    @classmethod
    def load_from_checkpoint(cls, ckpt_filepath, config):
        """Reads a Lightning checkpoint's weights (keys 'model.<denoiser key>')
        without executing anything from the file."""
        ck = torch.load(ckpt_filepath, map_location='cpu', weights_only=True)
        sd = ck['state_dict'] if 'state_dict' in ck else ck
        obj = cls(config)
        obj.model.load_state_dict({k[len('model.'):] if k.startswith('model.') else k: v for k, v in sd.items()})
        return obj


def load_pretrained_model(rootdir, name, epoch):
    """model_io.py:139-173: <rootdir>/<name>/configuration + checkpoints/epoch.<E>.ckpt."""
    basedir = os.path.join(rootdir, name)
    if not os.path.exists(basedir):
        print('Base directory not found at ' + basedir)
        raise SystemExit(0)
    config = Config(os.path.join(basedir, 'configuration'))
    ckpt = os.path.join(basedir, 'checkpoints', 'epoch.{}.ckpt'.format(epoch))
    if not os.path.exists(ckpt):
        print('Missing checkpoint file: ' + ckpt)
        raise SystemExit(0)
    return Genie.load_from_checkpoint(ckpt, config)
