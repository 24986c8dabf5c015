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
        # what a training run continues from (GenieTrainer.resume): Lightning keeps these next to the weights (train.py:35-39)
        obj.checkpoint_info = {k: ck[k] for k in ('epoch', 'global_step', 'optimizer_states') if isinstance(ck, dict) and k in ck}
        return obj


def load_pretrained_model(rootdir, name, epoch):
    """model_io.py:139-173: <rootdir>/<name>/configuration + checkpoints/epoch.<E>.ckpt."""
    basedir = os.path.join(rootdir, name)
    if not os.path.exists(basedir):
        print('Base directory not found at ' + basedir)
        raise SystemExit(1)          # the reference exits with 0 here (model_io.py:150-152); a missing model is a failure
    config = Config(os.path.join(basedir, 'configuration'))
    ckpt = os.path.join(basedir, 'checkpoints', 'epoch.{}.ckpt'.format(epoch))
    if not os.path.exists(ckpt):
        print('Missing checkpoint file: ' + ckpt)
        raise SystemExit(1)
    return Genie.load_from_checkpoint(ckpt, config)


# ---- training-directory layout (genie/utils/model_io.py:7-137) -----------------------------------------------------
def get_versions(rootdir, name):
    """model_io.py:7-25: <rootdir>/<name>/version_<v> directories, ascending."""
    import glob
    return sorted(int(d.split('_')[-1]) for d in glob.glob(os.path.join(rootdir, name, 'version_*')))


def get_epochs(rootdir, name, version):
    """model_io.py:27-48: version_<v>/checkpoints/epoch=<e>.ckpt, ascending."""
    import glob
    return sorted(int(p.split('=')[-1].split('.')[0])
                  for p in glob.glob(os.path.join(rootdir, name, 'version_{}'.format(version), 'checkpoints', '*.ckpt')))


def load_config(rootdir, name):
    return Config(os.path.join(rootdir, name, 'configuration'))


def load_default_model(rootdir, name):
    return Genie(load_config(rootdir, name))


def load_model(rootdir, name, version=None, epoch=None):
    """model_io.py:84-137: latest version / epoch by default; an untrained Genie when there is no checkpoint."""
    versions = get_versions(rootdir, name)
    if version is None:
        if not versions:
            print('No checkpoint available (version)')
            print('Using default untrained model')
            return load_default_model(rootdir, name)
        version = max(versions)
    else:
        assert version in versions, 'Missing checkpoint version: {}'.format(version)
    epochs = get_epochs(rootdir, name, version)
    if epoch is None:
        if not epochs:
            print('No checkpoint available (epoch)')
            print('Using default untrained model')
            return load_default_model(rootdir, name)
        epoch = max(epochs)
    else:
        assert epoch in epochs, 'Missing checkpoint epoch: {}'.format(epoch)
    ckpt = os.path.join(rootdir, name, 'version_{}'.format(version), 'checkpoints', 'epoch={}.ckpt'.format(epoch))
    print('Loading checkpoint: {}'.format(ckpt))
    return Genie.load_from_checkpoint(ckpt, config=load_config(rootdir, name))


def save_checkpoint(genie, ckpt_filepath, epoch=0, global_step=0, trainer=None):
    """Write a checkpoint in the layout the reference's Lightning `Genie.load_from_checkpoint(path, config=...)` reads and its
    `ModelCheckpoint(filename='{epoch}', save_top_k=-1)` writes (train.py:35-39): `state_dict` with `model.`-prefixed Denoiser keys,
    `optimizer_states` = [torch.optim.Adam.state_dict()] over the parameters in that order, `epoch`, `global_step`.  Containers,
    numbers and tensors only: loadable with weights_only=True.  With `trainer` (a GenieTrainer) the weights and the Adam moments
    come from its device blobs and the live module / engine are left alone; without, weights only (from `genie`)."""
    os.makedirs(os.path.dirname(os.path.abspath(ckpt_filepath)), exist_ok=True)
    src = trainer.state_dict() if trainer is not None else genie.model.state_dict()
    ck = {'epoch': int(epoch), 'global_step': int(global_step), 'pytorch-lightning_version': 'none',
          'state_dict': {'model.' + k: v.detach().cpu() for k, v in src.items()}}
    if trainer is not None:
        ck['optimizer_states'] = [trainer.optimizer_state_dict()]
        ck['lr_schedulers'] = []
    torch.save(ck, ckpt_filepath)


def mse(x_pred, x, mask, aggregate=None, eps=1e-10):
    """genie/utils/loss.py:4-36 (despite the name: masked per-residue L2 error, optionally averaged / summed per sample)."""
    errors = (eps + torch.sum((x_pred - x) ** 2, dim=-1)) ** 0.5
    if aggregate is None:
        return errors * mask
    if aggregate == 'mean':
        return torch.sum(errors * mask, dim=-1) / torch.sum(mask, dim=-1)
    if aggregate == 'sum':
        return torch.sum(errors * mask, dim=-1)
    print('Invalid aggregate method: {}'.format(aggregate))
    raise SystemExit(0)
