"""Samplers with the reference's API (genie/sampler/base.py, unconditional.py).

`BaseSampler._sample` keeps the reference's contract -- features from
`create_np_features`, T-step ancestral sampling, the same Gaussian draw order
(1 initial + one per step except the last, each [B,N,3] from the default
generator of the model's device) -- but the loop body never returns to the
host: all T steps are enqueued through `genie_sample_loop`.
"""
import os
from abc import ABC, abstractmethod

import torch

from . import features as F


class BaseSampler(ABC):
    def __init__(self, model):
        """model: a `genie2_amd.diffusion.Genie` (or anything with .device,
        .config, .model = Denoiser and .setup_schedule())."""
        self.model = model
        self.device = model.device
        self.required = ['scale', 'outdir', 'num_samples', 'prefix', 'offset']
        self.model.setup_schedule()
        self.setup()

    @abstractmethod
    def setup(self):
        raise NotImplementedError

    @abstractmethod
    def on_sample_start(self, params):
        raise NotImplementedError

    @abstractmethod
    def create_np_features(self, params):
        raise NotImplementedError

    @abstractmethod
    def on_sample_end(self, params, list_np_features):
        raise NotImplementedError

    def sample(self, params):
        self.validate_parameters(params)          # like the reference, the result is not acted upon (base.py:164)
        self.on_sample_start(params)
        list_np_features = self._sample(params)
        self.on_sample_end(params, list_np_features)

    def draw_noise(self, B, N, T):
        """The reference's draws (base.py:227,269), in its order, on its device."""
        return torch.stack([torch.randn(B, N, 3, device=self.device) for _ in range(T)])

    def _sample(self, params):
        feats = F.convert_np_features_to_tensor(
            F.batchify_np_features([self.create_np_features(params) for _ in range(params['num_samples'])]),
            self.device)
        B, N = feats['residue_mask'].shape
        T = self.model.config.diffusion['n_timestep']
        noise = params.get('noise')
        if noise is None:
            noise = self.draw_noise(B, N, T)
        denoiser = self.model.model
        eng = denoiser.bind(feats)
        trans, _, _ = eng.sample_loop(noise, params['scale'], quat_codes=params.get('quat_codes'))
        feats['atom_positions'] = trans.detach().cpu()
        return F.debatchify_np_features(F.convert_tensor_features_to_numpy(feats))

    def add_required_parameter(self, name):
        self.required.append(name)

    def validate_parameters(self, params):
        return all(name in params for name in self.required)


class UnconditionalSampler(BaseSampler):
    """genie/sampler/unconditional.py:12-137."""

    def setup(self):
        self.add_required_parameter('length')

    def on_sample_start(self, params):
        os.makedirs(os.path.join(params['outdir'], 'pdbs'), exist_ok=True)

    def create_np_features(self, params):
        return F.create_empty_np_features([params['length']])

    def on_sample_end(self, params, list_np_features):
        for i, np_features in enumerate(list_np_features):
            name = '{}_{}'.format(params['prefix'], params['offset'] + i)
            F.save_np_features_to_pdb(np_features, os.path.join(params['outdir'], 'pdbs', name + '.pdb'))


class ScaffoldSampler(BaseSampler):
    """genie/sampler/scaffold.py:12-169: motif scaffolding.  Conditioning enters only through
    the features (aatype, atom_positions, fixed_* masks); scaffold lengths are drawn per sample,
    so batches are ragged and the residue mask matters."""

    def setup(self):
        self.add_required_parameter('filepath')

    def on_sample_start(self, params):
        for sub in ('pdbs', 'motif_pdbs'):
            os.makedirs(os.path.join(params['outdir'], sub), exist_ok=True)

    def create_np_features(self, params):
        return F.create_np_features_from_motif_pdb(params['filepath'])

    def on_sample_end(self, params, list_np_features):
        from .motif import save_motif_pdb
        for i, np_features in enumerate(list_np_features):
            name = '{}_{}'.format(params['prefix'], params['offset'] + i)
            F.save_np_features_to_pdb(np_features, os.path.join(params['outdir'], 'pdbs', name + '.pdb'))
            save_motif_pdb(params['filepath'], np_features['fixed_sequence_mask'],
                           os.path.join(params['outdir'], 'motif_pdbs', name + '.pdb'))
