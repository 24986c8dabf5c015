"""Training data: the reference's dataset and data module (genie/data/dataset.py:13-249, genie/data/data_module.py:12-300)
without Lightning.  `GenieDataset.__getitem__` = create_np_features_from_pdb -> (with probability motif_prob) Algorithm 1's
motif masks -> pad_np_features; the random draws come from numpy's and Python's global generators in the reference's order,
so the same seeds give the same masks (tests/golden/dataset_items.npz was recorded from the reference's own class)."""
import glob
import os
import random

import numpy as np
from torch.utils.data import DataLoader, Dataset

from . import features as F


class GenieDataset(Dataset):
    def __init__(self, dataset_info, min_n_res, max_n_res, max_n_chain, motif_prob, motif_min_pct_res, motif_max_pct_res,
                 motif_min_n_seg, motif_max_n_seg):
        super().__init__()
        self.min_n_res, self.max_n_res, self.max_n_chain = min_n_res, max_n_res, max_n_chain
        self.motif_prob = motif_prob
        self.motif_min_pct_res, self.motif_max_pct_res = motif_min_pct_res, motif_max_pct_res
        self.motif_min_n_seg, self.motif_max_n_seg = motif_min_n_seg, motif_max_n_seg
        self.filepaths = self._get_filepaths(dataset_info)
        print('Dataset size: {}'.format(len(self.filepaths)))

    def __len__(self):
        return len(self.filepaths)

    def __getitem__(self, idx):
        np_features = F.create_np_features_from_pdb(self.filepaths[idx])
        if np.random.random() <= self.motif_prob:
            np_features = self._update_motif_masks(np_features)
        return F.pad_np_features(np_features, self.max_n_chain, self.max_n_res)

    def _get_filepaths(self, dataset_info):
        """dataset.py:149-171: <name>.pdb.gz first, then <name>.pdb, only files that exist, shuffled (Python's generator)."""
        paths = [os.path.join(dataset_info['datadir'], f'{name}.pdb.gz') for name in dataset_info['names']]
        paths.extend(os.path.join(dataset_info['datadir'], f'{name}.pdb') for name in dataset_info['names'])
        paths = [p for p in paths if os.path.exists(p)]
        random.shuffle(paths)
        return paths

    def _update_motif_masks(self, np_features):
        """dataset.py:173-249 (Algorithm 1): number of motif residues, number of segments, segment lengths, then a shuffle of
        segments and single scaffold residues.  fixed_group stays 0 (single-motif training)."""
        assert np_features['num_chains'] == 1, 'Input must be monomer'
        n = np_features['num_residues']
        motif_n_res = np.random.randint(np.floor(n * self.motif_min_pct_res), np.ceil(n * self.motif_max_pct_res))
        motif_n_seg = np.random.randint(self.motif_min_n_seg, min(self.motif_max_n_seg, motif_n_res) + 1)
        indices = sorted(np.random.choice(motif_n_res - 1, motif_n_seg - 1, replace=False) + 1)
        indices = [0] + indices + [motif_n_res]
        seg_lens = [indices[i + 1] - indices[i] for i in range(motif_n_seg)]
        segs = [''.join(['1'] * length) for length in seg_lens]
        segs.extend(['0'] * (n - motif_n_res))
        random.shuffle(segs)
        seq_mask = np.array([int(c) for c in ''.join(segs)]).astype(bool)
        np_features['fixed_sequence_mask'] = seq_mask
        np_features['fixed_structure_mask'] = (seq_mask[:, np.newaxis] * seq_mask[np.newaxis, :]).astype(bool)
        return np_features


class GenieDataModule:
    """data_module.py:12-300: filter by length, keep the train / validation split in <rootdir>/<name>/{train,validation}.txt."""

    def __init__(self, name, rootdir, datadir, min_n_res, max_n_res, max_n_chain, validation_split, batch_size, motif_prob,
                 motif_min_pct_res, motif_max_pct_res, motif_min_n_seg, motif_max_n_seg):
        self.name, self.rootdir, self.datadir = name, rootdir, datadir
        self.min_n_res, self.max_n_res, self.max_n_chain = min_n_res, max_n_res, max_n_chain
        self.validation_split, self.batch_size = validation_split, batch_size
        self.motif = (motif_prob, motif_min_pct_res, motif_max_pct_res, motif_min_n_seg, motif_max_n_seg)

    def setup(self, stage=None):
        train_fp = os.path.join(self.rootdir, self.name, 'train.txt')
        val_fp = os.path.join(self.rootdir, self.name, 'validation.txt')
        if os.path.exists(train_fp):
            if self.validation_split is not None:
                assert os.path.exists(val_fp)
            return
        print('INFO: creating dataset...')
        os.makedirs(os.path.dirname(train_fp), exist_ok=True)
        names = self._fetch_names(self.datadir)
        if self.validation_split is not None:
            train_names, val_names = self._split(names)
            self._save_names(train_names, train_fp)
            self._save_names(val_names, val_fp)
        else:
            self._save_names(names, train_fp)

    def _dataset(self, which):
        info = {'datadir': self.datadir, 'names': self._load_names(os.path.join(self.rootdir, self.name, which + '.txt'))}
        return GenieDataset(info, self.min_n_res, self.max_n_res, self.max_n_chain, *self.motif)

    def train_dataloader(self, sampler=None):
        ds = self._dataset('train')
        return DataLoader(ds, batch_size=self.batch_size, shuffle=sampler is None, sampler=sampler)

    def val_dataloader(self):
        return DataLoader(self._dataset('validation'), batch_size=self.batch_size, shuffle=False)

    @staticmethod
    def _load_names(filepath):
        with open(filepath) as fh:
            return [line.strip() for line in fh]

    @staticmethod
    def _save_names(names, filepath):
        with open(filepath, 'w') as fh:
            fh.write('\n'.join(names))

    def _fetch_names(self, datadir):
        """data_module.py:231-248 looks for *.pdb.gz only; plain *.pdb files are accepted too (the dataset reads both)."""
        names = []
        for fp in sorted(glob.glob(os.path.join(datadir, '*.pdb.gz')) + glob.glob(os.path.join(datadir, '*.pdb'))):
            if self._validate(fp):
                names.append(os.path.basename(fp).split('.')[0])
        return names

    def _split(self, names):
        k = int(len(names) * self.validation_split) if self.validation_split < 1 else int(self.validation_split)
        return names[:-k], names[-k:]

    def _validate(self, filepath):
        seqs, _ = F.parse_pdb(filepath)
        n = sum(len(s) for s in seqs)
        return self.min_n_res <= n <= self.max_n_res
