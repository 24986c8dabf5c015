"""Scaffold path (SURVEY §8f item 1 / BASELINE config 4): motif-problem parser, mask sampler,
motif feature builder, motif-PDB writer and ScaffoldSampler.  Expected values were produced by
the reference's own functions (oracle/make_goldens.py: gen_motif) on
tests/golden/motif_problem_6E6R.pdb."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, golden_features, load_golden
from oracle import genie_oracle as O

PROBLEM = os.path.join(GOLDEN, 'motif_problem_6E6R.pdb')
FEATURE_KEYS = ('aatype', 'num_chains', 'num_residues', 'num_residues_per_chain', 'atom_positions', 'residue_mask',
                'residue_index', 'chain_index', 'fixed_sequence_mask', 'fixed_structure_mask', 'fixed_group',
                'interface_mask')


def test_motif_spec_parser_matches_reference():
    from genie.utils.motif_utils import load_motif_spec
    g = load_golden('motif_features')
    spec = load_motif_spec(PROBLEM)
    assert spec['name'] == str(g['spec_name'])
    assert (spec['min_total_length'], spec['max_total_length']) == (int(g['spec_min']), int(g['spec_max']))
    rows = [[s['type'] == 'motif', s.get('min_length', s.get('start_index')), s.get('max_length', s.get('end_index')),
             ord(s.get('group', ' ')), ord(s.get('chain', ' '))] for s in spec['structures']]
    assert np.array_equal(np.array(rows), g['spec_structures'])


def test_parse_pdb_and_structure_features_match_reference():
    from genie.utils import feat_utils
    g = load_golden('motif_features')
    seqs, coords = feat_utils.parse_pdb(PROBLEM)
    assert len(seqs) == 1 and np.array_equal(np.array(seqs[0]), g['parse_seq'])
    assert np.array_equal(np.array(coords[0]), g['parse_coords'])
    f = feat_utils.create_np_features_from_pdb(PROBLEM)
    for k in FEATURE_KEYS:
        assert f[k].dtype == g['frompdb_' + k].dtype and np.array_equal(f[k], g['frompdb_' + k]), k


@pytest.mark.parametrize('seed', range(4))
def test_motif_features_and_motif_pdb_match_reference(seed, tmp_path):
    """Same np.random stream => same scaffold lengths, masks, groups and motif PDB bytes."""
    from genie.utils import feat_utils, motif_utils
    g = load_golden('motif_features')
    np.random.seed(seed)
    f = feat_utils.create_np_features_from_motif_pdb(PROBLEM)
    assert set(f) == set(FEATURE_KEYS)
    for k in FEATURE_KEYS:
        want = g[f'seed{seed}_{k}']
        assert f[k].dtype == want.dtype and f[k].shape == want.shape and np.array_equal(f[k], want), k
    n = len(f['residue_mask'])
    assert 60 <= n <= 80 and f['fixed_sequence_mask'].sum() == 13
    assert set(np.unique(f['fixed_group'])) == {0, 1, 2}
    assert not f['fixed_structure_mask'][f['fixed_group'] == 1][:, f['fixed_group'] == 2].any()
    out = tmp_path / 'm.pdb'
    motif_utils.save_motif_pdb(PROBLEM, f['fixed_sequence_mask'], str(out))
    assert out.read_bytes() == g[f'seed{seed}_motif_pdb'].tobytes()


def test_oracle_scaffold_trajectory_matches_reference_golden(base_weights):
    """ragged B=2 (68/74 residues), two motif groups, T=20 through the reference's ScaffoldSampler."""
    g = load_golden('trajectory_scaffold_t20')
    feats = golden_features(g)
    dims = dict(O.BASE_DIMS, n_timestep=20)
    noise, codes, final = (torch.from_numpy(g[k]) for k in ('noise', 'quat_codes', 'final'))
    mine, _, _ = O.sample_loop(base_weights, dims, feats, noise, float(g['scale']), 'closed', codes)
    assert float((mine - final).abs().max()) <= 1e-4 * max(1.0, float(final.pow(2).mean().sqrt()))


def test_scaffold_cli_flags_and_tasks(tmp_path):
    from genie2_amd.sample_scaffold import ScaffoldRunner, build_parser
    a = build_parser().parse_args(['--name', 'base', '--epoch', '40', '--scale', '0.4', '--outdir', 'o'])
    assert (a.rootdir, a.strength, a.num_samples, a.batch_size, a.motif_name, a.datadir, a.num_devices) == \
        ('results', 0, 100, 4, None, 'data/design25', 1)
    r = ScaffoldRunner()
    for n in ('1bcf', '6e6r_long'):
        (tmp_path / (n + '.pdb')).write_text('')
    tasks = r.create_tasks(dict(vars(a), datadir=str(tmp_path)))
    assert sorted(t['motif_name'] for t in tasks) == ['1bcf', '6e6r_long']
    assert r.create_tasks(dict(vars(a), motif_name='x')) == [{'motif_name': 'x'}]
    assert set(r.create_constants(vars(a))) == {'rootdir', 'name', 'epoch', 'scale', 'strength', 'outdir',
                                                'num_samples', 'batch_size', 'datadir'}


@pytest.mark.gpu
@pytest.mark.parametrize('math', ['hx', 'f32'])
def test_scaffold_sampler_matches_reference_golden(tmp_path, base_weights, math, monkeypatch):
    """genie.sampler.scaffold.ScaffoldSampler over the HIP engine against the reference's own
    ScaffoldSampler._sample run (same np/torch streams via explicit noise + recorded quaternion signs)."""
    from genie.config import Config
    from genie.sampler.scaffold import ScaffoldSampler
    from genie2_amd.diffusion import Genie
    monkeypatch.setenv('GENIE_MATH', math)          # initial arithmetic of the engine the façade creates
    g = load_golden('trajectory_scaffold_t20')
    cfg = Config()
    cfg.diffusion['n_timestep'] = 20
    model = Genie(cfg)
    model.model.load_state_dict(base_weights)
    sampler = ScaffoldSampler(model.eval().to('cuda:0'))
    params = {'filepath': PROBLEM, 'scale': float(g['scale']), 'strength': 0, 'num_samples': 2, 'outdir': str(tmp_path),
              'prefix': '6E6R_long', 'offset': 0, 'noise': torch.from_numpy(g['noise']),
              'quat_codes': torch.from_numpy(g['quat_codes'])}
    np.random.seed(int(g['seed']))
    got = sampler._sample(params)
    final = torch.from_numpy(g['final'])
    rms = float(final.pow(2).mean().sqrt())
    assert [len(x['residue_mask']) for x in got] == list(g['lengths'])
    for b, x in enumerate(got):
        n = len(x['residue_mask'])
        d = float((torch.tensor(x['atom_positions'], dtype=torch.float32) - final[b, :n]).abs().max())
        assert d <= 1e-4 * max(1.0, rms), (b, d)
    # without recorded signs the closed-form quaternion differs only in sign convention of the eigh solver
    np.random.seed(int(g['seed']))
    sampler.sample(dict(params))
    assert sorted(p.name for p in (tmp_path / 'pdbs').iterdir()) == ['6E6R_long_0.pdb', '6E6R_long_1.pdb']
    motif = (tmp_path / 'motif_pdbs' / '6E6R_long_1.pdb').read_text().splitlines()
    assert len(motif) == 65 and all(line[21] == 'A' for line in motif)
    out = (tmp_path / 'pdbs' / '6E6R_long_1.pdb').read_text().splitlines()
    assert len(out) == int(g['lengths'][1])
    groups = {line[72] for line in out}
    assert groups == {' ', 'A', 'B'}
