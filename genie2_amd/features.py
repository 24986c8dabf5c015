"""The reference's feature dictionaries (genie/utils/feat_utils.py), own code.

12 keys; numpy on the host side exactly like the reference so that user code
written against `genie.utils.feat_utils` keeps working.
"""
import numpy as np
import torch

RESTYPES = 'ARNDCQEGHILKMFPSTWYV'          # genie/constants/residue.py:8-29
RESTYPE_3 = ['ALA', 'ARG', 'ASN', 'ASP', 'CYS', 'GLN', 'GLU', 'GLY', 'HIS', 'ILE', 'LEU', 'LYS', 'MET', 'PHE',
             'PRO', 'SER', 'THR', 'TRP', 'TYR', 'VAL']

_PER_RESIDUE = ('aatype', 'atom_positions', 'residue_mask', 'residue_index', 'chain_index', 'fixed_sequence_mask',
                'fixed_group', 'interface_mask')
KEYS = ('aatype', 'num_chains', 'num_residues', 'num_residues_per_chain', 'atom_positions', 'residue_mask',
        'residue_index', 'chain_index', 'fixed_sequence_mask', 'fixed_structure_mask', 'fixed_group', 'interface_mask')


def create_empty_np_features(lengths):
    """feat_utils.py:17-65: an unconditional structure with the given chain lengths."""
    lengths = [int(x) for x in lengths]
    n = int(sum(lengths))
    return {
        'aatype': np.zeros((n, 20), dtype=int),
        'num_chains': np.array(len(lengths)).astype(int),
        'num_residues': np.array(n).astype(int),
        'num_residues_per_chain': np.array(lengths).astype(int),
        'atom_positions': np.zeros((n, 3), dtype=float),
        'residue_mask': np.ones(n, dtype=int),
        'residue_index': np.concatenate([np.arange(x) for x in lengths]).astype(int),
        'chain_index': np.concatenate([np.full(x, i) for i, x in enumerate(lengths)]).astype(int),
        'fixed_sequence_mask': np.zeros(n, dtype=bool),
        'fixed_structure_mask': np.zeros((n, n), dtype=bool),
        'fixed_group': np.zeros(n, dtype=int),
        'interface_mask': np.zeros(n, dtype=bool),
    }


def pad_np_features(f, max_n_chain, max_n_res):
    """feat_utils.py:192-232 (returns a new dict)."""
    n, nc = int(f['num_residues']), int(f['num_chains'])
    out = dict(f)
    out['num_residues_per_chain'] = np.concatenate(
        [f['num_residues_per_chain'], np.zeros(max_n_chain - nc, dtype=f['num_residues_per_chain'].dtype)])
    out['fixed_structure_mask'] = np.pad(f['fixed_structure_mask'], [(0, max_n_res - n), (0, max_n_res - n)])
    for k in _PER_RESIDUE:
        pad = np.zeros((max_n_res - n,) + f[k].shape[1:], dtype=f[k].dtype)
        out[k] = np.concatenate([f[k], pad])
    return out


def batchify_np_features(list_np_features):
    """feat_utils.py:234-272."""
    max_c = max(int(f['num_chains']) for f in list_np_features)
    max_r = max(int(f['num_residues']) for f in list_np_features)
    padded = [pad_np_features(f, max_c, max_r) for f in list_np_features]
    return {k: np.stack([p[k] for p in padded]) for k in list_np_features[0].keys()}


def debatchify_np_features(np_features):
    """feat_utils.py:274-302: strip the padding again."""
    out = []
    for i in range(np_features['aatype'].shape[0]):
        nc, n = int(np_features['num_chains'][i]), int(np_features['num_residues'][i])
        d = {'num_chains': np_features['num_chains'][i], 'num_residues': np_features['num_residues'][i],
             'num_residues_per_chain': np_features['num_residues_per_chain'][i, :nc],
             'fixed_structure_mask': np_features['fixed_structure_mask'][i, :n, :n]}
        for k in _PER_RESIDUE:
            d[k] = np_features[k][i, :n]
        out.append(d)
    return out


_T_DTYPES = {'num_chains': torch.int32, 'num_residues': torch.int32, 'num_residues_per_chain': torch.int32,
             'aatype': torch.int32, 'atom_positions': torch.float32, 'residue_mask': torch.int32,
             'residue_index': torch.int32, 'chain_index': torch.int32, 'fixed_sequence_mask': torch.bool,
             'fixed_structure_mask': torch.bool, 'fixed_group': torch.int32, 'interface_mask': torch.bool}
_NP_DTYPES = {k: (bool if v == torch.bool else float if v == torch.float32 else int) for k, v in _T_DTYPES.items()}


def convert_np_features_to_tensor(features, device):
    """feat_utils.py:304-321."""
    return {k: torch.as_tensor(np.asarray(features[k])).to(dt).to(device) for k, dt in _T_DTYPES.items()}


def convert_tensor_features_to_numpy(features):
    """feat_utils.py:323-340."""
    return {k: features[k].detach().cpu().numpy().astype(_NP_DTYPES[k]) for k in _T_DTYPES}


def prepare_tensor_features(features):
    """feat_utils.py:342-359: cast a batched tensor feature dict (e.g. from a DataLoader) to the types the model reads."""
    ints = ('num_chains', 'num_residues', 'num_residues_per_chain', 'aatype', 'residue_mask', 'residue_index', 'chain_index', 'fixed_group')
    out = {k: features[k].int() for k in ints}
    out['atom_positions'] = features['atom_positions'].float()
    for k in ('fixed_sequence_mask', 'fixed_structure_mask', 'interface_mask'):
        out[k] = features[k].bool()
    return out


def save_np_features_to_pdb(np_features, filepath):
    """feat_utils.py:136-186: C-alpha-only PDB, coordinates centred and rounded
    to 3 decimals, written with str(x).rjust(8) (so 14.61, not 14.610)."""
    coords = np_features['atom_positions']
    coords = np.around(coords - np.mean(coords, axis=0, keepdims=True), decimals=3)
    lines = []
    for i in range(coords.shape[0]):
        cols = [' '] * 80

        def put(at, text):
            cols[at:at + len(text)] = list(text)

        grp = int(np_features['fixed_group'][i])
        put(0, 'ATOM')
        put(6, str(i + 1).rjust(5))
        put(13, 'CA')
        put(17, RESTYPE_3[int(np.argmax(np_features['aatype'][i]))])
        put(21, chr(ord('A') + int(np_features['chain_index'][i])))
        put(22, str(int(np_features['residue_index'][i]) + 1).rjust(4))
        put(30, str(coords[i][0]).rjust(8))
        put(38, str(coords[i][1]).rjust(8))
        put(46, str(coords[i][2]).rjust(8))
        put(72, (' ' if grp == 0 else chr(grp - 1 + ord('A'))).ljust(4))
        put(77, 'C')
        lines.append(''.join(cols[:80]) + '\n')
    with open(filepath, 'w') as fh:
        fh.writelines(lines)


RESTYPE_ORDER = {c: i for i, c in enumerate(RESTYPES)}
_RESTYPE_3_TO_1 = {t: RESTYPES[i] for i, t in enumerate(RESTYPE_3)}


def parse_pdb(filepath):
    """feat_utils.py:377-416: per-chain residue-type indices and C-alpha coordinates."""
    import gzip
    seqs, coords, current = [], [], None
    opener = (lambda p: gzip.open(p, 'rt')) if filepath.endswith('.gz') else (lambda p: open(p, 'r'))
    with opener(filepath) as fh:
        for line in fh:
            if line.startswith('ATOM') and line[13:15].strip() == 'CA':
                chain = line[21]
                if current is None or chain != current:
                    seqs.append([])
                    coords.append([])
                    current = chain
                seqs[-1].append(RESTYPE_ORDER[_RESTYPE_3_TO_1[line[17:20]]])
                coords[-1].append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return seqs, coords


def create_np_features_from_pdb(filepath):
    """feat_utils.py:66-93: unconditional features of a structure file (coordinates centred)."""
    seqs, coords = parse_pdb(filepath)
    f = create_empty_np_features([len(s) for s in seqs])
    xyz = np.concatenate(coords)
    f['aatype'] = np.eye(20)[np.concatenate(seqs)].astype(int)
    f['atom_positions'] = (xyz - np.mean(xyz, axis=0, keepdims=True)).astype(float)
    return f


def create_np_features_from_motif_pdb(filepath):
    """feat_utils.py:95-130: sample a scaffold/motif arrangement satisfying the problem file and
    place the motif's residue types and C-alpha coordinates at the motif positions."""
    from .motif import load_motif_spec, sample_motif_mask
    spec = load_motif_spec(filepath)
    seqs, coords = parse_pdb(filepath)
    mask = sample_motif_mask(spec)
    f = create_empty_np_features([len(mask['sequence'])])
    f['aatype'][mask['sequence']] = np.eye(20)[np.concatenate(seqs)]
    f['atom_positions'][mask['sequence']] = np.concatenate(coords)
    f['fixed_sequence_mask'] = mask['sequence']
    f['fixed_structure_mask'] = mask['structure']
    f['fixed_group'] = mask['group']
    return f
