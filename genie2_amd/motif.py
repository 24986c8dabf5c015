"""Motif-scaffolding problem files (genie/utils/motif_utils.py): the fixed-column
`REMARK 999` specification + motif structure in PDB format (reference README.md:135-170)."""
import numpy as np


def load_motif_spec(filepath):
    """motif_utils.py:4-57.  INPUT lines: column 19 = chain (blank for a scaffold segment),
    20-23 / 24-27 = start / end residue index (motif) or min / max length (scaffold),
    29 = motif group (default 'A')."""
    spec = {'structures': []}
    with open(filepath) as fh:
        for line in fh:
            if line.startswith('REMARK 999 INPUT'):
                if line[18] == ' ':
                    spec['structures'].append({'type': 'scaffold', 'min_length': int(line[19:23]),
                                               'max_length': int(line[23:27])})
                else:
                    spec['structures'].append({
                        'type': 'motif', 'chain': line[18], 'start_index': int(line[19:23]), 'end_index': int(line[23:27]),
                        'group': line[28] if len(line) > 28 and line[28] != ' ' else 'A'})
            elif line.startswith('REMARK 999 NAME'):
                spec['name'] = line[18:]
            elif line.startswith('REMARK 999 MINIMUM TOTAL LENGTH'):
                spec['min_total_length'] = int(line[37:])
            elif line.startswith('REMARK 999 MAXIMUM TOTAL LENGTH'):
                spec['max_total_length'] = int(line[37:])
    return spec


def sample_motif_mask(spec):
    """motif_utils.py:59-129: draw scaffold segment lengths (np.random.randint, in order) until
    the total length satisfies the bounds; motif residues of one group condition each other."""
    while True:
        seq_mask, groups = [], []
        for st in spec['structures']:
            if st['type'] == 'scaffold':
                n = np.random.randint(st['min_length'], st['max_length'] + 1)
                seq_mask += [0] * n
                groups += [0] * n
            else:
                n = st['end_index'] - st['start_index'] + 1
                seq_mask += [1] * n
                groups += [ord(st['group']) - ord('A') + 1] * n
        if spec['min_total_length'] <= len(seq_mask) <= spec['max_total_length']:
            break
    groups = np.array(groups).astype(int)
    structure = np.zeros((len(groups), len(groups)))
    for g in range(1, 1 + int(np.max(groups))):
        m = groups == g
        structure += m[:, None] * m[None, :]
    return {'sequence': np.array(seq_mask).astype(bool), 'structure': structure.astype(bool), 'group': groups}


def save_motif_pdb(spec_filepath, mask, pdb_filepath):
    """motif_utils.py:131-190: the motif's ATOM records re-indexed to the generated structure
    (chain 'A', residue index = 1-based position of the motif residue, group in columns 73-76)."""
    spec = load_motif_spec(spec_filepath)
    spec_res = [(st['chain'], i, st['group']) for st in spec['structures'] if st['type'] == 'motif'
                for i in range(st['start_index'], st['end_index'] + 1)]
    pdb_res = [i + 1 for i, on in enumerate(mask) if on]
    assert len(pdb_res) == len(spec_res)
    remap = {'{}_{}'.format(c, i): (pdb_res[k], g) for k, (c, i, g) in enumerate(spec_res)}
    with open(spec_filepath) as fh:
        lines = [line for line in fh if line.startswith('ATOM')]
    out = []
    for line in lines:
        new_index, group = remap['{}_{}'.format(line[21], int(line[22:26]))]
        out.append(line[:21] + 'A' + str(new_index).rjust(4) + line[26:72] + group.ljust(4) + line[76:])
    with open(pdb_filepath, 'w') as fh:
        fh.write(''.join(out))
