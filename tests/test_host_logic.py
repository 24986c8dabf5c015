"""Host logic and the C-ABI surface.  CPU only: no compute call reaches the GPU."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def test_ctypes_mirrors_have_the_headers_struct_layout(tmp_path):
    """The structs cross the C ABI by pointer: size and every field offset of the ctypes mirrors (genie2_amd/capi.py) against what a C
    compiler makes of include/genie_hip.h."""
    import ctypes as C
    import subprocess
    from genie2_amd import capi
    pairs = {'genie_dims_t': capi.GenieDims, 'genie_features_t': capi.GenieFeatures, 'genie_taps_t': capi.GenieTaps,
             'genie_train_opts_t': capi.GenieTrainOpts, 'genie_gemm_desc_t': capi.GenieGemmDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "genie_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for ln in out.splitlines():
        cname, what, val = ln.split()
        cls = pairs[cname]
        if what == 'size':
            assert C.sizeof(cls) == int(val), (cname, C.sizeof(cls), val)
        else:
            assert getattr(cls, what).offset == int(val), (cname, what, getattr(cls, what).offset, val)
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in pairs.values())


def test_library_exports_every_declared_symbol():
    from genie2_amd import build, capi
    build.build()
    lib = capi.load_library()
    header = open(os.path.join(ROOT, 'include', 'genie_hip.h')).read()
    declared = set(re.findall(r'\b(genie_[a-z_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))
    for name in declared:
        assert getattr(lib, name) is not None


def test_create_fails_loudly_without_gpu_or_bad_dims():
    import ctypes as C
    from genie2_amd import capi, pack
    lib = capi.load_library()
    bad = dict(pack.BASE_DIMS, c_p=64)
    h = C.c_void_p()
    rc = lib.genie_create(C.byref(capi.GenieDims(**bad)), 0, C.byref(h))
    assert rc == -1 and b'c_p' in lib.genie_last_error(None)
    if not torch.cuda.is_available():
        from genie2_amd.engine import GenieEngine
        with pytest.raises(capi.GenieError):
            GenieEngine(pack.BASE_DIMS, pack.random_state_dict(pack.BASE_DIMS), 'cuda:0')
        with pytest.raises(capi.GenieError):
            GenieEngine(pack.BASE_DIMS, {}, 'cpu')


def test_weight_layout_matches_reference_state_dict():
    from genie2_amd import pack
    g = load_golden('weights_recipe')
    lay = pack.weight_layout(pack.BASE_DIMS)
    assert [k for k, _ in lay] == list(g['keys'])
    assert [str(tuple(s)) for _, s in lay] == list(g['shapes'])
    assert sum(int(np.prod(s)) for _, s in lay) == int(g['n_params']) == 15732080


def test_weight_count_matches_c_library():
    import ctypes as C
    from genie2_amd import capi, pack
    from oracle import genie_oracle as O
    lib = capi.load_library()
    for dims in (pack.BASE_DIMS, O.small_dims(), O.small_dims(c_s=256, n_head_ipa=8, n_v_point=4, pair_transition_n=2)):
        n = sum(int(np.prod(s)) for _, s in pack.weight_layout(dims))
        assert n == lib.genie_weight_count(C.byref(capi.GenieDims(**{k: dims[k] for k in pack.DIM_KEYS})))


def test_flatten_state_dict_checks_keys_and_shapes():
    from genie2_amd import pack
    from oracle import genie_oracle as O
    d = O.small_dims()
    sd = pack.random_state_dict(d)
    blob = pack.flatten_state_dict({'model.' + k: v for k, v in sd.items()}, d)      # checkpoint prefix accepted
    assert blob.numel() == sum(v.numel() for v in sd.values())
    bad = dict(sd)
    bad.pop('single_feature_net.linear.weight')
    with pytest.raises(KeyError):
        pack.flatten_state_dict(bad, d)
    bad = dict(sd)
    bad['single_feature_net.linear.weight'] = torch.zeros(3, 3)
    with pytest.raises(ValueError):
        pack.flatten_state_dict(bad, d)


def test_tables_and_schedule_match_reference():
    from genie2_amd import pack
    g = load_golden('encoding')
    for nm, (vmax, N, D) in dict(pos=(256, 256, 256), chain=(4, 1, 64), t1000=(1001, 1000, 512)).items():
        tab = pack.sinusoidal_table(vmax, N, D)
        assert torch.equal(tab[torch.from_numpy(g[nm + '_rows'])], torch.from_numpy(g[nm + '_vals']))
    s = load_golden('schedule')
    assert torch.equal(pack.cosine_betas(1000), torch.from_numpy(s['betas_1000']))
    blk = pack.schedule_block(pack.schedule_tensors(100))
    assert blk.shape == (4, 101) and blk[0, 0] == 1 and blk[3, 0] == 0


def test_feature_dicts_and_pdb_writer():
    from genie2_amd import features as F
    g = load_golden('pdb_writer')
    f = F.create_empty_np_features([7])
    f['atom_positions'] = g['atom_positions']
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'x.pdb')
        F.save_np_features_to_pdb(f, p)
        assert open(p, 'rb').read() == g['pdb_bytes'].tobytes()      # byte-exact with the reference's writer
    b = F.batchify_np_features([F.create_empty_np_features([5]), F.create_empty_np_features([3, 4])])
    assert b['aatype'].shape == (2, 7, 20) and b['fixed_structure_mask'].shape == (2, 7, 7)
    assert b['residue_mask'].tolist() == [[1, 1, 1, 1, 1, 0, 0], [1] * 7]
    assert b['chain_index'][1].tolist() == [0, 0, 0, 1, 1, 1, 1] and b['residue_index'][1].tolist() == [0, 1, 2, 0, 1, 2, 3]
    tt = F.convert_np_features_to_tensor(b, 'cpu')
    assert tt['residue_mask'].dtype == torch.int32 and tt['fixed_structure_mask'].dtype == torch.bool
    back = F.debatchify_np_features(F.convert_tensor_features_to_numpy(tt))
    assert back[0]['atom_positions'].shape == (5, 3) and back[1]['num_residues_per_chain'].tolist() == [3, 4]


def test_config_parser_defaults_and_file():
    from genie2_amd.config import Config
    c = Config()
    assert c.model['c_s'] == 384 and c.model['n_pair_transform_layer'] == 5 and c.diffusion['n_timestep'] == 1000
    assert c.io['max_n_res'] == 256 and c.model['include_tri_att'] is False
    with tempfile.NamedTemporaryFile('w', suffix='.cfg', delete=False) as fh:
        fh.write('name base\n\nnumPairTransformLayers 3\nincludeTriangularAttention False\nrescale 2.5\nbad line here\n')
    c = Config(fh.name)
    os.unlink(fh.name)
    assert c.io['name'] == 'base' and c.model['n_pair_transform_layer'] == 3 and c.model['rescale'] == 2.5


def test_api_surface_mirrors_reference_paths():
    from genie.sampler.base import BaseSampler
    from genie.sampler.unconditional import UnconditionalSampler
    from genie.model.model import Denoiser
    from genie.config import Config
    from genie.utils.affine_utils import T
    from genie.diffusion.schedule import get_betas
    from genie.utils.multiprocessor import MultiProcessor  # noqa: F401
    assert issubclass(UnconditionalSampler, BaseSampler)
    assert get_betas(10, 'cosine').shape == (11,)
    c = Config()
    m = Denoiser(**c.model, n_timestep=1000, max_n_res=256, max_n_chain=1)
    g = load_golden('weights_recipe')
    assert list(m.state_dict().keys()) == list(g['keys'])
    t = T(None, torch.zeros(2, 5, 3))
    assert t.rots.shape == (2, 5, 3, 3) and t[0].trans.shape == (5, 3)
    from genie2_amd.capi import GenieError
    with pytest.raises(GenieError):            # the product path has no CPU fallback
        m(t, torch.ones(2, dtype=torch.int32), {})


def test_cli_flags_and_task_split():
    from genie2_amd.sample_unconditional import UnconditionalRunner, build_parser
    from genie2_amd.multiprocessor import split_tasks
    a = build_parser().parse_args(['--name', 'base', '--epoch', '40', '--scale', '0.6', '--outdir', 'o'])
    assert (a.num_samples, a.batch_size, a.min_length, a.max_length, a.length_step, a.num_devices) == (5, 4, 50, 256, 1, 1) and a.resume is False
    tasks = UnconditionalRunner().create_tasks(dict(min_length=50, max_length=256, length_step=16))
    assert [t['length'] for t in tasks] == list(range(256, 49, -16)) and len(tasks) == 13      # 50 itself is never visited
    bins = split_tasks(tasks, 8)
    assert [len(b) for b in bins] == [2, 2, 2, 2, 2, 2, 1, 0]                                    # reference's imbalance quirk
    assert sum(bins, []) == tasks


_DIST_WORKER = r'''
import os, sys, torch, torch.distributed as td
sys.path.insert(0, sys.argv[1])
from genie2_amd import distributed as D
td.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=2)
r = td.get_rank()
tasks = list(range(5))
mine = D.rank_tasks(tasks)
assert mine == ([0, 1, 2] if r == 0 else [3, 4]), mine
t = D.max_over_ranks(1.0 + r)
assert t == 2.0
x = torch.full((2, 4, 3), float(r))
g = D.gather_coordinates(x)
assert g.shape == (4, 4, 3) and g[:2].eq(0).all() and g[2:].eq(1).all()
td.destroy_process_group()
print('ok', r)
'''


def test_two_rank_gloo_path():
    """world_size 2 on CPU: task sharding, max-over-ranks timing, result gather."""
    with tempfile.NamedTemporaryFile('w', suffix='.py', delete=False) as fh:
        fh.write(_DIST_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
        procs.append(subprocess.Popen([sys.executable, fh.name, ROOT], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=120) for p in procs]
    os.unlink(fh.name)
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]


def test_checkpoint_directory_layouts_round_trip(tmp_path):
    """genie/utils/model_io.py: pretrained layout (checkpoints/epoch.<E>.ckpt), training layout
    (version_<v>/checkpoints/epoch=<e>.ckpt, latest by default), Lightning-style state_dict keys."""
    import shutil
    import torch
    from genie.utils import model_io
    from genie2_amd.diffusion import Genie
    from genie2_amd.config import Config
    root = tmp_path / 'runs'
    (root / 'tiny').mkdir(parents=True)
    cfg_text = 'name tiny\nnumberOfPairTransformLayers 1\nnumberOfStructureLayers 1\n'
    (root / 'tiny' / 'configuration').write_text(cfg_text)
    cfg = Config(str(root / 'tiny' / 'configuration'))
    g = Genie(cfg)
    with torch.no_grad():
        for p in g.model.parameters():
            p.normal_(0, 0.02)
    assert model_io.get_versions(str(root), 'tiny') == []
    fresh = model_io.load_model(str(root), 'tiny')                   # no checkpoint: untrained model
    assert isinstance(fresh, Genie)
    model_io.save_checkpoint(g, str(root / 'tiny' / 'version_0' / 'checkpoints' / 'epoch=3.ckpt'), epoch=3)
    model_io.save_checkpoint(fresh, str(root / 'tiny' / 'version_0' / 'checkpoints' / 'epoch=1.ckpt'), epoch=1)
    model_io.save_checkpoint(fresh, str(root / 'tiny' / 'version_2' / 'checkpoints' / 'epoch=0.ckpt'))
    assert model_io.get_versions(str(root), 'tiny') == [0, 2] and model_io.get_epochs(str(root), 'tiny', 0) == [1, 3]
    ck = torch.load(root / 'tiny' / 'version_0' / 'checkpoints' / 'epoch=3.ckpt', weights_only=True)
    assert all(k.startswith('model.') for k in ck['state_dict']) and len(ck['state_dict']) == len(g.model.state_dict())
    got = model_io.load_model(str(root), 'tiny', version=0)          # latest epoch of version 0
    for k, v in g.model.state_dict().items():
        assert torch.equal(got.model.state_dict()[k], v), k
    (root / 'tiny' / 'checkpoints').mkdir()
    shutil.copy(root / 'tiny' / 'version_0' / 'checkpoints' / 'epoch=3.ckpt', root / 'tiny' / 'checkpoints' / 'epoch.3.ckpt')
    pre = model_io.load_pretrained_model(str(root), 'tiny', 3)
    assert torch.equal(pre.model.state_dict()['single_feature_net.linear.weight'], g.model.state_dict()['single_feature_net.linear.weight'])


def test_small_reference_utilities_under_their_import_paths():
    """genie.utils.encoding / genie.constants.residue / genie.utils.loss: the host-side helpers user code imports."""
    import torch
    from conftest import load_golden
    from genie.utils.encoding import sinusoidal_encoding
    from genie.constants.residue import RESTYPES, RESTYPE_ORDER, RESTYPE_1_TO_3, RESTYPE_3_TO_1
    from genie.utils.loss import mse
    g = load_golden('encoding')
    for nm, (vmax, N, D) in dict(pos=(256, 256, 256), chain=(4, 1, 64), t1000=(1001, 1000, 512)).items():
        e = sinusoidal_encoding(torch.arange(vmax, dtype=torch.int32), N, D)
        assert torch.equal(e[torch.from_numpy(g[nm + '_rows'])], torch.from_numpy(g[nm + '_vals']))      # the reference's own values
    assert sinusoidal_encoding(torch.zeros(2, 3, dtype=torch.int32), 10, 8).shape == (2, 3, 8)
    assert ''.join(RESTYPES) == 'ARNDCQEGHILKMFPSTWYV' and RESTYPE_ORDER['V'] == 19
    assert RESTYPE_1_TO_3['W'] == 'TRP' and RESTYPE_3_TO_1['GLY'] == 'G'
    x = torch.zeros(2, 3, 3)
    y = torch.ones(2, 3, 3)
    m = torch.tensor([[1., 1., 0.], [1., 0., 0.]])
    e = mse(y, x, m)
    assert e.shape == (2, 3) and torch.allclose(e[0, :2], torch.full((2,), 3 ** 0.5)) and float(e[0, 2]) == 0.0
    assert torch.allclose(mse(y, x, m, 'mean'), torch.full((2,), 3 ** 0.5)) and torch.allclose(mse(y, x, m, 'sum'), torch.tensor([2 * 3 ** 0.5, 3 ** 0.5]))


def test_asm_register_rings_are_not_touched_by_the_compiler(tmp_path):
    """k_gemm_rows / k_gemm_rows_hx stream weights through registers with asm-issued loads and counted vmcnt waits
    (csrc/common.h).  tools/check_asm_ring.py audits the compiled ISA: between such a load and the wait that retires
    it no compiler-scheduled instruction may read or write its destination registers."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    src = os.path.join(ROOT, 'genie2_amd', 'csrc', 'single_kernels.hip')
    out = tmp_path / 'single_kernels.s'
    r = subprocess.run([hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-DGENIE_BUILD', '-S', '--cuda-device-only', '-o', str(out), src],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    chk = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_asm_ring.py'), str(out)], capture_output=True, text=True)
    assert chk.returncode == 0 and ' 0 violations' in chk.stdout, chk.stdout[-2000:]
    n_loads = int(chk.stdout.split(':')[-1].split('asm loads')[0])
    assert n_loads >= 40          # both kernels' rings were actually found


def test_multiprocessor_spawn_path_and_exit_codes(tmp_path):
    """MultiProcessor.run with num_devices=2 (genie/utils/multiprocessor.py:73-100): one spawned child per device, contiguous
    bins of ceil(n/2) tasks, every task executed exactly once; a dying worker makes run() raise (the reference drops exit codes)."""
    import json
    from _mp_stub import StubRunner
    params = dict(min_length=50, max_length=130, length_step=16, outdir=str(tmp_path))
    StubRunner().run(params, 2, sequential_order=True)
    got = {n: json.load(open(tmp_path / n)) for n in sorted(os.listdir(tmp_path))}
    assert sorted(got) == ['cuda_0.json', 'cuda_1.json']
    assert [t['length'] for t in got['cuda_0.json']['tasks']] == [130, 114, 98]
    assert [t['length'] for t in got['cuda_1.json']['tasks']] == [82, 66, 50]
    assert got['cuda_0.json']['pid'] != got['cuda_1.json']['pid'] != os.getpid()
    # shuffled order (the default): same multiset of tasks
    for n in got:
        os.unlink(tmp_path / n)
    StubRunner().run(params, 2)
    lens = sorted(t['length'] for n in os.listdir(tmp_path) for t in json.load(open(tmp_path / n))['tasks'])
    assert lens == [50, 66, 82, 98, 114, 130]
    with pytest.raises(RuntimeError, match='cuda:1 exit code 3'):
        StubRunner().run(dict(params, fail_on='cuda:1'), 2)


def test_missing_checkpoint_exits_non_zero(tmp_path):
    from genie2_amd.diffusion import load_pretrained_model
    with pytest.raises(SystemExit) as e:
        load_pretrained_model(str(tmp_path), 'nope', 1)
    assert e.value.code == 1
    (tmp_path / 'm').mkdir()
    (tmp_path / 'm' / 'configuration').write_text('name m\n')
    with pytest.raises(SystemExit) as e:
        load_pretrained_model(str(tmp_path), 'm', 1)
    assert e.value.code == 1


def test_dataset_items_match_reference_golden(tmp_path):
    """genie.data.dataset.GenieDataset (dataset.py:80-249) on a structure file the reference ships: the motif masks of Algorithm 1
    under three seeds and the unconditional item, against what the reference's own class returned; and the data module's
    filter / split files (data_module.py:98-300)."""
    import random
    import shutil
    from conftest import GOLDEN
    from genie.data.dataset import GenieDataset
    from genie.data.data_module import GenieDataModule
    g = load_golden('dataset_items')
    d = tmp_path / 'pdbs'
    d.mkdir()
    shutil.copy(os.path.join(GOLDEN, 'dataset_100_0.pdb'), d / 'x100.pdb')
    random.seed(0)
    ds = GenieDataset({'datadir': str(d), 'names': ['x100']}, 20, 128, 1, 1.0, 0.05, 0.5, 1, 4)
    assert len(ds) == 1
    for seed in (1, 2, 3):
        np.random.seed(seed)
        random.seed(seed)
        it = ds[0]
        for k, v in it.items():
            assert np.array_equal(np.asarray(v), g[f's{seed}_{k}']), (seed, k)
    ds.motif_prob = 0.0
    np.random.seed(9)
    it = ds[0]
    for k, v in it.items():
        assert np.array_equal(np.asarray(v), g[f'u_{k}']), k
    # data module: length filter, split files, loader batches
    shutil.copy(os.path.join(GOLDEN, 'dataset_100_0.pdb'), d / 'y100.pdb')
    shutil.copy(os.path.join(GOLDEN, 'motif_problem_6E6R.pdb'), d / 'short.pdb')      # 13 motif residues: below min_n_res
    dm = GenieDataModule('run', str(tmp_path / 'runs'), str(d), 20, 128, 1, 1, 2, 0.8, 0.05, 0.5, 1, 4)
    dm.setup()
    assert open(tmp_path / 'runs' / 'run' / 'train.txt').read().split() == ['x100'] and open(tmp_path / 'runs' / 'run' / 'validation.txt').read().split() == ['y100']
    batch = next(iter(dm.train_dataloader()))
    assert batch['atom_positions'].shape == (1, 128, 3) and batch['fixed_structure_mask'].shape == (1, 128, 128)
    from genie.utils.feat_utils import prepare_tensor_features
    f = prepare_tensor_features(batch)
    assert f['residue_mask'].dtype == torch.int32 and f['fixed_sequence_mask'].dtype == torch.bool and int(f['num_residues'][0]) == 100


_DDP_WORKER = r'''
import os, sys, torch, torch.distributed as td
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
from _oracle_backend import OracleBackend, small_config
from genie2_amd.diffusion import Genie
from genie2_amd.training import GenieTrainer
from genie2_amd import features as F
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if world > 1:
    td.init_process_group('gloo', rank=rank, world_size=world)
torch.manual_seed(0)
cfg = small_config()
genie = Genie(cfg)                                   # same random_state_dict(seed 0) on every rank
tr = GenieTrainer(genie, backend=OracleBackend(genie.model.dims), train_mode=False)
tr.lr = 1e-3
# four structures; rank r of 2 takes structures 2r, 2r+1; the single process takes all four
g = torch.Generator().manual_seed(5)
feats = []
for n in (20, 17, 23, 12):
    f = F.create_empty_np_features([n])
    f['atom_positions'] = (torch.randn(n, 3, generator=g) * 4).numpy()
    feats.append(F.pad_np_features(f, 1, 24))
mine = feats if world == 1 else feats[2 * rank: 2 * rank + 2]
import numpy as np
batch = {k: torch.as_tensor(np.stack([f[k] for f in mine])) for k in mine[0]}          # (what the DataLoader's default collate does)
draws_s = torch.tensor([7, 31, 18, 44]); draws_z = torch.randn(4, 24, 3, generator=g)
import genie2_amd.training as T
sl = slice(0, 4) if world == 1 else slice(2 * rank, 2 * rank + 2)
orig_randint, orig_randn_like = torch.randint, torch.randn_like
torch.randint = lambda *a, **k: draws_s[sl] - 1               # the step's draws, so that both runs see the same noise
torch.randn_like = lambda x: draws_z[sl].to(x.dtype)
for _ in range(2):
    loss = tr.training_step(batch)
    tr.optimizer_step()
torch.randint, torch.randn_like = orig_randint, orig_randn_like
torch.save({'w': tr.w, 'loss': float(loss)}, sys.argv[2] + f'.{world}.{rank}')
if world > 1:
    td.destroy_process_group()
'''


def test_ddp_training_step_two_ranks_equal_single_process(tmp_path):
    """GenieTrainer under torch.distributed (gloo, world_size 2; gradients from the oracle's autograd through the test backend):
    two ranks with different halves of a batch end two optimizer steps with identical weights, equal to the single process that
    saw the concatenated batch -- DistributedDataParallel's mean all-reduce (train.py:57-59), here over the flat gradient blob."""
    with tempfile.NamedTemporaryFile('w', suffix='.py', delete=False) as fh:
        fh.write(_DDP_WORKER)
    out = str(tmp_path / 'w')
    env1 = dict(os.environ, RANK='0', WORLD_SIZE='1', OMP_NUM_THREADS='2')
    single = subprocess.Popen([sys.executable, fh.name, ROOT, out], env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', OMP_NUM_THREADS='2')
        procs.append(subprocess.Popen([sys.executable, fh.name, ROOT, out], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in [single] + procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e.decode()[-3000:]
    os.unlink(fh.name)
    w1 = torch.load(out + '.1.0')
    a, b = torch.load(out + '.2.0'), torch.load(out + '.2.1')
    assert torch.equal(a['w'], b['w'])                                   # ranks stay in lockstep
    # the mean over ranks of per-rank mean losses = the mean over the concatenated batch (equal batch sizes)
    # (Adam divides by sqrt(v): where a gradient is rounding noise -- linear_b.bias under the softmax's shift invariance -- the two
    #  summation orders may move a weight by a fraction of lr = 1e-3 per step; everywhere else the runs agree to f32 rounding)
    d = (a['w'] - w1['w']).abs()
    assert float(d.max()) <= 2e-4 and float(d.mean()) <= 1e-7
    assert abs(a['loss'] - b['loss']) >= 0.0 and float((a['w'] - torch.load(out + '.1.0')['w']).abs().max()) < 1e-3


def test_smc_helpers_follow_the_quoted_reference_lines():
    """genie2_amd.smc: systematic resampling against a literal restatement of unconditional_smc.py:258-283 (the while loops), the
    weight normalisation / ESS helpers (:25-43) on known values."""
    from genie.sampler.unconditional_smc import systematic_resampling, compute_ess_from_log_w, normalize_log_weights
    g = torch.Generator().manual_seed(0)
    for n in (4, 7, 16):
        w = torch.rand(n, generator=g) ** 3 + 1e-3
        particles = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
        for u in (0.0, 0.31 / n, 0.999 / n):
            wn = w / w.sum()
            cs = torch.cat([torch.tensor([0.0]), torch.cumsum(wn, 0)])
            pts = [u + i / n for i in range(n)]
            idx, j = [], 0
            for i in range(n):
                while pts[i] > float(cs[j + 1]) and j < n - 1:
                    j += 1
                idx.append(j)
            got, neww, gi = systematic_resampling(particles, w, u)
            assert gi.tolist() == idx and torch.equal(got, particles[torch.tensor(idx)]) and float(neww.abs().max()) == 0.0
    lw = torch.log(torch.tensor([0.5, 0.25, 0.25]))
    assert abs(float(compute_ess_from_log_w(lw)) - 1.0 / (0.25 + 0.0625 * 2)) < 1e-5
    assert torch.allclose(torch.exp(normalize_log_weights(lw + 100.0, 0)), torch.tensor([0.5, 0.25, 0.25]), atol=1e-6)
    assert abs(float(compute_ess_from_log_w(torch.zeros(8))) - 8.0) < 1e-5


def _tiny_batch(lengths=(20, 17), pad=24, seed=5, motif=False):
    from genie2_amd import features as F
    g = torch.Generator().manual_seed(seed)
    feats = []
    for n in lengths:
        f = F.create_empty_np_features([n])
        f['atom_positions'] = (torch.randn(n, 3, generator=g) * 4).numpy()
        feats.append(F.pad_np_features(f, 1, pad))
    batch = {k: torch.as_tensor(np.stack([f[k] for f in feats])) for k in feats[0]}
    if motif:       # sample 0 carries a 4-residue motif (conditioned), sample 1 stays unconditional
        batch['fixed_sequence_mask'][0, 2:6] = True
    return batch


def test_checkpoint_carries_adam_state_and_resume_is_bit_exact(tmp_path):
    """Lightning's ModelCheckpoint keeps `optimizer_states` next to the weights (train.py:35-39) and a new run opens the next
    version_* directory: two steps -> save_checkpoint(trainer=) -> load_model (latest version / epoch, model_io.py:84-137) ->
    GenieTrainer.resume -> a third step gives the weights of the uninterrupted three-step run BIT FOR BIT (gradients from the
    oracle's autograd through the CPU test backend).  The file holds tensors and plain containers only (weights_only=True) and its
    optimizer entry is what torch.optim.Adam.load_state_dict accepts over the Denoiser's parameters in state_dict order."""
    from _oracle_backend import OracleBackend
    from genie2_amd.diffusion import load_model, save_checkpoint, get_versions, get_epochs
    from genie2_amd.training import GenieTrainer
    from genie2_amd import pack
    root = tmp_path / 'runs'
    (root / 'tiny').mkdir(parents=True)
    (root / 'tiny' / 'configuration').write_text('name tiny\nnumPairTransformLayers 1\nnumStructureLayers 1\nnumTimesteps 50\nmaximumNumResidues 32\nlearningRate 0.001\n')
    batch = _tiny_batch()

    def steps(tr, which):
        for k in which:
            torch.manual_seed(1000 + k)          # the step's draws of s and z (genie.py:72-79) come from the global generator
            tr.training_step(batch)
            tr.optimizer_step()

    g_full = load_model(str(root), 'tiny')                # no checkpoint yet: the untrained default
    full = GenieTrainer(g_full, backend=OracleBackend(g_full.model.dims), train_mode=False)
    steps(full, (0, 1, 2))

    g_a = load_model(str(root), 'tiny')
    a = GenieTrainer(g_a, backend=OracleBackend(g_a.model.dims), train_mode=False)
    steps(a, (0, 1))
    a.epoch = 4
    ck = root / 'tiny' / 'version_0' / 'checkpoints' / 'epoch=3.ckpt'
    save_checkpoint(g_a, str(ck), epoch=3, global_step=a.step, trainer=a)
    steps(a, (2,))                                        # training goes on after a checkpoint (the module / backend were not touched)
    assert torch.equal(a.w, full.w)

    blob = torch.load(str(ck), map_location='cpu', weights_only=True)
    assert {'state_dict', 'optimizer_states', 'epoch', 'global_step', 'lr_schedulers'} <= set(blob)
    layout = pack.weight_layout(g_a.model.dims)
    assert list(blob['state_dict']) == ['model.' + k for k, _ in layout]
    params = [torch.nn.Parameter(torch.zeros(shape)) for _, shape in layout]
    opt = torch.optim.Adam(params, lr=1e-4)
    opt.load_state_dict(blob['optimizer_states'][0])
    assert float(opt.state[params[0]]['step']) == 2.0 and opt.param_groups[0]['lr'] == 1e-3
    assert opt.state[params[3]]['exp_avg'].shape == params[3].shape

    assert get_versions(str(root), 'tiny') == [0] and get_epochs(str(root), 'tiny', 0) == [3]
    g_b = load_model(str(root), 'tiny')
    assert g_b.checkpoint_info['epoch'] == 3 and g_b.checkpoint_info['global_step'] == 2
    b = GenieTrainer(g_b, backend=OracleBackend(g_b.model.dims), train_mode=False)
    b.resume(g_b.checkpoint_info)
    assert b.step == 2 and b.epoch == 4
    steps(b, (2,))
    assert torch.equal(b.w, full.w)                       # bit for bit
    # a weights-only checkpoint (what the reference's own resume amounts to) restarts Adam
    save_checkpoint(g_a, str(root / 'tiny' / 'version_1' / 'checkpoints' / 'epoch=0.ckpt'), epoch=0)
    g_c = load_model(str(root), 'tiny')
    c = GenieTrainer(g_c, backend=OracleBackend(g_c.model.dims), train_mode=False)
    c.resume(g_c.checkpoint_info)
    assert c.step == 0 and c.epoch == 1 and float(c.m.abs().max()) == 0.0


def test_training_loss_log_follows_the_reference_split():
    """genie.py:106-118: per step unweighted / weighted loss; per sample motif + scaffold losses when the sample is conditioned,
    the unconditional loss otherwise, each over its own residue count."""
    from _oracle_backend import OracleBackend, small_config
    from genie2_amd.diffusion import Genie, mse
    from genie2_amd.training import GenieTrainer, format_log
    from genie2_amd.features import prepare_tensor_features
    genie = Genie(small_config())
    tr = GenieTrainer(genie, backend=OracleBackend(genie.model.dims), train_mode=False)
    batch = _tiny_batch(motif=True)
    torch.manual_seed(3)
    tr.training_step(batch)
    log = tr.loss_log()
    assert len(log['motif_mse_loss']) == 1 and len(log['scaffold_mse_loss']) == 1 and len(log['unconditional_mse_loss']) == 1
    f = prepare_tensor_features(batch)
    call = tr.backend.calls[-1]
    zp = tr.last['z']
    cm = f['residue_mask'] * f['fixed_sequence_mask']
    im = f['residue_mask'] * ~f['fixed_sequence_mask']
    cond = mse(zp, call['z'], cm, aggregate='sum') / cm.sum(-1)
    infill = mse(zp, call['z'], im, aggregate='sum') / im.sum(-1)
    assert abs(log['motif_mse_loss'][0] - float(cond[0])) < 1e-5 and abs(log['scaffold_mse_loss'][0] - float(infill[0])) < 1e-5
    assert abs(log['unconditional_mse_loss'][0] - float(infill[1])) < 1e-5
    line = format_log(0, 1, log)
    assert all(k in line for k in ('unweighted_loss', 'weighted_loss', 'motif_mse_loss', 'scaffold_mse_loss', 'unconditional_mse_loss'))


def test_motif_placements_enumerate_like_the_reference():
    """unconditional_smc.py:172-232 on hand-enumerable cases: placements in ascending order of the segment starts, non-overlapping,
    in order, inside the sequence; the mask layout [n_placement, n_segment, N, 3]; thinning to max_offsets with one choice() draw."""
    from genie.sampler.unconditional_smc import get_all_motif_locations, generate_motif_index_mask
    from genie2_amd.smc import placement_masks
    assert get_all_motif_locations(4, [2]) == [[(0, 1)], [(1, 2)], [(2, 3)]]
    assert get_all_motif_locations(5, [2, 1]) == [[(0, 1), (2, 2)], [(0, 1), (3, 3)], [(0, 1), (4, 4)], [(1, 2), (3, 3)], [(1, 2), (4, 4)],
                                                  [(2, 3), (4, 4)]]
    assert get_all_motif_locations(3, [2, 2]) == [] and get_all_motif_locations(4, [2, 2]) == [[(0, 1), (2, 3)]]
    # count = C(L - total + k, k); order = lexicographic in the starts
    import math
    locs = get_all_motif_locations(12, [3, 2, 2])
    assert len(locs) == math.comb(12 - 7 + 3, 3)
    starts = [tuple(s for s, _ in pl) for pl in locs]
    assert starts == sorted(starts) and len(set(starts)) == len(starts)
    for pl in locs:
        assert pl[0][0] >= 0 and pl[-1][1] <= 11 and all(pl[i][1] < pl[i + 1][0] for i in range(2))
        assert [e - s + 1 for s, e in pl] == [3, 2, 2]
    m = generate_motif_index_mask([torch.zeros(2, 3), torch.zeros(1, 3)], 5)
    assert m.shape == (6, 2, 5, 3) and m.dtype == torch.bool
    assert m[3, 0, :, 0].tolist() == [False, True, True, False, False] and m[3, 1, :, 1].tolist() == [False, False, False, True, False]
    assert placement_masks(m).sum(-1).tolist() == [3] * 6
    # more placements than max_offsets: one numpy choice(n, max_offsets, replace=False) picks which survive
    rng = np.random.RandomState(4)
    few = get_all_motif_locations(12, [3, 2, 2], max_offsets=10, rng=rng)
    pick = np.random.RandomState(4).choice(len(locs), 10, replace=False)
    assert few == [locs[i] for i in pick]


def test_train_entry_point_parses_the_reference_flags():
    """genie/train.py is an entry point like the reference's (train.py:70-81): -c required, -d / -n / -t accepted."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'genie', 'train.py'), '--help'], capture_output=True, text=True)
    assert r.returncode == 0 and all(f in r.stdout for f in ('--devices', '--num_nodes', '--config', '--test'))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'genie', 'train.py')], capture_output=True, text=True)
    assert r.returncode == 2 and 'config' in r.stderr
