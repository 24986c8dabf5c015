"""`python -m genie2_amd.sample_scaffold` -- same flags and output layout as the reference CLI
(genie/sample_scaffold.py): outdir/motif=<name>/{pdbs,motif_pdbs}/<name>_<index>.pdb."""
import argparse
import glob
import os

from tqdm import tqdm

from .diffusion import load_pretrained_model
from .multiprocessor import MultiProcessor
from .sampler import ScaffoldSampler


class ScaffoldRunner(MultiProcessor):
    def create_tasks(self, params):
        # (the reference reads the global `args` here, sample_scaffold.py:34-49; same values)
        if params.get('motif_name') is not None:
            names = [params['motif_name']]
        else:
            names = [p.split('/')[-1].split('.')[0] for p in glob.glob(os.path.join(params['datadir'], '*.pdb'))]
        return [{'motif_name': n} for n in names]

    def create_constants(self, params):
        return {k: params[k] for k in ('rootdir', 'name', 'epoch', 'scale', 'strength', 'outdir', 'num_samples',
                                       'batch_size', 'datadir')}

    def load_model(self, constants, device):
        return load_pretrained_model(constants['rootdir'], constants['name'], constants['epoch']).eval().to(device)

    def execute(self, constants, tasks, device):
        sampler = ScaffoldSampler(self.load_model(constants, device))
        for task in tqdm(tasks, desc=device):
            outdir = os.path.join(constants['outdir'], 'motif={}'.format(task['motif_name']))
            remaining = constants['num_samples']
            while remaining > 0:
                batch = min(constants['batch_size'], remaining)
                sampler.sample({
                    'filepath': os.path.join(constants['datadir'], '{}.pdb'.format(task['motif_name'])),
                    'scale': constants['scale'], 'strength': constants['strength'], 'num_samples': batch,
                    'outdir': outdir, 'prefix': task['motif_name'], 'offset': constants['num_samples'] - remaining})
                remaining -= batch


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--name', type=str, help='Model name', required=True)
    p.add_argument('--epoch', type=int, help='Model epoch', required=True)
    p.add_argument('--rootdir', type=str, help='Root directory', default='results')
    p.add_argument('--scale', type=float, help='Sampling noise scale', required=True)
    p.add_argument('--outdir', type=str, help='Output directory', required=True)
    p.add_argument('--strength', type=float, help='Sampling classifier-free strength', default=0)
    p.add_argument('--num_samples', type=int, help='Number of samples per length', default=100)
    p.add_argument('--batch_size', type=int, help='Batch size', default=4)
    p.add_argument('--motif_name', type=str, help='Motif name', default=None)
    p.add_argument('--datadir', type=str, help='Data directory', default='data/design25')
    p.add_argument('--num_devices', type=int, help='Number of GPU devices', default=1)
    return p


def main(args):
    ScaffoldRunner().run(vars(args), args.num_devices)


if __name__ == '__main__':
    main(build_parser().parse_args())
