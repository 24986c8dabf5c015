"""`python -m genie2_amd.sample_unconditional` -- same flags and output layout as
the reference CLI (genie/sample_unconditional.py:132-159):
outdir/pdbs/{length}_{index}.pdb."""
import argparse
import os

from tqdm import tqdm

from .diffusion import load_pretrained_model
from .multiprocessor import MultiProcessor
from .sampler import UnconditionalSampler


class UnconditionalRunner(MultiProcessor):
    def create_tasks(self, params):
        # downward from max_length (sample_unconditional.py:33-44)
        return [{'length': length}
                for length in range(params['max_length'], params['min_length'] - 1, -params['length_step'])]

    def create_constants(self, params):
        return {k: params.get(k) for k in ('rootdir', 'name', 'epoch', 'scale', 'outdir', 'num_samples', 'batch_size', 'resume')}

    def load_model(self, constants, device):
        return load_pretrained_model(constants['rootdir'], constants['name'], constants['epoch']).eval().to(device)

    def execute(self, constants, tasks, device):
        sampler = UnconditionalSampler(self.load_model(constants, device))
        for task in tqdm(tasks, desc=device):
            remaining = constants['num_samples']
            while remaining > 0:
                batch = min(constants['batch_size'], remaining)
                offset = constants['num_samples'] - remaining
                if constants.get('resume') and all(
                        os.path.exists(os.path.join(constants['outdir'], 'pdbs', '{}_{}.pdb'.format(task['length'], offset + i)))
                        for i in range(batch)):
                    remaining -= batch           # this batch was written by an earlier (interrupted) run
                    continue
                sampler.sample({
                    'length': task['length'], 'scale': constants['scale'], 'num_samples': batch,
                    'outdir': constants['outdir'], 'prefix': str(task['length']),
                    'offset': constants['num_samples'] - remaining})
                remaining -= batch


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--name', type=str, help='Model name', required=True)
    p.add_argument('--epoch', type=int, help='Model epoch', required=True)
    p.add_argument('--rootdir', type=str, help='Root directory', default='results')
    p.add_argument('--scale', type=float, help='Sampling noise scale', required=True)
    p.add_argument('--outdir', type=str, help='Output directory', required=True)
    p.add_argument('--num_samples', type=int, help='Number of samples per length', default=5)
    p.add_argument('--batch_size', type=int, help='Batch size', default=4)
    p.add_argument('--min_length', type=int, help='Minimum sequence length', default=50)
    p.add_argument('--max_length', type=int, help='Maximum sequence length', default=256)
    p.add_argument('--length_step', type=int, help='Length step size', default=1)
    p.add_argument('--num_devices', type=int, help='Number of GPU devices', default=1)
    p.add_argument('--sequential_order', action='store_true', help='Run in increasing order of length')
    p.add_argument('--resume', action='store_true', help='Skip batches whose PDB files already exist (not in the reference CLI)')
    return p


def main(args):
    UnconditionalRunner().run(vars(args), args.num_devices, args.sequential_order)


if __name__ == '__main__':
    main(build_parser().parse_args())
