"""One process per GPU, static task split, no communication
(genie/utils/multiprocessor.py:13-100)."""
import math
import random
from abc import ABC, abstractmethod

import torch.multiprocessing as mp


def split_tasks(tasks, num_devices):
    """Contiguous bins of ceil(len/num_devices) (multiprocessor.py:84-91);
    trailing devices may get nothing, exactly like the reference."""
    binsize = math.ceil(len(tasks) / num_devices) if tasks else 0
    return [tasks[binsize * i: binsize * (i + 1)] for i in range(num_devices)]


class MultiProcessor(ABC):
    @abstractmethod
    def create_tasks(self, params):
        raise NotImplementedError

    @abstractmethod
    def create_constants(self, params):
        raise NotImplementedError

    @abstractmethod
    def execute(self, constants, tasks, device):
        raise NotImplementedError

    def run(self, params, num_devices, sequential_order=False):
        tasks = self.create_tasks(params)
        if num_devices > 1 and not sequential_order:
            random.shuffle(tasks)
        constants = self.create_constants(params)
        bins = split_tasks(tasks, num_devices)
        if num_devices == 1:
            self.execute(constants, bins[0], 'cuda:0')
            return
        # spawn, not fork: a forked child must not inherit an initialised HIP runtime
        ctx = mp.get_context('spawn')
        procs = [ctx.Process(target=self.execute, args=(constants, bins[i], f'cuda:{i}')) for i in range(num_devices)]
        for p in procs:
            p.start()
        for p in procs:
            p.join()
        # unlike the reference (multiprocessor.py:97-100, exit codes dropped) a dead worker is an error here: there is no CPU
        # fallback behind the HIP path, so a partial outdir must not look like success
        failed = [(i, p.exitcode) for i, p in enumerate(procs) if p.exitcode != 0]
        if failed:
            raise RuntimeError('worker(s) failed: ' + ', '.join('cuda:%d exit code %s' % f for f in failed))
