"""Stub MultiProcessor subclass for the CPU test of the spawn path (importable from a spawned child)."""
import json
import os

from genie2_amd.multiprocessor import MultiProcessor


class StubRunner(MultiProcessor):
    def create_tasks(self, params):
        return [{'length': n} for n in range(params['max_length'], params['min_length'] - 1, -params['length_step'])]

    def create_constants(self, params):
        return {'outdir': params['outdir'], 'fail_on': params.get('fail_on')}

    def execute(self, constants, tasks, device):
        if constants['fail_on'] == device:
            raise SystemExit(3)
        with open(os.path.join(constants['outdir'], device.replace(':', '_') + '.json'), 'w') as fh:
            json.dump({'device': device, 'pid': os.getpid(), 'tasks': tasks}, fh)
