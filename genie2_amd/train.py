"""`python -m genie2_amd.train -c <config> [-d N]` (also `python genie/train.py -c <config>`) -- the reference's train.py
(genie/train.py:14-75) without Lightning.  Multi-GPU: launch one process per GPU with torch.distributed.run
(`--nproc-per-node N --master-addr 127.0.0.1`); each rank reads its shard of the training set (DistributedSampler), gradients
are averaged over RCCL.

Run directories as Lightning's logger + ModelCheckpoint lay them out (train.py:22-39, utils/model_io.py:7-48): every run opens
`<rootdir>/<name>/version_{n+1}/checkpoints/` and writes `epoch=<e>.ckpt` there; `load_model` starts it from the latest
version's latest epoch.  The reference reloads weights only (a fresh Trainer.fit without ckpt_path: Adam restarts, epochs count
from 0 again); here the checkpoint's Adam moments / step are restored too and the epoch count continues (`--weights_only`
gives the reference's behaviour)."""
import argparse
import os
import random

import numpy as np
import torch
import torch.distributed as td

from .config import Config
from .data import GenieDataModule
from .diffusion import get_versions, load_model, save_checkpoint
from .training import GenieTrainer, format_log


def main(args):
    config = Config(filename=args.config)
    world, rank, local = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    ddp = 'RANK' in os.environ                              # launched by torch.distributed.run (one rank is a valid rehearsal of the RCCL path)
    if ddp:
        td.init_process_group('nccl', device_id=torch.device('cuda', local))
    seed = config.training['seed']                       # seed_everything(config.training['seed'], workers=True)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    dm = GenieDataModule(**config.io, batch_size=config.training['batch_size'])
    if rank == 0:
        dm.setup()
    if ddp:
        td.barrier()
    model = load_model(config.io['rootdir'], config.io['name']).to(f'cuda:{local}')
    trainer = GenieTrainer(model, force_overlap=args.force_overlap)
    info = getattr(model, 'checkpoint_info', None)
    if info and not args.weights_only:
        trainer.resume(info)
        if rank == 0:
            print('Resuming at epoch {} (Adam step {})'.format(trainer.epoch, trainer.step))
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dm._dataset('train'), num_replicas=world, rank=rank, shuffle=True, seed=seed)
    loader = dm.train_dataloader(sampler=sampler)
    every = config.training['checkpoint_every_n_epoch']
    versions = get_versions(config.io['rootdir'], config.io['name'])
    ckdir = os.path.join(config.io['rootdir'], config.io['name'], 'version_{}'.format(max(versions) + 1 if versions else 0), 'checkpoints')
    first_epoch = trainer.epoch
    for epoch in range(first_epoch, first_epoch + config.training['n_epoch']):
        if sampler is not None:
            sampler.set_epoch(epoch)
        for i, batch in enumerate(loader):
            trainer.training_step(batch, i)
            trainer.optimizer_step()
            if rank == 0 and trainer.step % config.training['log_every_n_step'] == 0:
                print(format_log(epoch, trainer.step, trainer.loss_log()))
        trainer.epoch = epoch + 1
        if rank == 0 and (epoch + 1) % every == 0:      # from the trainer's blobs: the live module and its engine are not touched
            save_checkpoint(model, os.path.join(ckdir, 'epoch={}.ckpt'.format(epoch)), epoch=epoch, global_step=trainer.step, trainer=trainer)
    if ddp:
        td.destroy_process_group()


def cli(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('-d', '--devices', type=int, help='Number of GPU devices to use (informational: ranks come from torch.distributed.run)')
    p.add_argument('-n', '--num_nodes', type=int, help='Number of nodes')
    p.add_argument('-c', '--config', type=str, help='Path for configuration file', required=True)
    p.add_argument('-t', '--test', action='store_true', help='Enable test mode', default=False)
    p.add_argument('--weights_only', action='store_true', help="Restart Adam and the epoch count when continuing from a checkpoint (the reference's behaviour)")
    p.add_argument('--force_overlap', action='store_true', help='Take the bucketed side-stream all-reduce path even with one rank (rehearsal)')
    main(p.parse_args(argv))


if __name__ == '__main__':
    cli()
