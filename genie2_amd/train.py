"""`python -m genie2_amd.train -c <config> [-d N]` -- the reference's train.py (genie/train.py:14-75) without Lightning.
Multi-GPU: launch one process per GPU with torch.distributed.run (`--nproc-per-node N --master-addr 127.0.0.1`); each rank
reads its shard of the training set (DistributedSampler), gradients are averaged over RCCL."""
import argparse
import os
import random

import numpy as np
import torch
import torch.distributed as td

from .config import Config
from .data import GenieDataModule
from .diffusion import load_model, save_checkpoint
from .training import GenieTrainer


def main(args):
    config = Config(filename=args.config)
    world, rank, local = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0)), int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    if world > 1:
        td.init_process_group('nccl', device_id=torch.device('cuda', local))
    seed = config.training['seed']                       # seed_everything(config.training['seed'], workers=True)
    random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
    dm = GenieDataModule(**config.io, batch_size=config.training['batch_size'])
    if rank == 0:
        dm.setup()
    if world > 1:
        td.barrier()
    model = load_model(config.io['rootdir'], config.io['name']).to(f'cuda:{local}')
    trainer = GenieTrainer(model)
    sampler = None
    if world > 1:
        from torch.utils.data.distributed import DistributedSampler
        sampler = DistributedSampler(dm._dataset('train'), num_replicas=world, rank=rank, shuffle=True, seed=seed)
    loader = dm.train_dataloader(sampler=sampler)
    every = config.training['checkpoint_every_n_epoch']
    ckdir = os.path.join(config.io['rootdir'], config.io['name'], 'version_0', 'checkpoints')
    for epoch in range(config.training['n_epoch']):
        if sampler is not None:
            sampler.set_epoch(epoch)
        for i, batch in enumerate(loader):
            loss = trainer.training_step(batch, i)
            trainer.optimizer_step()
            if rank == 0 and trainer.step % config.training['log_every_n_step'] == 0:
                print('epoch {} step {} weighted_loss {:.5f}'.format(epoch, trainer.step, float(loss)))
        if rank == 0 and (epoch + 1) % every == 0:
            save_checkpoint(trainer.sync_to_model(), os.path.join(ckdir, 'epoch={}.ckpt'.format(epoch)), epoch=epoch, global_step=trainer.step)
    if world > 1:
        td.destroy_process_group()


if __name__ == '__main__':
    p = argparse.ArgumentParser()
    p.add_argument('-d', '--devices', type=int, help='Number of GPU devices to use (informational: ranks come from torch.distributed.run)')
    p.add_argument('-n', '--num_nodes', type=int, help='Number of nodes')
    p.add_argument('-c', '--config', type=str, help='Path for configuration file', required=True)
    p.add_argument('-t', '--test', action='store_true', help='Enable test mode', default=False)
    main(p.parse_args())
