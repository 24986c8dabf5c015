"""The training loop around libgenie_hip's forward / backward pass: what `Genie.training_step`, `DDPM.configure_optimizers`
(genie/diffusion/genie.py:60-120, ddpm.py:73-77) and Lightning's `Trainer(strategy='ddp')` (train.py:54-65) do in the
reference, without Lightning:

    trainer = GenieTrainer(genie)                       # genie2_amd.diffusion.Genie on a GPU
    for batch in dataloader:
        loss = trainer.training_step(batch)             # noise, q_sample + frames, Denoiser forward + backward (HIP)
        trainer.optimizer_step()                        # DDP mean all-reduce of the gradient blob (RCCL), Adam (HIP)

One process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI); the only exchange is the gradient all-reduce:
the flat fp32 blob (62.9 MB for the base model) in two buckets -- the structure_net tail, whose gradients are final first, is
reduced on a side stream while the pair stack's backward pass still runs; the rest follows.
"""
import torch
import torch.distributed as td

from . import pack
from .features import prepare_tensor_features


def _world():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


class HipBackend:
    """device side of a training step: the bound GenieEngine"""

    def __init__(self, genie):
        self.genie = genie
        self.engine = genie.model.engine()
        self.device = self.engine.device

    def bind(self, features):
        self.engine.bind_features(features)

    def q_sample(self, x0, z, c0, c1):
        return self.engine.q_sample(x0, z, c0, c1)

    def forward_backward(self, w, g, trans, rots, s, z, cond_w, seed, opts, event):
        return self.engine.train_forward_backward(w, trans, rots, s, z, cond_w, grads=g, seed=seed, struct_done_event=event, **opts)

    def adam(self, w, g, m, v, lr, step):
        from .engine import adam_step
        adam_step(w, g, m, v, lr, step)


class GenieTrainer:
    def __init__(self, genie, backend=None, train_mode=True, fast_math=0, seed=None):
        self.genie = genie
        self.config = genie.config
        self.dims = genie.model.dims
        self.backend = backend if backend is not None else HipBackend(genie)
        self.device = self.backend.device
        self.w = pack.flatten_state_dict(genie.model.state_dict(), self.dims).to(self.device)
        self.g = torch.zeros_like(self.w)
        self.m = torch.zeros_like(self.w)
        self.v = torch.zeros_like(self.w)
        self.step = 0
        self.lr = float(self.config.optimization['lr'])
        m = self.config.model
        self.opts = dict(train_mode=train_mode, fast_math=fast_math, tri_dropout=float(m['tri_dropout']), ipa_dropout=float(m['ipa_dropout']),
                         transition_dropout=float(m['structure_transition_dropout']))
        self.seed = int(self.config.training['seed']) if seed is None else int(seed)
        self.schedule = {k: v.to(self.device) for k, v in pack.schedule_tensors(self.config.diffusion['n_timestep']).items()}
        # first float of the structure_net.* tensors: the tail bucket of the all-reduce
        off, self.struct_offset = 0, None
        for key, shape in pack.weight_layout(self.dims):
            if key.startswith('structure_net.') and self.struct_offset is None:
                self.struct_offset = off
            n = 1
            for s in shape:
                n *= s
            off += n
        self._side = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None
        self._event = torch.cuda.Event() if self.device.type == 'cuda' else None
        self._tail_work = None
        self.last = None

    # ------------------------------------------------------------------ genie.py:60-120
    def training_step(self, batch, batch_idx=0):
        f = {k: v.to(self.device) for k, v in prepare_tensor_features(batch).items()}
        B = f['atom_positions'].shape[0]
        T = self.config.diffusion['n_timestep']
        s = torch.randint(T, size=(B,)).to(self.device) + 1                                  # genie.py:72-76 (drawn on the host, as there)
        z = torch.randn_like(f['atom_positions']) * f['residue_mask'].unsqueeze(-1)          # genie.py:79
        self.backend.bind(f)
        trans_s, rots_s = self.backend.q_sample(f['atom_positions'], z, self.schedule['sqrt_alphas_cumprod'][s],
                                                self.schedule['sqrt_one_minus_alphas_cumprod'][s])
        overlap = self._event is not None and _world() > 1
        out = self.backend.forward_backward(self.w, self.g, trans_s, rots_s, s.int(), z, float(self.config.training['condition_loss_weight']),
                                            self.seed + self.step, self.opts, self._event if overlap else None)
        if overlap:     # tail bucket: reduce the structure_net gradients while the pair stack's backward pass is still running
            self._side.wait_event(self._event)
            with torch.cuda.stream(self._side):
                self._tail_work = td.all_reduce(self.g[self.struct_offset:], async_op=True)
        self.last = out
        return out['weighted_loss']

    # ------------------------------------------------------------------ Lightning's DDP + ddpm.py:73-77
    def sync_gradients(self):
        """mean over ranks, as DistributedDataParallel does"""
        world = _world()
        if world == 1:
            return
        if self._tail_work is not None:
            head = td.all_reduce(self.g[:self.struct_offset], async_op=True)
            self._tail_work.wait()
            head.wait()
            self._tail_work = None
            if self._side is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._side)
        else:
            td.all_reduce(self.g[self.struct_offset:])
            td.all_reduce(self.g[:self.struct_offset])
        self.g.mul_(1.0 / world)

    def optimizer_step(self):
        self.sync_gradients()
        self.step += 1
        self.backend.adam(self.w, self.g, self.m, self.v, self.lr, self.step)

    # ------------------------------------------------------------------ weights back into the module (checkpoints, sampling)
    def sync_to_model(self):
        sd, o = {}, 0
        blob = self.w.detach()
        for key, shape in pack.weight_layout(self.dims):
            n = 1
            for s in shape:
                n *= s
            sd[key] = blob[o:o + n].reshape(shape).clone()
            o += n
        self.genie.model.load_state_dict(sd)
        return self.genie

    def fit(self, dataloader, n_epoch=1, log_every=0):
        for epoch in range(n_epoch):
            for i, batch in enumerate(dataloader):
                loss = self.training_step(batch, i)
                self.optimizer_step()
                if log_every and self.step % log_every == 0:
                    print('epoch {} step {} weighted_loss {:.5f}'.format(epoch, self.step, float(loss)))
        return self.sync_to_model()
