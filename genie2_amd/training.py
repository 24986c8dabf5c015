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

The reference trains in fp32 (its `Trainer(...)` sets no `precision=`, train.py:54-65): `fast_math=0` (f32-grade three-piece
products) is the default and the only mode figures are quoted in; modes 1 / 2 are narrower than the reference.
"""
import torch
import torch.distributed as td

from . import pack
from .features import prepare_tensor_features


def _world():
    return td.get_world_size() if td.is_available() and td.is_initialized() else 1


class HipBackend:
    """device side of a training step: the Denoiser's GenieEngine.  The engine is fetched from the module on every use: a
    `load_state_dict` / `.to()` on the module (checkpointing through `sync_to_model`) closes the old handle and the next call
    here gets the new one."""

    def __init__(self, genie):
        self.genie = genie
        self.device = self.engine.device

    @property
    def engine(self):
        return self.genie.model.engine()

    def bind(self, features):
        self.genie.model._bound = None        # the Denoiser's "same tensors as last time" shortcut must not survive a foreign bind
        self.engine.bind_features(features)

    def q_sample(self, x0, z, c0, c1):
        return self.engine.q_sample(x0, z, c0, c1)

    def forward_backward(self, w, g, trans, rots, s, z, cond_w, seed, opts, event):
        return self.engine.train_forward_backward(w, trans, rots, s, z, cond_w, grads=g, seed=seed, struct_done_event=event, **opts)

    def adam(self, w, g, m, v, lr, step):
        from .engine import adam_step
        adam_step(w, g, m, v, lr, step)


class GenieTrainer:
    """`force_overlap`: take the event / side-stream bucket path even with one rank (what the 1-rank RCCL test on the GPU runs)."""

    BETAS, EPS = (0.9, 0.999), 1e-8           # torch.optim.Adam defaults, as ddpm.py:73-77 leaves them

    def __init__(self, genie, backend=None, train_mode=True, fast_math=0, seed=None, force_overlap=False):
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
        self.epoch = 0
        self.lr = float(self.config.optimization['lr'])
        m = self.config.model
        self.opts = dict(train_mode=train_mode, fast_math=fast_math, tri_dropout=float(m['tri_dropout']), ipa_dropout=float(m['ipa_dropout']),
                         transition_dropout=float(m['structure_transition_dropout']))
        self.seed = int(self.config.training['seed']) if seed is None else int(seed)
        self.schedule = {k: v.to(self.device) for k, v in pack.schedule_tensors(self.config.diffusion['n_timestep']).items()}
        # first float of the structure_net.* tensors: the tail bucket of the all-reduce
        self._layout = [(key, tuple(shape), _numel(shape)) for key, shape in pack.weight_layout(self.dims)]
        off, self.struct_offset = 0, None
        for key, _, n in self._layout:
            if key.startswith('structure_net.') and self.struct_offset is None:
                self.struct_offset = off
            off += n
        self.force_overlap = bool(force_overlap)
        self._side = self._event = None
        if self.device.type == 'cuda':
            self._side = torch.cuda.Stream(self.device)
            # torch creates the hipEvent lazily, at the first record(): without one, `cuda_event` is 0, the library would be
            # handed NULL and the side stream would wait on nothing
            self._event = torch.cuda.Event(enable_timing=True)
            self._event.record(torch.cuda.current_stream(self.device))
            assert self._event.cuda_event != 0
            self._tail_done = torch.cuda.Event(enable_timing=True)
        self._tail_pending = False
        self.last = None
        self.logged = {}

    # ------------------------------------------------------------------ hooks the tests replace
    def _world(self):
        return _world()

    def _all_reduce(self, t):
        """sum over ranks, ordered behind the work already on the current stream (RCCL over xGMI on GPUs: the call returns once the
        collective is enqueued, the current stream then waits for it; gloo on the CPU: blocking)"""
        td.all_reduce(t)

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
        overlap = self._event is not None and (self._world() > 1 or self.force_overlap)
        out = self.backend.forward_backward(self.w, self.g, trans_s, rots_s, s.int(), z, float(self.config.training['condition_loss_weight']),
                                            self.seed + self.step, self.opts, self._event if overlap else None)
        if overlap:     # tail bucket: reduce the structure_net gradients while the pair stack's backward pass is still running
            self._side.wait_event(self._event)
            with torch.cuda.stream(self._side):
                self._all_reduce(self.g[self.struct_offset:])
                self._tail_done.record(self._side)
            self._tail_pending = True
        self.last = out
        self._features = f
        return out['weighted_loss']

    def loss_log(self):
        """What training_step logs (genie.py:106-118): unweighted / weighted loss of the step and, per sample, the motif and scaffold
        losses of a conditioned sample or the unconditional loss of an unconditioned one (each divided by its residue count).
        Returns {name: [values]} for the last step; synchronises."""
        out, f = self.last, self._features
        log = {'unweighted_loss': [float(out['unweighted_loss'])], 'weighted_loss': [float(out['weighted_loss'])],
               'motif_mse_loss': [], 'scaffold_mse_loss': [], 'unconditional_mse_loss': []}
        rm = f['residue_mask'].to(torch.float32)
        fs = f['fixed_sequence_mask'].to(torch.float32)
        n_cond = (rm * fs).sum(-1).cpu()
        n_infill = (rm * (1.0 - fs)).sum(-1).cpu()
        cond, infill = out['condition_losses'].cpu(), out['infill_losses'].cpu()
        for i in range(cond.shape[0]):
            if n_cond[i] > 0:
                log['motif_mse_loss'].append(float(cond[i] / n_cond[i]))
                log['scaffold_mse_loss'].append(float(infill[i] / n_infill[i]))
            else:
                log['unconditional_mse_loss'].append(float(infill[i] / n_infill[i]))
        self.logged = log
        return log

    # ------------------------------------------------------------------ Lightning's DDP + ddpm.py:73-77
    def sync_gradients(self):
        """mean over ranks, as DistributedDataParallel does"""
        world = self._world()
        if self._tail_pending:          # the tail bucket is already being reduced on the side stream: the head follows on this one
            self._all_reduce(self.g[:self.struct_offset])
            torch.cuda.current_stream(self.device).wait_stream(self._side)
            self._tail_pending = False
        elif world > 1:
            self._all_reduce(self.g)
        if world > 1:
            self.g.mul_(1.0 / world)

    def optimizer_step(self):
        self.sync_gradients()
        self.step += 1
        self.backend.adam(self.w, self.g, self.m, self.v, self.lr, self.step)

    # ------------------------------------------------------------------ weights / optimizer state out and in (checkpoints, sampling)
    def _split(self, blob):
        sd, o = {}, 0
        for key, shape, n in self._layout:
            sd[key] = blob[o:o + n].reshape(shape).clone()
            o += n
        return sd

    def state_dict(self):
        """the Denoiser's state_dict from the trained blob, on the host; leaves the live module and its engine alone"""
        return self._split(self.w.detach().cpu())

    def optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() over the Denoiser's parameters in state_dict order -- one entry of the `optimizer_states` list
        of a Lightning checkpoint (train.py:35-39 keeps them with save_top_k=-1)."""
        m, v = self._split(self.m.detach().cpu()), self._split(self.v.detach().cpu())
        state = {i: {'step': torch.tensor(float(self.step)), 'exp_avg': m[k], 'exp_avg_sq': v[k]} for i, (k, _, _) in enumerate(self._layout)}
        group = {'lr': self.lr, 'betas': self.BETAS, 'eps': self.EPS, 'weight_decay': 0, 'amsgrad': False, 'maximize': False, 'foreach': None,
                 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(self._layout)))}
        return {'state': state if self.step > 0 else {}, 'param_groups': [group]}

    def load_optimizer_state_dict(self, osd):
        state = osd.get('state', {})
        if not state:
            self.m.zero_(); self.v.zero_(); self.step = 0
            return
        assert len(state) == len(self._layout), 'optimizer state does not cover the Denoiser parameters'
        ms, vs = [], []
        for i, (k, shape, n) in enumerate(self._layout):
            e = state[i] if i in state else state[str(i)]
            assert tuple(e['exp_avg'].shape) == shape, k
            ms.append(e['exp_avg'].reshape(-1).float()); vs.append(e['exp_avg_sq'].reshape(-1).float())
        self.m.copy_(torch.cat(ms)); self.v.copy_(torch.cat(vs))
        e0 = state[0] if 0 in state else state['0']
        self.step = int(float(e0['step']))

    def resume(self, ckpt):
        """continue from what diffusion.save_checkpoint(..., trainer=) wrote: Adam moments and step, epoch counter.  (The reference's
        train.py reloads weights only -- load_model + a fresh Trainer.fit, no ckpt_path -- and so restarts Adam; restoring is a superset.)"""
        if ckpt.get('optimizer_states'):
            self.load_optimizer_state_dict(ckpt['optimizer_states'][0])
        self.epoch = int(ckpt.get('epoch', -1)) + 1           # a weights-only checkpoint leaves Adam at step 0, as in the reference

    def sync_to_model(self):
        """load the trained blob into the module (for sampling with it).  The module drops its engine on load_state_dict; the backend
        picks up the new one at its next use."""
        self.genie.model.load_state_dict({k: v.to(self.device) for k, v in self.state_dict().items()})
        return self.genie

    def fit(self, dataloader, n_epoch=1, log_every=0):
        for epoch in range(n_epoch):
            for i, batch in enumerate(dataloader):
                loss = self.training_step(batch, i)
                self.optimizer_step()
                if log_every and self.step % log_every == 0:
                    print(format_log(epoch, self.step, self.loss_log()))
            self.epoch += 1
        return self.sync_to_model()


def format_log(epoch, step, log):
    parts = ['epoch {} step {}'.format(epoch, step)]
    for k, v in log.items():
        if v:
            parts.append('{} {:.5f}'.format(k, sum(v) / len(v)))
    return ' '.join(parts)


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n
