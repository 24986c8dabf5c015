"""GenieEngine: one libgenie_hip handle bound to one GPU.

PyTorch is plumbing here (device buffers and the current stream); every
tensor op of the denoising path runs inside the HIP library.
"""
import ctypes as C

import torch

from . import capi, pack

_FEATURE_DTYPES = {
    'aatype': torch.int32, 'atom_positions': torch.float32, 'residue_mask': torch.int32,
    'residue_index': torch.int32, 'chain_index': torch.int32,
    'fixed_sequence_mask': torch.bool, 'fixed_structure_mask': torch.bool, 'interface_mask': torch.bool,
}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def adam_step(p, g, m, v, lr, step, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam's update (ddpm.py:73-77) in place on flat fp32 device tensors; `step` counts from 1."""
    lib = capi.load_library()
    for t in (p, g, m, v):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    rc = lib.genie_adam_step(C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream), p.numel(), _ptr(p), _ptr(g), _ptr(m), _ptr(v),
                             float(lr), float(betas[0]), float(betas[1]), float(eps), int(step))
    if rc != 0:
        raise capi.GenieError('genie_adam_step failed (%d)' % rc)


def compute_frenet_frames(coords, chains, mask):
    """genie/utils/geo_utils.py:21-85 with the reference's own signature: coords [B,N,3] on a GPU, chains / mask [B,N]."""
    if not coords.is_cuda:
        raise capi.GenieError('compute_frenet_frames runs on the GPU (libgenie_hip); there is no CPU path')
    lib = capi.load_library()
    x = coords.to(torch.float32).contiguous()
    ch = chains.to(device=x.device, dtype=torch.int32).contiguous()
    mk = mask.to(device=x.device, dtype=torch.int32).contiguous()
    B, N = x.shape[:2]
    rots = torch.empty(B, N, 3, 3, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.genie_frenet_frames(C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), B, N, _ptr(x), _ptr(ch), _ptr(mk), _ptr(rots))
    if rc != 0:
        raise capi.GenieError('genie_frenet_frames failed (%d)' % rc)
    return rots


class GenieEngine:
    def __init__(self, dims, state_dict, device='cuda:0', n_pos=None, n_chain=None, math=None):
        self.lib = capi.load_library()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise capi.GenieError('GenieEngine needs a GPU device (there is no CPU path)')
        if not torch.cuda.is_available():
            raise capi.GenieError('no HIP device visible to PyTorch')
        self.dims = {k: dims[k] for k in pack.DIM_KEYS}
        cd = capi.GenieDims(**self.dims)
        self._h = C.c_void_p()
        rc = self.lib.genie_create(C.byref(cd), self.device.index or 0, C.byref(self._h))
        capi.check(None, rc, 'genie_create')
        blob = pack.flatten_state_dict(state_dict, self.dims)
        assert blob.numel() == self.lib.genie_weight_count(C.byref(cd))
        capi.check(self._h, self.lib.genie_load_weights(self._h, _ptr(blob), blob.numel()), 'genie_load_weights')
        self.n_pos = int(n_pos or max(self.dims['max_n_res'], 1))
        self.n_chain = int(n_chain or max(self.dims['max_n_chain'], 8))
        self.schedule = pack.schedule_tensors(self.dims['n_timestep'])
        pos = pack.sinusoidal_table(self.n_pos, self.dims['max_n_res'], self.dims['c_pos_emb'])
        chn = pack.sinusoidal_table(self.n_chain, self.dims['max_n_chain'], self.dims['c_chain_emb'])
        tt = pack.sinusoidal_table(self.dims['n_timestep'] + 1, self.dims['n_timestep'], self.dims['c_timestep_emb'])
        sb = pack.schedule_block(self.schedule)
        capi.check(self._h, self.lib.genie_set_tables(self._h, _ptr(pos), self.n_pos, _ptr(chn), self.n_chain,
                                                      _ptr(tt), _ptr(sb)), 'genie_set_tables')
        self.B = self.N = None
        if math is not None:
            self.set_math(math)

    MATH_MODES = {'f32': 0, 'hx': 1}

    def set_math(self, mode):
        """'hx' (default): pair-stack GEMMs as three f16 MFMAs on split operands; 'f32': exact f32 MFMA."""
        capi.check(self._h, self.lib.genie_set_math(self._h, self.MATH_MODES[mode]), 'genie_set_math')

    @property
    def math(self):
        return 'hx' if self.lib.genie_get_math(self._h) == 1 else 'f32'

    def close(self):
        if getattr(self, '_h', None):
            self.lib.genie_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype):
        return t.to(device=self.device, dtype=dtype).contiguous()

    def bind_features(self, features):
        """features: the reference's batched feature dict (torch tensors)."""
        f = {k: self._dev(features[k], dt) for k, dt in _FEATURE_DTYPES.items()}
        B, N = f['residue_mask'].shape
        if int(f['residue_index'].max()) >= self.n_pos or int(f['chain_index'].max()) >= self.n_chain:
            raise capi.GenieError('residue_index / chain_index exceed the uploaded encoding tables')
        gf = capi.GenieFeatures(**{k: f[k].data_ptr() for k in _FEATURE_DTYPES})
        with torch.cuda.device(self.device):
            rc = self.lib.genie_prepare_features(self._h, self._stream(), B, N, C.byref(gf))
        capi.check(self._h, rc, 'genie_prepare_features')
        self._bound = f           # keep alive until the async copies ran
        self.B, self.N = B, N

    def frenet(self, trans):
        trans = self._dev(trans, torch.float32)
        rots = torch.empty(self.B, self.N, 3, 3, device=self.device)
        capi.check(self._h, self.lib.genie_frenet(self._h, self._stream(), _ptr(trans), _ptr(rots)), 'genie_frenet')
        return rots

    def q_sample(self, x0, z, c_x0, c_z):
        """genie.py:80-87 for the bound batch: (trans_s, rots_s) from x0, masked noise z and the two schedule terms [B]."""
        x0, z = self._dev(x0, torch.float32), self._dev(z, torch.float32)
        c0, c1 = self._dev(c_x0, torch.float32), self._dev(c_z, torch.float32)
        trans = torch.empty(self.B, self.N, 3, device=self.device)
        rots = torch.empty(self.B, self.N, 3, 3, device=self.device)
        rc = self.lib.genie_q_sample(self._h, self._stream(), _ptr(x0), _ptr(z), _ptr(c0), _ptr(c1), _ptr(trans), _ptr(rots))
        capi.check(self._h, rc, 'genie_q_sample')
        return trans, rots

    def training_loss(self, z_pred, z, condition_loss_weight, grad=True):
        """genie.py:90-105: dict(unweighted_loss, weighted_loss, condition_losses, infill_losses[, grad = d weighted / d z_pred])."""
        zp, z = self._dev(z_pred, torch.float32), self._dev(z, torch.float32)
        B = self.B
        out = torch.empty(2 + 2 * B, device=self.device)
        g = torch.empty(B, self.N, 3, device=self.device) if grad else None
        rc = self.lib.genie_training_loss(self._h, self._stream(), _ptr(zp), _ptr(z), float(condition_loss_weight), _ptr(out), _ptr(g))
        capi.check(self._h, rc, 'genie_training_loss')
        res = {'unweighted_loss': out[0], 'weighted_loss': out[1], 'condition_losses': out[2:2 + B], 'infill_losses': out[2 + B:]}
        if grad:
            res['grad'] = g
        return res

    def train_forward_backward(self, weights, trans, rots, timesteps, z_target, condition_loss_weight=1.0, quat_codes=None, grads=None,
                               train_mode=True, seed=0, tri_dropout=0.25, ipa_dropout=0.1, transition_dropout=0.1, fast_math=False,
                               struct_done_event=None):
        """Forward + backward pass of Genie.training_step through the Denoiser (genie.py:88-105) for the bound batch.
        `weights`: flat fp32 device tensor in state_dict order (pack.flatten_state_dict(...).to(device)); returns
        dict(weighted_loss, unweighted_loss, condition_losses, infill_losses, z, grads) with `grads` a flat tensor of the same layout
        (overwritten when passed in) -- what the optimizer (adam_step) and the DDP all-reduce act on."""
        B, N = self.B, self.N
        w = self._dev(weights, torch.float32)
        assert w.dim() == 1 and w.numel() == self.lib.genie_weight_count(C.byref(capi.GenieDims(**self.dims)))
        g = torch.empty_like(w) if grads is None else grads
        assert g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and g.numel() == w.numel()
        tr, ro = self._dev(trans, torch.float32), self._dev(rots, torch.float32)
        ts, zt = self._dev(timesteps, torch.int32), self._dev(z_target, torch.float32)
        codes = self._dev(quat_codes, torch.int8) if quat_codes is not None else None
        if struct_done_event is not None and not struct_done_event.cuda_event:
            raise capi.GenieError('struct_done_event has no hipEvent behind it yet (torch creates it at the first record()): record it once')
        opts = capi.GenieTrainOpts(float(tri_dropout), float(ipa_dropout), float(transition_dropout), int(seed) & 0xFFFFFFFF,
                                   1 if train_mode else 0, int(fast_math),
                                   C.c_void_p(struct_done_event.cuda_event) if struct_done_event is not None else None)
        out = torch.empty(2 + 2 * B, device=self.device)
        zp = torch.empty(B, N, 3, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.genie_train_forward_backward(self._h, self._stream(), _ptr(w), _ptr(g), _ptr(tr), _ptr(ro), _ptr(ts), _ptr(zt), _ptr(codes),
                                                       float(condition_loss_weight), C.byref(opts), _ptr(out), _ptr(zp))
        capi.check(self._h, rc, 'genie_train_forward_backward')
        return {'unweighted_loss': out[0], 'weighted_loss': out[1], 'condition_losses': out[2:2 + B], 'infill_losses': out[2 + B:],
                'z': zp, 'grads': g}

    def denoise_vjp(self, weights, trans, rots, timesteps, dz, quat_codes=None):
        """(z, d <dz, z> / d trans) with the frames held fixed: torch.autograd.grad through the denoiser as the twisted-diffusion
        samplers use it (unconditional_smc.py:465-482).  `weights`: flat fp32 device tensor in state_dict order."""
        w = self._dev(weights, torch.float32)
        tr, ro = self._dev(trans, torch.float32), self._dev(rots, torch.float32)
        ts, g = self._dev(timesteps, torch.int32), self._dev(dz, torch.float32)
        codes = self._dev(quat_codes, torch.int8) if quat_codes is not None else None
        z = torch.empty(self.B, self.N, 3, device=self.device)
        dt = torch.empty(self.B, self.N, 3, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.genie_denoise_vjp(self._h, self._stream(), _ptr(w), _ptr(tr), _ptr(ro), _ptr(ts), _ptr(codes), _ptr(g), _ptr(z), _ptr(dt))
        capi.check(self._h, rc, 'genie_denoise_vjp')
        return z, dt

    def denoise(self, trans, rots, timesteps, quat_codes=None, taps=()):
        """Denoiser.forward.  Returns {'z': ..., <tap>: ...}."""
        B, N = self.B, self.N
        trans = self._dev(trans, torch.float32)
        rots = self._dev(rots, torch.float32)
        ts = self._dev(timesteps, torch.int32)
        codes = self._dev(quat_codes, torch.int8) if quat_codes is not None else None
        z = torch.empty(B, N, 3, device=self.device)
        shapes = {'s': (B, N, self.dims['c_s']), 'p': (B, N, N, self.dims['c_p']), 's_final': (B, N, self.dims['c_s']),
                  'rots_out': (B, N, 3, 3), 'trans_out': (B, N, 3), 'p_init': (B, N, N, self.dims['c_p']),
                  'p_layer0': (B, N, N, self.dims['c_p']), 'p_trimul_out0': (B, N, N, self.dims['c_p']),
                  'ipa_cat0': (B, N, self.dims['n_head_ipa'] * (self.dims['c_hidden_ipa'] + 4 * self.dims['n_v_point'] + self.dims['c_p'])),
                  'states': (1 + self.dims['n_structure_block'] * self.dims['n_structure_layer'], B, N, self.dims['c_s'])}
        out = {k: torch.empty(shapes[k], device=self.device) for k in taps}
        ct = capi.GenieTaps(**{k: v.data_ptr() for k, v in out.items()})
        rc = self.lib.genie_denoise(self._h, self._stream(), _ptr(trans), _ptr(rots), _ptr(ts), _ptr(codes), _ptr(z),
                                    C.byref(ct) if taps else None)
        capi.check(self._h, rc, 'genie_denoise')
        out['z'] = z
        return out

    def p_sample(self, step, scale, trans, z, eps):
        """In place on `trans`; returns the new frames."""
        rots = torch.empty(self.B, self.N, 3, 3, device=self.device)
        rc = self.lib.genie_p_sample(self._h, self._stream(), int(step), float(scale), _ptr(trans), _ptr(rots), _ptr(z),
                                     _ptr(eps))
        capi.check(self._h, rc, 'genie_p_sample')
        return rots

    def sample_loop(self, noise, scale, quat_codes=None, first_step=None, last_step=1, state=None, record=False):
        """noise [T,B,N,3] on device.  Returns (trans, rots, record or None)."""
        T = self.dims['n_timestep']
        first_step = T if first_step is None else first_step
        noise = self._dev(noise, torch.float32)
        n_it = first_step - last_step + 1
        codes = self._dev(quat_codes, torch.int8) if quat_codes is not None else None
        if state is None:
            trans = torch.empty(self.B, self.N, 3, device=self.device)
            rots = torch.empty(self.B, self.N, 3, 3, device=self.device)
        else:
            trans, rots = state
        rec = torch.empty(n_it, self.B, self.N, 3, device=self.device) if record else None
        rc = self.lib.genie_sample_loop(self._h, self._stream(), float(scale), _ptr(noise), _ptr(codes), first_step,
                                        last_step, _ptr(trans), _ptr(rots), _ptr(rec))
        capi.check(self._h, rc, 'genie_sample_loop')
        return trans, rots, rec

    # ------------------------------------------------------------------
    def profile(self, enable):
        self.lib.genie_profile_enable(self._h, 1 if enable else 0)

    def profile_read(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_double * 32)()
        cnt = (C.c_int64 * 32)()
        n = self.lib.genie_profile_read(self._h, names, ms, cnt, 32)
        return {names[i].decode(): (ms[i], cnt[i]) for i in range(n)}

    def workspace_bytes(self):
        return int(self.lib.genie_workspace_bytes(self._h))
