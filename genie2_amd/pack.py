"""Host-side packing for libgenie_hip.so: the weight blob, the sinusoidal
tables and the diffusion schedule.

Key names / shapes follow the reference `Denoiser.state_dict()`
(genie/model/model.py:76-123 and sub-modules); checkpoint
files prefix them with 'model.' (genie/diffusion/ddpm.py:26,
genie/utils/model_io.py:159-173).
"""
import math

import torch

DIM_KEYS = (
    'c_s', 'c_p', 'c_pos_emb', 'c_chain_emb', 'c_timestep_emb', 'relpos_k', 'template_dist_n_bin',
    'template_dist_min', 'template_dist_step', 'n_pair_transform_layer', 'c_hidden_mul', 'pair_transition_n',
    'n_structure_layer', 'n_structure_block', 'c_hidden_ipa', 'n_head_ipa', 'n_qk_point', 'n_v_point',
    'rescale', 'n_timestep', 'max_n_res', 'max_n_chain')


def _linear(out, prefix, o, i, bias=True):
    out.append((prefix + '.weight', (o, i)))
    if bias:
        out.append((prefix + '.bias', (o,)))


def _norm(out, prefix, c):
    out.append((prefix + '.weight', (c,)))
    out.append((prefix + '.bias', (c,)))


def weight_layout(dims):
    """Ordered [(key, shape)] of every Denoiser parameter."""
    d = dims
    c_s, c_p, ch = d['c_s'], d['c_p'], d['c_hidden_mul']
    H, C, Pq, Pv = d['n_head_ipa'], d['c_hidden_ipa'], d['n_qk_point'], d['n_v_point']
    nbin = d['template_dist_n_bin']
    lay = []
    _linear(lay, 'single_feature_net.linear', c_s, d['c_pos_emb'] + d['c_chain_emb'] + d['c_timestep_emb'] + 23, False)
    pf = 'pair_feature_net.'
    _linear(lay, pf + 'linear_s_p_i', c_p, c_s, False)
    _linear(lay, pf + 'linear_s_p_j', c_p, c_s, False)
    _linear(lay, pf + 'linear_relpos', c_p, 2 * d['relpos_k'] + 3, False)
    _linear(lay, pf + 'linear_template', c_p, nbin + 6, False)
    _linear(lay, pf + 'linear_motif_template', c_p, nbin + 2, False)
    for layer in range(d['n_pair_transform_layer']):
        base = f'pair_transform_net.net.{layer}.'
        for direction in ('tri_mul_out.', 'tri_mul_in.'):
            t = base + direction
            for name in ('linear_a_p', 'linear_a_g', 'linear_b_p', 'linear_b_g'):
                _linear(lay, t + name, ch, c_p)
            _linear(lay, t + 'linear_g', c_p, c_p)
            _linear(lay, t + 'linear_z', c_p, ch)
            _norm(lay, t + 'layer_norm_in', c_p)
            _norm(lay, t + 'layer_norm_out', ch)
        t = base + 'pair_transition.'
        _norm(lay, t + 'layer_norm', c_p)
        _linear(lay, t + 'linear_1', d['pair_transition_n'] * c_p, c_p)
        _linear(lay, t + 'linear_2', c_p, d['pair_transition_n'] * c_p)
    for layer in range(d['n_structure_layer']):
        base = f'structure_net.net.{layer}.'
        lay.append((base + 'ipa.head_weights', (H,)))
        _linear(lay, base + 'ipa.linear_q', H * C, c_s)
        _linear(lay, base + 'ipa.linear_kv', 2 * H * C, c_s)
        _linear(lay, base + 'ipa.linear_q_points', 3 * H * Pq, c_s)
        _linear(lay, base + 'ipa.linear_kv_points', 3 * H * (Pq + Pv), c_s)
        _linear(lay, base + 'ipa.linear_b', H, c_p)
        _linear(lay, base + 'ipa.linear_out', c_s, H * (c_p + C + 4 * Pv))
        _norm(lay, base + 'ipa_layer_norm', c_s)
        for k in (1, 2, 3):
            _linear(lay, base + f'transition.layers.0.linear_{k}', c_s, c_s)
        _norm(lay, base + 'transition.layer_norm', c_s)
        _linear(lay, base + 'bb_update.linear', 6, c_s)
    return lay


def flatten_state_dict(state_dict, dims):
    """state_dict -> one contiguous fp32 CPU tensor in `weight_layout` order
    (the `blob` argument of genie_load_weights).  Accepts keys with or without
    the checkpoint's 'model.' prefix; refuses missing keys and wrong shapes."""
    parts = []
    for key, shape in weight_layout(dims):
        t = state_dict.get(key)
        if t is None:
            t = state_dict.get('model.' + key)
        if t is None:
            raise KeyError(f'state_dict lacks {key}')
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f'{key}: expected shape {tuple(shape)}, got {tuple(t.shape)}')
        parts.append(t.detach().to('cpu', torch.float32).reshape(-1))
    return torch.cat(parts).contiguous()


def sinusoidal_table(n_rows, N, D):
    """Row v = sinusoidal_encoding(v, N, D) for v = 0..n_rows-1, evaluated with
    the same fp32 torch expression as genie/utils/encoding.py:5-25 (even
    columns cos(v*pi / N^(2(k-1)/D)), odd columns sin(v*pi / N^(2k/D)), k 1-based)."""
    v = torch.arange(n_rows, dtype=torch.int32)
    k = torch.arange(1, D + 1)
    arg_sin = v.unsqueeze(-1) * math.pi / (N ** (2 * k / D))
    arg_cos = v.unsqueeze(-1) * math.pi / (N ** (2 * (k - 1) / D))
    tab = torch.sin(arg_sin)
    tab[:, 0::2] = torch.cos(arg_cos)[:, 0::2]
    return tab.float().contiguous()


def cosine_betas(n_timestep):
    """genie/diffusion/schedule.py:27-49: length T+1, beta_0 = 0."""
    steps = n_timestep + 1
    x = torch.linspace(0, n_timestep, steps)
    ac = torch.cos((x / steps) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.cat([torch.zeros(1), torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)])


def schedule_tensors(n_timestep):
    """The DDPM terms BaseSampler reads (genie/diffusion/ddpm.py:40-56)."""
    betas = cosine_betas(n_timestep)
    alphas = 1. - betas
    alphas_cumprod = torch.cumprod(alphas, 0)
    return {
        'betas': betas,
        'alphas': alphas,
        'alphas_cumprod': alphas_cumprod,
        'sqrt_betas': torch.sqrt(betas),
        'sqrt_alphas': torch.sqrt(alphas),
        'sqrt_alphas_cumprod': torch.sqrt(alphas_cumprod),
        'sqrt_one_minus_alphas_cumprod': torch.sqrt(1. - alphas_cumprod),
    }


def schedule_block(sched):
    """[4][T+1] block for genie_set_tables."""
    return torch.stack([sched['alphas'], sched['sqrt_alphas'], sched['sqrt_one_minus_alphas_cumprod'],
                        sched['sqrt_betas']]).float().contiguous()


def random_state_dict(dims, seed=0):
    """Random-init weights of the Denoiser architecture for benchmarking
    (trained checkpoints are not available offline).  Unlike the reference's
    default init (primitives.py:76-83,157-158: 'final' layers are zero) every
    matrix is non-zero so that no kernel runs on trivial operands."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in weight_layout(dims):
        if key.endswith('head_weights'):
            t = 0.5413 + 0.3 * torch.randn(shape, generator=g)
        elif 'layer_norm' in key:
            t = (1.0 if key.endswith('weight') else 0.0) + 0.1 * torch.randn(shape, generator=g)
        elif key.endswith('bias'):
            t = 0.1 * torch.randn(shape, generator=g) + (1.0 if ('_g.bias' in key or 'linear_g.bias' in key) else 0.0)
        else:
            t = torch.randn(shape, generator=g) * ((0.1 if 'bb_update' in key else 1.0) / math.sqrt(shape[1]))
        sd[key] = t.float().contiguous()
    return sd


BASE_DIMS = dict(
    c_s=384, c_p=128, rescale=1.0, c_pos_emb=256, c_chain_emb=64, c_timestep_emb=512,
    relpos_k=32, template_dist_min=2.0, template_dist_step=0.5, template_dist_n_bin=37,
    n_pair_transform_layer=5, c_hidden_mul=128, pair_transition_n=4,
    n_structure_layer=8, n_structure_block=1, c_hidden_ipa=16, n_head_ipa=12, n_qk_point=4, n_v_point=8,
    n_timestep=1000, max_n_res=256, max_n_chain=1)


def sinusoidal_encoding(v, N, D):
    """genie/utils/encoding.py:5-25 for arbitrary index tensors `v` [*] -> [*, D] (the device tables above are
    this function evaluated on 0..n-1)."""
    k = torch.arange(1, D + 1, device=v.device)
    shape = (1,) * v.dim() + (D,)
    sin_enc = torch.sin(v.unsqueeze(-1) * math.pi / (N ** (2 * k / D)).view(shape))
    cos_enc = torch.cos(v.unsqueeze(-1) * math.pi / (N ** (2 * (k - 1) / D)).view(shape))
    enc = sin_enc.clone()
    enc[..., 0::2] = cos_enc[..., 0::2]
    return enc
