"""
ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU (PyTorch fp32/fp64) restatement of the Genie 2 denoising hot path of
marvinli00/genie2.  It is the checker for the HIP kernels in
`genie2_amd/csrc` and the `cpu_baseline` ("port") leg of `bench.py`.
Nothing under `genie2_amd/` imports this file; the product path never falls
back to it.

Parity status: PINNED.  `oracle/make_goldens.py` (runs only in the development
container, where /root/reference is importable) drives the real reference
modules on seeded inputs and stores their outputs as fixtures under
`tests/golden/`; `tests/test_oracle_golden.py` checks every function here
against those fixtures.  The reference itself ships no tests or golden vectors
(SURVEY.md section 4).

Every function cites the reference file:line it restates (paths relative to
the reference root).  The code is written functionally over a flat
`state_dict` with the reference's key names instead of
the reference's nn.Module tree.
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# configuration (genie/config.py:41-80 defaults, results/base/configuration)
# --------------------------------------------------------------------------

BASE_DIMS = dict(
    c_s=384, c_p=128, rescale=1.0,
    c_pos_emb=256, c_chain_emb=64, c_timestep_emb=512,
    relpos_k=32, template_dist_min=2.0, template_dist_step=0.5, template_dist_n_bin=37,
    n_pair_transform_layer=5, c_hidden_mul=128, pair_transition_n=4,
    n_structure_layer=8, n_structure_block=1, c_hidden_ipa=16, n_head_ipa=12,
    n_qk_point=4, n_v_point=8,
    n_timestep=1000, max_n_res=256, max_n_chain=1,
)


def small_dims(**over):
    """A reduced-depth config for quick tests (same per-layer maths)."""
    d = dict(BASE_DIMS)
    d.update(n_pair_transform_layer=2, n_structure_layer=2, n_timestep=100)
    d.update(over)
    return d


# --------------------------------------------------------------------------
# state_dict template + synthetic weights (hazard 3)
# --------------------------------------------------------------------------

def state_dict_template(dims):
    """[(key, shape)] in the order of the reference Denoiser.state_dict()
    (genie/model/model.py:76-123 and the sub-module constructors)."""
    d = dims
    c_s, c_p, ch = d['c_s'], d['c_p'], d['c_hidden_mul']
    H, C, Pq, Pv = d['n_head_ipa'], d['c_hidden_ipa'], d['n_qk_point'], d['n_v_point']
    nb = d['template_dist_n_bin']
    out = []
    n_single_in = d['c_pos_emb'] + d['c_chain_emb'] + d['c_timestep_emb'] + 20 + 3
    out.append(('single_feature_net.linear.weight', (c_s, n_single_in)))
    out.append(('pair_feature_net.linear_s_p_i.weight', (c_p, c_s)))
    out.append(('pair_feature_net.linear_s_p_j.weight', (c_p, c_s)))
    out.append(('pair_feature_net.linear_relpos.weight', (c_p, 2 * d['relpos_k'] + 3)))
    out.append(('pair_feature_net.linear_template.weight', (c_p, nb + 6)))
    out.append(('pair_feature_net.linear_motif_template.weight', (c_p, nb + 2)))
    for l in range(d['n_pair_transform_layer']):
        for tm in ('tri_mul_out', 'tri_mul_in'):
            p = f'pair_transform_net.net.{l}.{tm}.'
            for nm, (o, i) in (('linear_a_p', (ch, c_p)), ('linear_a_g', (ch, c_p)),
                               ('linear_b_p', (ch, c_p)), ('linear_b_g', (ch, c_p)),
                               ('linear_g', (c_p, c_p)), ('linear_z', (c_p, ch))):
                out.append((p + nm + '.weight', (o, i)))
                out.append((p + nm + '.bias', (o,)))
            out.append((p + 'layer_norm_in.weight', (c_p,)))
            out.append((p + 'layer_norm_in.bias', (c_p,)))
            out.append((p + 'layer_norm_out.weight', (ch,)))
            out.append((p + 'layer_norm_out.bias', (ch,)))
        p = f'pair_transform_net.net.{l}.pair_transition.'
        n = d['pair_transition_n']
        out.append((p + 'layer_norm.weight', (c_p,)))
        out.append((p + 'layer_norm.bias', (c_p,)))
        out.append((p + 'linear_1.weight', (n * c_p, c_p)))
        out.append((p + 'linear_1.bias', (n * c_p,)))
        out.append((p + 'linear_2.weight', (c_p, n * c_p)))
        out.append((p + 'linear_2.bias', (c_p,)))
    for l in range(d['n_structure_layer']):
        p = f'structure_net.net.{l}.'
        out.append((p + 'ipa.head_weights', (H,)))
        for nm, o, i in (('linear_q', H * C, c_s), ('linear_kv', 2 * H * C, c_s),
                         ('linear_q_points', H * Pq * 3, c_s),
                         ('linear_kv_points', H * (Pq + Pv) * 3, c_s),
                         ('linear_b', H, c_p),
                         ('linear_out', c_s, H * (c_p + C + Pv * 4))):
            out.append((p + 'ipa.' + nm + '.weight', (o, i)))
            out.append((p + 'ipa.' + nm + '.bias', (o,)))
        out.append((p + 'ipa_layer_norm.weight', (c_s,)))
        out.append((p + 'ipa_layer_norm.bias', (c_s,)))
        for k in (1, 2, 3):
            out.append((p + f'transition.layers.0.linear_{k}.weight', (c_s, c_s)))
            out.append((p + f'transition.layers.0.linear_{k}.bias', (c_s,)))
        out.append((p + 'transition.layer_norm.weight', (c_s,)))
        out.append((p + 'transition.layer_norm.bias', (c_s,)))
        out.append((p + 'bb_update.linear.weight', (6, c_s)))
        out.append((p + 'bb_update.linear.bias', (6,)))
    return out


def synthetic_state_dict(dims, seed=0):
    """Deterministic 'everything is live' weights (SURVEY.md hazard 3: the
    reference's default init zeroes every final linear, so default-init
    goldens would not exercise the kernels).  Matrices ~ N(0, 1/fan_in) scaled
    so activations stay O(1); LN gamma ~ 1 +- 0.1, beta ~ +-0.1; biases small;
    IPA head weights random around softplus^-1(1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in state_dict_template(dims):
        if key.endswith('head_weights'):
            t = 0.5413 + 0.3 * torch.randn(shape, generator=g)
        elif 'layer_norm' in key and key.endswith('weight'):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif 'layer_norm' in key and key.endswith('bias'):
            t = 0.1 * torch.randn(shape, generator=g)
        elif key.endswith('bias'):
            t = 0.1 * torch.randn(shape, generator=g)
            if '_g.bias' in key or key.endswith('linear_g.bias'):
                t = t + 1.0  # gating init: b = 1 (primitives.py:150-153)
        else:
            fan_in = shape[1]
            std = 1.0 / math.sqrt(fan_in)
            if 'bb_update' in key:
                std *= 0.1  # keep frame updates small
            t = std * torch.randn(shape, generator=g)
        sd[key] = t.float().contiguous()
    return sd


# --------------------------------------------------------------------------
# schedule (genie/diffusion/schedule.py:27-49, genie/diffusion/ddpm.py:36-66)
# --------------------------------------------------------------------------

def cosine_beta_schedule(n_timestep):
    steps = n_timestep + 1
    x = torch.linspace(0, n_timestep, steps)
    ac = torch.cos((x / steps) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = 1 - (ac[1:] / ac[:-1])
    return torch.cat([torch.zeros((1,)), torch.clip(betas, 0, 0.999)])


def setup_schedule(n_timestep):
    """The subset of ddpm.py:36-66 the sampler reads (base.py:249-270)."""
    betas = cosine_beta_schedule(n_timestep)
    alphas = 1. - betas
    alphas_cumprod = torch.cumprod(alphas, 0)
    return dict(
        betas=betas, alphas=alphas, alphas_cumprod=alphas_cumprod,
        sqrt_betas=torch.sqrt(betas), sqrt_alphas=torch.sqrt(alphas),
        sqrt_one_minus_alphas_cumprod=torch.sqrt(1. - alphas_cumprod),
    )


# --------------------------------------------------------------------------
# geometry (genie/utils/geo_utils.py, genie/utils/affine_utils.py)
# --------------------------------------------------------------------------

def compute_frenet_frames(coords, chains, mask, eps=1e-10):
    """geo_utils.py:21-85.  torch.cross is given dim=-1 explicitly: the
    reference omits `dim`, which picks the first size-3 dim and so computes a
    different product when B == 3 (a reference quirk we
    do not reproduce)."""
    t = coords[:, 1:] - coords[:, :-1]
    t = t / torch.sqrt(eps + torch.sum(t ** 2, dim=-1)).unsqueeze(-1)
    b = torch.cross(t[:, :-1], t[:, 1:], dim=-1)
    b = b / torch.sqrt(eps + torch.sum(b ** 2, dim=-1)).unsqueeze(-1)
    n = torch.cross(b, t[:, 1:], dim=-1)
    tbn = torch.stack([t[:, 1:], b, n], dim=-1)  # columns t, b, n  (det = -1)
    B, N = mask.shape
    rots = torch.eye(3, dtype=coords.dtype).repeat(B, N, 1, 1)
    for i in range(B):
        length = int(torch.sum(mask[i]))
        if length > 2:
            rots[i, 1:length - 1] = tbn[i, :length - 2]
        ch = chains[i].tolist()
        for j in range(length):          # start of chain (geo_utils.py:69-72)
            if j == 0 or ch[j] != ch[j - 1]:
                rots[i, j] = rots[i, j + 1]
        for j in range(length):          # end of chain (geo_utils.py:74-77)
            if j == length - 1 or ch[j] != ch[j + 1]:
                rots[i, j] = rots[i, j - 1]
    return rots


def rot_matmul(a, b):
    """affine_utils.py:24-42 (explicit sums of products, row by row)."""
    rows = []
    for i in range(3):
        rows.append(torch.stack([
            a[..., i, 0] * b[..., 0, j] + a[..., i, 1] * b[..., 1, j] + a[..., i, 2] * b[..., 2, j]
            for j in range(3)], dim=-1))
    return torch.stack(rows, dim=-2)


def rot_vec_mul(r, t):
    """affine_utils.py:44-52."""
    x, y, z = t[..., 0], t[..., 1], t[..., 2]
    return torch.stack([
        r[..., 0, 0] * x + r[..., 0, 1] * y + r[..., 0, 2] * z,
        r[..., 1, 0] * x + r[..., 1, 1] * y + r[..., 1, 2] * z,
        r[..., 2, 0] * x + r[..., 2, 1] * y + r[..., 2, 2] * z], dim=-1)


def quat_to_rot(q):
    """affine_utils.py:299-334 (Hamilton, w first), written out."""
    a, b, c, d = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    rows = [
        [a * a + b * b - c * c - d * d, 2 * b * c - 2 * a * d, 2 * b * d + 2 * a * c],
        [2 * b * c + 2 * a * d, a * a - b * b + c * c - d * d, 2 * c * d - 2 * a * b],
        [2 * b * d - 2 * a * c, 2 * c * d + 2 * a * b, a * a - b * b - c * c + d * d],
    ]
    return torch.stack([torch.stack(r, dim=-1) for r in rows], dim=-2)


def rot_to_quat_eigh(rot):
    """affine_utils.py:336-355: top eigenvector of K/3 via LAPACK eigh.
    Its SIGN is backend dependent (SURVEY.md hazard 1)."""
    xx, xy, xz = rot[..., 0, 0], rot[..., 0, 1], rot[..., 0, 2]
    yx, yy, yz = rot[..., 1, 0], rot[..., 1, 1], rot[..., 1, 2]
    zx, zy, zz = rot[..., 2, 0], rot[..., 2, 1], rot[..., 2, 2]
    k = [[xx + yy + zz, zy - yz, xz - zx, yx - xy],
         [zy - yz, xx - yy - zz, xy + yx, xz + zx],
         [xz - zx, xy + yx, yy - xx - zz, yz + zy],
         [yx - xy, xz + zx, yz + zy, zz - xx - yy]]
    k = (1. / 3.) * torch.stack([torch.stack(t, dim=-1) for t in k], dim=-2)
    _, vec = torch.linalg.eigh(k)
    return vec[..., -1]


def rot_to_quat_closed(rot):
    """Closed form of the same eigenvector for a proper rotation: K = 4qq^T - I,
    so q is any normalised column of (K + I); take the column with the largest
    diagonal (Shepperd).  Canonical sign: that component is positive.  This is
    what the HIP kernel computes; `quat_sign_codes` ties it to eigh's sign."""
    xx, xy, xz = rot[..., 0, 0], rot[..., 0, 1], rot[..., 0, 2]
    yx, yy, yz = rot[..., 1, 0], rot[..., 1, 1], rot[..., 1, 2]
    zx, zy, zz = rot[..., 2, 0], rot[..., 2, 1], rot[..., 2, 2]
    one = torch.ones_like(xx)
    cols = [
        torch.stack([one + xx + yy + zz, zy - yz, xz - zx, yx - xy], -1),
        torch.stack([zy - yz, one + xx - yy - zz, xy + yx, xz + zx], -1),
        torch.stack([xz - zx, xy + yx, one + yy - xx - zz, yz + zy], -1),
        torch.stack([yx - xy, xz + zx, yz + zy, one + zz - xx - yy], -1),
    ]
    M = torch.stack(cols, dim=-1)                      # [..., 4(row), 4(col)]
    diag = torch.diagonal(M, dim1=-2, dim2=-1)
    m = torch.argmax(diag, dim=-1)
    col = torch.gather(M, -1, m[..., None, None].expand(*m.shape, 4, 1))[..., 0]
    return col / torch.sqrt(torch.sum(col * col, dim=-1, keepdim=True))


def quat_sign_codes(q_ref):
    """int8 code per pair that pins a quaternion's sign robustly:
    code = 1 + 2*m + (q_ref[m] < 0), m = argmax |q_ref|.  0 means 'canonical'.
    The kernel flips its closed-form q so that component m has that sign."""
    m = torch.argmax(q_ref.abs(), dim=-1)
    neg = torch.gather(q_ref, -1, m[..., None])[..., 0] < 0
    return (1 + 2 * m + neg.long()).to(torch.int8)


def apply_sign_codes(q, codes):
    """Flip q (closed form) according to `quat_sign_codes` output."""
    codes = codes.long()
    m = torch.clamp((codes - 1) // 2, min=0)
    want_neg = ((codes - 1) % 2) == 1
    comp = torch.gather(q, -1, m[..., None])[..., 0]
    flip = (codes > 0) & ((comp < 0) != want_neg)
    return torch.where(flip[..., None], -q, q)


# --------------------------------------------------------------------------
# encodings (genie/utils/encoding.py:5-25)
# --------------------------------------------------------------------------

def sinusoidal_encoding(v, N, D):
    k = torch.arange(1, D + 1)
    sin_div = N ** (2 * k / D)
    cos_div = N ** (2 * (k - 1) / D)
    sin_enc = torch.sin(v.unsqueeze(-1) * math.pi / sin_div)
    cos_enc = torch.cos(v.unsqueeze(-1) * math.pi / cos_div)
    enc = torch.zeros_like(sin_enc)
    enc[..., 0::2] = cos_enc[..., 0::2]
    enc[..., 1::2] = sin_enc[..., 1::2]
    return enc


# --------------------------------------------------------------------------
# denoiser pieces
# --------------------------------------------------------------------------

def _lin(sd, key, x):
    return F.linear(x, sd[key + '.weight'], sd.get(key + '.bias'))


def _ln(sd, key, x):
    return F.layer_norm(x, (x.shape[-1],), sd[key + '.weight'], sd[key + '.bias'], 1e-5)


def single_feature_net(sd, dims, timesteps, features, n_res):
    """single_feature_net.py:58-142."""
    pos = sinusoidal_encoding(features['residue_index'], dims['max_n_res'], dims['c_pos_emb'])
    chn = sinusoidal_encoding(features['chain_index'], dims['max_n_chain'], dims['c_chain_emb'])
    s = timesteps.unsqueeze(-1).repeat(1, n_res)
    tem = sinusoidal_encoding(s, dims['n_timestep'], dims['c_timestep_emb'])
    fsm = features['fixed_sequence_mask']
    aat = features['aatype'] * fsm.unsqueeze(-1)
    x = torch.cat([pos, chn, tem, aat, fsm.unsqueeze(-1), fsm.unsqueeze(-1),
                   features['interface_mask'].unsqueeze(-1)], dim=-1)
    return F.linear(x, sd['single_feature_net.linear.weight']) * features['residue_mask'].unsqueeze(-1)


def soft_distance_bins(dims, coords, mask):
    """pair_feature_net.py:223-269 (fork-specific soft one-hot, alpha = 4);
    distance has eps inside the sqrt (geo_utils.py:19)."""
    diff = coords.unsqueeze(2) - coords.unsqueeze(1)
    d = (1e-10 + torch.sum(diff ** 2, dim=-1)) ** 0.5
    v = dims['template_dist_min'] + torch.arange(0, dims['template_dist_n_bin']) * dims['template_dist_step']
    oh = F.softmax(-4.0 * torch.abs(d.unsqueeze(-1) - v), dim=-1)
    pm = mask.unsqueeze(1) * mask.unsqueeze(2)
    return oh * pm.unsqueeze(-1)


def pair_orientations(rots, mask, quat_mode='eigh', sign_codes=None):
    """pair_feature_net.py:271-301: r[b,i,j] = R_j . R_i (NOT R_i^T R_j)."""
    r = torch.matmul(rots.unsqueeze(1), rots.unsqueeze(2))
    if quat_mode == 'eigh':
        q = rot_to_quat_eigh(r)
    else:
        q = rot_to_quat_closed(r)
        if sign_codes is not None:
            q = apply_sign_codes(q, sign_codes)
    pm = mask.unsqueeze(1) * mask.unsqueeze(2)
    return q * pm.unsqueeze(-1), q


def relpos(sd, dims, features):
    """pair_feature_net.py:166-221."""
    ri, ci = features['residue_index'], features['chain_index']
    k = dims['relpos_k']
    same = ci[:, :, None] == ci[:, None, :]
    d_same = torch.clip(ri[:, :, None] - ri[:, None, :] + k, 0, 2 * k)
    d = d_same * same + (2 * k + 1) * (~same)
    oh = F.one_hot(d.long(), num_classes=2 * k + 2).float()
    return F.linear(torch.cat([oh, same.unsqueeze(-1).float()], dim=-1),
                    sd['pair_feature_net.linear_relpos.weight'])


def pair_feature_net(sd, dims, s, rots, trans, features, quat_mode='eigh', sign_codes=None, taps=None):
    """pair_feature_net.py:72-160."""
    rm = features['residue_mask']
    pm = rm.unsqueeze(1) * rm.unsqueeze(2)
    p_i = F.linear(s, sd['pair_feature_net.linear_s_p_i.weight'])
    p_j = F.linear(s, sd['pair_feature_net.linear_s_p_j.weight'])
    p = p_i[:, :, None, :] + p_j[:, None, :, :]
    p = p + relpos(sd, dims, features)
    fsm2 = features['fixed_structure_mask'].unsqueeze(-1).float()
    qm, q_raw = pair_orientations(rots, rm, quat_mode, sign_codes)
    if taps is not None:
        taps['quat'] = q_raw
    p = p + F.linear(torch.cat([soft_distance_bins(dims, trans, rm), qm, fsm2, fsm2], dim=-1),
                     sd['pair_feature_net.linear_template.weight'])
    p = p + F.linear(torch.cat([
        soft_distance_bins(dims, features['atom_positions'], features['fixed_sequence_mask']) * fsm2,
        fsm2, fsm2], dim=-1), sd['pair_feature_net.linear_motif_template.weight'])
    return p * pm.unsqueeze(-1)


def triangle_multiplication(sd, pfx, z, mask, outgoing):
    """modules/triangular_multiplicative_update.py:57-110."""
    m = mask.unsqueeze(-1)
    zn = _ln(sd, pfx + 'layer_norm_in', z)
    a = _lin(sd, pfx + 'linear_a_p', zn) * torch.sigmoid(_lin(sd, pfx + 'linear_a_g', zn)) * m
    b = _lin(sd, pfx + 'linear_b_p', zn) * torch.sigmoid(_lin(sd, pfx + 'linear_b_g', zn)) * m
    if outgoing:   # x[i,j,c] = sum_k a[i,k,c] b[j,k,c]
        x = torch.matmul(a.permute(0, 3, 1, 2), b.permute(0, 3, 2, 1))
    else:          # x[i,j,c] = sum_k a[k,i,c] b[k,j,c]
        x = torch.matmul(a.permute(0, 3, 2, 1), b.permute(0, 3, 1, 2))
    x = x.permute(0, 2, 3, 1)
    x = _lin(sd, pfx + 'linear_z', _ln(sd, pfx + 'layer_norm_out', x))
    return x * torch.sigmoid(_lin(sd, pfx + 'linear_g', zn))


def pair_transition(sd, pfx, z, mask):
    """modules/pair_transition.py:48-87 (eval-mode chunking is an identity)."""
    zn = _ln(sd, pfx + 'layer_norm', z)
    h = F.relu(_lin(sd, pfx + 'linear_1', zn))
    return _lin(sd, pfx + 'linear_2', h) * mask.unsqueeze(-1)


def pair_transform_net(sd, dims, p, features, taps=None, dropout_masks=None):
    """pair_transform_net.py:91-119,224-232.  Dropouts are identity in eval; `dropout_masks` (train mode) maps
    ('tri', layer, 0 | 1) to the row-shared keep-mask [B,1,N,C] (already scaled by 1 / (1 - rate), modules/dropout.py:23-76)."""
    rm = features['residue_mask']
    pm = (rm.unsqueeze(1) * rm.unsqueeze(2)).float()
    dm = dropout_masks or {}
    for l in range(dims['n_pair_transform_layer']):
        pfx = f'pair_transform_net.net.{l}.'
        p = p + triangle_multiplication(sd, pfx + 'tri_mul_out.', p, pm, True) * dm.get(('tri', l, 0), 1.0)
        if taps is not None and l == 0:
            taps['p_after_trimul_out0'] = p
        p = p + triangle_multiplication(sd, pfx + 'tri_mul_in.', p, pm, False) * dm.get(('tri', l, 1), 1.0)
        if taps is not None and l == 0:
            taps['p_after_trimul_in0'] = p
        p = p + pair_transition(sd, pfx + 'pair_transition.', p, pm)
        p = p * pm.unsqueeze(-1)
        if taps is not None and l == 0:
            taps['p_after_layer0'] = p
    return p


def invariant_point_attention(sd, dims, pfx, s, z, rots, trans, mask, taps=None):
    """modules/invariant_point_attention.py:100-260."""
    H, C, Pq, Pv = dims['n_head_ipa'], dims['c_hidden_ipa'], dims['n_qk_point'], dims['n_v_point']
    B, N = s.shape[:2]
    q = _lin(sd, pfx + 'linear_q', s).view(B, N, H, C)
    kv = _lin(sd, pfx + 'linear_kv', s).view(B, N, H, 2 * C)
    k, v = kv[..., :C], kv[..., C:]

    def points(key, n_pts):
        x = _lin(sd, pfx + key, s)
        x = torch.stack(torch.split(x, x.shape[-1] // 3, dim=-1), dim=-1)  # [B,N,H*P,3]
        x = rot_vec_mul(rots.unsqueeze(-3), x) + trans.unsqueeze(-2)
        return x.view(B, N, H, n_pts, 3)

    q_pts = points('linear_q_points', Pq)
    kv_pts = points('linear_kv_points', Pq + Pv)
    k_pts, v_pts = kv_pts[..., :Pq, :], kv_pts[..., Pq:, :]

    b = _lin(sd, pfx + 'linear_b', z)                       # [B,N,N,H]
    a = torch.matmul(q.permute(0, 2, 1, 3), k.permute(0, 2, 3, 1))
    a = a * math.sqrt(1. / (3 * C))
    a = a + math.sqrt(1. / 3) * b.permute(0, 3, 1, 2)
    pt = q_pts.unsqueeze(-4) - k_pts.unsqueeze(-5)          # [B,N,N,H,Pq,3]
    pt = torch.sum(pt ** 2, dim=-1)
    hw = F.softplus(sd[pfx + 'head_weights']).view(1, 1, 1, H, 1)
    hw = hw * math.sqrt(1. / (3 * (Pq * 9. / 2)))
    pt = torch.sum(pt * hw, dim=-1) * (-0.5)                # [B,N,N,H]
    sq = mask.unsqueeze(-1) * mask.unsqueeze(-2)
    sq = 1e5 * (sq - 1)
    a = a + pt.permute(0, 3, 1, 2) + sq.unsqueeze(-3)
    a = F.softmax(a, dim=-1)                                # [B,H,N,N]
    if taps is not None:
        taps['ipa_att'] = a

    o = torch.matmul(a, v.transpose(-2, -3)).transpose(-2, -3).reshape(B, N, H * C)
    o_pt = torch.matmul(a.unsqueeze(-3), v_pts.permute(0, 2, 4, 1, 3))   # [B,H,3,N,Pv]
    o_pt = o_pt.permute(0, 3, 1, 4, 2)                                   # [B,N,H,Pv,3]
    o_pt = rot_vec_mul(rots.transpose(-1, -2)[:, :, None, None], o_pt - trans[:, :, None, None, :])
    o_pt_norm = torch.sqrt(torch.sum(o_pt ** 2, dim=-1) + 1e-8).reshape(B, N, H * Pv)
    o_pt = o_pt.reshape(B, N, H * Pv, 3)
    o_pair = torch.matmul(a.transpose(-2, -3), z).reshape(B, N, H * z.shape[-1])
    cat = torch.cat((o, *torch.unbind(o_pt, dim=-1), o_pt_norm, o_pair), dim=-1)
    if taps is not None:
        taps['ipa_cat'] = cat
    return _lin(sd, pfx + 'linear_out', cat)


def backbone_update(sd, pfx, s):
    """modules/backbone_update.py:40-66."""
    params = _lin(sd, pfx + 'linear', s)
    quats, t_upd = params[..., :3], params[..., 3:]
    denom = torch.sqrt(torch.sum(quats ** 2, dim=-1) + 1)
    quats = torch.cat((torch.ones_like(quats[..., :1]), quats), dim=-1) / denom.unsqueeze(-1)
    return quat_to_rot(quats), t_upd


def structure_net(sd, dims, s, p, rots, trans, features, taps=None, dropout_masks=None):
    """structure_net.py:76-116,189-243 and modules/structure_transition.py:34-70.  `dropout_masks` (train mode):
    ('ipa', i) on s + ipa(s) (structure_net.py:109) and ('transition', i) before the transition's LayerNorm
    (structure_transition.py:66), i = running layer index, each [B,N,c_s] scaled by 1 / (1 - rate)."""
    mask = features['residue_mask'].float()
    dm = dropout_masks or {}
    states = [s]                                         # structure_net.py:236-243: input + s after every layer
    li = 0
    for _ in range(dims['n_structure_block']):
        for l in range(dims['n_structure_layer']):
            pfx = f'structure_net.net.{l}.'
            s = s + invariant_point_attention(sd, dims, pfx + 'ipa.', s, p, rots, trans, mask,
                                              taps if l == 0 else None)
            s = s * dm.get(('ipa', li), 1.0)
            s = _ln(sd, pfx + 'ipa_layer_norm', s)
            t0 = s
            h = F.relu(_lin(sd, pfx + 'transition.layers.0.linear_1', s))
            h = F.relu(_lin(sd, pfx + 'transition.layers.0.linear_2', h))
            s = _lin(sd, pfx + 'transition.layers.0.linear_3', h) + t0
            s = s * dm.get(('transition', li), 1.0)
            li += 1
            s = _ln(sd, pfx + 'transition.layer_norm', s)
            if taps is not None and l == 0:
                taps['s_after_layer0'] = s
            states.append(s)
            r_upd, t_upd = backbone_update(sd, pfx + 'bb_update.', s)
            trans = rot_vec_mul(rots, t_upd) + trans       # affine_utils.py:109-116
            rots = rot_matmul(rots, r_upd)
    if taps is not None:
        taps['states'] = torch.stack(states)
    return s, rots, trans


def denoiser_forward(sd, dims, rots, trans, timesteps, features, quat_mode='eigh',
                     sign_codes=None, taps=None, dropout_masks=None):
    """model/model.py:125-192.  Returns dict(z, s, p, s_final, rots, trans)."""
    f = prepare_features(features)
    trans0 = trans
    trans = trans * dims['rescale']
    N = trans.shape[1]
    s = single_feature_net(sd, dims, timesteps, f, N)
    p = pair_feature_net(sd, dims, s, rots, trans, f, quat_mode, sign_codes, taps)
    if taps is not None:
        taps['p_init'] = p
    if dims['n_pair_transform_layer'] > 0:
        p = pair_transform_net(sd, dims, p, f, taps, dropout_masks)
    s_fin, r_out, t_out = structure_net(sd, dims, s, p, rots, trans, f, taps, dropout_masks)
    t_out = t_out * (1. / dims['rescale'])
    return dict(z=trans0 - t_out, s=s, p=p, s_final=s_fin, rots=r_out, trans=t_out)


# --------------------------------------------------------------------------
# training step, the two ends around the denoiser (genie/diffusion/genie.py:66-105, utils/loss.py:4-36)
# --------------------------------------------------------------------------

def training_schedule(n_timestep):
    """ddpm.py:40-57: the two terms training_step indexes (genie.py:80-81)."""
    ac = setup_schedule(n_timestep)['alphas_cumprod']
    return dict(sqrt_alphas_cumprod=torch.sqrt(ac), sqrt_one_minus_alphas_cumprod=torch.sqrt(1. - ac))


def q_sample(x0, s, z, chains, mask, sched):
    """genie.py:77-87: noised coordinates and their Frenet frames.  z is already masked (genie.py:77)."""
    trans = sched['sqrt_alphas_cumprod'][s.long()].view(-1, 1, 1) * x0 + \
        sched['sqrt_one_minus_alphas_cumprod'][s.long()].view(-1, 1, 1) * z
    return trans, compute_frenet_frames(trans, chains, mask)


def mse(x_pred, x, mask, aggregate=None, eps=1e-10):
    """utils/loss.py:4-36 (a masked per-residue L2 error despite the name)."""
    errors = (eps + torch.sum((x_pred - x) ** 2, dim=-1)) ** 0.5
    if aggregate is None:
        return errors * mask
    if aggregate == 'mean':
        return torch.sum(errors * mask, dim=-1) / torch.sum(mask, dim=-1)
    assert aggregate == 'sum'
    return torch.sum(errors * mask, dim=-1)


def training_loss(z_pred, z, features, condition_loss_weight):
    """genie.py:90-105.  Returns dict(unweighted_loss, weighted_loss (what training_step returns), and the
    per-structure condition / infill sums)."""
    rm = features['residue_mask'].float()
    fs = features['fixed_sequence_mask'].bool()
    condition_mask = rm * fs
    infill_mask = rm * ~fs
    cond = mse(z_pred, z, condition_mask, aggregate='sum')
    infill = mse(z_pred, z, infill_mask, aggregate='sum')
    unweighted = (cond + infill) / features['num_residues'].reshape(-1).float()
    w = condition_loss_weight
    weighted = (w * cond + infill) / (w * torch.sum(condition_mask, dim=-1) + torch.sum(infill_mask, dim=-1))
    return dict(unweighted_loss=torch.mean(unweighted), weighted_loss=torch.mean(weighted), condition_losses=cond,
                infill_losses=infill)


def train_dropout_masks(dims, B, N, seed, tri_dropout=0.25, ipa_dropout=0.1, transition_dropout=0.1):
    """The keep-masks the HIP training path derives from `seed` (csrc/train.h drop_scale: a counter-based hash of (seed, site, element
    index)), rebuilt here so that the oracle can run the SAME train-mode forward: site tags 4 l + {0, 1} for the two triangle
    multiplications of pair layer l (row-shared: index (b N + j) C + c), 1000 + 2 i + {0, 1} for structure layer i."""
    import numpy as np

    def scale(tag, n, rate):
        idx = np.arange(n, dtype=np.uint64)
        with np.errstate(over='ignore'):
            x = ((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) * np.uint32(0x9E3779B1)) ^ ((idx >> np.uint64(32)).astype(np.uint32) * np.uint32(0x85EBCA77)) \
                ^ (np.uint32(seed & 0xFFFFFFFF) * np.uint32(0xC2B2AE3D)) ^ (np.uint32(tag) * np.uint32(0x27D4EB2F))
            x ^= x >> np.uint32(16); x *= np.uint32(0x85EBCA6B); x ^= x >> np.uint32(13); x *= np.uint32(0xC2B2AE35); x ^= x >> np.uint32(16)
        u = (x >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        return torch.from_numpy(np.where(u < np.float32(rate), np.float32(0), np.float32(1.0) / (np.float32(1.0) - np.float32(rate))).astype(np.float32))

    cp, cs = dims['c_p'], dims['c_s']
    m = {}
    for l in range(dims['n_pair_transform_layer']):
        for k in (0, 1):
            m[('tri', l, k)] = scale(4 * l + k, B * N * cp, tri_dropout).reshape(B, 1, N, cp)
    for i in range(dims['n_structure_layer'] * dims['n_structure_block']):
        m[('ipa', i)] = scale(1000 + 2 * i, B * N * cs, ipa_dropout).reshape(B, N, cs)
        m[('transition', i)] = scale(1001 + 2 * i, B * N * cs, transition_dropout).reshape(B, N, cs)
    return m


def prepare_features(features):
    """Same dtypes the model sees after feat_utils.py:304-321."""
    out = dict(features)
    for k in ('residue_mask', 'residue_index', 'chain_index', 'aatype'):
        out[k] = features[k].int() if k != 'aatype' else features[k].int()
    for k in ('fixed_sequence_mask', 'fixed_structure_mask', 'interface_mask'):
        out[k] = features[k].bool()
    out['atom_positions'] = features['atom_positions'].float()
    return out


# --------------------------------------------------------------------------
# features (genie/utils/feat_utils.py:17-65, 192-321) -- torch tensors direct
# --------------------------------------------------------------------------

def empty_features(lengths_per_sample, n_pad=None, chains_per_sample=None):
    """Batched unconditional features.  lengths_per_sample: list of total
    lengths; chains_per_sample: optional list of per-sample chain-length lists."""
    B = len(lengths_per_sample)
    N = n_pad or max(lengths_per_sample)
    f = dict(
        aatype=torch.zeros(B, N, 20, dtype=torch.int32),
        atom_positions=torch.zeros(B, N, 3),
        residue_mask=torch.zeros(B, N, dtype=torch.int32),
        residue_index=torch.zeros(B, N, dtype=torch.int32),
        chain_index=torch.zeros(B, N, dtype=torch.int32),
        fixed_sequence_mask=torch.zeros(B, N, dtype=torch.bool),
        fixed_structure_mask=torch.zeros(B, N, N, dtype=torch.bool),
        fixed_group=torch.zeros(B, N, dtype=torch.int32),
        interface_mask=torch.zeros(B, N, dtype=torch.bool),
        num_residues=torch.tensor(lengths_per_sample, dtype=torch.int32),
    )
    for b, L in enumerate(lengths_per_sample):
        f['residue_mask'][b, :L] = 1
        chains = chains_per_sample[b] if chains_per_sample else [L]
        off = 0
        for ci, cl in enumerate(chains):
            f['residue_index'][b, off:off + cl] = torch.arange(cl, dtype=torch.int32)
            f['chain_index'][b, off:off + cl] = ci
            off += cl
    f['num_chains'] = torch.tensor([len(c) for c in chains_per_sample] if chains_per_sample
                                   else [1] * B, dtype=torch.int32)
    return f


def add_motif(features, b, motif_positions, motif_index, aatype_idx=None):
    """Mark residues `motif_index` of sample b as a fixed motif (one group):
    what create_np_features_from_motif_pdb (feat_utils.py:95-130) produces."""
    idx = torch.as_tensor(motif_index)
    features['atom_positions'][b, idx] = torch.as_tensor(motif_positions, dtype=torch.float32)
    features['fixed_sequence_mask'][b, idx] = True
    features['fixed_structure_mask'][b, idx[:, None], idx[None, :]] = True
    features['fixed_group'][b, idx] = 1
    if aatype_idx is None:
        aatype_idx = torch.arange(len(idx)) % 20
    features['aatype'][b, idx, :] = 0
    features['aatype'][b, idx, torch.as_tensor(aatype_idx)] = 1
    return features


# --------------------------------------------------------------------------
# reverse loop (genie/sampler/base.py:169-289)
# --------------------------------------------------------------------------

def p_sample_step(sched, step, scale, trans, z_pred, eps, features):
    """base.py:249-282 for one step value (same for the whole batch).
    Returns (new_trans, new_rots).  eps=None on the last step (step == 1)."""
    mask = features['residue_mask'].unsqueeze(-1)
    w_z = (1. - sched['alphas'][step]) / sched['sqrt_one_minus_alphas_cumprod'][step]
    mean = (1. / sched['sqrt_alphas'][step]) * (trans - w_z * z_pred)
    mean = mean * mask
    if step == 1:
        new = mean
    else:
        new = (mean + scale * sched['sqrt_betas'][step] * eps) * mask
    rots = compute_frenet_frames(new, features['chain_index'], features['residue_mask'])
    return new, rots


def sample_loop(sd, dims, features, noise, scale, quat_mode='eigh', sign_codes_per_step=None,
                record_every=0, denoiser=None):
    """base.py:218-287 with the Gaussian draws supplied explicitly:
    noise[0] is the initial trans (base.py:227), noise[k] (k = 1..T-1) is the
    draw of the k-th loop iteration (step = T-k+1 ... 2); step 1 draws nothing.
    `denoiser` lets make_goldens.py plug in the real reference model."""
    T = dims['n_timestep']
    sched = setup_schedule(T)
    f = prepare_features(features)
    trans = noise[0].clone()
    rots = compute_frenet_frames(trans, f['chain_index'], f['residue_mask'])
    B = trans.shape[0]
    rec = []
    for it, step in enumerate(range(T, 0, -1)):
        ts = torch.full((B,), step, dtype=torch.int32)
        codes = None if sign_codes_per_step is None else sign_codes_per_step[it]
        if denoiser is not None:
            z = denoiser(rots, trans, ts)
        else:
            z = denoiser_forward(sd, dims, rots, trans, ts, f, quat_mode, codes)['z']
        eps = None if step == 1 else noise[it + 1]
        trans, rots = p_sample_step(sched, step, scale, trans, z, eps, f)
        if record_every and (it + 1) % record_every == 0:
            rec.append(trans.clone())
    return trans, rots, rec
