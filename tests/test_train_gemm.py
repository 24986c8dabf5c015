"""The training path's GEMM (csrc/train_kernels.hip: k_gemm_big / k_gemm_fast / k_gemm) through the C-ABI entry genie_train_gemm, against
float64 torch.matmul.  The parity tests of the training step run at N = 16 (512 pair rows) and never reach the 128 x 128-tile kernel
or most edge paths; these shapes do.  Tolerances: 3 pieces reproduce f32 products (error of an f32 accumulation), 2 pieces carry 16
significand bits, 1 piece is bf16."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from genie2_amd import capi
    return capi, capi.load_library()


def _run(lib, capi, a, b, c, M, N, K, sa, sb, sc, batch=1, nb2=1, bs=((0, 0), (0, 0), (0, 0)), nsplit=1, mode=0, terms=3, relu=0, alpha=1.0,
         bias=None, gate=None, asum=None):
    d = capi.GenieGemmDesc(M=M, N=N, K=K, batch=batch, nb2=nb2, nsplit=nsplit, mode=mode, terms=terms, relu=relu, am=sa[0], ak=sa[1], bk=sb[0],
                           bn=sb[1], cm=sc[0], cn=sc[1], a1=bs[0][0], a2=bs[0][1], b1=bs[1][0], b2=bs[1][1], c1=bs[2][0], c2=bs[2][1], alpha=alpha)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    rc = lib.genie_train_gemm(None, C.byref(d), p(a), p(b), p(c), p(bias), p(gate), p(asum))
    torch.cuda.synchronize()
    assert rc == 0
    return c


TOL = {3: 2e-6, 2: 3e-4, 1: 2e-2}     # relative to |A| |B| row-column products


def _check(got, ref, a64, b64, terms):
    scale = (a64.abs() @ b64.abs()).clamp_min(1e-30)
    err = ((got.double() - ref).abs() / scale).max().item()
    assert err <= TOL[terms], err


# (M, N, K): 128-tile kernel (>= 256 tiles of 128 x 128), 64-tile kernel, edge shapes of the generic kernel
SHAPES = [(2048, 2048, 64), (4096, 1024, 96), (256, 384, 128), (64, 64, 32), (100, 67, 39), (33, 130, 70), (512, 6, 384), (1, 1, 1)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('layout', ['kk', 'km', 'mk', 'mm'])      # which dimension of A / B is contiguous: k or the row index
@pytest.mark.parametrize('terms', [3, 1])
def test_store_matches_matmul(shape, layout, terms):
    capi, lib = _lib()
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 31 + N * 7 + K)
    a64 = torch.randn(M, K, generator=g, dtype=torch.float64)
    b64 = torch.randn(K, N, generator=g, dtype=torch.float64)
    a32, b32 = a64.float(), b64.float()
    a64, b64 = a32.double(), b32.double()
    if layout[0] == 'k': a, sa = a32.contiguous().cuda(), (K, 1)
    else: a, sa = a32.t().contiguous().cuda(), (1, M)                 # stored [K][M]
    if layout[1] == 'k': b, sb = b32.t().contiguous().cuda(), (1, K)   # stored [N][K]: element (k, n) at n K + k
    else: b, sb = b32.contiguous().cuda(), (N, 1)
    c = torch.full((M, N), float('nan'), device='cuda')
    bias = torch.randn(N, generator=g).cuda()
    _run(lib, capi, a, b, c, M, N, K, sa, sb, (N, 1), terms=terms, bias=bias, alpha=0.5)
    _check(c.cpu() - bias.cpu(), 0.5 * (a64 @ b64), a64, b64, terms)


def test_misaligned_operands_and_odd_leading_dimensions():
    """tensors inside the flat state_dict blob start at any float offset; activations have padded rows"""
    capi, lib = _lib()
    M, N, K = 256, 128, 64
    g = torch.Generator().manual_seed(3)
    for off_a, off_b, lda, ldb in [(1, 3, K + 8, K), (2, 1, K, K + 4), (3, 2, K + 1, K + 3)]:
        A = torch.randn(off_a + M * lda, generator=g).cuda()
        Bm = torch.randn(off_b + N * ldb, generator=g).cuda()
        a = A[off_a:]
        b = Bm[off_b:]
        c = torch.empty(M, N, device='cuda')
        _run(lib, capi, a, b, c, M, N, K, (lda, 1), (1, ldb), (N, 1))
        a64 = a.cpu()[:M * lda].view(M, lda)[:, :K].double()
        b64 = b.cpu()[:N * ldb].view(N, ldb)[:, :K].double().t()
        _check(c.cpu(), a64 @ b64, a64, b64, 3)


@pytest.mark.parametrize('shape,nsplit', [((128, 128, 32768), 192), ((512, 128, 8192), 48), ((128, 256, 100000), 64), ((12, 128, 4096), 16), ((100, 67, 5000), 7)])
def test_split_k_atomic_accumulation_and_row_sums(shape, nsplit):
    """the weight-gradient form: A = dY stored [K][M] (am 1), B = X stored [K][N], C += over splits; asum = column sums of dY"""
    capi, lib = _lib()
    M, N, K = shape
    g = torch.Generator().manual_seed(K)
    dy = torch.randn(K, M, generator=g)
    x = torch.randn(K, N, generator=g)
    c0 = torch.randn(M, N, generator=g)
    s0 = torch.randn(M, generator=g)
    c = c0.clone().cuda()
    asum = s0.clone().cuda()
    _run(lib, capi, dy.cuda(), x.cuda(), c, M, N, K, (1, M), (N, 1), (N, 1), nsplit=nsplit, mode=2, asum=asum)
    a64, b64 = dy.double().t(), x.double()
    _check(c.cpu() - c0, a64 @ b64, a64, b64, 3)
    ref = dy.double().sum(0)
    assert ((asum.cpu().double() - s0.double() - ref).abs() / dy.double().abs().sum(0)).max().item() <= 2e-6


def test_modes_relu_and_gate():
    capi, lib = _lib()
    g = torch.Generator().manual_seed(11)
    for M, N, K in [(2048, 2048, 64), (256, 128, 64), (70, 50, 33)]:
        a = torch.randn(M, K, generator=g)
        b = torch.randn(N, K, generator=g)
        ref = a.double() @ b.double().t()
        c0 = torch.randn(M, N, generator=g)
        c = c0.clone().cuda()
        _run(lib, capi, a.cuda(), b.cuda(), c, M, N, K, (K, 1), (1, K), (N, 1), mode=1)                         # add
        _check(c.cpu() - c0, ref, a.double(), b.double().t(), 3)
        c = torch.empty(M, N, device='cuda')
        _run(lib, capi, a.cuda(), b.cuda(), c, M, N, K, (K, 1), (1, K), (N, 1), relu=1)
        _check(c.cpu(), ref.clamp_min(0), a.double(), b.double().t(), 3)
        gate = torch.randn(M, N, generator=g)
        c = torch.empty(M, N, device='cuda')
        _run(lib, capi, a.cuda(), b.cuda(), c, M, N, K, (K, 1), (1, K), (N, 1), gate=gate.cuda())
        _check(c.cpu(), torch.where(gate > 0, ref, torch.zeros_like(ref)), a.double(), b.double().t(), 3)


def test_batched_channel_major_contraction():
    """the triangle contraction's form: B ch matrices of N x N, z = z1 nb2 + z2 (genie_train.hip: tri_fwd), both orientations"""
    capi, lib = _lib()
    Bn, ch, N = 2, 128, 128            # 256 matrices: the 128-tile kernel's threshold
    g = torch.Generator().manual_seed(5)
    a = torch.randn(Bn, ch, N, N, generator=g)
    b = torch.randn(Bn, ch, N, N, generator=g)
    bs = ((ch * N * N, N * N),) * 3
    for outgoing in (True, False):
        c = torch.empty(Bn, ch, N, N, device='cuda')
        if outgoing: sa, sb = (N, 1), (1, N)      # x[i][j] = sum_k a[i][k] b[j][k]
        else: sa, sb = (1, N), (N, 1)             # x[i][j] = sum_k a[k][i] b[k][j]
        _run(lib, capi, a.cuda(), b.cuda(), c, N, N, N, sa, sb, (N, 1), batch=Bn * ch, nb2=ch, bs=bs)
        ref = torch.einsum('bcik,bcjk->bcij', a.double(), b.double()) if outgoing else torch.einsum('bcki,bckj->bcij', a.double(), b.double())
        scale = torch.einsum('bcik,bcjk->bcij', a.double().abs(), b.double().abs()) if outgoing else torch.einsum('bcki,bckj->bcij', a.double().abs(), b.double().abs())
        assert ((c.cpu().double() - ref).abs() / scale).max().item() <= TOL[3]


def test_two_piece_split_and_bad_arguments():
    capi, lib = _lib()
    g = torch.Generator().manual_seed(2)
    M, N, K = 256, 128, 96
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    c = torch.empty(M, N, device='cuda')
    _run(lib, capi, a.cuda(), b.cuda(), c, M, N, K, (K, 1), (1, K), (N, 1), terms=2)
    _check(c.cpu(), a.double() @ b.double().t(), a.double(), b.double().t(), 2)
    d = capi.GenieGemmDesc(M=M, N=N, K=K, batch=1, nb2=1, nsplit=4, mode=0, terms=3, relu=0, am=K, ak=1, bk=1, bn=K, cm=N, cn=1, alpha=1.0)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert lib.genie_train_gemm(None, C.byref(d), p(a.cuda()), p(b.cuda()), p(c), None, None, None) == -1      # split-K needs atomic mode


@pytest.mark.parametrize('case', ['fwd_640', 'wgrad_640', 'fwd_small', 'wgrad_small'])
def test_c_in_separately_placed_blocks(case):
    """Several Linears sharing an input as ONE GEMM: the product's column blocks (forward) or row blocks (weight gradients, split-K with
    row sums) land at separate places -- the tiled kernels take the block table, other shapes fall back to one GEMM per block."""
    capi, lib = _lib()
    g = torch.Generator().manual_seed(11)
    nb = 5
    if case.startswith('fwd'):
        R, K, O = (4096, 128, 128) if case == 'fwd_640' else (96, 24, 24)
        x = torch.randn(R, K, generator=g); w = torch.randn(nb * O, K, generator=g); bias = torch.randn(nb * O, generator=g)
        # results interleaved with gaps inside one buffer, in shuffled order
        order = [3, 0, 4, 1, 2]
        buf = torch.full((nb * R * O + 5 * 64,), float('nan'), device='cuda')
        offs = [order[k] * (R * O + 64) for k in range(nb)]
        d = capi.GenieGemmDesc(M=R, N=nb * O, K=K, batch=1, nb2=1, nsplit=1, mode=0, terms=3, relu=0, am=K, ak=1, bk=1, bn=K, cm=O, cn=1, alpha=1.0,
                               cblk=O, cblk_m=0)
        for k in range(nb): d.ctab[k] = offs[k]
        p = lambda t: C.c_void_p(t.data_ptr())
        assert lib.genie_train_gemm(None, C.byref(d), p(x.cuda()), p(w.cuda()), p(buf), p(bias.cuda()), None, None) == 0
        torch.cuda.synchronize()
        ref = x.double() @ w.double().t() + bias.double()
        scale = (x.double().abs() @ w.double().abs().t()).clamp_min(1e-30)
        for k in range(nb):
            got = buf[offs[k]:offs[k] + R * O].view(R, O).cpu().double()
            assert ((got - ref[:, k * O:(k + 1) * O]).abs() / scale[:, k * O:(k + 1) * O]).max().item() <= TOL[3]
    else:
        R, K, O = (8192, 128, 128) if case == 'wgrad_640' else (200, 24, 24)
        dy = torch.randn(R, nb * O, generator=g); x = torch.randn(R, K, generator=g)
        blob = torch.zeros(nb * (O * K + O) + 64, device='cuda')             # [w0 | b0 | w1 | b1 | ...] like a state_dict blob
        woff = [k * (O * K + O) for k in range(nb)]; boff = [k * (O * K + O) + O * K for k in range(nb)]
        nsplit = 16 if case == 'wgrad_640' else 2
        d = capi.GenieGemmDesc(M=nb * O, N=K, K=R, batch=1, nb2=1, nsplit=nsplit, mode=2, terms=3, relu=0, am=1, ak=nb * O, bk=K, bn=1, cm=K, cn=1,
                               alpha=1.0, cblk=O, cblk_m=1)
        for k in range(nb): d.ctab[k] = woff[k]; d.atab[k] = boff[k]
        p = lambda t: C.c_void_p(t.data_ptr())
        assert lib.genie_train_gemm(None, C.byref(d), p(dy.cuda()), p(x.cuda()), p(blob), None, None, p(blob)) == 0
        torch.cuda.synchronize()
        ref = dy.double().t() @ x.double()
        scale = (dy.double().abs().t() @ x.double().abs()).clamp_min(1e-30)
        for k in range(nb):
            got = blob[woff[k]:woff[k] + O * K].view(O, K).cpu().double()
            assert ((got - ref[k * O:(k + 1) * O]).abs() / scale[k * O:(k + 1) * O]).max().item() <= 4 * TOL[3]
            gb = blob[boff[k]:boff[k] + O].cpu().double()
            rb = dy.double()[:, k * O:(k + 1) * O].sum(0)
            assert ((gb - rb).abs() / dy.double()[:, k * O:(k + 1) * O].abs().sum(0)).max().item() <= 1e-5

