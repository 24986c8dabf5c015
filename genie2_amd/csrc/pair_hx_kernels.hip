// Pair-stack GEMM kernels in "hx" arithmetic (hx.h): the WL structure of pair_wl_kernels.hip
// (a wave owns a 32-pair tile in fragment registers, weights arrive through LDS-DMA stages, the
// work-groups are persistent) with every f32 GEMM carried by three f16 MFMAs on split operands.
// 16x less matrix-pipe time per f32 FLOP than v_mfma_f32_32x32x2_f32 x 3 products = 5.3x, and the
// VALU work (LayerNorm, splits, gates) now overlaps the MFMAs instead of competing for the FP32 lanes.
#include "hx_pair.h"


// ---------------------------------------------------------------------------------------------
// Pair transition + end-of-layer mask (modules/pair_transition.py:48-56, pair_transform_net.py:116-117):
//   z = (z + W2 relu(W1 LN(z) + b1) + b2) * mask.
// Stage = hidden block of 32: units 0..7 = W1 k-chunks (A operand: rows = hidden units),
// units 8..15 = W2 (k-chunk c of the block, output block ob) at 8 + 4c + ob (B operand, chained-k
// permutation).  D'[hidden][pair] = Sx S1 (W1 zn + b1) accumulates from the scaled bias; relu and
// the rescale c1 = Sh / (Sx S1) give the second GEMM's A operand; out = acc * c2 (c2 = 1 / (Sh S2)).
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_pair_transition_hx(
    float* __restrict__ z, const float* __restrict__ rmask, const unsigned char* __restrict__ wimg,
    const float* __restrict__ b1s, const float* __restrict__ b2s, int N, long long M, int n_hb, float sx, float c1, float c2, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    float* sb1 = reinterpret_cast<float*>(smb + 2 * HX_STAGE_BYTES);
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int n_wt = (int)((M + 31) / 32);
    const int n_tiles = (n_wt + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, (unsigned)(n_hb * HX_STAGE_BYTES));
    const rsrc_t rz = hx_rsrc(z, (unsigned)(M * 512));               // rows >= M fall off the end: loads give 0, stores drop
    const rsrc_t rnull = hx_rsrc(z, 0u);
    const int lane16 = lane * 16;
    constexpr int PW = 32 / NW;                   // 1-KiB pieces of a stage per wave
    auto issue = [&](int hb, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, hb * HX_STAGE_BYTES + p * 1024);
        }
    };
    unsigned char* zt = smb + 2 * HX_STAGE_BYTES + 2048 + wave * HX_ZT_BYTES;
    auto tile_geom = [&](int tile, int& soff, int& nv) {      // wave-tile NW tile + wave (clamped): first row (bytes), valid rows
        const long long r0 = (long long)min(HX_PHYS(tile) * NW + wave, n_wt - 1) * 32;
        nv = (int)min((long long)32, M - r0);
        soff = (int)(r0 * 512);
    };
    // next tile's rows: half 0 requested in stage hq0, moved to registers (and half 1 requested) in stage hq1, half 1 read at
    // the tile boundary; with fewer than 2 stages per tile the tile is loaded synchronously instead
    const bool prefetch = n_hb >= 2;
    const int hq0 = max(0, n_hb / 2 - 2), hq1 = min(n_hb - 1, hq0 + 4);
    int tile = blockIdx.x;
    HX_TS_DECL(2);
    issue(0, 0);
    for (int u = threadIdx.x; u < n_hb * 32; u += NW * 64) sb1[u] = b1s[u];
    const float cb0 = b2s[pl], cb1 = b2s[32 + pl], cb2 = b2s[64 + pl], cb3 = b2s[96 + pl];
    float4 raw[16];
    int n_soff, n_nv;
    auto load_sync = [&](int t) {
        tile_geom(t, n_soff, n_nv);
        hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 0); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 0); hx_lds_done();
        hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 1); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 1);
    };
    load_sync(tile);
    __syncthreads();
    h8 zh[8], zl[8];
    float zres[4][16];
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
        const bool act = wt_raw < n_wt;
        const long long row0 = (long long)(act ? wt_raw : n_wt - 1) * 32;
        const int nrows = (int)min((long long)32, M - row0);
        const bool more = tile + (int)gridDim.x < n_tiles;
        float m_own = 0.f;
        if (pl < nrows) {
            const long long idx = row0 + pl;
            const int bb = (int)(idx / ((long long)N * N));
            const int rem = (int)(idx - (long long)bb * N * N);
            m_own = rmask[bb * N + rem / N] * rmask[bb * N + rem % N];
        }
        hx_norm_split(zh, zl, raw, sx);
        f32x16 o[4];
        {
            float e0 = cb0, e1 = cb1, e2 = cb2, e3 = cb3;
            asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));   // keep hipcc from hoisting (and spilling) the splats
#pragma unroll
            for (int r = 0; r < 16; ++r) { o[0][r] = e0; o[1][r] = e1; o[2][r] = e2; o[3][r] = e3; }
        }
        const rsrc_t rzz = act ? rz : rnull;
        const int voff = (4 * h * 128 + pl) * 4;
        const int srow0 = (int)(row0 * 512);
        auto stage_body = [&](int hb, auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            HX_TS();
            if (more && prefetch) {
                if (hb == hq0) { tile_geom(tile + gridDim.x, n_soff, n_nv); hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 0); }
                if (hb == hq1) { hx_zt_read(raw, zt, pl, h, 0); hx_lds_done(); hx_zt_dma(rz, zt, lane, n_soff, 512, n_nv, 1); }
            }
            if (!LAST) issue(hb + 1, (hb + 1) & 1);
            else if (more) issue(0, 0);
            HX_TS2();
            const unsigned char* stage = smb + (hb & 1) * HX_STAGE_BYTES;
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = sb1[hb * 32 + acc_row(r, lane)];
            HX_TS2();
            {
                h8 wh = hx_frag(stage, 0, 0, lane), wl = hx_frag(stage, 0, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const h8 nh = hx_frag(stage, min(kc + 1, 7), 0, lane), nl = hx_frag(stage, min(kc + 1, 7), 1, lane);
                    PIPE_FENCE();
                    MFH3(wh, wl, zh[kc], zl[kc], d);
                    PIPE_FENCE();
                    if (kc == 0 || kc == 3) HX_TS2();
                    wh = nh; wl = nl;
                }
            }
            HX_TS();
            if (LAST) {     // zh / zl are dead from here on: the first half of the residual rows is fetched while the last GEMM runs
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int so = srow0 + ((r & 3) + 8 * (r >> 2)) * 512;
                    zres[0][r] = hx_load(rzz, voff, so); zres[1][r] = hx_load(rzz, voff, so + 128);
                    zres[2][r] = hx_load(rzz, voff, so + 256); zres[3][r] = hx_load(rzz, voff, so + 384);
                }
            }
            h8 ah[2], al[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = fmaxf(d[8 * c + e], 0.f);
                hx_split8(x, c1, ah[c], al[c]);
            }
            {
                h8 bh = hx_frag(stage, 8, 0, lane), bl = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const h8 nh = hx_frag(stage, 8 + min(u + 1, 7), 0, lane), nl = hx_frag(stage, 8 + min(u + 1, 7), 1, lane);
                    PIPE_FENCE();
                    MFH3(ah[u >> 2], al[u >> 2], bh, bl, o[u & 3]);
                    PIPE_FENCE();
                    bh = nh; bl = nl;
                }
            }
            HX_TS();
            __syncthreads();
            HX_TS();
        };
#pragma unroll 1
        for (int hb = 0; hb + 1 < n_hb; ++hb) stage_body(hb, std::false_type{});
        stage_body(n_hb - 1, std::true_type{});
        // epilogue: row t = acc_row(r, lane) of the tile, channel 32 ob + pl; second half of the residual rows requested first
#pragma unroll
        for (int r = 8; r < 16; ++r) {
            const int so = srow0 + ((r & 3) + 8 * (r >> 2)) * 512;
            zres[0][r] = hx_load(rzz, voff, so); zres[1][r] = hx_load(rzz, voff, so + 128);
            zres[2][r] = hx_load(rzz, voff, so + 256); zres[3][r] = hx_load(rzz, voff, so + 384);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rc = (r & 3) + 8 * (r >> 2);
            const float m = __shfl(m_own, rc + 4 * h);
            const int so = srow0 + rc * 512;
            const float v0 = o[0][r], v1 = o[1][r], v2 = o[2][r], v3 = o[3][r];
            hx_store(rzz, fmaf(v0, c2, zres[0][r]) * m, voff, so);
            hx_store(rzz, fmaf(v1, c2, zres[1][r]) * m, voff, so + 128);
            hx_store(rzz, fmaf(v2, c2, zres[2][r]) * m, voff, so + 256);
            hx_store(rzz, fmaf(v3, c2, zres[3][r]) * m, voff, so + 384);
        }
        if (more) {
            if (prefetch) hx_zt_read(raw, zt, pl, h, 1);
            else load_sync(tile + gridDim.x);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, projections (modules/triangular_multiplicative_update.py:93-103):
//   a = (W_ap zn + b) sigmoid(W_ag zn + b) mask,  b likewise;  zn = LN_in(z).
// Stage = pass: units 2kc = the pass's 32 projection rows, 2kc + 1 = its 32 gate rows (pre-scaled by
// -log2 e), k-chunk kc.  D rows = channels, D cols = pairs, so every store is a 128-B run of the
// channel-major operand image the contraction reads; a and b are stored already SPLIT
// (hi | lo << 16 of a S_a), 4 bytes per element as before.
// ---------------------------------------------------------------------------------------------
template <bool OUTGOING, int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_trimul_proj_hx(
    const float* __restrict__ z, const float* __restrict__ rmask, const unsigned char* __restrict__ wimg,
    const float* __restrict__ bias, unsigned* __restrict__ acm, unsigned* __restrict__ bcm, int N, int NP, int n_wtiles,
    unsigned cm_bytes, unsigned z_bytes, float sx, float cpa, float cpb, float cg, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    float* sbias = reinterpret_cast<float*>(smb + 2 * HX_STAGE_BYTES);      // [512]
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, 8 * HX_STAGE_BYTES);
    const rsrc_t ra = hx_rsrc(acm, cm_bytes), rb = hx_rsrc(bcm, cm_bytes);
    const int lane16 = lane * 16;
    const int sstride = NP * NP * 4;                      // bytes per channel
    constexpr int PW = 32 / NW;
    auto issue = [&](int pass, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, pass * HX_STAGE_BYTES + p * 1024);
        }
    };
    unsigned char* zt = smb + 2 * HX_STAGE_BYTES + 2048 + wave * HX_ZT_BYTES;
    const rsrc_t rz = hx_rsrc(z, z_bytes);
    const int zstride = OUTGOING ? 512 : N * 512;         // bytes between consecutive pairs of a tile
    auto tile_geom = [&](int tile, int& soff, int& nv) {  // wave-tile NW tile + wave (clamped): byte offset of its first row, valid rows
        const int wt = min(HX_PHYS(tile) * NW + wave, n_wtiles - 1);
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        nv = min(32, N - st * 32);
        soff = OUTGOING ? ((b * N + line) * N + st * 32) * 512 : ((b * N + st * 32) * N + line) * 512;
    };
    int tile = blockIdx.x;
    HX_TS_DECL(OUTGOING ? 1 : 0);
    issue(0, 0);
    for (int u = threadIdx.x; u < 512; u += NW * 64) sbias[u] = bias[u];
    __syncthreads();
    unsigned wst[8];
    f32x16 apA, agA, apB, agB;                            // set A: even passes, set B: odd passes
#pragma unroll
    for (int r = 0; r < 16; ++r) { apB[r] = 0.f; agB[r] = 0.f; apA[r] = sbias[acc_row(r, lane)]; agA[r] = sbias[256 + acc_row(r, lane)]; }
    h8 zh[8], zl[8];
    float4 raw[16];
    int n_soff, n_nv;
    tile_geom(tile, n_soff, n_nv);
    hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 0); hx_lds_done();
    hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1); hx_vm_done(); hx_zt_read(raw, zt, pl, h, 1);
    int p_voff = 0x7FFFFFF0, p_so = 0;                    // epilogue state of the previous tile's pass 7 (none yet: stores dropped)
    float p_pm = 0.f;
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;      // idle waves shadow the last tile (stores dropped)
        const int st = wt % ntile, line = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0i = st * 32;
        const int nvalid = act ? min(32, N - t0i) : 0;
        const float msk = (pl < nvalid) ? rmask[b * N + line] * rmask[b * N + t0i + pl] : 0.f;
        const float ma = msk * cpa, mb = msk * cpb;
        // channel-major store: element ((b*128 + ch)*NP + line)*NP + t0 + pl, ch = 32 (pass & 3) + row
        const int voff = (pl < nvalid) ? (4 * h * NP * NP + pl) * 4 : 0x7FFFFFF0;   // out-of-range offset: store dropped
        const int sbase = ((b * 128 * NP + line) * NP + t0i) * 4;
        const bool more = tile + (int)gridDim.x < n_tiles;
        HX_TS();
        hx_norm_split(zh, zl, raw, sx);
#pragma unroll 1
        for (int pp = 0; pp < 4; ++pp) {
            {   // even pass 2pp -> set A; epilogue of pass 2pp - 1 (set B; for pp = 0 the previous tile's pass 7)
                const int pass = 2 * pp;
                HX_TS();
                if (more) {                               // next tile's rows: first half -> raw at pass 4 (requested at the ends of passes 0..2)
                    if (pp == 0) tile_geom(tile + gridDim.x, n_soff, n_nv);
                    if (pp == 2) { hx_zt_read(raw, zt, pl, h, 0); hx_lds_done(); }
                }
                issue(pass + 1, 1);
                const unsigned char* stage = smb;
                const rsrc_t e_rd = (pp == 0 || pp > 2) ? rb : ra;
                const int e_voff = pp == 0 ? p_voff : voff;
                const int e_so = pp == 0 ? p_so : sbase + ((pass - 1) & 3) * 32 * sstride;
                const float e_pm = pp == 0 ? p_pm : (pp > 2 ? mb : ma);
                HX_PROJ_STAGE(apA, agA, apB, agB);
                HX_TS();
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // weights of the next stage + the row-tile pieces requested a stage ago
                if (more) {                               // row-tile requests AFTER the stage's last store: passes 0, 2 -> half 0; 4, 6 -> half 1
                    if (pp == 0) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0, 0, 3);
                    else if (pp == 1) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0, 6, 2);
                    else if (pp == 2) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1, 0, 3);
                    else hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1, 6, 2);
                }
                HX_TS(); hx_stage_barrier();
                HX_TS();
            }
            {   // odd pass 2pp + 1 -> set B; epilogue of pass 2pp (set A)
                const int pass = 2 * pp + 1;
                HX_TS();
                if (pp < 3) issue(pass + 1, 0);
                else if (more) issue(0, 0);
                const unsigned char* stage = smb + HX_STAGE_BYTES;
                const rsrc_t e_rd = pp < 2 ? ra : rb;
                const int e_voff = voff;
                const int e_so = sbase + ((pass - 1) & 3) * 32 * sstride;
                const float e_pm = pp < 2 ? ma : mb;
                HX_PROJ_STAGE(apB, agB, apA, agA);
                HX_TS();
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                if (more) {                               // passes 1 -> half 0, 5 -> half 1 (none after passes 3 and 7: the halves are read next)
                    if (pp == 0) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 0, 3, 3);
                    else if (pp == 2) hx_zt_dma(rz, zt, lane, n_soff, zstride, n_nv, 1, 3, 3);
                }
                HX_TS(); hx_stage_barrier();
                HX_TS();
            }
        }
        p_voff = voff; p_so = sbase + 3 * 32 * sstride; p_pm = mb;
        if (more) hx_zt_read(raw, zt, pl, h, 1);          // second half: read at the tile boundary (register pressure)
    }
    {   // drain: epilogue of the last pass 7 (set B)
        const rsrc_t e_rd = rb;
        const int e_voff = p_voff, e_so = p_so;
        const float e_pm = p_pm;
        float t[16], u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            HX_PROJ_PIECE_A(agB, r, t[r]);
            HX_PROJ_PIECE_B(apB, r, t[r], u[r]);
        }
        PIPE_FENCE();           // (the packs are asm: keep them clear of the v_rcp results' forwarding window)
#pragma unroll
        for (int r = 0; r < 16; ++r) HX_PROJ_PIECE_C_NOW(r, t[r], u[r]);
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, contraction (triangular_multiplicative_update.py:57-82):
//   x_cm[bc][i][j] = sum_k a_cm[bc][i][k] b_cm[bc][j][k] / (S_a S_b),  operands stored split.
// WG tile (64 WT)^2, 4 waves of (32 WT)^2, K streamed 16 at a time: each thread loads 16 B (4 split
// values) per 64 rows, de-interleaves them with two v_perm (hi halves / lo halves) and writes the
// two 8-B pieces into the hi / lo planes of the double-buffered LDS stage (row stride 48 B: the
// ds_read_b128 fragment reads are conflict free).  HBM-bound in this arithmetic (36 FLOP/B).
// Persistent + XCD-aware tile order as in pair_kernels.hip.
// ---------------------------------------------------------------------------------------------
#define CX_ROWB 48
template <int WT>
__global__ __launch_bounds__(256, 2) void k_trimul_contract_hx(const unsigned* __restrict__ acm, const unsigned* __restrict__ bcm,
                                                               float* __restrict__ xcm, int NP, int n_mat, unsigned cm_bytes, float cx, int rev) {
    constexpr int TM = 64 * WT;
    constexpr int PLANE = TM * CX_ROWB;                 // bytes of one (operand, half) plane
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];   // [2 buf][A hi | A lo | B hi | B lo]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = UNI(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles = (NP + TM - 1) / TM;
    const int T = tiles * tiles;
    const int n_tiles = T * ((n_mat + 7) / 8) * 8;
    const int nk = NP / 16;
    const rsrc_t ra = hx_rsrc(acm, cm_bytes), rb = hx_rsrc(bcm, cm_bytes), rx = hx_rsrc(xcm, cm_bytes);
    const int lr = tid >> 2, c4 = tid & 3;
    const int vload = (lr * NP + c4 * 4) * 4;           // per-lane byte offset inside a 64-row block of a panel
    const int slds = lr * CX_ROWB + c4 * 8;
    u32x4 rA[WT], rB[WT];

    auto decode = [&](int wl, int& mi, int& i0, int& j0) {      // tile id -> (matrix, tile origin); `rev`: last matrices first
        const int w = rev ? n_tiles - 1 - wl : wl;
        const int grp = w / (8 * T), rem = w % (8 * T);
        const int tile = rem >> 3;
        mi = grp * 8 + (rem & 7);
        i0 = (tile / tiles) * TM;
        j0 = (tile % tiles) * TM;
    };
    auto gload = [&](int w, int kc) {
        int mi, i0, j0;
        decode(w, mi, i0, j0);
        const int mclamp = min(mi, n_mat - 1);
        const int mbase = mclamp * NP * NP * 4 + kc * 64;
#pragma unroll
        for (int u = 0; u < WT; ++u) {     // rows past NP: voffset pushed out of the buffer -> the load returns 0
            const int va = (i0 + 64 * u + lr < NP) ? vload : 0x7FFFFFF0, vb = (j0 + 64 * u + lr < NP) ? vload : 0x7FFFFFF0;
            rA[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, va, mbase + (i0 + 64 * u) * NP * 4, 0));
            rB[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, vb, mbase + (j0 + 64 * u) * NP * 4, 0));
        }
    };
    auto swrite = [&](int buf) {
        unsigned char* sa = smb + buf * 4 * PLANE + slds;
#pragma unroll
        for (int u = 0; u < WT; ++u) {
            uint2 ahi, alo, bhi, blo;
            ahi.x = __builtin_amdgcn_perm(rA[u].y, rA[u].x, 0x05040100u); ahi.y = __builtin_amdgcn_perm(rA[u].w, rA[u].z, 0x05040100u);
            alo.x = __builtin_amdgcn_perm(rA[u].y, rA[u].x, 0x07060302u); alo.y = __builtin_amdgcn_perm(rA[u].w, rA[u].z, 0x07060302u);
            bhi.x = __builtin_amdgcn_perm(rB[u].y, rB[u].x, 0x05040100u); bhi.y = __builtin_amdgcn_perm(rB[u].w, rB[u].z, 0x05040100u);
            blo.x = __builtin_amdgcn_perm(rB[u].y, rB[u].x, 0x07060302u); blo.y = __builtin_amdgcn_perm(rB[u].w, rB[u].z, 0x07060302u);
            *reinterpret_cast<uint2*>(sa + 64 * u * CX_ROWB) = ahi;
            *reinterpret_cast<uint2*>(sa + PLANE + 64 * u * CX_ROWB) = alo;
            *reinterpret_cast<uint2*>(sa + 2 * PLANE + 64 * u * CX_ROWB) = bhi;
            *reinterpret_cast<uint2*>(sa + 3 * PLANE + 64 * u * CX_ROWB) = blo;
        }
    };

    f32x16 acc[WT][WT];
#pragma unroll
    for (int m = 0; m < WT; ++m)
#pragma unroll
        for (int n = 0; n < WT; ++n) acc[m][n] = zero16();

    int w = blockIdx.x;
    if (w >= n_tiles) return;
    const int my_tiles = (n_tiles - 1 - w) / gridDim.x + 1;
    const int total = my_tiles * nk;
    gload(w, 0);
    swrite(0);
    __syncthreads();
    int kc = 0;
    const int foff = (lane & 31) * CX_ROWB + (lane >> 5) * 16;
    for (int it = 0; it < total; ++it) {
        const bool last_chunk = kc == nk - 1;
        const int wn_next = last_chunk ? w + gridDim.x : w;
        const int kc_next = last_chunk ? 0 : kc + 1;
        if (it + 1 < total) gload(wn_next, kc_next);
        const unsigned char* sa = smb + (it & 1) * 4 * PLANE + foff;
        h8 ah[WT], al[WT], bh[WT], bl[WT];
#pragma unroll
        for (int m = 0; m < WT; ++m) {
            ah[m] = *reinterpret_cast<const h8*>(sa + (wm * WT + m) * 32 * CX_ROWB);
            al[m] = *reinterpret_cast<const h8*>(sa + PLANE + (wm * WT + m) * 32 * CX_ROWB);
        }
#pragma unroll
        for (int n = 0; n < WT; ++n) {
            bh[n] = *reinterpret_cast<const h8*>(sa + 2 * PLANE + (wn * WT + n) * 32 * CX_ROWB);
            bl[n] = *reinterpret_cast<const h8*>(sa + 3 * PLANE + (wn * WT + n) * 32 * CX_ROWB);
        }
#pragma unroll
        for (int m = 0; m < WT; ++m)
#pragma unroll
            for (int n = 0; n < WT; ++n) MFH3(ah[m], al[m], bh[n], bl[n], acc[m][n]);
        if (last_chunk) {       // store the finished tile (row = acc_row(r): 4 (lane>>5) in voffset, the rest scalar)
            int mi, i0, j0;
            decode(w, mi, i0, j0);
            if (mi < n_mat) {
                const int vst = ((4 * (lane >> 5)) * NP + (lane & 31)) * 4;
#pragma unroll
                for (int m = 0; m < WT; ++m)
#pragma unroll
                    for (int n = 0; n < WT; ++n) {
                        const int ib = i0 + (wm * WT + m) * 32, jb = j0 + (wn * WT + n) * 32;
                        if (ib < NP && jb < NP) {
                            const int sbase = ((mi * NP + ib) * NP + jb) * 4;
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const float v = acc[m][n][r] * cx;     // (bit_cast straight from a vector element picks element 0)
                                hx_store(rx, v, vst, sbase + ((r & 3) + 8 * (r >> 2)) * NP * 4);
                            }
                        }
                        acc[m][n] = zero16();
                    }
            }
        }
        if (it + 1 < total) swrite((it + 1) & 1);
        __syncthreads();
        w = wn_next;
        kc = kc_next;
    }
}

// The same contraction with ONE 512-thread work-group per matrix (b, channel) owning the whole NP x NP output (NP <= 256): every
// operand byte leaves HBM exactly once (the 128^2 tiling above fetches each panel twice: 0.96 GB measured against 0.81 algorithmic).
// Waves 4 (rows) x 2 (columns), wave tile 64 x 128 = 8 accumulator blocks (128 registers); K in chunks of 32: each thread stages
// 4 + 4 b128 loads (whole 128-B lines of a row: 8 lanes x 16 B) in registers while the previous chunk is multiplied, de-interleaves
// them with v_perm and writes hi / lo planes at row stride 80 B (conflict-free ds_read_b128 fragments); two barriers per 32-k chunk,
// 48 MFMAs per wave between them.  Persistent over matrices, next matrix's first chunk requested before the epilogue's stores.
#define CB_ROWB 80
#define CB_PLANE (256 * CB_ROWB)
__global__ __launch_bounds__(512, 1) void k_trimul_contract_hx_big(const unsigned* __restrict__ acm, const unsigned* __restrict__ bcm,
                                                                   float* __restrict__ xcm, int NP, int n_mat, unsigned cm_bytes, float cx, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];   // [A hi | A lo | B hi | B lo], 256 rows x 80 B each
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = UNI(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;              // rows 64 wm .., columns 128 wn ..
    const int nk = NP / 32;                               // (NP is a multiple of 32)
    const rsrc_t ra = hx_rsrc(acm, cm_bytes), rb = hx_rsrc(bcm, cm_bytes), rx = hx_rsrc(xcm, cm_bytes);
    // row within a 64-row block, 16-B piece of the row's 128-B chunk.  Consecutive 8-lane groups take rows r and r + 4 (not r + 1):
    // the two rows a 16-lane ds_write_b64 group covers are then 320 B apart = 16 banks of the 32 a store sees, so they do not collide
    const int gq = tid >> 3, pc = tid & 7;
    const int lr = (gq & 0x38) | ((gq & 1) << 2) | ((gq >> 1) & 3);
    u32x4 rA[2][4], rB[2][4];       // two chunks in flight: chunk it + 2 is requested while chunk it is multiplied
    auto mat_of = [&](int w) { return rev ? n_mat - 1 - w : w; };
    auto gload = [&](int w, int kc, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        const int mbase = mat_of(w) * NP * NP * 4 + kc * 128;
#pragma unroll
        for (int u = 0; u < 4; ++u) {     // rows past NP: voffset pushed out of the buffer -> zeros
            const int row = 64 * u + lr;
            const int vo = row < NP ? (row * NP * 4 + pc * 16) : 0x7FFFFFF0;
            rA[S][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, vo, mbase, 0));
            rB[S][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, vo, mbase, 0));
        }
    };
    auto swrite = [&](auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        unsigned char* sa = smb + lr * CB_ROWB + pc * 8;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint2 ahi, alo, bhi, blo;
            ahi.x = __builtin_amdgcn_perm(rA[S][u].y, rA[S][u].x, 0x05040100u); ahi.y = __builtin_amdgcn_perm(rA[S][u].w, rA[S][u].z, 0x05040100u);
            alo.x = __builtin_amdgcn_perm(rA[S][u].y, rA[S][u].x, 0x07060302u); alo.y = __builtin_amdgcn_perm(rA[S][u].w, rA[S][u].z, 0x07060302u);
            bhi.x = __builtin_amdgcn_perm(rB[S][u].y, rB[S][u].x, 0x05040100u); bhi.y = __builtin_amdgcn_perm(rB[S][u].w, rB[S][u].z, 0x05040100u);
            blo.x = __builtin_amdgcn_perm(rB[S][u].y, rB[S][u].x, 0x07060302u); blo.y = __builtin_amdgcn_perm(rB[S][u].w, rB[S][u].z, 0x07060302u);
            *reinterpret_cast<uint2*>(sa + 64 * u * CB_ROWB) = ahi;
            *reinterpret_cast<uint2*>(sa + CB_PLANE + 64 * u * CB_ROWB) = alo;
            *reinterpret_cast<uint2*>(sa + 2 * CB_PLANE + 64 * u * CB_ROWB) = bhi;
            *reinterpret_cast<uint2*>(sa + 3 * CB_PLANE + 64 * u * CB_ROWB) = blo;
        }
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = zero16();
    int w = blockIdx.x;
    if (w >= n_mat) return;
    const int my = (n_mat - 1 - w) / gridDim.x + 1;
    const int total = my * nk;
    // position (matrix, chunk) of the loads issued last: two chunks ahead of the one being multiplied
    int lw = w, lkc = 0, issued = 0;
    auto advance = [&]() { if (lkc == nk - 1) { lw += gridDim.x; lkc = 0; } else ++lkc; };
    gload(lw, lkc, std::integral_constant<int, 0>{}); advance(); ++issued;
    if (issued < total) { gload(lw, lkc, std::integral_constant<int, 1>{}); advance(); ++issued; }
    int kc = 0;
    const int foff = (lane & 31) * CB_ROWB + (lane >> 5) * 16;
    const bool wave_live = 64 * wm < NP && 128 * wn < NP;          // waves whose tile lies past NP only help loading
    auto body = [&](int it, auto set_tag) {
        const bool last_chunk = kc == nk - 1;
        swrite(set_tag);
        __syncthreads();
        if (issued < total) { gload(lw, lkc, set_tag); advance(); ++issued; }
        if (wave_live) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const unsigned char* sa = smb + foff + c * 32;
                h8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ah[m] = *reinterpret_cast<const h8*>(sa + (wm * 2 + m) * 32 * CB_ROWB);
                    al[m] = *reinterpret_cast<const h8*>(sa + CB_PLANE + (wm * 2 + m) * 32 * CB_ROWB);
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    bh[n] = *reinterpret_cast<const h8*>(sa + 2 * CB_PLANE + (wn * 4 + n) * 32 * CB_ROWB);
                    bl[n] = *reinterpret_cast<const h8*>(sa + 3 * CB_PLANE + (wn * 4 + n) * 32 * CB_ROWB);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) MFH3(ah[m], al[m], bh[n], bl[n], acc[m][n]);
            }
        }
        if (last_chunk) {
            const int mi = mat_of(w);
            const int vst = ((4 * (lane >> 5)) * NP + (lane & 31)) * 4;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int ib = (wm * 2 + m) * 32, jb = (wn * 4 + n) * 32;
                    if (ib < NP && jb < NP) {
                        const int sbase = ((mi * NP + ib) * NP + jb) * 4;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = acc[m][n][r] * cx;
#if HX_ABL & 512      // developer knock-out (timing only): one store in sixteen
                            if (r == 0)
#endif
                            hx_store(rx, v, vst, sbase + ((r & 3) + 8 * (r >> 2)) * NP * 4);
                        }
                    }
                    acc[m][n] = zero16();
                }
            w += gridDim.x;
            kc = 0;
        } else
            ++kc;
        __syncthreads();
    };
    for (int it = 0; it < total; it += 2) {
        body(it, std::integral_constant<int, 0>{});
        if (it + 1 < total) body(it + 1, std::integral_constant<int, 1>{});
    }
}

// ---------------------------------------------------------------------------------------------
// Triangle multiplication, output (modules/triangular_multiplicative_update.py:105-108 + residual):
//   z += (W_z LN_out(x) + b_z) * sigmoid(W_g LN_in(z) + b_g).
// A operand = the pair tile (zn, then xn), B = weights; stages W_g{0,1}, W_z{0,1}, W_g{2,3}, W_z{2,3},
// unit = 8 (output block within the stage) + k-chunk.  x arrives channel-major: lane (p, h) reads
// x[c][p] for its 64 channels c = 16kc + 8h + e (each load = two 128-B runs).
// ---------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_trimul_out_hx(
    float* __restrict__ z, const float* __restrict__ xcm, const unsigned char* __restrict__ wimg,
    const float* __restrict__ bgs, const float* __restrict__ bzs, int N, int NP, int n_wtiles, unsigned cm_bytes,
    unsigned z_bytes, float sx, float cg, float cz, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, 4 * HX_STAGE_BYTES);
    const rsrc_t rx = hx_rsrc(xcm, cm_bytes), rz = hx_rsrc(z, z_bytes);
    const int lane16 = lane * 16;
    constexpr int PW = 32 / NW;
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, s * HX_STAGE_BYTES + p * 1024);
        }
    };
    unsigned char* zt = smb + 2 * HX_STAGE_BYTES + 2048 + wave * HX_ZT_BYTES;
    int tile = blockIdx.x;
    issue(0, 0);
    __syncthreads();
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;
        const int st = wt % ntile, i = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0 = st * 32;
        const int nvalid = act ? min(32, N - t0) : 0;
        const int prow0 = (b * N + i) * N + t0;                  // first pair row of the tile
        const bool more = tile + (int)gridDim.x < n_tiles;
        h8 zh[8], zl[8], xh[8], xl[8];
        {   // x_cm[((b*128 + c)*NP + i)*NP + t0 + pl], c = 16kc + 8h + e   (t0 + pl < NP always); the z rows come through the
            // coalesced LDS loader (two halves), the second half's latency is spent on LayerNorm + split of x
            float4 raw[16], rawz[16];
            const int cs = NP * NP * 4;
            const int vx = (8 * h * NP * NP + pl) * 4;
            const int sxo = ((b * 128 * NP + i) * NP + t0) * 4;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                raw[2 * kc].x = hx_load(rx, vx, sxo + (16 * kc + 0) * cs); raw[2 * kc].y = hx_load(rx, vx, sxo + (16 * kc + 1) * cs);
                raw[2 * kc].z = hx_load(rx, vx, sxo + (16 * kc + 2) * cs); raw[2 * kc].w = hx_load(rx, vx, sxo + (16 * kc + 3) * cs);
                raw[2 * kc + 1].x = hx_load(rx, vx, sxo + (16 * kc + 4) * cs); raw[2 * kc + 1].y = hx_load(rx, vx, sxo + (16 * kc + 5) * cs);
                raw[2 * kc + 1].z = hx_load(rx, vx, sxo + (16 * kc + 6) * cs); raw[2 * kc + 1].w = hx_load(rx, vx, sxo + (16 * kc + 7) * cs);
            }
            const int zsoff = prow0 * 512, znv = min(32, N - t0);
            hx_zt_dma(rz, zt, lane, zsoff, 512, znv, 0); hx_vm_done(); hx_zt_read(rawz, zt, pl, h, 0); hx_lds_done();
            hx_zt_dma(rz, zt, lane, zsoff, 512, znv, 1);
            hx_norm_split(xh, xl, raw, sx);
            hx_vm_done(); hx_zt_read(rawz, zt, pl, h, 1);
            hx_norm_split(zh, zl, rawz, sx);
        }
        const int voff = (4 * h * 128 + pl) * 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 ga, gb;
            {   // gate: A = zn fragments (i = pair), B = W_g (j = channel)
                issue(2 * half + 1, 1);
                const unsigned char* stage = smb;
                float c0 = bgs[(2 * half) * 32 + pl], c1 = bgs[(2 * half + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));      // keep hipcc from hoisting (and spilling) the 16-register splats
#pragma unroll
                for (int r = 0; r < 16; ++r) { ga[r] = c0; gb[r] = c1; }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(zh[kc], zl[kc], f0, f0l, ga);
                    MFH3(zh[kc], zl[kc], f1, f1l, gb);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ga[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ga[r] * cg));
                    gb[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb[r] * cg));
                }
                __syncthreads();
            }
            {   // update of channel blocks 2 half, 2 half + 1
                if (half == 0) issue(2, 0);
                else if (more) issue(0, 0);
                const unsigned char* stage = smb + HX_STAGE_BYTES;
                const int ob = 2 * half;
                f32x16 a0, a1;
                float c0 = bzs[ob * 32 + pl], c1 = bzs[(ob + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = c0; a1[r] = c1; }
                const int h4 = 4 * h;
                float zp0[16], zp1[16];
                if (half == 1) {    // zh / zl are dead after the second gate: the residual rows of this half are requested before its GEMM
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rc = (r & 3) + 8 * (r >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const int vr = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        zp0[r] = hx_load(rz, vr, so);
                        zp1[r] = hx_load(rz, vr, so + 128);
                    }
                }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(xh[kc], xl[kc], f0, f0l, a0);
                    MFH3(xh[kc], xl[kc], f1, f1l, a1);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
                hx_stage_landed();
                // residual + store, 8 rows (16 loads) in flight at a time; rows past the tile's valid pairs belong to
                // the next line: their offset is pushed out of the buffer (load gives 0, store is dropped)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float zr0[8], zr1[8];
                    int vo[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        vo[q] = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        if (half == 1) { zr0[q] = zp0[r0 + q]; zr1[q] = zp1[r0 + q]; }
                        else { zr0[q] = hx_load(rz, vo[q], so); zr1[q] = hx_load(rz, vo[q], so + 128); }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const float u0 = a0[r0 + q], u1 = a1[r0 + q], g0 = ga[r0 + q], g1 = gb[r0 + q];
                        hx_store(rz, fmaf(u0, g0, zr0[q]), vo[q], so);
                        hx_store(rz, fmaf(u1, g1, zr1[q]), vo[q], so + 128);
                    }
                    PIPE_FENCE();
                }
                hx_stage_barrier();
            }
        }
    }
}

// The same kernel with all four weight stages resident in LDS (128 KiB: no weight stream, no barrier inside the tile loop --
// the eight waves of a work-group run their tiles independently) and the z rows loaded straight into operand registers
// (no LDS left for the row-tile loader).
template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 1 : 2) void k_trimul_out_hx_r(
    float* __restrict__ z, const float* __restrict__ xcm, const unsigned char* __restrict__ wimg,
    const float* __restrict__ bgs, const float* __restrict__ bzs, int N, int NP, int n_wtiles, unsigned cm_bytes,
    unsigned z_bytes, float sx, float cg, float cz, int rev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    const int lane = threadIdx.x & 63, wave = UNI(threadIdx.x >> 6);
    const int h = lane >> 5, pl = lane & 31;
    const int ntile = (N + 31) >> 5;
    const int n_tiles = (n_wtiles + NW - 1) / NW;
    const rsrc_t rw = hx_rsrc(wimg, 4 * HX_STAGE_BYTES);
    const rsrc_t rx = hx_rsrc(xcm, cm_bytes), rz = hx_rsrc(z, z_bytes);
    const int lane16 = lane * 16;
    constexpr int PW = 32 / NW;
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int p = PW * wave + q;
            hx_dma(rw, smb + buf * HX_STAGE_BYTES + p * 1024, lane16, s * HX_STAGE_BYTES + p * 1024);
        }
    };
    int tile = blockIdx.x;
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);      // all four weight stages stay in LDS (128 KiB) for every tile
    hx_stage_landed();
    __syncthreads();
#pragma unroll 1
    for (; tile < n_tiles; tile += gridDim.x) {
        const int wt_raw = HX_PHYS(tile) * NW + wave;
        const bool act = wt_raw < n_wtiles;
        const int wt = act ? wt_raw : n_wtiles - 1;
        const int st = wt % ntile, i = (wt / ntile) % N, b = wt / (ntile * N);
        const int t0 = st * 32;
        const int nvalid = act ? min(32, N - t0) : 0;
        const int prow0 = (b * N + i) * N + t0;                  // first pair row of the tile
        h8 zh[8], zl[8], xh[8], xl[8];
        {   // x_cm[((b*128 + c)*NP + i)*NP + t0 + pl], c = 16kc + 8h + e   (t0 + pl < NP always); the z rows come through the
            // coalesced LDS loader (two halves), the second half's latency is spent on LayerNorm + split of x
            float4 raw[16], rawz[16];
            const int cs = NP * NP * 4;
            const int vx = (8 * h * NP * NP + pl) * 4;
            const int sxo = ((b * 128 * NP + i) * NP + t0) * 4;
#pragma unroll
            for (int kc = 0; kc < 8; ++kc) {
                raw[2 * kc].x = hx_load(rx, vx, sxo + (16 * kc + 0) * cs); raw[2 * kc].y = hx_load(rx, vx, sxo + (16 * kc + 1) * cs);
                raw[2 * kc].z = hx_load(rx, vx, sxo + (16 * kc + 2) * cs); raw[2 * kc].w = hx_load(rx, vx, sxo + (16 * kc + 3) * cs);
                raw[2 * kc + 1].x = hx_load(rx, vx, sxo + (16 * kc + 4) * cs); raw[2 * kc + 1].y = hx_load(rx, vx, sxo + (16 * kc + 5) * cs);
                raw[2 * kc + 1].z = hx_load(rx, vx, sxo + (16 * kc + 6) * cs); raw[2 * kc + 1].w = hx_load(rx, vx, sxo + (16 * kc + 7) * cs);
            }
            {   // this lane's pair row, channels 16 kc + 8 h .. + 7 (rows past the tile: clamped)
                const float* zrow = z + ((size_t)(prow0 + min(pl, min(32, N - t0) - 1))) * 128 + 8 * h;
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    rawz[2 * kc] = *reinterpret_cast<const float4*>(zrow + 16 * kc);
                    rawz[2 * kc + 1] = *reinterpret_cast<const float4*>(zrow + 16 * kc + 4);
                }
            }
            hx_norm_split(xh, xl, raw, sx);
            hx_norm_split(zh, zl, rawz, sx);
        }
        const int voff = (4 * h * 128 + pl) * 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 ga, gb;
            {   // gate: A = zn fragments (i = pair), B = W_g (j = channel)
                const unsigned char* stage = smb + (2 * half) * HX_STAGE_BYTES;
                float c0 = bgs[(2 * half) * 32 + pl], c1 = bgs[(2 * half + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));      // keep hipcc from hoisting (and spilling) the 16-register splats
#pragma unroll
                for (int r = 0; r < 16; ++r) { ga[r] = c0; gb[r] = c1; }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(zh[kc], zl[kc], f0, f0l, ga);
                    MFH3(zh[kc], zl[kc], f1, f1l, gb);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    ga[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(ga[r] * cg));
                    gb[r] = cz * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gb[r] * cg));
                }
            }
            {   // update of channel blocks 2 half, 2 half + 1
                const unsigned char* stage = smb + (2 * half + 1) * HX_STAGE_BYTES;
                const int ob = 2 * half;
                f32x16 a0, a1;
                float c0 = bzs[ob * 32 + pl], c1 = bzs[(ob + 1) * 32 + pl];
                asm volatile("" : "+v"(c0), "+v"(c1));
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = c0; a1[r] = c1; }
                const int h4 = 4 * h;
                float zp0[16], zp1[16];
                if (half == 1) {    // zh / zl are dead after the second gate: the residual rows of this half are requested before its GEMM
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rc = (r & 3) + 8 * (r >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const int vr = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        zp0[r] = hx_load(rz, vr, so);
                        zp1[r] = hx_load(rz, vr, so + 128);
                    }
                }
                h8 f0 = hx_frag(stage, 0, 0, lane), f0l = hx_frag(stage, 0, 1, lane), f1 = hx_frag(stage, 8, 0, lane),
                   f1l = hx_frag(stage, 8, 1, lane);
#pragma unroll
                for (int kc = 0; kc < 8; ++kc) {
                    const int kn = min(kc + 1, 7);
                    const h8 n0 = hx_frag(stage, kn, 0, lane), n0l = hx_frag(stage, kn, 1, lane), n1 = hx_frag(stage, 8 + kn, 0, lane),
                             n1l = hx_frag(stage, 8 + kn, 1, lane);
                    PIPE_FENCE();
                    MFH3(xh[kc], xl[kc], f0, f0l, a0);
                    MFH3(xh[kc], xl[kc], f1, f1l, a1);
                    PIPE_FENCE();
                    f0 = n0; f0l = n0l; f1 = n1; f1l = n1l;
                }
                // residual + store, 8 rows (16 loads) in flight at a time; rows past the tile's valid pairs belong to
                // the next line: their offset is pushed out of the buffer (load gives 0, store is dropped)
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += 8) {
                    float zr0[8], zr1[8];
                    int vo[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        vo[q] = (h4 < nvalid - rc) ? voff : 0x7FFFFFF0;
                        if (half == 1) { zr0[q] = zp0[r0 + q]; zr1[q] = zp1[r0 + q]; }
                        else { zr0[q] = hx_load(rz, vo[q], so); zr1[q] = hx_load(rz, vo[q], so + 128); }
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int rc = ((r0 + q) & 3) + 8 * ((r0 + q) >> 2);
                        const int so = (prow0 + rc) * 512 + ob * 128;
                        const float u0 = a0[r0 + q], u1 = a1[r0 + q], g0 = ga[r0 + q], g1 = gb[r0 + q];
                        hx_store(rz, fmaf(u0, g0, zr0[q]), vo[q], so);
                        hx_store(rz, fmaf(u1, g1, zr1[q]), vo[q], so + 128);
                    }
                    PIPE_FENCE();
                }
            }
        }
    }
}

// (Tried: the output kernel transposed, D[channel][pair] with A = weights, so that a lane holds its pair row on both sides of
//  the GEMMs -- four v_permlane32_swap per k-chunk turn the loaded z row into the accumulator's row set, which is then both the
//  residual and, with the gate weights packed in that k order, the gate's B operand; LayerNorm applied while splitting, twice
//  per operand, to stay inside 256 registers; one float4 store per lane and 4-channel group.  Parity green, no second read of
//  z (-268 MB of L2 reads per launch), but 0.254 ms against 0.237: the doubled split work and the row-per-lane 16-B stores
//  cost more than the re-read.  tools/probe/swap_probe.hip is the check of the swap's semantics.)
// ---------------------------------------------------------------------------------------------
static int g_hx_cu = 0;
static int hx_num_cu() {
    if (!g_hx_cu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); (void)hipGetDeviceProperties(&pr, dev); g_hx_cu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    return g_hx_cu;
}
static int hx_nw() { return 8; }   // waves per work-group (one work-group per CU: 130 KiB of LDS)
// `cus`: CUs this launch may fill (all of them, or half when two halves of the batch run side by side)
static unsigned hx_grid(long long n_tiles, int nw, int cus) {
    const long long cap = (nw == 8 ? 1LL : 2LL) * cus;
    return (unsigned)(n_tiles < cap ? n_tiles : cap);
}

// A contiguous slice of the batch (structures b0 .. b0 + nb - 1) on its own stream: every pair-stack kernel is separable over
// the batch (outermost dimension of p, a, b, x and of the mask), so a slice is a pointer offset and a smaller tile count.
struct HxSlice { int b0, nb; hipStream_t st; int cus; unsigned launches; bool prof; };

static void pair_transition_slice(genie_ctx* h, HxSlice& v, const PairLayerW& w) {
    const int N = h->N;
    const long long M = (long long)v.nb * N * N;
    const long long n_wt = (M + 31) / 32;
    const int n_hb = h->d.pair_transition_n * 4;
    const HxTransW& x = w.hx_pt;
    hipLaunchKernelGGL(k_pair_transition_hx<8>, dim3(hx_grid((n_wt + 7) / 8, 8, v.cus)), dim3(512), HX_LDS_BYTES, v.st,
                       h->p + (size_t)v.b0 * N * N * 128, h->rmaskf + (size_t)v.b0 * N, x.img, x.b1s, x.b2s, N, M, n_hb, x.sx, x.c1, x.c2,
                       (int)(v.launches++ & 1));
}

struct HxTriGeom {
    int N, NP, ntile, nw, n_wt;
    unsigned *acm, *bcm; float* xcm; float* zs; const float* ms;
    unsigned cm_bytes, z_bytes;
};
static HxTriGeom tri_geom(genie_ctx* h, const HxSlice& v) {
    HxTriGeom g;
    g.N = h->N; g.NP = h->NP; g.ntile = (g.N + 31) / 32; g.nw = hx_nw();
    const size_t cm_off = (size_t)v.b0 * 128 * g.NP * g.NP;
    g.acm = reinterpret_cast<unsigned*>(h->acm) + cm_off;
    g.bcm = reinterpret_cast<unsigned*>(h->bcm) + cm_off;
    g.xcm = h->xcm + cm_off;
    g.n_wt = v.nb * g.N * g.ntile;
    g.cm_bytes = (unsigned)((size_t)v.nb * 128 * g.NP * g.NP * 4);
    g.z_bytes = (unsigned)((size_t)v.nb * g.N * g.N * 512);
    g.zs = h->p + (size_t)v.b0 * g.N * g.N * 128;
    g.ms = h->rmaskf + (size_t)v.b0 * g.N;
    return g;
}

static void trimul_proj_slice(genie_ctx* h, HxSlice& v, const TriMulW& w, bool outgoing) {
    const HxTriGeom g = tri_geom(h, v);
    const HxTriW& x = w.hx;
    hipStream_t st = v.st;
    ProfScope ps(h, st, KC_TRIMUL_PROJ, v.prof);
    const dim3 grid(hx_grid((g.n_wt + g.nw - 1) / g.nw, g.nw, v.cus)), block(g.nw * 64);
    const int rev = (int)(v.launches++ & 1);
#define HX_PROJ(OUT, NWV) hipLaunchKernelGGL((k_trimul_proj_hx<OUT, NWV>), grid, block, HX_LDS_BYTES, st, g.zs, g.ms, x.img_proj, \
                                         x.bias_proj, g.acm, g.bcm, g.N, g.NP, g.n_wt, g.cm_bytes, g.z_bytes, x.sx, x.cpa, x.cpb, x.cg, rev)
    if (outgoing) HX_PROJ(true, 8);
    else HX_PROJ(false, 8);
#undef HX_PROJ
}

// x = a b^T per channel; `transposed`: the operands swapped, which leaves x^T (what the column-tile chain A reads)
static void trimul_contract_slice(genie_ctx* h, HxSlice& v, const TriMulW& w, bool transposed) {
    const HxTriGeom g = tri_geom(h, v);
    const HxTriW& x = w.hx;
    hipStream_t st = v.st;
    ProfScope ps(h, st, KC_TRIMUL_CONTRACT, v.prof);
    const int BC = v.nb * h->d.c_hidden_mul;
    const int ncu = v.cus;
    const int rev = (int)(v.launches++ & 1);
    const unsigned* pa = transposed ? g.bcm : g.acm;
    const unsigned* pb = transposed ? g.acm : g.bcm;
    if (g.NP > 192 && g.NP <= 256 && !getenv("GENIE_CONTRACT_TILED")) {
        hipLaunchKernelGGL(k_trimul_contract_hx_big, dim3(BC < ncu ? BC : ncu), dim3(512), 4 * CB_PLANE, st, pa, pb, g.xcm, g.NP, BC, g.cm_bytes,
                           x.cx, rev);
    } else if (g.NP >= 128) {
        const int tiles = (g.NP + 127) / 128;
        const int n_tiles = tiles * tiles * ((BC + 7) / 8) * 8;
        hipLaunchKernelGGL(k_trimul_contract_hx<2>, dim3(n_tiles < 3 * ncu ? n_tiles : 3 * ncu), dim3(256), 2 * 4 * 128 * CX_ROWB, st,
                           pa, pb, g.xcm, g.NP, BC, g.cm_bytes, x.cx, rev);
    } else {
        const int tiles = (g.NP + 63) / 64;
        const int n_tiles = tiles * tiles * ((BC + 7) / 8) * 8;
        hipLaunchKernelGGL(k_trimul_contract_hx<1>, dim3(n_tiles < 4 * ncu ? n_tiles : 4 * ncu), dim3(256), 2 * 4 * 64 * CX_ROWB, st,
                           pa, pb, g.xcm, g.NP, BC, g.cm_bytes, x.cx, rev);
    }
}

static void trimul_out_slice(genie_ctx* h, HxSlice& v, const TriMulW& w) {
    const HxTriGeom g = tri_geom(h, v);
    const HxTriW& x = w.hx;
    hipStream_t st = v.st;
    ProfScope ps(h, st, KC_TRIMUL_OUT, v.prof);
    const bool resident = getenv("GENIE_OUT_STREAMED") == nullptr;     // default: weights resident in LDS (0.233 vs 0.240 ms per launch)
    if (resident)
        hipLaunchKernelGGL(k_trimul_out_hx_r<8>, dim3(hx_grid((g.n_wt + 7) / 8, 8, v.cus)), dim3(512), 4 * HX_STAGE_BYTES, st, g.zs, g.xcm,
                           x.img_out, x.bgs, x.bzs, g.N, g.NP, g.n_wt, g.cm_bytes, g.z_bytes, x.sx, x.cgo, x.cz, (int)(v.launches++ & 1));
    else
        hipLaunchKernelGGL(k_trimul_out_hx<8>, dim3(hx_grid((g.n_wt + 7) / 8, 8, v.cus)), dim3(512), HX_LDS_BYTES, st, g.zs, g.xcm, x.img_out,
                           x.bgs, x.bzs, g.N, g.NP, g.n_wt, g.cm_bytes, g.z_bytes, x.sx, x.cgo, x.cz, (int)(v.launches++ & 1));
}

static void trimul_slice(genie_ctx* h, HxSlice& v, const TriMulW& w, bool outgoing) {
    trimul_proj_slice(h, v, w, outgoing);
    trimul_contract_slice(h, v, w, false);
    trimul_out_slice(h, v, w);
}

// The whole pair transform net with the row-local chains fused (pair_fused_kernels.hip):
//   proj(out_0) | per block: contract^T, chain A, contract, chain B (last block: output + transition)
// 4 launches per block instead of 7; z is read and written once per chain.  GENIE_NO_PAIR_FUSE=1 keeps the separate launches.
void launch_pair_fused(genie_ctx* h, hipStream_t st, const HxFusedW& f, const HxTriW& o, const HxTransW* t, const HxTriW* p, bool col);
bool launch_pair_stack_fused(genie_ctx* h, hipStream_t st, float* tap_trimul_out0, float* tap_layer0) {
    const int L = h->d.n_pair_transform_layer;
    if (!h->hx || L < 1 || getenv("GENIE_NO_PAIR_FUSE") || (getenv("GENIE_HX_SLICE") && atoi(getenv("GENIE_HX_SLICE")) > 0)) return false;
    if ((h->d.pair_transition_n * 4) & 1) return false;
    const size_t pbytes = (size_t)h->B * h->N * h->N * h->d.c_p * 4;
    HxSlice v{0, h->B, st, hx_num_cu(), h->hx_launches, h->prof};
    trimul_proj_slice(h, v, h->pair[0].out, true);
    for (int l = 0; l < L; ++l) {
        const PairLayerW& W = h->pair[l];
        trimul_contract_slice(h, v, W.out, true);
        h->hx_launches = v.launches;
        { ProfScope ps(h, st, KC_PAIR_FUSED_A); launch_pair_fused(h, st, W.fa, W.out.hx, nullptr, &W.in.hx, true); }
        v.launches = h->hx_launches;
        if (l == 0 && tap_trimul_out0) (void)hipMemcpyAsync(tap_trimul_out0, h->p, pbytes, hipMemcpyDeviceToDevice, st);
        trimul_contract_slice(h, v, W.in, false);
        if (l + 1 < L) {
            h->hx_launches = v.launches;
            { ProfScope ps(h, st, KC_PAIR_FUSED_B); launch_pair_fused(h, st, W.fb, W.in.hx, &W.hx_pt, &h->pair[l + 1].out.hx, false); }
            v.launches = h->hx_launches;
        } else {       // last block: the same chain without projections
            h->hx_launches = v.launches;
            { ProfScope ps(h, st, KC_PAIR_FUSED_B); launch_pair_fused(h, st, W.fb, W.in.hx, &W.hx_pt, nullptr, false); }
            v.launches = h->hx_launches;
        }
        if (l == 0 && tap_layer0) (void)hipMemcpyAsync(tap_layer0, h->p, pbytes, hipMemcpyDeviceToDevice, st);
    }
    h->hx_launches = v.launches;
    return true;
}

void launch_pair_transition_hx(genie_ctx* h, hipStream_t st, const PairLayerW& w) {
    HxSlice v{0, h->B, st, hx_num_cu(), h->hx_launches, h->prof};
    pair_transition_slice(h, v, w);
    h->hx_launches = v.launches;
}

// GENIE_HX_SLICE=n runs the three kernels of a triangle multiplication per slice of n structures, so that (n = 2, N = 256:
// a + b + x = 201 MB) the operands stay in the 256-MiB Infinity Cache between producer and consumer.  Measured: no gain
// (82.1 / 80.4 / 83.0 / 80.9 batch-steps/s for n = 8 / 4 / 2 / 1), so the default is the whole batch; kept as a switch for larger N.
static int g_hx_slice = 0;
static int hx_slice() {
    if (!g_hx_slice) { const char* e = getenv("GENIE_HX_SLICE"); g_hx_slice = e && atoi(e) > 0 ? atoi(e) : 1 << 20; }
    return g_hx_slice;
}

void launch_trimul_hx(genie_ctx* h, hipStream_t st, const TriMulW& w, bool outgoing) {
    const int SB = hx_slice();
    for (int b0 = 0; b0 < h->B; b0 += SB) {
        HxSlice v{b0, h->B - b0 < SB ? h->B - b0 : SB, st, hx_num_cu(), h->hx_launches, h->prof};
        trimul_slice(h, v, w, outgoing);
        h->hx_launches = v.launches;
    }
}

// (Tried on top of the slices: the whole pair transform net with the batch in two halves on two streams, each on half of the
//  CUs, the second half one or two kernels behind the first, so that a memory-bound TriMul kernel and the matrix-bound
//  transition run side by side.  89.6 - 94.2 batch-steps/s against 95.1: a kernel on half of the CUs takes twice as long,
//  whether it is bound by memory or by the matrix pipe.)

// ---------------------------------------------------------------------------------------------
// Pair bias of all IPA layers in one pass over p (invariant_point_attention.py:181), hx arithmetic:
//   out[o][b][i][j] = sum_c W[o][c] p[b][i][j][c] + bias[o],  o = layer * H + head  (L H = 96 rows, three 32-row blocks).
// The f32 form (pair_kernels.hip k_ipa_bias) staged p through LDS for 192 v_mfma_f32_32x32x2 per wave and tile.  Here a wave
// takes 32 consecutive j of one (b, i): lane n = j holds its own row -- 8 consecutive channels per k-step are exactly the B
// fragment of v_mfma_f32_32x32x16_f16 (k = 16 s + 8 (lane >> 5) + e), so p goes from HBM to the matrix pipe through
// registers only (a 128-B line is used up by four consecutive k-steps of the same wave), the block-floating-point scale of
// a row is an in-lane maximum plus one exchange with the lane holding the row's other channels, and it comes back out as a
// per-column (= per-lane) factor.  W (48 KiB as 24 hx units) sits in LDS.  By-product: max |p| for the attention kernel's split.
// ---------------------------------------------------------------------------------------------
#define IB_UNITS 24          // 3 row blocks x 8 k-steps
__global__ __launch_bounds__(256) void k_ipa_bias_hx(const float* __restrict__ z, const unsigned char* __restrict__ wimg, float inv_sw,
                                                     const float* __restrict__ bias, float* __restrict__ out, int B, int N, int LH,
                                                     int n_tiles, int rev, unsigned* pmax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ibw[];          // [IB_UNITS][hi 1 KiB | lo 1 KiB], then bias[96]
    float* sbias = reinterpret_cast<float*>(ibw + IB_UNITS * 2048);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int u = tid; u < IB_UNITS * 128; u += 256) reinterpret_cast<uint4*>(ibw)[u] = reinterpret_cast<const uint4*>(wimg)[u];
    if (tid < 96) sbias[tid] = tid < LH ? bias[tid] : 0.f;
    __syncthreads();
    const int n = lane & 31, hf = lane >> 5;
    const int jt = (N + 31) >> 5;                        // 32-wide j tiles per (b, i)
    float amax = 0.f;
    for (int t = blockIdx.x * 4 + wave; t < n_tiles; t += gridDim.x * 4) {
        const int tt = rev ? n_tiles - 1 - t : t;
        const int j0 = (tt % jt) * 32;
        const int bi = tt / jt;                          // b * N + i
        const int j = min(j0 + n, N - 1);
        const float* row = z + ((size_t)bi * N + j) * 128 + 8 * hf;
        float4 x[8][2];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            x[s][0] = *reinterpret_cast<const float4*>(row + 16 * s);
            x[s][1] = *reinterpret_cast<const float4*>(row + 16 * s + 4);
        }
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            m = fmaxf(m, fmaxf(fmaxf(fabsf(x[s][0].x), fabsf(x[s][0].y)), fmaxf(fabsf(x[s][0].z), fabsf(x[s][0].w))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(x[s][1].x), fabsf(x[s][1].y)), fmaxf(fabsf(x[s][1].z), fabsf(x[s][1].w))));
        }
        m = fmaxf(m, __shfl_xor(m, 32));                 // the row's other 64 channels
        amax = fmaxf(amax, m);
        const int e = (int)((__builtin_bit_cast(unsigned, m) >> 23) & 255u);
        const bool tiny = e < 16;
        const float sc = tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(268 - e) << 23);      // max * sc in [2^14, 2^15)
        const float isc = (tiny ? 1.0f : __builtin_bit_cast(float, (unsigned)(e - 14) << 23)) * inv_sw;
        f32x16 acc[3];
#pragma unroll
        for (int blk = 0; blk < 3; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float xs[8] = {x[s][0].x, x[s][0].y, x[s][0].z, x[s][0].w, x[s][1].x, x[s][1].y, x[s][1].z, x[s][1].w};
            h8 bh, bl;
            hx_split8(xs, sc, bh, bl);
#pragma unroll
            for (int blk = 0; blk < 3; ++blk) {
                const h8 ah = hx_frag(ibw, blk * 8 + s, 0, lane), al = hx_frag(ibw, blk * 8 + s, 1, lane);
                MFH3(ah, al, bh, bl, acc[blk]);
            }
        }
        if (j0 + n < N) {
            const int b = bi / N, i = bi - b * N;
#pragma unroll
            for (int blk = 0; blk < 3; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (o < LH) out[((((size_t)o * B + b) * N + i) * N) + j0 + n] = fmaf(acc[blk][r], isc, sbias[o]);
                }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    if (lane == 0 && __float_as_uint(amax) > *reinterpret_cast<volatile unsigned*>(pmax)) atomicMax(pmax, __float_as_uint(amax));
}

bool launch_ipa_bias_hx(genie_ctx* h, hipStream_t st) {
    const int LH = h->d.n_structure_layer * h->d.n_head_ipa;
    if (!h->hx || LH > 96 || h->d.c_p != 128 || getenv("GENIE_IPA_BIAS_F32")) return false;
    const HxGemmW* w = nullptr;
    for (int i = 0; i < h->n_hxg; ++i)
        if (h->hxg[i].w == h->ipa_bias_w) w = &h->hxg[i];
    if (!w) return false;
    const int N = h->N, n_tiles = h->B * N * ((N + 31) / 32);
    const int grid = n_tiles / 4 < 2 * hx_num_cu() ? (n_tiles + 3) / 4 : 2 * hx_num_cu();      // 191 registers: two work-groups per CU
    (void)hipMemsetAsync(h->pmax, 0, sizeof(unsigned), st);
    hipLaunchKernelGGL(k_ipa_bias_hx, dim3(grid), dim3(256), IB_UNITS * 2048 + 96 * 4, st, h->p, w->img, w->inv_s, h->ipa_bias_b, h->ipa_bias,
                       h->B, N, LH, n_tiles, (int)(h->hx_launches & 1), h->pmax);
    return true;
}

void pair_hx_kernels_init() {
#define HX_ATTR(k) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, HX_LDS_BYTES)
    HX_ATTR((k_trimul_proj_hx<true, 8>)); HX_ATTR((k_trimul_proj_hx<false, 8>));
    HX_ATTR(k_trimul_out_hx<8>); HX_ATTR(k_trimul_out_hx_r<8>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_trimul_contract_hx_big), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * CB_PLANE);
#undef HX_ATTR
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_transition_hx<8>), hipFuncAttributeMaxDynamicSharedMemorySize, HX_LDS_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_ipa_bias_hx), hipFuncAttributeMaxDynamicSharedMemorySize, IB_UNITS * 2048 + 96 * 4);
}
