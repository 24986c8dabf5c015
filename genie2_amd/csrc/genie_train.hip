// The training step's forward and backward pass through the Denoiser (genie/diffusion/genie.py:88-105 around
// genie/model/model.py:125-192), on the plain state_dict blob the caller owns (weights in, gradients out, both in
// Denoiser.state_dict() order -- the layout of genie_load_weights).  Train mode: the reference's four dropout sites
// (pair_transform_net.py:109-110 row-shared; structure_net.py:109, structure_transition.py:66 elementwise) with counter-based
// masks (train.h drop_scale).  Everything is enqueued on the caller's stream; activations live in a workspace owned by the handle.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "common.h"
#include "train.h"

namespace {
struct TriOff { size_t ap_w, ap_b, ag_w, ag_b, bp_w, bp_b, bg_w, bg_b, g_w, g_b, z_w, z_b, lni_g, lni_b, lno_g, lno_b; };
struct PairOff { TriOff out, in; size_t ln_g, ln_b, w1, b1, w2, b2; };
struct StructOff {
    size_t head, q_w, q_b, kv_w, kv_b, qp_w, qp_b, kvp_w, kvp_b, b_w, b_b, o_w, o_b, ln1_g, ln1_b, t1w, t1b, t2w, t2b, t3w, t3b, ln2_g, ln2_b,
        bb_w, bb_b;
};
struct Offs { size_t single_w, wi, wj, wrel, wt, wm; std::vector<PairOff> pair; std::vector<StructOff> st; size_t total; };

Offs make_offsets(const genie_dims_t& d) {
    Offs o;
    size_t c = 0;
    auto take = [&](size_t n) { const size_t r = c; c += n; return r; };
    const size_t cs = d.c_s, cp = d.c_p, ch = d.c_hidden_mul;
    const size_t nsi = d.c_pos_emb + d.c_chain_emb + d.c_timestep_emb + 23;
    o.single_w = take(cs * nsi);
    o.wi = take(cp * cs); o.wj = take(cp * cs);
    o.wrel = take(cp * (2 * d.relpos_k + 3));
    o.wt = take(cp * (d.template_dist_n_bin + 6));
    o.wm = take(cp * (d.template_dist_n_bin + 2));
    for (int l = 0; l < d.n_pair_transform_layer; ++l) {
        PairOff p;
        for (int dir = 0; dir < 2; ++dir) {
            TriOff& t = dir == 0 ? p.out : p.in;
            t.ap_w = take(ch * cp); t.ap_b = take(ch); t.ag_w = take(ch * cp); t.ag_b = take(ch);
            t.bp_w = take(ch * cp); t.bp_b = take(ch); t.bg_w = take(ch * cp); t.bg_b = take(ch);
            t.g_w = take(cp * cp); t.g_b = take(cp); t.z_w = take(cp * ch); t.z_b = take(cp);
            t.lni_g = take(cp); t.lni_b = take(cp); t.lno_g = take(ch); t.lno_b = take(ch);
        }
        const size_t nh = (size_t)d.pair_transition_n * cp;
        p.ln_g = take(cp); p.ln_b = take(cp); p.w1 = take(nh * cp); p.b1 = take(nh); p.w2 = take(cp * nh); p.b2 = take(cp);
        o.pair.push_back(p);
    }
    const size_t H = d.n_head_ipa, C = d.c_hidden_ipa, Pq = d.n_qk_point, Pv = d.n_v_point;
    const size_t ncat = H * (cp + C + 4 * Pv);
    for (int l = 0; l < d.n_structure_layer; ++l) {
        StructOff s;
        s.head = take(H);
        s.q_w = take(H * C * cs); s.q_b = take(H * C);
        s.kv_w = take(2 * H * C * cs); s.kv_b = take(2 * H * C);
        s.qp_w = take(3 * H * Pq * cs); s.qp_b = take(3 * H * Pq);
        s.kvp_w = take(3 * H * (Pq + Pv) * cs); s.kvp_b = take(3 * H * (Pq + Pv));
        s.b_w = take(H * cp); s.b_b = take(H);
        s.o_w = take(cs * ncat); s.o_b = take(cs);
        s.ln1_g = take(cs); s.ln1_b = take(cs);
        s.t1w = take(cs * cs); s.t1b = take(cs); s.t2w = take(cs * cs); s.t2b = take(cs); s.t3w = take(cs * cs); s.t3b = take(cs);
        s.ln2_g = take(cs); s.ln2_b = take(cs);
        s.bb_w = take(6 * cs); s.bb_b = take(6);
        o.st.push_back(s);
    }
    o.total = c;
    return o;
}

// bump allocator; `dry`: only measures.  Two of them: kept-for-backward and per-sublayer scratch (stack discipline).
struct Arena {
    char* base = nullptr; size_t off = 0, peak = 0; bool dry = true;
    float* f(size_t n) {
        const size_t bytes = (n * sizeof(float) + 255) & ~(size_t)255;
        float* r = dry ? nullptr : reinterpret_cast<float*>(base + off);
        off += bytes;
        if (off > peak) peak = off;
        return r;
    }
};

struct TriSave { float *xhat, *rstd, *ap, *ag, *bp, *bg, *acm, *bcm, *xhat_o, *rstd_o, *u, *g; };
struct TransSave { float *xhat, *rstd, *h; unsigned* hmask; };     // hmask: sign pattern of h, one bit per element (null: not kept)
struct PairSave { TriSave out, in; TransSave tr; };
struct StructSave {
    float *s_in, *q, *kv, *qplin, *kvplin, *qp, *kp, *vp, *att, *cat, *xhat1, *rstd1, *s2, *h1, *h2, *xhat2, *rstd2, *s4, *bb, *R, *T;
};
}  // namespace

struct genie_train_ws { void* kept = nullptr; size_t kept_bytes = 0; void* tmp = nullptr; size_t tmp_bytes = 0; FoldEntry* fold_tab = nullptr; int fold_n = 0; };

void train_ws_free(genie_ctx* h) {
    if (!h->train) return;
    if (h->train->kept) (void)hipFree(h->train->kept);
    if (h->train->tmp) (void)hipFree(h->train->tmp);
    if (h->train->fold_tab) (void)hipFree(h->train->fold_tab);
    delete h->train;
    h->train = nullptr;
}

#define TR_ERR(...) do { snprintf(h->err, sizeof(h->err), __VA_ARGS__); } while (0)

namespace {
struct Run {
    genie_ctx* h; hipStream_t st; bool dry; int terms;
    const float* W; float* G;
    Arena K, T;        // kept / scratch
    int B, N, M; long long P;

    // ---- GEMM shapes -------------------------------------------------------------------------------------------------------
    void gemm(const GemmP& p) {
        if (dry) return;
        h->train_gemm_flop += 2.0 * p.M * p.N * (double)p.K * p.batch;
        ProfScope ps_(h, st, KC_TR_GEMM);
#ifndef GENIE_DEV
        launch_gemm(st, p, terms);
#else
        static const bool log = getenv("GENIE_TRAIN_GEMM_LOG") != nullptr;      // developer aid (-DGENIE_DEV builds): one line per GEMM with its own duration; synchronises
        if (!log) { launch_gemm(st, p, terms); return; }
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0, st);
        launch_gemm(st, p, terms);
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        fprintf(stderr, "GEMM M %d N %d K %d batch %d nsplit %d ak %lld bk %lld cn %lld mode %d us %.1f\n", p.M, p.N, p.K, p.batch, p.nsplit, p.ak, p.bk,
                p.cn, p.mode, ms * 1e3f);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
#endif
    }
    // Y[R][O] = X[R][K] (ld ldx) W[O][K]^T + b
    void lin_fwd(const float* X, long long ldx, long long R, int K, size_t w, long long b_off, int O, float* Y, int mode = 0, bool relu = false) {
        GemmP p{X, W + w, Y, b_off >= 0 ? W + b_off : nullptr, (int)R, O, K, ldx, 1, 1, K, O, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, mode};
        p.relu = relu ? 1 : 0;
        gemm(p);
    }
    // dX[R][K] (+)= dY[R][O] W[O][K]
    // `relu_out` (laid out like dX, not with accumulate): the forward ReLU's output -- dX is zeroed where it is not positive
    void lin_bwd_x(const float* dY, long long R, int O, size_t w, int K, float* dX, long long lddx, bool accumulate, const float* relu_out = nullptr,
                   const unsigned* relu_mask = nullptr) {
        GemmP p{dY, W + w, dX, nullptr, (int)R, K, O, O, 1, K, 1, lddx, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, accumulate ? 1 : 0};
        if (relu_mask) p.mask_in = relu_mask;       // (the forward output's sign pattern, GemmP::mask_in)
        else p.gate = relu_out;
        gemm(p);
    }
    // dX[R][K] = dY[R][O] Wc[O][K] with the weight at a raw device pointer (a concatenation of several Linears' matrices)
    void lin_bwd_x_ptr(const float* dY, long long R, int O, const float* Wc, int K, float* dX, long long lddx) {
        GemmP p{dY, Wc, dX, nullptr, (int)R, K, O, O, 1, K, 1, lddx, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, 0};
        gemm(p);
    }
    // dW[O][K] += dY[R][O]^T X[R][K];  db[O] += column sums of dY
    void lin_bwd_w(const float* dY, long long R, int O, const float* X, long long ldx, int K, size_t w, long long b_off) {
        if (!G) return;                     // input-gradient only (genie_denoise_vjp)
        GemmP p{dY, X, G + w, nullptr, O, K, (int)R, 1, O, ldx, 1, K, 1, 1, 1, 0, 0, 0, 0, 0, 0, gemm_splits(O, K, R, 1), 1.0f, 2};
        p.asum = b_off >= 0 ? G + b_off : nullptr;      // db: the column sums of dY, taken while the GEMM streams it
        gemm(p);
    }
    // Y[R][O] = (xhat gamma + beta) W^T + b without forming xhat gamma + beta: the Linear's weights folded with the LayerNorm's affine
    // (`fold`: O K + O floats written by launch_fold_ln_table at the start of the pass) applied to xhat
    void lin_fwd_ln(const float* Xhat, long long R, int K, int O, float* Y, const float* fold, bool relu = false, const float* fold_b = nullptr,
                    unsigned* mask_out = nullptr) {
        GemmP p{Xhat, fold, Y, fold_b ? fold_b : fold ? fold + (size_t)O * K : nullptr, (int)R, O, K, K, 1, 1, K, O, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, 0};
        p.relu = relu ? 1 : 0;
        p.mask_out = mask_out;
        gemm(p);
    }
    // ... `nb` of them on the same xhat in one GEMM (xhat is read once): folded weights stacked [nb O][K], biases [nb O]; result k to Y[k]
    void lin_fwd_ln_cat(const float* Xhat, long long R, int K, int O, int nb, float* const* Y, const float* fold, const float* fold_b) {
        GemmP p{Xhat, fold, Y[0], fold_b, (int)R, nb * O, K, K, 1, 1, K, O, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, 0};
        p.cblk = O; p.cblk_m = 0;
        for (int k = 0; k < nb; ++k) p.ctab[k] = Y[k] - Y[0];
        gemm(p);
    }
    // dW[O][K] += dY^T (xhat gamma + beta) without forming xhat gamma + beta:  gamma[k] (dY^T xhat)[o][k]  +  beta[k] db[o]
    // (db = the bias gradient the same GEMM accumulates; each Linear of the pair stack owns its bias, so db is final when the GEMM is)
    // (lddy: row stride of dY -- it may be a column block of a wider matrix)
    void lin_bwd_w_ln(const float* dY, long long R, int O, const float* Xhat, int K, size_t w, size_t b_off, size_t g_off, size_t beta_off, long long lddy = 0) {
        if (!G) return;
        GemmP p{dY, Xhat, G + w, nullptr, O, K, (int)R, 1, lddy ? lddy : O, K, 1, K, 1, 1, 1, 0, 0, 0, 0, 0, 0, gemm_splits(O, K, R, 1), 1.0f, 2};
        p.asum = G + b_off;
        p.colscale = W + g_off;
        gemm(p);          // (the beta part, dW[o][k] += beta[k] db[o], is added for all such Linears at once at the end of the pass: launch_rank1_table)
    }
    // ... of `nb` Linears of O outputs each on the same xhat, their output gradients the column blocks of dYcat [R][nb O]: one GEMM
    // (xhat is read once), its row blocks landing on the nb weight gradients inside the blob
    void lin_bwd_w_ln_cat(const float* dYcat, long long R, int O, int nb, const float* Xhat, int K, const size_t* w, const size_t* b_off, size_t g_off,
                          size_t beta_off) {
        if (!G) return;
        GemmP p{dYcat, Xhat, G, nullptr, nb * O, K, (int)R, 1, (long long)nb * O, K, 1, K, 1, 1, 1, 0, 0, 0, 0, 0, 0, gemm_splits((long long)nb * O, K, R, 1), 1.0f, 2};
        p.asum = G;
        p.colscale = W + g_off;
        p.cblk = O; p.cblk_m = 1;
        for (int k = 0; k < nb; ++k) { p.ctab[k] = (long long)w[k]; p.atab[k] = (long long)b_off[k]; }
        gemm(p);
    }
    void ln_fwd(const float* x, size_t g, size_t b, float* y, float* xhat, float* rstd, long long R, int C) {
        if (dry) return;
        ProfScope ps_(h, st, KC_TR_LN);
        launch_ln_fwd(st, x, W + g, W + b, y, xhat, rstd, R, C);
    }
    // dx (+)= LN backward; gamma / beta gradients
    void ln_bwd(const float* dy, const float* xhat, const float* rstd, size_t g, size_t b, float* dx, long long R, int C, bool accumulate) {
        if (dry) return;
        ProfScope ps_(h, st, KC_TR_LN);
        launch_ln_bwd(st, dy, xhat, rstd, W + g, dx, R, C, accumulate ? 1 : 0, G ? G + g : nullptr, G ? G + b : nullptr);
    }
    template <class F> void ew(long long n, F f) { if (dry) return; ProfScope ps_(h, st, KC_TR_EW); launch_ew(st, n, f); }
    void transpose(const float* in, float* out, int Bt, int R, int C, bool to_cm) { if (dry) return; ProfScope ps_(h, st, KC_TR_TRANSPOSE); launch_transpose(st, in, out, Bt, R, C, to_cm); }
};
}  // namespace

// losses_out [2 + 2B] as genie_training_loss; z_pred_out [B,N,3] optional
// dz_in != nullptr: vector-Jacobian mode -- no loss; the cotangent of z is dz_in and dtrans_out receives d <dz_in, z> / d trans with
// the frames held fixed (Gd may be nullptr: no weight gradients)
static int train_run(genie_ctx* h, hipStream_t st, bool dry, const float* Wd, float* Gd, const float* trans0, const float* rots0,
                     const int32_t* ts_dev, const float* z_target, const int8_t* codes, float cond_w, const genie_train_opts_t& opt,
                     float* losses_out, float* z_pred_out, size_t* kept_bytes, size_t* tmp_bytes, const float* dz_in = nullptr,
                     float* dtrans_out = nullptr) {
    const genie_dims_t& d = h->d;
    const Offs O = make_offsets(d);
    Run r;
    r.h = h; r.st = st; r.dry = dry; r.terms = opt.fast_math == 1 ? 1 : (opt.fast_math == 2 ? 2 : 3); r.W = Wd; r.G = Gd;
    r.B = h->B; r.N = h->N; r.M = h->B * h->N; r.P = (long long)r.M * h->N;
    r.K.dry = r.T.dry = dry;
    if (!dry) { r.K.base = (char*)h->train->kept; r.T.base = (char*)h->train->tmp; }
    const int B = r.B, N = r.N, M = r.M;
    const long long P = r.P;
    const int cs = d.c_s, cp = d.c_p, ch = d.c_hidden_mul;
    const int nh = d.pair_transition_n * cp;
    const int H = d.n_head_ipa, C = d.c_hidden_ipa, Pq = d.n_qk_point, Pv = d.n_v_point;
    const int ncat = H * (cp + C + 4 * Pv);
    const int nbin = d.template_dist_n_bin, kt = nbin + 6, km = nbin + 2, kr = 2 * d.relpos_k + 3, nf = kt + km + kr;
    const int nsi = d.c_pos_emb + d.c_chain_emb + d.c_timestep_emb + 23, ldx = (nsi + 7) / 8 * 8;
    const bool train = opt.train_mode != 0;
    const float r_tri = train ? opt.tri_dropout : 0.f, r_ipa = train ? opt.ipa_dropout : 0.f, r_tr = train ? opt.transition_dropout : 0.f;
    const uint32_t seed = opt.seed;
    const float* rm = h->rmaskf;
    const int L = d.n_pair_transform_layer, SL = d.n_structure_layer;
    Arena& K = r.K; Arena& T = r.T;

    // =========================================================================================== forward
    if (!dry) launch_single_input(h, st, ts_dev);
    const float* xs = h->xsingle;
    float* tr = K.f((size_t)M * 3);                      // scaled input translations (model.py:171)
    { const float sc = d.rescale; r.ew((long long)M * 3, [=] __device__(long long i) { tr[i] = trans0[i] * sc; }); }
    float* s0 = K.f((size_t)M * cs);
    r.lin_fwd(xs, ldx, M, nsi, O.single_w, -1, cs, s0);
    r.ew((long long)M * cs, [=] __device__(long long i) { s0[i] *= rm[i / cs]; });
    float* Fm = K.f((size_t)P * nf);
    if (!dry) launch_pair_features(st, tr, rots0, codes, rm, h->f_fstm, h->f_fsm, h->f_pos, h->f_ridx, h->f_cidx, Fm, B, N, nbin,
                                   d.template_dist_min, d.template_dist_step, d.relpos_k);
    float* z = K.f((size_t)P * cp);                      // the pair representation, updated in place
    {
        size_t mark = T.off;
        float* pi = T.f((size_t)M * cp); float* pj = T.f((size_t)M * cp);
        r.lin_fwd(s0, cs, M, cs, O.wi, -1, cp, pi);
        r.lin_fwd(s0, cs, M, cs, O.wj, -1, cp, pj);
        r.lin_fwd(Fm, nf, P, kt, O.wt, -1, cp, z, 0);
        r.lin_fwd(Fm + kt, nf, P, km, O.wm, -1, cp, z, 1);
        r.lin_fwd(Fm + kt + km, nf, P, kr, O.wrel, -1, cp, z, 1);
        if (cp % 4 == 0) {
            r.ew(P * cp / 4, [=] __device__(long long e4) {
                const long long e = e4 * 4, row = e / cp; const int c = (int)(e % cp);
                const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
                const float m = rm[bi] * rm[b * N + j];
                float4 zv = *reinterpret_cast<const float4*>(z + e);
                const float4 a = *reinterpret_cast<const float4*>(pi + bi * cp + c), bb = *reinterpret_cast<const float4*>(pj + (b * N + j) * cp + c);
                zv.x = (zv.x + a.x + bb.x) * m; zv.y = (zv.y + a.y + bb.y) * m; zv.z = (zv.z + a.z + bb.z) * m; zv.w = (zv.w + a.w + bb.w) * m;
                *reinterpret_cast<float4*>(z + e) = zv;
            });
        } else
        r.ew(P * cp, [=] __device__(long long e) {
            const long long row = e / cp; const int c = (int)(e % cp);
            const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
            z[e] = (z[e] + pi[bi * cp + c] + pj[(b * N + j) * cp + c]) * (rm[bi] * rm[b * N + j]);
        });
        T.off = mark;
    }
    std::vector<PairSave> ps(L);
    // Every Linear of the pair stack that follows a LayerNorm runs on xhat with the LayerNorm's affine folded into its weights; all
    // 6 L x 2 + L folds in ONE launch here (the table of blob offsets depends on the dims only and is uploaded once per handle)
    std::vector<FoldEntry> fold_plan;
    std::vector<size_t> fold_tri((size_t)L * 2 * 7), fold_tr(L);
    size_t fold_floats = 0;
    {
        // (folded weights [O][K] at the returned offset, the folded bias right behind them)
        auto plan = [&](size_t w, size_t b, size_t g, size_t be, int Oo, int Kk) {
            fold_plan.push_back(FoldEntry{(long long)w, (long long)b, (long long)g, (long long)be, (long long)fold_floats,
                                          (long long)(fold_floats + (size_t)Oo * Kk), -1, Oo, Kk});
            const size_t at = fold_floats;
            fold_floats += ((size_t)Oo * Kk + Oo + 63) / 64 * 64;
            return at;
        };
        // five Linears on the same input whose folded weights form ONE matrix [5 O][K] (then the five biases, then the five unfolded
        // matrices stacked the same way, f[6]: the B operand of their one input-gradient GEMM): they run as one GEMM
        auto plan5 = [&](const size_t (&w)[5], const size_t (&b)[5], size_t g, size_t be, int Oo, int Kk, size_t* f) {
            const size_t raw0 = fold_floats + ((size_t)5 * Oo * Kk + 5 * Oo + 63) / 64 * 64;
            for (int k = 0; k < 5; ++k) {
                f[k] = fold_floats + (size_t)k * Oo * Kk;
                fold_plan.push_back(FoldEntry{(long long)w[k], (long long)b[k], (long long)g, (long long)be, (long long)f[k],
                                              (long long)(fold_floats + (size_t)5 * Oo * Kk + (size_t)k * Oo), (long long)(raw0 + (size_t)k * Oo * Kk), Oo, Kk});
            }
            f[6] = raw0;
            fold_floats = raw0 + (size_t)5 * Oo * Kk;
        };
        for (int l = 0; l < L; ++l) {
            for (int dir = 0; dir < 2; ++dir) {
                const TriOff& t = dir == 0 ? O.pair[l].out : O.pair[l].in;
                size_t* f = &fold_tri[((size_t)l * 2 + dir) * 7];
                if (ch == cp) {
                    const size_t w5[5] = {t.ap_w, t.ag_w, t.bp_w, t.bg_w, t.g_w}, b5[5] = {t.ap_b, t.ag_b, t.bp_b, t.bg_b, t.g_b};
                    plan5(w5, b5, t.lni_g, t.lni_b, ch, cp, f);
                } else {
                    f[0] = plan(t.ap_w, t.ap_b, t.lni_g, t.lni_b, ch, cp); f[1] = plan(t.ag_w, t.ag_b, t.lni_g, t.lni_b, ch, cp);
                    f[2] = plan(t.bp_w, t.bp_b, t.lni_g, t.lni_b, ch, cp); f[3] = plan(t.bg_w, t.bg_b, t.lni_g, t.lni_b, ch, cp);
                    f[4] = plan(t.g_w, t.g_b, t.lni_g, t.lni_b, cp, cp);
                }
                f[5] = plan(t.z_w, t.z_b, t.lno_g, t.lno_b, cp, ch);
            }
            fold_tr[l] = plan(O.pair[l].w1, O.pair[l].b1, O.pair[l].ln_g, O.pair[l].ln_b, nh, cp);
        }
    }
    float* foldbuf = K.f(fold_floats);
    if (!dry && !fold_plan.empty()) {
        genie_train_ws* tw = h->train;
        if (tw->fold_n != (int)fold_plan.size()) {
            if (tw->fold_tab) (void)hipFree(tw->fold_tab);
            (void)hipMalloc((void**)&tw->fold_tab, fold_plan.size() * sizeof(FoldEntry));
            (void)hipMemcpyAsync(tw->fold_tab, fold_plan.data(), fold_plan.size() * sizeof(FoldEntry), hipMemcpyHostToDevice, st);
            (void)hipStreamSynchronize(st);          // (once per handle: the host vector goes out of scope)
            tw->fold_n = (int)fold_plan.size();
        }
        ProfScope ps_(h, st, KC_TR_EW);
        launch_fold_ln_table(st, Wd, foldbuf, tw->fold_tab, tw->fold_n, nh > ch ? (nh > cp ? nh : cp) : (ch > cp ? ch : cp));
    }
    // ch == 128: the gate / LayerNorm passes next to the contraction are fused with its layout changes (train_layout_kernels.hip) and the
    // projections ap, bp are temporaries (the backward pass works from a, b themselves)
    const bool cmf = train_cm_fusable(ch);
    auto tri_fwd = [&](const TriOff& t, TriSave& s, bool outgoing, uint32_t tag, const size_t* fo) {
        s.xhat = K.f(P * cp); s.rstd = K.f(P); s.ag = K.f(P * ch); s.bg = K.f(P * ch);
        if (!cmf) { s.ap = K.f(P * ch); s.bp = K.f(P * ch); }
        s.acm = K.f(P * ch); s.bcm = K.f(P * ch); s.xhat_o = K.f(P * ch); s.rstd_o = K.f(P); s.u = K.f(P * cp); s.g = K.f(P * cp);
        size_t mark = T.off;
        if (cmf) { s.ap = T.f(P * ch); s.bp = T.f(P * ch); }
        r.ln_fwd(z, t.lni_g, t.lni_b, nullptr, s.xhat, s.rstd, P, cp);                                  // (xhat only: the Linears below run on folded weights)
        auto fw = [&](int k) -> const float* { return dry ? nullptr : foldbuf + fo[k]; };
        const bool cat5 = ch == cp;                 // the five Linears on xhat as one GEMM (fold_plan stacked their folded weights)
        if (cat5) {
            float* const Y5[5] = {s.ap, s.ag, s.bp, s.bg, s.g};
            r.lin_fwd_ln_cat(s.xhat, P, cp, ch, 5, Y5, fw(0), dry ? nullptr : fw(0) + (size_t)5 * ch * cp);
        } else {
            r.lin_fwd_ln(s.xhat, P, cp, ch, s.ap, fw(0));
            r.lin_fwd_ln(s.xhat, P, cp, ch, s.ag, fw(1));
            r.lin_fwd_ln(s.xhat, P, cp, ch, s.bp, fw(2));
            r.lin_fwd_ln(s.xhat, P, cp, ch, s.bg, fw(3));
        }
        float *xcm, *xn;
        if (cmf) {
            if (!dry) { ProfScope ps_(h, st, KC_TR_TRANSPOSE); launch_gate_to_cm(st, s.ap, s.ag, s.bp, s.bg, rm, s.acm, s.bcm, B, N, ch); }
            xcm = T.f(P * ch);
            xn = s.ap;          // (dead from here on)
        } else {
            float* arm = T.f(P * ch); float* brm = T.f(P * ch);
            const float *ap = s.ap, *ag = s.ag, *bp = s.bp, *bg = s.bg;
            r.ew(P * ch, [=] __device__(long long e) {
                const long long row = e / ch; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
                const float m = rm[bi] * rm[b * N + j];
                arm[e] = ap[e] * m / (1.0f + expf(-ag[e]));
                brm[e] = bp[e] * m / (1.0f + expf(-bg[e]));
            });
            r.transpose(arm, s.acm, B, N * N, ch, true); r.transpose(brm, s.bcm, B, N * N, ch, true);
            xcm = arm; xn = arm;
        }
        {
            GemmP g{s.acm, s.bcm, xcm, nullptr, N, N, N, 0, 0, 0, 0, N, 1, B * ch, ch, (long long)ch * N * N, (long long)N * N,
                    (long long)ch * N * N, (long long)N * N, (long long)ch * N * N, (long long)N * N, 1, 1.0f, 0};
            if (outgoing) { g.am = N; g.ak = 1; g.bk = 1; g.bn = N; }      // x[i][j] = sum_k a[i][k] b[j][k]
            else { g.am = 1; g.ak = N; g.bk = N; g.bn = 1; }               // x[i][j] = sum_k a[k][i] b[k][j]
            r.gemm(g);
        }
        (void)xn;
        if (cmf) {
            if (!dry) { ProfScope ps_(h, st, KC_TR_LN); launch_ln_from_cm(st, xcm, Wd + t.lno_g, Wd + t.lno_b, nullptr, s.xhat_o, s.rstd_o, B, N * N, ch); }
        } else {
            float* xrm = T.f(P * ch);
            r.transpose(xcm, xrm, B, N * N, ch, false);
            r.ln_fwd(xrm, t.lno_g, t.lno_b, nullptr, s.xhat_o, s.rstd_o, P, ch);
        }
        r.lin_fwd_ln(s.xhat_o, P, ch, cp, s.u, fw(5));
        if (!cat5) r.lin_fwd_ln(s.xhat, P, cp, cp, s.g, fw(4));
        {
            float *u = s.u, *g = s.g;
            if (cp % 4 == 0) {      // four channels of one pair row per thread: 16-byte accesses (five passes of the pair tensor, 83 -> 6x us)
                r.ew(P * cp / 4, [=] __device__(long long e4) {
                    const long long e = e4 * 4, row = e / cp; const int c = (int)(e % cp);
                    const int j = (int)(row % N); const long long b = row / ((long long)N * N);
                    float4 gv = *reinterpret_cast<const float4*>(g + e);
                    const float4 uv = *reinterpret_cast<const float4*>(u + e);
                    float4 zv = *reinterpret_cast<const float4*>(z + e);
                    gv.x = 1.0f / (1.0f + expf(-gv.x)); gv.y = 1.0f / (1.0f + expf(-gv.y)); gv.z = 1.0f / (1.0f + expf(-gv.z)); gv.w = 1.0f / (1.0f + expf(-gv.w));
                    *reinterpret_cast<float4*>(g + e) = gv;
                    const uint64_t di = (uint64_t)((b * N + j) * cp + c);
                    const bool dr = r_tri > 0.f;
                    zv.x += uv.x * gv.x * (dr ? drop_scale(seed, tag, di, r_tri) : 1.0f);
                    zv.y += uv.y * gv.y * (dr ? drop_scale(seed, tag, di + 1, r_tri) : 1.0f);
                    zv.z += uv.z * gv.z * (dr ? drop_scale(seed, tag, di + 2, r_tri) : 1.0f);
                    zv.w += uv.w * gv.w * (dr ? drop_scale(seed, tag, di + 3, r_tri) : 1.0f);
                    *reinterpret_cast<float4*>(z + e) = zv;
                });
            } else
            r.ew(P * cp, [=] __device__(long long e) {
                const float gg = 1.0f / (1.0f + expf(-g[e]));
                g[e] = gg;
                const long long row = e / cp; const int c = (int)(e % cp);
                const int j = (int)(row % N); const long long b = row / ((long long)N * N);
                const float ds = r_tri > 0.f ? drop_scale(seed, tag, (uint64_t)((b * N + j) * cp + c), r_tri) : 1.0f;
                z[e] += u[e] * gg * ds;
            });
        }
        T.off = mark;
    };
    for (int l = 0; l < L; ++l) {
        tri_fwd(O.pair[l].out, ps[l].out, true, (uint32_t)(4 * l), &fold_tri[((size_t)l * 2) * 7]);
        tri_fwd(O.pair[l].in, ps[l].in, false, (uint32_t)(4 * l + 1), &fold_tri[((size_t)l * 2 + 1) * 7]);
        TransSave& s = ps[l].tr;
        const PairOff& o = O.pair[l];
        s.xhat = K.f(P * cp); s.rstd = K.f(P); s.h = K.f(P * nh);
        // where the 128-tile kernel runs both the Linear + ReLU and the ReLU's backward GEMM, the forward pass leaves h's sign pattern as
        // one bit per element and the backward GEMM's epilogue reads that instead of h itself (268 MB per layer at N = 256, batch 2)
        {
            const GemmP gf{nullptr, nullptr, nullptr, nullptr, (int)P, nh, cp, cp, 1, 1, cp, nh, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, 0};       // lin_fwd_ln
            const GemmP gb{nullptr, nullptr, nullptr, nullptr, (int)P, nh, cp, cp, 1, nh, 1, nh, 1, 1, 1, 0, 0, 0, 0, 0, 0, 1, 1.0f, 0};       // lin_bwd_x through w2
            s.hmask = (nh % 32 == 0 && gemm_takes_mask(gf) && gemm_takes_mask(gb)) ? reinterpret_cast<unsigned*>(K.f(P * (nh / 32))) : nullptr;
        }
        size_t mark = T.off;
        float* ot = T.f(P * cp);
        r.ln_fwd(z, o.ln_g, o.ln_b, nullptr, s.xhat, s.rstd, P, cp);
        r.lin_fwd_ln(s.xhat, P, cp, nh, s.h, dry ? nullptr : foldbuf + fold_tr[l], true, nullptr, dry ? nullptr : s.hmask);          // LayerNorm's affine folded into Linear + ReLU
        r.lin_fwd(s.h, nh, P, nh, o.w2, o.b2, cp, ot);
        if (cp % 4 == 0) {
            r.ew(P * cp / 4, [=] __device__(long long e4) {
                const long long e = e4 * 4, row = e / cp; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
                const float m = rm[bi] * rm[b * N + j];
                float4 zv = *reinterpret_cast<const float4*>(z + e);
                const float4 o4 = *reinterpret_cast<const float4*>(ot + e);
                zv.x = (zv.x + o4.x * m) * m; zv.y = (zv.y + o4.y * m) * m; zv.z = (zv.z + o4.z * m) * m; zv.w = (zv.w + o4.w * m) * m;
                *reinterpret_cast<float4*>(z + e) = zv;
            });
        } else
        r.ew(P * cp, [=] __device__(long long e) {
            const long long row = e / cp; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
            const float m = rm[bi] * rm[b * N + j];
            z[e] = (z[e] + ot[e] * m) * m;
        });
        T.off = mark;
    }
    // ---- structure net
    std::vector<StructSave> ss(SL * d.n_structure_block);
    float* s = K.f((size_t)M * cs);
    if (!dry) (void)hipMemcpyAsync(s, s0, (size_t)M * cs * 4, hipMemcpyDeviceToDevice, st);
    float* Rc = K.f((size_t)M * 9); float* Tc = K.f((size_t)M * 3);
    if (!dry) { (void)hipMemcpyAsync(Rc, rots0, (size_t)M * 9 * 4, hipMemcpyDeviceToDevice, st); (void)hipMemcpyAsync(Tc, tr, (size_t)M * 3 * 4, hipMemcpyDeviceToDevice, st); }
    int li = 0;
    for (int blk = 0; blk < d.n_structure_block; ++blk)
        for (int l = 0; l < SL; ++l, ++li) {
            const StructOff& o = O.st[l];
            StructSave& v = ss[li];
            v.s_in = s; v.R = Rc; v.T = Tc;
            v.q = K.f((size_t)M * H * C); v.kv = K.f((size_t)M * 2 * H * C); v.qplin = K.f((size_t)M * 3 * H * Pq); v.kvplin = K.f((size_t)M * 3 * H * (Pq + Pv));
            v.qp = K.f((size_t)M * H * Pq * 3); v.kp = K.f((size_t)M * H * Pq * 3); v.vp = K.f((size_t)M * H * Pv * 3);
            v.att = K.f((size_t)B * H * N * N); v.cat = K.f((size_t)M * ncat);
            v.xhat1 = K.f((size_t)M * cs); v.rstd1 = K.f(M); v.s2 = K.f((size_t)M * cs); v.h1 = K.f((size_t)M * cs); v.h2 = K.f((size_t)M * cs);
            v.xhat2 = K.f((size_t)M * cs); v.rstd2 = K.f(M); v.s4 = K.f((size_t)M * cs); v.bb = K.f((size_t)M * 6);
            float* Rn = K.f((size_t)M * 9); float* Tn = K.f((size_t)M * 3);
            size_t mark = T.off;
            r.lin_fwd(s, cs, M, cs, o.q_w, o.q_b, H * C, v.q); r.lin_fwd(s, cs, M, cs, o.kv_w, o.kv_b, 2 * H * C, v.kv);
            r.lin_fwd(s, cs, M, cs, o.qp_w, o.qp_b, 3 * H * Pq, v.qplin); r.lin_fwd(s, cs, M, cs, o.kvp_w, o.kvp_b, 3 * H * (Pq + Pv), v.kvplin);
            if (!dry) { launch_points_fwd(st, v.qplin, Rc, Tc, v.qp, nullptr, M, H, Pq, 0); launch_points_fwd(st, v.kvplin, Rc, Tc, v.kp, v.vp, M, H, Pq, Pv); }
            float* bias = T.f((size_t)P * H);
            r.lin_fwd(z, cp, P, cp, o.b_w, o.b_b, H, bias);
            IpaArgs a{};
            a.B = B; a.N = N; a.H = H; a.C = C; a.Pq = Pq; a.Pv = Pv; a.cp = cp;
            a.q = v.q; a.kv = v.kv; a.qp = v.qp; a.kp = v.kp; a.vp = v.vp; a.bias = bias; a.p = z; a.rots = Rc; a.trans = Tc; a.rmask = rm;
            a.head_w = Wd + o.head; a.att = v.att; a.cat = v.cat;
            if (!dry) { ProfScope ps_(h, st, KC_TR_IPA); launch_ipa_fwd(st, a); }
            float* s1 = T.f((size_t)M * cs);
            r.lin_fwd(v.cat, ncat, M, ncat, o.o_w, o.o_b, cs, s1);
            {
                const float* sin = s; const uint32_t tag = 1000u + 2u * (uint32_t)li;
                r.ew((long long)M * cs, [=] __device__(long long e) {
                    const float ds = r_ipa > 0.f ? drop_scale(seed, tag, (uint64_t)e, r_ipa) : 1.0f;
                    s1[e] = (s1[e] + sin[e]) * ds;
                });
            }
            r.ln_fwd(s1, o.ln1_g, o.ln1_b, v.s2, v.xhat1, v.rstd1, M, cs);
            r.lin_fwd(v.s2, cs, M, cs, o.t1w, o.t1b, cs, v.h1, 0, true);       // Linear + ReLU
            r.lin_fwd(v.h1, cs, M, cs, o.t2w, o.t2b, cs, v.h2, 0, true);
            float* s3 = T.f((size_t)M * cs);
            r.lin_fwd(v.h2, cs, M, cs, o.t3w, o.t3b, cs, s3);
            {
                const float* s2 = v.s2; const uint32_t tag = 1001u + 2u * (uint32_t)li;
                r.ew((long long)M * cs, [=] __device__(long long e) {
                    const float ds = r_tr > 0.f ? drop_scale(seed, tag, (uint64_t)e, r_tr) : 1.0f;
                    s3[e] = (s3[e] + s2[e]) * ds;
                });
            }
            r.ln_fwd(s3, o.ln2_g, o.ln2_b, v.s4, v.xhat2, v.rstd2, M, cs);
            r.lin_fwd(v.s4, cs, M, cs, o.bb_w, o.bb_b, 6, v.bb);
            if (!dry) launch_frames_fwd(st, v.bb, Rc, Tc, Rn, Tn, M);
            s = v.s4; Rc = Rn; Tc = Tn;
            T.off = mark;
        }
    // ---- z = trans_in - trans_out / rescale (model.py:184-187), loss and d loss / d z (genie.py:90-105)
    float* zp = K.f((size_t)M * 3); float* dzp = K.f((size_t)M * 3);
    { const float inv = 1.0f / d.rescale; const float* Tf = Tc; r.ew((long long)M * 3, [=] __device__(long long i) { zp[i] = trans0[i] - Tf[i] * inv; }); }
    if (!dry) {
        if (dz_in) (void)hipMemcpyAsync(dzp, dz_in, (size_t)M * 3 * 4, hipMemcpyDeviceToDevice, st);
        else launch_training_loss(h, st, zp, z_target, cond_w, losses_out, dzp);
        if (z_pred_out) (void)hipMemcpyAsync(z_pred_out, zp, (size_t)M * 3 * 4, hipMemcpyDeviceToDevice, st);
    }

    // =========================================================================================== backward
    if (!dry && Gd) (void)hipMemsetAsync(Gd, 0, O.total * sizeof(float), st);
    float* gsink = K.f(64);                      // where the IPA kernels' small atomics go when no weight gradients are wanted
    float* ds = K.f((size_t)M * cs);             // gradient wrt the single representation leaving a structure layer
    float* dRn = K.f((size_t)M * 9); float* dTn = K.f((size_t)M * 3);      // ... wrt the frames leaving it
    float* dRl = K.f((size_t)M * 9); float* dTl = K.f((size_t)M * 3);      // ... wrt the frames entering it
    float* dP = K.f((size_t)P * cp);             // ... wrt the pair representation
    if (!dry) {
        (void)hipMemsetAsync(ds, 0, (size_t)M * cs * 4, st); (void)hipMemsetAsync(dRn, 0, (size_t)M * 9 * 4, st);
        (void)hipMemsetAsync(dP, 0, (size_t)P * cp * 4, st);
    }
    { const float inv = -1.0f / d.rescale; r.ew((long long)M * 3, [=] __device__(long long i) { dTn[i] = dzp[i] * inv; }); }
    for (li = (int)ss.size() - 1; li >= 0; --li) {
        const StructOff& o = O.st[li % SL];
        StructSave& v = ss[li];
        size_t mark = T.off;
        float* dbb = T.f((size_t)M * 6);
        if (!dry) {
            (void)hipMemsetAsync(dRl, 0, (size_t)M * 9 * 4, st); (void)hipMemsetAsync(dTl, 0, (size_t)M * 3 * 4, st);
            launch_frames_bwd(st, v.bb, v.R, dRn, dTn, dbb, dRl, dTl, M);
        }
        r.lin_bwd_w(dbb, M, 6, v.s4, cs, cs, o.bb_w, (long long)o.bb_b);
        r.lin_bwd_x(dbb, M, 6, o.bb_w, cs, ds, cs, true);                  // ds = d s4 (already holds the next layer's contribution)
        float* d3 = T.f((size_t)M * cs);
        r.ln_bwd(ds, v.xhat2, v.rstd2, o.ln2_g, o.ln2_b, d3, M, cs, false);
        {
            const uint32_t tag = 1001u + 2u * (uint32_t)li;
            r.ew((long long)M * cs, [=] __device__(long long e) { if (r_tr > 0.f) d3[e] *= drop_scale(seed, tag, (uint64_t)e, r_tr); });
        }
        float* d2 = T.f((size_t)M * cs);           // gradient wrt s2 (MLP input + residual)
        if (!dry) (void)hipMemcpyAsync(d2, d3, (size_t)M * cs * 4, hipMemcpyDeviceToDevice, st);
        r.lin_bwd_w(d3, M, cs, v.h2, cs, cs, o.t3w, (long long)o.t3b);
        float* dh = T.f((size_t)M * cs);
        r.lin_bwd_x(d3, M, cs, o.t3w, cs, dh, cs, false, v.h2);            // through the ReLU
        r.lin_bwd_w(dh, M, cs, v.h1, cs, cs, o.t2w, (long long)o.t2b);
        float* dh1 = d3;        // reuse
        r.lin_bwd_x(dh, M, cs, o.t2w, cs, dh1, cs, false, v.h1);
        r.lin_bwd_w(dh1, M, cs, v.s2, cs, cs, o.t1w, (long long)o.t1b);
        r.lin_bwd_x(dh1, M, cs, o.t1w, cs, d2, cs, true);
        float* d1 = dh;         // gradient wrt s1 (dropped sum)
        r.ln_bwd(d2, v.xhat1, v.rstd1, o.ln1_g, o.ln1_b, d1, M, cs, false);
        {
            const uint32_t tag = 1000u + 2u * (uint32_t)li;
            r.ew((long long)M * cs, [=] __device__(long long e) { if (r_ipa > 0.f) d1[e] *= drop_scale(seed, tag, (uint64_t)e, r_ipa); });
        }
        // d1 = gradient wrt (s_in + ipa_out): residual into ds, and through linear_out
        if (!dry) (void)hipMemcpyAsync(ds, d1, (size_t)M * cs * 4, hipMemcpyDeviceToDevice, st);
        r.lin_bwd_w(d1, M, cs, v.cat, ncat, ncat, o.o_w, (long long)o.o_b);
        float* dcat = T.f((size_t)M * ncat);
        r.lin_bwd_x(d1, M, cs, o.o_w, ncat, dcat, ncat, false);
        float* dlg = T.f((size_t)B * H * N * N);
        float* dq = T.f((size_t)M * H * C); float* dkv = T.f((size_t)M * 2 * H * C);
        float* dqp = T.f((size_t)M * H * Pq * 3); float* dkp = T.f((size_t)M * H * Pq * 3); float* dvp = T.f((size_t)M * H * Pv * 3);
        float* doptg = T.f((size_t)M * H * Pv * 3);
        IpaArgs a{};
        a.B = B; a.N = N; a.H = H; a.C = C; a.Pq = Pq; a.Pv = Pv; a.cp = cp;
        a.q = v.q; a.kv = v.kv; a.qp = v.qp; a.kp = v.kp; a.vp = v.vp; a.p = z; a.rots = v.R; a.trans = v.T; a.rmask = rm;
        a.head_w = Wd + o.head; a.wb = Wd + o.b_w; a.att = v.att; a.cat = v.cat; a.dcat = dcat;
        a.dlg = dlg; a.dq = dq; a.dqp = dqp; a.doptg = doptg; a.dP = dP; a.dhead = Gd ? Gd + o.head : gsink; a.dbb = Gd ? Gd + o.b_b : gsink + 32; a.dR = dRl; a.dT = dTl;
        a.dkv = dkv; a.dkp = dkp; a.dvp = dvp;
        if (!dry) { ProfScope ps_(h, st, KC_TR_IPA); launch_ipa_bwd(st, a); }
        {   // linear_b weight: dWb[h][c] += c_b sum_{b,i,j} dlogit[b,h,i,j] p[b,i,j,c]
            GemmP g{dlg, z, Gd ? Gd + o.b_w : nullptr, nullptr, H, cp, N * N, (long long)N * N, 1, cp, 1, cp, 1, B, 1, (long long)H * N * N, 0, (long long)N * N * cp, 0, 0, 0,
                    gemm_splits(H, cp, (long long)N * N, B), sqrtf(1.0f / 3.0f), 2};
            if (Gd) r.gemm(g);
        }
        float* dqplin = T.f((size_t)M * 3 * H * Pq); float* dkvplin = T.f((size_t)M * 3 * H * (Pq + Pv));
        if (!dry) {
            launch_points_bwd(st, v.qplin, v.R, dqp, nullptr, dqplin, dRl, dTl, M, H, Pq, 0);
            launch_points_bwd(st, v.kvplin, v.R, dkp, dvp, dkvplin, dRl, dTl, M, H, Pq, Pv);
        }
        r.lin_bwd_w(dq, M, H * C, v.s_in, cs, cs, o.q_w, (long long)o.q_b); r.lin_bwd_x(dq, M, H * C, o.q_w, cs, ds, cs, true);
        r.lin_bwd_w(dkv, M, 2 * H * C, v.s_in, cs, cs, o.kv_w, (long long)o.kv_b); r.lin_bwd_x(dkv, M, 2 * H * C, o.kv_w, cs, ds, cs, true);
        r.lin_bwd_w(dqplin, M, 3 * H * Pq, v.s_in, cs, cs, o.qp_w, (long long)o.qp_b); r.lin_bwd_x(dqplin, M, 3 * H * Pq, o.qp_w, cs, ds, cs, true);
        r.lin_bwd_w(dkvplin, M, 3 * H * (Pq + Pv), v.s_in, cs, cs, o.kvp_w, (long long)o.kvp_b);
        r.lin_bwd_x(dkvplin, M, 3 * H * (Pq + Pv), o.kvp_w, cs, ds, cs, true);
        std::swap(dRn, dRl); std::swap(dTn, dTl);
        T.off = mark;
    }
    if (!dry && opt.struct_done_event) (void)hipEventRecord((hipEvent_t)opt.struct_done_event, st);      // structure_net.* gradients are final
    // ---- pair transform net, backwards.  dP = gradient wrt the pair representation leaving the current sub-layer.
    auto tri_bwd = [&](const TriOff& t, TriSave& sv, bool outgoing, uint32_t tag, const size_t* fo) {
        size_t mark = T.off;
        // ch == 128: the gradients of the five Linears on LN_in(z) are the column blocks [d ap | d ag | d bp | d bg | d g] of ONE row-major
        // matrix, so that their input gradients are one GEMM against the five weight matrices stacked (no read-modify-write of dzn)
        const int ncat5 = 4 * ch + cp;
        float* du = T.f(P * cp);
        float* dycat = cmf ? T.f(P * ncat5) : nullptr;
        float* dgl = cmf ? dycat + 4 * ch : T.f(P * cp);
        const int ldg = cmf ? ncat5 : cp;
        {
            const float *u = sv.u, *g = sv.g;
            if (cp % 4 == 0 && ldg % 4 == 0) {
                r.ew(P * cp / 4, [=] __device__(long long e4) {
                    const long long e = e4 * 4, row = e / cp; const int c = (int)(e % cp);
                    const int j = (int)(row % N); const long long b = row / ((long long)N * N);
                    const float4 dv = *reinterpret_cast<const float4*>(dP + e), uv = *reinterpret_cast<const float4*>(u + e), gv = *reinterpret_cast<const float4*>(g + e);
                    const uint64_t di = (uint64_t)((b * N + j) * cp + c);
                    const bool dr = r_tri > 0.f;
                    const float4 dout = make_float4(dv.x * (dr ? drop_scale(seed, tag, di, r_tri) : 1.0f), dv.y * (dr ? drop_scale(seed, tag, di + 1, r_tri) : 1.0f),
                                                    dv.z * (dr ? drop_scale(seed, tag, di + 2, r_tri) : 1.0f), dv.w * (dr ? drop_scale(seed, tag, di + 3, r_tri) : 1.0f));
                    *reinterpret_cast<float4*>(du + e) = make_float4(dout.x * gv.x, dout.y * gv.y, dout.z * gv.z, dout.w * gv.w);
                    *reinterpret_cast<float4*>(dgl + row * ldg + c) = make_float4(dout.x * uv.x * gv.x * (1.0f - gv.x), dout.y * uv.y * gv.y * (1.0f - gv.y),
                                                                                  dout.z * uv.z * gv.z * (1.0f - gv.z), dout.w * uv.w * gv.w * (1.0f - gv.w));
                });
            } else
            r.ew(P * cp, [=] __device__(long long e) {
                const long long row = e / cp; const int c = (int)(e % cp);
                const int j = (int)(row % N); const long long b = row / ((long long)N * N);
                const float ds_ = r_tri > 0.f ? drop_scale(seed, tag, (uint64_t)((b * N + j) * cp + c), r_tri) : 1.0f;
                const float dout = dP[e] * ds_;
                du[e] = dout * g[e];
                dgl[row * ldg + c] = dout * u[e] * g[e] * (1.0f - g[e]);
            });
        }
        // linear_z: its weight gradient against xhat_o (gamma as a column scale, beta through the bias gradient)
        r.lin_bwd_w_ln(du, P, cp, sv.xhat_o, ch, t.z_w, t.z_b, t.lno_g, t.lno_b);
        float* dxn = T.f(P * ch);
        r.lin_bwd_x(du, P, cp, t.z_w, ch, dxn, ch, false);
        float* dxcm = T.f(P * ch);
        if (cmf) {
            if (!dry) { ProfScope ps_(h, st, KC_TR_LN); launch_ln_bwd_to_cm(st, dxn, sv.xhat_o, sv.rstd_o, Wd + t.lno_g, dxcm, B, N * N, ch, Gd ? Gd + t.lno_g : nullptr, Gd ? Gd + t.lno_b : nullptr); }
        } else {
            float* dxrm = T.f(P * ch);
            r.ln_bwd(dxn, sv.xhat_o, sv.rstd_o, t.lno_g, t.lno_b, dxrm, P, ch, false);
            r.transpose(dxrm, dxcm, B, N * N, ch, true);
        }
        float* dacm = T.f(P * ch); float* dbcm = T.f(P * ch);
        {
            const long long bs1 = (long long)ch * N * N, bs2 = (long long)N * N;
            GemmP ga{nullptr, nullptr, dacm, nullptr, N, N, N, 0, 0, 0, 0, N, 1, B * ch, ch, bs1, bs2, bs1, bs2, bs1, bs2, 1, 1.0f, 0};
            GemmP gb = ga; gb.C = dbcm;
            if (outgoing) {     // da[i][k] = sum_j dx[i][j] b[j][k];  db[j][k] = sum_i dx[i][j] a[i][k]
                ga.A = dxcm; ga.am = N; ga.ak = 1; ga.B = sv.bcm; ga.bk = N; ga.bn = 1;
                gb.A = dxcm; gb.am = 1; gb.ak = N; gb.B = sv.acm; gb.bk = N; gb.bn = 1;
            } else {            // da[k][i] = sum_j b[k][j] dx[i][j];  db[k][j] = sum_i a[k][i] dx[i][j]
                ga.A = sv.bcm; ga.am = N; ga.ak = 1; ga.B = dxcm; ga.bk = 1; ga.bn = N;
                gb.A = sv.acm; gb.am = N; gb.ak = 1; gb.B = dxcm; gb.bk = N; gb.bn = 1;
            }
            r.gemm(ga); r.gemm(gb);
        }
        float *dap, *dag, *dbp, *dbg;
        float* dzn;
        if (cmf) {              // a = ap m s, so  d ap = da m s  and  d ag = da a (1 - s): the projections themselves are not needed
            dap = dycat; dag = dycat + ch; dbp = dycat + 2 * ch; dbg = dycat + 3 * ch;
            if (!dry) { ProfScope ps_(h, st, KC_TR_TRANSPOSE); launch_gate_bwd_from_cm(st, dacm, dbcm, sv.acm, sv.bcm, sv.ag, sv.bg, rm, dap, dag, dbp, dbg, B, N, ch, ncat5); }
            // the five weight matrices stacked [4 ch + cp][cp] (in the blob their biases sit between them): with ch == cp the fold launch
            // at the start of the pass has left that stack next to the folded weights
            float* wcat = ch == cp ? (dry ? nullptr : foldbuf + fo[6]) : T.f((size_t)ncat5 * cp);
            if (!dry && ch != cp) {
                const size_t offs[5] = {t.ap_w, t.ag_w, t.bp_w, t.bg_w, t.g_w};
                for (int k = 0; k < 5; ++k)
                    (void)hipMemcpyAsync(wcat + (size_t)k * ch * cp, Wd + offs[k], (size_t)(k < 4 ? ch : cp) * cp * 4, hipMemcpyDeviceToDevice, st);
            }
            dzn = T.f(P * cp);
            if (ch == cp) {
                const size_t w5[5] = {t.ap_w, t.ag_w, t.bp_w, t.bg_w, t.g_w}, b5[5] = {t.ap_b, t.ag_b, t.bp_b, t.bg_b, t.g_b};
                r.lin_bwd_w_ln_cat(dycat, P, ch, 5, sv.xhat, cp, w5, b5, t.lni_g, t.lni_b);
            } else {
                r.lin_bwd_w_ln(dap, P, ch, sv.xhat, cp, t.ap_w, t.ap_b, t.lni_g, t.lni_b, ncat5);
                r.lin_bwd_w_ln(dag, P, ch, sv.xhat, cp, t.ag_w, t.ag_b, t.lni_g, t.lni_b, ncat5);
                r.lin_bwd_w_ln(dbp, P, ch, sv.xhat, cp, t.bp_w, t.bp_b, t.lni_g, t.lni_b, ncat5);
                r.lin_bwd_w_ln(dbg, P, ch, sv.xhat, cp, t.bg_w, t.bg_b, t.lni_g, t.lni_b, ncat5);
                r.lin_bwd_w_ln(dgl, P, cp, sv.xhat, cp, t.g_w, t.g_b, t.lni_g, t.lni_b, ncat5);
            }
            r.lin_bwd_x_ptr(dycat, P, ncat5, wcat, cp, dzn, cp);
        } else {
            float* darm = dxn; float* dbrm = T.f(P * ch);
            r.transpose(dacm, darm, B, N * N, ch, false); r.transpose(dbcm, dbrm, B, N * N, ch, false);
            dap = dacm; dag = dbcm; dbp = darm; dbg = dbrm;      // in place where the shapes allow
            const float *ap = sv.ap, *ag = sv.ag, *bp = sv.bp, *bg = sv.bg;
            r.ew(P * ch, [=] __device__(long long e) {
                const long long row = e / ch; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
                const float m = rm[bi] * rm[b * N + j];
                const float sa = 1.0f / (1.0f + expf(-ag[e])), sb = 1.0f / (1.0f + expf(-bg[e]));
                const float da = darm[e] * m, db = dbrm[e] * m;
                dap[e] = da * sa; dag[e] = da * ap[e] * sa * (1.0f - sa);
                dbp[e] = db * sb; dbg[e] = db * bp[e] * sb * (1.0f - sb);
            });
            // the five Linears on LN_in(z): weight gradients against xhat, input gradients summed into dzn
            dzn = T.f(P * cp);
            r.lin_bwd_w_ln(dap, P, ch, sv.xhat, cp, t.ap_w, t.ap_b, t.lni_g, t.lni_b); r.lin_bwd_x(dap, P, ch, t.ap_w, cp, dzn, cp, false);
            r.lin_bwd_w_ln(dag, P, ch, sv.xhat, cp, t.ag_w, t.ag_b, t.lni_g, t.lni_b); r.lin_bwd_x(dag, P, ch, t.ag_w, cp, dzn, cp, true);
            r.lin_bwd_w_ln(dbp, P, ch, sv.xhat, cp, t.bp_w, t.bp_b, t.lni_g, t.lni_b); r.lin_bwd_x(dbp, P, ch, t.bp_w, cp, dzn, cp, true);
            r.lin_bwd_w_ln(dbg, P, ch, sv.xhat, cp, t.bg_w, t.bg_b, t.lni_g, t.lni_b); r.lin_bwd_x(dbg, P, ch, t.bg_w, cp, dzn, cp, true);
            r.lin_bwd_w_ln(dgl, P, cp, sv.xhat, cp, t.g_w, t.g_b, t.lni_g, t.lni_b); r.lin_bwd_x(dgl, P, cp, t.g_w, cp, dzn, cp, true);
        }
        r.ln_bwd(dzn, sv.xhat, sv.rstd, t.lni_g, t.lni_b, dP, P, cp, true);
        T.off = mark;
    };
    for (int l = L - 1; l >= 0; --l) {
        const PairOff& o = O.pair[l];
        TransSave& sv = ps[l].tr;
        size_t mark = T.off;
        // z_out = (z + o m) m:  dz = dP m (kept in dP);  do = dz m
        float* dot = T.f(P * cp);
        if (cp % 4 == 0) {
            r.ew(P * cp / 4, [=] __device__(long long e4) {
                const long long e = e4 * 4, row = e / cp; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
                const float m = rm[bi] * rm[b * N + j];
                float4 g = *reinterpret_cast<const float4*>(dP + e);
                g.x *= m; g.y *= m; g.z *= m; g.w *= m;
                *reinterpret_cast<float4*>(dP + e) = g;
                *reinterpret_cast<float4*>(dot + e) = make_float4(g.x * m, g.y * m, g.z * m, g.w * m);
            });
        } else
        r.ew(P * cp, [=] __device__(long long e) {
            const long long row = e / cp; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
            const float m = rm[bi] * rm[b * N + j];
            const float g = dP[e] * m;
            dP[e] = g; dot[e] = g * m;
        });
        r.lin_bwd_w(dot, P, cp, sv.h, nh, nh, o.w2, (long long)o.b2);
        float* dh = T.f(P * nh);
        r.lin_bwd_x(dot, P, cp, o.w2, nh, dh, nh, false, sv.h, dry ? nullptr : sv.hmask);          // through the ReLU
        r.lin_bwd_w_ln(dh, P, nh, sv.xhat, cp, o.w1, o.b1, o.ln_g, o.ln_b);
        float* dzn = dot;
        r.lin_bwd_x(dh, P, nh, o.w1, cp, dzn, cp, false);
        r.ln_bwd(dzn, sv.xhat, sv.rstd, o.ln_g, o.ln_b, dP, P, cp, true);
        T.off = mark;
        tri_bwd(o.in, ps[l].in, false, (uint32_t)(4 * l + 1), &fold_tri[((size_t)l * 2 + 1) * 7]);
        tri_bwd(o.out, ps[l].out, true, (uint32_t)(4 * l), &fold_tri[((size_t)l * 2) * 7]);
    }
    // ---- pair feature net and single feature net
    {
        size_t mark = T.off;
        if (cp % 4 == 0) {
            r.ew(P * cp / 4, [=] __device__(long long e4) {
                const long long e = e4 * 4, row = e / cp; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
                const float m = rm[bi] * rm[b * N + j];
                float4 g = *reinterpret_cast<const float4*>(dP + e);
                g.x *= m; g.y *= m; g.z *= m; g.w *= m;
                *reinterpret_cast<float4*>(dP + e) = g;
            });
        } else
        r.ew(P * cp, [=] __device__(long long e) {
            const long long row = e / cp; const int j = (int)(row % N); const long long bi = row / N; const long long b = bi / N;
            dP[e] *= rm[bi] * rm[b * N + j];
        });
        float* dpi = T.f((size_t)M * cp); float* dpj = T.f((size_t)M * cp);
        if (!dry) launch_pair_sum_bwd(st, dP, dpi, dpj, B, N, cp);
        if (dtrans_out) {   // d / d trans: direct term of z = trans - t_out / rescale, the initial frames' translations, the template distance bins
            float* dFt = T.f((size_t)P * kt); float* dtr = T.f((size_t)M * 3);
            r.lin_bwd_x(dP, P, cp, O.wt, kt, dFt, kt, false);
            if (!dry) {
                (void)hipMemsetAsync(dtr, 0, (size_t)M * 3 * 4, st);
                launch_pair_features_bwd(st, dFt, kt, tr, rm, dtr, B, N, nbin, d.template_dist_min, d.template_dist_step);
            }
            const float sc = d.rescale; const float* dT0 = dTn;
            r.ew((long long)M * 3, [=] __device__(long long i) { dtrans_out[i] = dzp[i] + sc * (dT0[i] + dtr[i]); });
        }
        r.lin_bwd_w(dP, P, cp, Fm, nf, kt, O.wt, -1);
        r.lin_bwd_w(dP, P, cp, Fm + kt, nf, km, O.wm, -1);
        r.lin_bwd_w(dP, P, cp, Fm + kt + km, nf, kr, O.wrel, -1);
        r.lin_bwd_w(dpi, M, cp, s0, cs, cs, O.wi, -1); r.lin_bwd_x(dpi, M, cp, O.wi, cs, ds, cs, true);
        r.lin_bwd_w(dpj, M, cp, s0, cs, cs, O.wj, -1); r.lin_bwd_x(dpj, M, cp, O.wj, cs, ds, cs, true);
        r.ew((long long)M * cs, [=] __device__(long long i) { ds[i] *= rm[i / cs]; });
        r.lin_bwd_w(ds, M, cs, xs, ldx, nsi, O.single_w, -1);
        T.off = mark;
    }
    // every weight gradient taken against a LayerNorm's xhat (lin_bwd_w_ln: exactly the Linears of the fold table) still lacks its beta
    // part, dW[o][k] += beta[k] db[o]; the bias gradients are final now
    if (!dry && Gd && !fold_plan.empty()) {
        int max_ok = 0;
        for (const FoldEntry& e : fold_plan) max_ok = std::max(max_ok, e.O * e.K);
        ProfScope ps_(h, st, KC_TR_EW);
        launch_rank1_table(st, Gd, Wd, h->train->fold_tab, (int)fold_plan.size(), max_ok);
    }
    if (kept_bytes) *kept_bytes = K.peak;
    if (tmp_bytes) *tmp_bytes = T.peak;
    return GENIE_OK;
}

static int train_alloc(genie_ctx* h, hipStream_t st, size_t kb, size_t tb);
int genie_train_forward_backward(genie_handle_t h, genie_stream_t stream, const float* weights, float* grads, const float* trans, const float* rots,
                                 const int32_t* timesteps, const float* z_target, const int8_t* quat_codes, float condition_loss_weight,
                                 const genie_train_opts_t* opts, float* losses_out, float* z_pred_out) {
    if (!h) return GENIE_E_ARG;
    if (!h->have_tables || !h->have_feats) { TR_ERR("genie_train_forward_backward: call genie_set_tables and genie_prepare_features first"); return GENIE_E_STATE; }
    if (!weights || !grads || !trans || !rots || !timesteps || !z_target || !opts || !losses_out) { TR_ERR("genie_train_forward_backward: null argument"); return GENIE_E_ARG; }
    if (h->d.c_s > 512 || h->d.c_p > 512 || h->d.c_hidden_mul > 512) { TR_ERR("training path: channel widths above 512 are not supported"); return GENIE_E_ARG; }
    if (hipSetDevice(h->device) != hipSuccess) return GENIE_E_HIP;
    hipStream_t st = (hipStream_t)stream;
    size_t kb = 0, tb = 0;
    train_run(h, st, true, weights, grads, trans, rots, timesteps, z_target, quat_codes, condition_loss_weight, *opts, losses_out, z_pred_out, &kb, &tb);
    if (int rc0 = train_alloc(h, st, kb, tb)) return rc0;
    h->train_gemm_flop = 0.0;
    const int rc = train_run(h, st, false, weights, grads, trans, rots, timesteps, z_target, quat_codes, condition_loss_weight, *opts, losses_out,
                             z_pred_out, nullptr, nullptr);
    if (rc) return rc;
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { TR_ERR("genie_train_forward_backward: %s", hipGetErrorString(e)); return GENIE_E_HIP; }
    return GENIE_OK;
}

static int train_alloc(genie_ctx* h, hipStream_t st, size_t kb, size_t tb) {
    if (!h->train) h->train = new genie_train_ws();
    if (kb > h->train->kept_bytes) {
        if (h->train->kept) { (void)hipStreamSynchronize(st); (void)hipFree(h->train->kept); h->train->kept = nullptr; h->train->kept_bytes = 0; }
        if (hipMalloc(&h->train->kept, kb) != hipSuccess) { TR_ERR("training workspace: hipMalloc(%zu MB) failed", kb >> 20); return GENIE_E_NOMEM; }
        h->train->kept_bytes = kb;
    }
    if (tb > h->train->tmp_bytes) {
        if (h->train->tmp) { (void)hipStreamSynchronize(st); (void)hipFree(h->train->tmp); h->train->tmp = nullptr; h->train->tmp_bytes = 0; }
        if (hipMalloc(&h->train->tmp, tb) != hipSuccess) { TR_ERR("training scratch: hipMalloc(%zu MB) failed", tb >> 20); return GENIE_E_NOMEM; }
        h->train->tmp_bytes = tb;
    }
    return GENIE_OK;
}

int genie_denoise_vjp(genie_handle_t h, genie_stream_t stream, const float* weights, const float* trans, const float* rots, const int32_t* timesteps,
                      const int8_t* quat_codes, const float* dz, float* z_out, float* dtrans_out) {
    if (!h) return GENIE_E_ARG;
    if (!h->have_tables || !h->have_feats) { TR_ERR("genie_denoise_vjp: call genie_set_tables and genie_prepare_features first"); return GENIE_E_STATE; }
    if (!weights || !trans || !rots || !timesteps || !dz || !dtrans_out) { TR_ERR("genie_denoise_vjp: null argument"); return GENIE_E_ARG; }
    if (hipSetDevice(h->device) != hipSuccess) return GENIE_E_HIP;
    hipStream_t st = (hipStream_t)stream;
    genie_train_opts_t opt{};
    size_t kb = 0, tb = 0;
    train_run(h, st, true, weights, nullptr, trans, rots, timesteps, nullptr, quat_codes, 1.f, opt, nullptr, z_out, &kb, &tb, dz, dtrans_out);
    if (int rc = train_alloc(h, st, kb, tb)) return rc;
    train_run(h, st, false, weights, nullptr, trans, rots, timesteps, nullptr, quat_codes, 1.f, opt, nullptr, z_out, nullptr, nullptr, dz, dtrans_out);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { TR_ERR("genie_denoise_vjp: %s", hipGetErrorString(e)); return GENIE_E_HIP; }
    return GENIE_OK;
}

size_t genie_train_workspace_bytes(genie_handle_t h) { return h && h->train ? h->train->kept_bytes + h->train->tmp_bytes : 0; }
size_t genie_train_kept_bytes(genie_handle_t h) { return h && h->train ? h->train->kept_bytes : 0; }
double genie_train_gemm_flop(genie_handle_t h) { return h ? h->train_gemm_flop : 0.0; }

int genie_train_gemm(genie_stream_t stream, const genie_gemm_desc_t* q, const float* a, const float* b, float* c, const float* bias,
                     const float* gate, float* asum) {
    if (!q || !a || !b || !c || q->M <= 0 || q->N <= 0 || q->K <= 0 || q->batch <= 0 || q->nb2 <= 0 || q->batch % q->nb2 != 0 || q->nsplit <= 0)
        return -1;
    if ((q->nsplit > 1 && q->mode != 2) || q->mode < 0 || q->mode > 2) return -1;
    if (asum && !(q->batch == 1 && q->am == 1 && q->ak == q->M)) return -1;
    GemmP p{a, b, c, bias, q->M, q->N, q->K, q->am, q->ak, q->bk, q->bn, q->cm, q->cn, q->batch, q->nb2, q->a1, q->a2, q->b1, q->b2, q->c1, q->c2,
            q->nsplit, q->alpha, q->mode};
    p.relu = q->relu; p.gate = gate; p.asum = asum;
    if (q->cblk > 0) {
        const int64_t span = q->cblk_m ? q->M : q->N;
        if (q->batch != 1 || gate || (span + q->cblk - 1) / q->cblk > 8) return -1;
        p.cblk = q->cblk; p.cblk_m = q->cblk_m ? 1 : 0;
        for (int t = 0; t < 8; ++t) { p.ctab[t] = q->ctab[t]; p.atab[t] = q->atab[t]; }
    }
    launch_gemm(static_cast<hipStream_t>(stream), p, q->terms);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
