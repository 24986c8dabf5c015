// Host side of libgenie_hip.so: handle, weight repacking, workspace, the
// per-step launch sequence and the device-resident reverse loop.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <cmath>
#include <new>
#include <string>
#include <vector>

#include "common.h"

void pair_kernels_init();
void pair_fused_kernels_init();
void single_kernels_init(const genie_dims_t& d, int n_max);
void launch_scale_copy(genie_ctx* h, hipStream_t st, const float* in, float* out, int n, float s);
size_t ipa_attn_lds(const genie_dims_t& d, int N);

static char g_create_err[512] = "";

#define SET_ERR(h, ...) do { snprintf((h)->err, sizeof((h)->err), __VA_ARGS__); } while (0)
#define HIP_TRY(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    SET_ERR(h, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return GENIE_E_HIP; } } while (0)

static const char* kKernelNames[KC_COUNT] = {
    "single_input", "gemm_rows", "layernorm_rows", "pair_static", "pair_init", "trimul_proj", "trimul_contract",
    "trimul_out", "pair_transition", "ipa_bias", "ipa_prep", "ipa_attn", "bb_update", "struct_rows", "p_sample_frenet", "misc",
    "pair_fused_a", "pair_fused_b", "train_gemm", "train_elementwise", "train_layernorm", "train_transpose", "train_ipa", "train_misc"};

// ------------------------------------------------------------------ profiling
void prof_begin(genie_ctx* h, hipStream_t st, int cls) {
    if (h->prof_n == h->prof_cap) {
        const int nc = h->prof_cap ? h->prof_cap * 2 : 1024;
        h->prof_recs = (genie_ctx::ProfRec*)realloc(h->prof_recs, nc * sizeof(genie_ctx::ProfRec));
        h->prof_cap = nc;
    }
    genie_ctx::ProfRec& r = h->prof_recs[h->prof_n];
    r.cls = cls;
    (void)hipEventCreate(&r.a);
    (void)hipEventCreate(&r.b);
    (void)hipEventRecord(r.a, st);
}
void prof_end(genie_ctx* h, hipStream_t st) {
    (void)hipEventRecord(h->prof_recs[h->prof_n].b, st);
    h->prof_n++;
}

// ------------------------------------------------------------------ dims
static int single_in(const genie_dims_t& d) { return d.c_pos_emb + d.c_chain_emb + d.c_timestep_emb + 23; }
static int ipa_proj_n(const genie_dims_t& d) {
    return d.n_head_ipa * (3 * d.c_hidden_ipa + 3 * d.n_qk_point + 3 * (d.n_qk_point + d.n_v_point));
}
static int ipa_cat_n(const genie_dims_t& d) { return d.n_head_ipa * (d.c_p + d.c_hidden_ipa + 4 * d.n_v_point); }

static const char* check_dims(const genie_dims_t& d) {
    if (d.c_p != 128 || d.c_hidden_mul != 128) return "kernels are specialised for c_p == c_hidden_mul == 128";
    if (d.c_s % 8 || d.c_s > 512) return "c_s must be a multiple of 8 and <= 512";
    if (d.template_dist_n_bin > 40 || d.template_dist_n_bin < 1) return "template_dist_n_bin must be in [1, 40]";
    if (d.n_head_ipa > 16) return "n_head_ipa must be <= 16";
    if (d.pair_transition_n < 1 || d.n_timestep < 1 || d.relpos_k < 0) return "bad dims";
    if (d.n_structure_layer < 1 || d.n_structure_block < 1 || d.n_pair_transform_layer < 0) return "bad layer counts";
    if ((d.n_head_ipa * d.c_hidden_ipa) % 4 || ipa_cat_n(d) % 4 || ipa_proj_n(d) % 4) return "IPA widths must be multiples of 4";
    return nullptr;
}

size_t genie_weight_count(const genie_dims_t* dp) {
    const genie_dims_t& d = *dp;
    size_t n = 0;
    const size_t cs = d.c_s, cp = d.c_p, ch = d.c_hidden_mul;
    n += cs * single_in(d);
    n += 2 * cp * cs + cp * (2 * d.relpos_k + 3) + cp * (d.template_dist_n_bin + 6) + cp * (d.template_dist_n_bin + 2);
    const size_t tm = 4 * (ch * cp + ch) + (cp * cp + cp) + (cp * ch + cp) + 2 * cp + 2 * ch;
    const size_t pt = 2 * cp + (d.pair_transition_n * cp * cp + d.pair_transition_n * cp) + (cp * d.pair_transition_n * cp + cp);
    n += (size_t)d.n_pair_transform_layer * (2 * tm + pt);
    const size_t H = d.n_head_ipa, C = d.c_hidden_ipa, Pq = d.n_qk_point, Pv = d.n_v_point;
    size_t sl = H;
    sl += (H * C) * cs + H * C + (2 * H * C) * cs + 2 * H * C + (3 * H * Pq) * cs + 3 * H * Pq + (3 * H * (Pq + Pv)) * cs + 3 * H * (Pq + Pv);
    sl += H * cp + H + cs * ipa_cat_n(d) + cs;
    sl += 2 * cs + 3 * (cs * cs + cs) + 2 * cs + 6 * cs + 6;
    n += (size_t)d.n_structure_layer * sl;
    return n;
}

// ------------------------------------------------------------------ create / destroy
int genie_create(const genie_dims_t* dims, int device, genie_handle_t* out) {
    if (!dims || !out) { snprintf(g_create_err, sizeof g_create_err, "genie_create: null argument"); return GENIE_E_ARG; }
    if (const char* m = check_dims(*dims)) { snprintf(g_create_err, sizeof g_create_err, "genie_create: %s", m); return GENIE_E_ARG; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        snprintf(g_create_err, sizeof g_create_err, "genie_create: no HIP device %d (found %d)", device, ndev);
        return GENIE_E_HIP;
    }
    genie_ctx* h = new (std::nothrow) genie_ctx();
    if (!h) return GENIE_E_NOMEM;
    memset(h, 0, sizeof(*h));
    h->d = *dims;
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete h; snprintf(g_create_err, sizeof g_create_err, "hipSetDevice failed"); return GENIE_E_HIP; }
    h->pair = new PairLayerW[dims->n_pair_transform_layer > 0 ? dims->n_pair_transform_layer : 1]();
    h->st = new StructLayerW[dims->n_structure_layer]();
    pair_kernels_init();
    pair_fused_kernels_init();
    {   // GENIE_MATH=f32: exact-f32 MFMA kernels; default: split-f16 ("hx", same accuracy class, see hx.h)
        const char* m = getenv("GENIE_MATH");
        h->hx = !(m && !strcmp(m, "f32"));
    }
    // second stream of the structure net (denoise_internal); a failure here only disables the split
    {   // Created with a NON-default priority: with the default one, a process that initialised RCCL first gets this stream on the
        // hardware queue of the caller's stream, and the two-stream structure net then runs the step 9 % SLOWER than one stream
        // (100.8 vs 108.0 batch-steps/s; highest or lowest priority: 109.3; without RCCL the priority changes nothing: 110.3).
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&h->st2, hipStreamNonBlocking, hi) != hipSuccess) h->st2 = nullptr;
    }
    if (h->st2 && (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
                   hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess)) {
        (void)hipStreamDestroy(h->st2);
        h->st2 = nullptr;
    }
    *out = h;
    return GENIE_OK;
}

static void free_batch(genie_ctx* h) {
    if (h->ws) { (void)hipFree(h->ws); h->ws = nullptr; h->ws_bytes = 0; }
    h->have_feats = false;
}

void genie_destroy(genie_handle_t h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    free_batch(h);
    train_ws_free(h);
    if (h->wdev) (void)hipFree(h->wdev);
    if (h->hxdev) (void)hipFree(h->hxdev);
    if (h->pos_tab) (void)hipFree(h->pos_tab);
    free(h->sched_host);
    for (int i = 0; i < h->prof_n; ++i) { (void)hipEventDestroy(h->prof_recs[i].a); (void)hipEventDestroy(h->prof_recs[i].b); }
    free(h->prof_recs);
    if (h->st2) { (void)hipStreamSynchronize(h->st2); (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join); (void)hipStreamDestroy(h->st2); }
    delete[] h->pair;
    delete[] h->st;
    delete h;
}

const char* genie_last_error(genie_handle_t h) { return h ? h->err : g_create_err; }
size_t genie_workspace_bytes(genie_handle_t h) { return h ? h->ws_bytes : 0; }

// ------------------------------------------------------------------ weights
namespace {
struct Img {
    std::vector<float> data;
    size_t add(size_t n) {            // 256-B aligned segment
        const size_t off = (data.size() + 63) & ~(size_t)63;
        data.resize(off + n, 0.f);
        return off;
    }
    // raw copy
    size_t raw(const float* src, size_t n) { const size_t o = add(n); memcpy(&data[o], src, n * sizeof(float)); return o; }
    // fragment-major pack of W[rows][cols] (row-major, ld = cols)
    size_t pack(const float* W, int rows, int cols) {
        const int NB = (rows + 31) / 32, KB = (cols + 7) / 8;
        const size_t o = add((size_t)NB * KB * 256);
        float* p = &data[o];
        for (int nb = 0; nb < NB; ++nb)
            for (int kb = 0; kb < KB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 4; ++e) {
                        const int r = nb * 32 + (lane & 31), k = kb * 8 + 4 * (lane >> 5) + e;
                        p[(((size_t)nb * KB + kb) * 64 + lane) * 4 + e] = (r < rows && k < cols) ? W[(size_t)r * cols + k] : 0.f;
                    }
        return o;
    }
};
// ---- "hx" images (hx.h): f16 hi/lo halves in MFMA fragment order, units in consumption order ----
uint16_t f2h(float f) {                 // f32 -> f16, round to nearest even, subnormals kept
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000);
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00 : 0x7C00));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00);           // >= 65520 rounds to infinity
    if (x < 0x38800000u) {                                            // below 2^-14: subnormal half = round(|f| 2^24)
        float a; memcpy(&a, &x, 4);
        return (uint16_t)(sign | (uint16_t)nearbyintf(a * 16777216.0f));
    }
    const uint32_t mant = x & 0x7FFFFFu, rem = mant & 0x1FFFu;
    uint32_t r = (((x >> 23) - 112u) << 10) | (mant >> 13);
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) ++r;           // a carry runs into the exponent as it should
    return (uint16_t)(sign | r);
}
float h2f(uint16_t v) {
    const int e = (v >> 10) & 31, m = v & 1023;
    float r = e == 0 ? ldexpf((float)m, -24) : e == 31 ? (m ? NAN : INFINITY) : ldexpf((float)(1024 + m), e - 25);
    return (v & 0x8000) ? -r : r;
}
float p2floor(double x) { return (x > 0 && std::isfinite(x)) ? (float)ldexp(1.0, (int)floor(log2(x))) : 1.0f; }
struct HxImg {
    std::vector<uint16_t> d;
    size_t begin() { d.resize((d.size() + 127) & ~(size_t)127); return d.size() * 2; }       // 256-B aligned byte offset
    // one unit: get(idx = lane & 31, half-wave h, e) -> scaled f32 value of slot (lane, e)
    template <class F> void unit(F get) {
        const size_t o = d.size();
        d.resize(o + 1024);
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const float v = get(lane & 31, lane >> 5, e);
                const uint16_t hi = f2h(v);
                d[o + lane * 8 + e] = hi;
                d[o + 512 + lane * 8 + e] = f2h(v - h2f(hi));
            }
    }
};
double max_abs(const float* w, size_t n) { double m = 0; for (size_t i = 0; i < n; ++i) m = std::max(m, (double)fabsf(w[i])); return m; }
// Cauchy-Schwarz bound of |W xhat + b| over LayerNorm outputs (|xhat|_2 <= sqrt(cols))
double hx_bound(const float* W, const float* b, int rows, int cols) {
    double m = 0;
    for (int r = 0; r < rows; ++r) {
        double ss = 0;
        for (int k = 0; k < cols; ++k) ss += (double)W[(size_t)r * cols + k] * W[(size_t)r * cols + k];
        m = std::max(m, sqrt(ss * cols) + fabs((double)b[r]));
    }
    return m;
}
constexpr float HX_SX = 1024.0f;        // LayerNorm outputs: |xhat| <= sqrt(128) -> < 11586 after scaling

struct Cur {
    const float* p; size_t left;
    const float* take(size_t n) { if (n > left) { left = 0; return nullptr; } const float* r = p; p += n; left -= n; return r; }
};
// LayerNorm affine folded into the Linear that consumes it:  W (g * xhat + beta) + b = (W diag g) xhat + (b + W beta).
// The kernels then only normalise (no gamma / beta traffic or FMAs per tile).  Sums in double.
void fold_ln(std::vector<float>& W, std::vector<float>& b, int rows, int cols, const float* g, const float* beta) {
    for (int r = 0; r < rows; ++r) {
        double acc = b[r];
        for (int k = 0; k < cols; ++k) acc += (double)W[(size_t)r * cols + k] * beta[k];
        b[r] = (float)acc;
        for (int k = 0; k < cols; ++k) W[(size_t)r * cols + k] *= g[k];
    }
}
std::vector<float> vcat(std::initializer_list<std::pair<const float*, size_t>> parts) {
    std::vector<float> v;
    for (auto& pr : parts) v.insert(v.end(), pr.first, pr.first + pr.second);
    return v;
}
}  // namespace

int genie_load_weights(genie_handle_t h, const float* blob, size_t n_floats) {
    if (!h || !blob) return GENIE_E_ARG;
    const genie_dims_t& d = h->d;
    if (n_floats != genie_weight_count(&d)) {
        SET_ERR(h, "genie_load_weights: got %zu floats, dims need %zu", n_floats, genie_weight_count(&d));
        return GENIE_E_ARG;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t cs = d.c_s, cp = d.c_p, ch = d.c_hidden_mul;
    Cur c{blob, n_floats};
    Img img;
    HxImg hx;
    std::vector<std::pair<float**, size_t>> fix;     // (pointer slot, offset)
    std::vector<std::pair<const unsigned char**, size_t>> hxfix;
    auto slot = [&](float** s, size_t off) { fix.push_back({s, off}); };
    // row-GEMM weight: the f32 fragment pack plus its hx image (units (nb, k16): 32 output columns x 16 k)
    struct GemmFix { float** slot; size_t hx_off; float inv_s; };
    std::vector<GemmFix> gfix;
    auto pack_gemm = [&](float** s, const float* W, int rows, int cols) {
        slot(s, img.pack(W, rows, cols));
        const float sw = p2floor(16384.0 / max_abs(W, (size_t)rows * cols));
        gfix.push_back({s, hx.begin(), 1.0f / sw});
        const int NB = (rows + 31) / 32, KC = (cols + 15) / 16;
        for (int nb = 0; nb < NB; ++nb)
            for (int kc = 0; kc < KC; ++kc)
                hx.unit([&](int j, int hh, int e) {
                    const int r = 32 * nb + j, k = 16 * kc + 8 * hh + e;
                    return (r < rows && k < cols) ? W[(size_t)r * cols + k] * sw : 0.f; });
    };

    size_t ones_off, zeros_off;
    {
        std::vector<float> one(512, 1.f), zero(512, 0.f);
        ones_off = img.raw(one.data(), one.size());
        zeros_off = img.raw(zero.data(), zero.size());
    }
    const int nsi = single_in(d);
    pack_gemm(&h->single_w, c.take(cs * nsi), (int)cs, nsi);
    {
        const float* wi = c.take(cp * cs);
        const float* wj = c.take(cp * cs);
        auto v = vcat({{wi, cp * cs}, {wj, cp * cs}});
        pack_gemm(&h->pij_w, v.data(), (int)(2 * cp), (int)cs);
    }
    {
        const int nr = 2 * d.relpos_k + 3;
        const float* w = c.take(cp * nr);
        std::vector<float> t((size_t)nr * cp);
        for (int o = 0; o < (int)cp; ++o) for (int k = 0; k < nr; ++k) t[(size_t)k * cp + o] = w[(size_t)o * nr + k];
        slot(&h->relpos_t, img.raw(t.data(), t.size()));
    }
    {   // template: K padded to 48, motif: K padded to 40 (LDS tile widths of k_pair_init / k_pair_static)
        const int kt = d.template_dist_n_bin + 6, km = d.template_dist_n_bin + 2;
        const float* wt = c.take(cp * kt);
        const float* wm = c.take(cp * km);
        std::vector<float> t((size_t)cp * 48, 0.f), m((size_t)cp * 40, 0.f);
        for (int o = 0; o < (int)cp; ++o) {
            for (int k = 0; k < kt; ++k) t[(size_t)o * 48 + k] = wt[(size_t)o * kt + k];
            for (int k = 0; k < km; ++k) m[(size_t)o * 40 + k] = wm[(size_t)o * km + k];
        }
        slot(&h->templ_w, img.pack(t.data(), (int)cp, 48));
        slot(&h->motif_w, img.pack(m.data(), (int)cp, 40));
    }
    struct TmSave { std::vector<float> w, gw, zw; float sp, sg, sgo, szo; };
    struct PtSave { std::vector<float> w1v, w2; float s1, s2; };
    std::vector<TmSave> tm_save(2 * (size_t)d.n_pair_transform_layer);
    std::vector<PtSave> pt_save(d.n_pair_transform_layer);
    for (int l = 0; l < d.n_pair_transform_layer; ++l) {
        PairLayerW& L = h->pair[l];
        for (int dir = 0; dir < 2; ++dir) {
            TriMulW& T = dir == 0 ? L.out : L.in;
            const float* ap_w = c.take(ch * cp); const float* ap_b = c.take(ch);
            const float* ag_w = c.take(ch * cp); const float* ag_b = c.take(ch);
            const float* bp_w = c.take(ch * cp); const float* bp_b = c.take(ch);
            const float* bg_w = c.take(ch * cp); const float* bg_b = c.take(ch);
            const float* g_w = c.take(cp * cp);  const float* g_b = c.take(cp);
            const float* z_w = c.take(cp * ch);  const float* z_b = c.take(cp);
            const float* li_g = c.take(cp); const float* li_b = c.take(cp);
            const float* lo_g = c.take(ch); const float* lo_b = c.take(ch);
            auto w = vcat({{ap_w, ch * cp}, {bp_w, ch * cp}, {ag_w, ch * cp}, {bg_w, ch * cp}});
            auto bb = vcat({{ap_b, ch}, {bp_b, ch}, {ag_b, ch}, {bg_b, ch}});
            fold_ln(w, bb, (int)(4 * ch), (int)cp, li_g, li_b);
            // gate halves (rows 2ch..4ch) pre-scaled by -log2(e): sigmoid(x) = 1 / (1 + exp2(x')) in the kernel
            for (size_t r = 2 * ch; r < 4 * ch; ++r) {
                bb[r] = (float)(-1.4426950408889634 * bb[r]);
                for (size_t k = 0; k < cp; ++k) w[r * cp + k] = (float)(-1.4426950408889634 * w[r * cp + k]);
            }
            std::vector<float> gw(g_w, g_w + cp * cp), gb(g_b, g_b + cp), zw(z_w, z_w + cp * ch), zb(z_b, z_b + cp);
            fold_ln(gw, gb, (int)cp, (int)cp, li_g, li_b);
            for (size_t r = 0; r < cp; ++r) {       // gate pre-scaled by -log2(e), as above
                gb[r] = (float)(-1.4426950408889634 * gb[r]);
                for (size_t k = 0; k < cp; ++k) gw[r * cp + k] = (float)(-1.4426950408889634 * gw[r * cp + k]);
            }
            fold_ln(zw, zb, (int)cp, (int)ch, lo_g, lo_b);
            slot(&T.proj_w, img.pack(w.data(), (int)(4 * ch), (int)cp));
            slot(&T.proj_b, img.raw(bb.data(), bb.size()));
            slot(&T.g_w, img.pack(gw.data(), (int)cp, (int)cp)); slot(&T.g_b, img.raw(gb.data(), cp));
            slot(&T.z_w, img.pack(zw.data(), (int)cp, (int)ch)); slot(&T.z_b, img.raw(zb.data(), cp));
            {   // hx images (hx.h).  a / b are bounded by their projection rows (the gate is in (0, 1)).
                HxTriW& X = T.hx;
                const float sp = p2floor(16384.0 / max_abs(w.data(), 2 * ch * cp)), sg = p2floor(16384.0 / max_abs(w.data() + 2 * ch * cp, 2 * ch * cp));
                const float sa = p2floor(32768.0 / hx_bound(w.data(), bb.data(), (int)ch, (int)cp));
                const float sb = p2floor(32768.0 / hx_bound(w.data() + ch * cp, bb.data() + ch, (int)ch, (int)cp));
                const float sgo = p2floor(16384.0 / max_abs(gw.data(), cp * cp)), szo = p2floor(16384.0 / max_abs(zw.data(), cp * ch));
                X.sx = HX_SX; X.cpa = sa / (HX_SX * sp); X.cpb = sb / (HX_SX * sp); X.cg = 1.0f / (HX_SX * sg); X.cx = 1.0f / (sa * sb);
                X.cgo = 1.0f / (HX_SX * sgo); X.cz = 1.0f / (HX_SX * szo);
                hxfix.push_back({&X.img_proj, hx.begin()});
                for (size_t pass = 0; pass < 8; ++pass)
                    for (int kc = 0; kc < 8; ++kc) {
                        hx.unit([&](int i, int hh, int e) { return w[(32 * pass + i) * cp + 16 * kc + 8 * hh + e] * sp; });
                        hx.unit([&](int i, int hh, int e) { return w[(2 * ch + 32 * pass + i) * cp + 16 * kc + 8 * hh + e] * sg; });
                    }
                hxfix.push_back({&X.img_out, hx.begin()});
                for (int st = 0; st < 4; ++st)
                    for (int obl = 0; obl < 2; ++obl)
                        for (int kc = 0; kc < 8; ++kc) {
                            const std::vector<float>& m = (st & 1) ? zw : gw;
                            const float sc = (st & 1) ? szo : sgo;
                            const size_t ob = 2 * (st >> 1) + obl;
                            hx.unit([&](int j, int hh, int e) { return m[(32 * ob + j) * cp + 16 * kc + 8 * hh + e] * sc; });
                        }
                std::vector<float> bp(4 * ch), bgs(cp), bzs(cp);
                for (size_t r = 0; r < 4 * ch; ++r) bp[r] = bb[r] * HX_SX * (r < 2 * ch ? sp : sg);
                for (size_t r = 0; r < cp; ++r) { bgs[r] = gb[r] * HX_SX * sgo; bzs[r] = zb[r] * HX_SX * szo; }
                slot(&X.bias_proj, img.raw(bp.data(), bp.size()));
                slot(&X.bgs, img.raw(bgs.data(), cp)); slot(&X.bzs, img.raw(bzs.data(), cp));
                tm_save[2 * l + dir] = TmSave{w, gw, zw, sp, sg, sgo, szo};
            }
            // the affine now lives in the weights: kernels that still take gamma / beta get (1, 0)
            slot(&T.ln_in_g, ones_off); slot(&T.ln_in_b, zeros_off);
            slot(&T.ln_out_g, ones_off); slot(&T.ln_out_b, zeros_off);
        }
        const size_t nh = (size_t)d.pair_transition_n * cp;
        const float* lg = c.take(cp); const float* lb = c.take(cp);
        const float* w1 = c.take(nh * cp); const float* b1 = c.take(nh);
        const float* w2 = c.take(cp * nh); const float* b2 = c.take(cp);
        {
            std::vector<float> w1v(w1, w1 + nh * cp), b1v(b1, b1 + nh);
            fold_ln(w1v, b1v, (int)nh, (int)cp, lg, lb);
            slot(&L.pt_ln_g, ones_off); slot(&L.pt_ln_b, zeros_off);
            slot(&L.pt_w1, img.pack(w1v.data(), (int)nh, (int)cp)); slot(&L.pt_b1, img.raw(b1v.data(), nh));
            // hx image: per hidden block of 32, 8 units of W1 (k-chunks) then 8 of W2 (chunk c, output block ob)
            HxTransW& X = L.hx_pt;
            const float s1 = p2floor(16384.0 / max_abs(w1v.data(), nh * cp)), s2 = p2floor(16384.0 / max_abs(w2, cp * nh));
            const float sh = p2floor(32768.0 / hx_bound(w1v.data(), b1v.data(), (int)nh, (int)cp));
            X.sx = HX_SX; X.c1 = sh / (HX_SX * s1); X.c2 = 1.0f / (sh * s2);
            hxfix.push_back({&X.img, hx.begin()});
            for (size_t hb = 0; hb < nh / 32; ++hb) {
                for (int kc = 0; kc < 8; ++kc)
                    hx.unit([&](int i, int hh, int e) { return w1v[(32 * hb + i) * cp + 16 * kc + 8 * hh + e] * s1; });
                for (int cc = 0; cc < 2; ++cc)
                    for (int ob = 0; ob < 4; ++ob)
                        hx.unit([&](int j, int hh, int e) {
                            return w2[(size_t)(32 * ob + j) * nh + 32 * hb + 16 * cc + (e & 3) + 8 * (e >> 2) + 4 * hh] * s2; });
            }
            std::vector<float> b1s(nh), b2s(cp);
            for (size_t r = 0; r < nh; ++r) b1s[r] = b1v[r] * HX_SX * s1;
            for (size_t r = 0; r < cp; ++r) b2s[r] = b2[r] * sh * s2;
            slot(&X.b1s, img.raw(b1s.data(), nh)); slot(&X.b2s, img.raw(b2s.data(), cp));
            pt_save[l] = PtSave{w1v, std::vector<float>(w2, w2 + cp * nh), s1, s2};
        }
        slot(&L.pt_w2, img.pack(w2, (int)cp, (int)nh)); slot(&L.pt_b2, img.raw(b2, cp));
    }
    {   // fused row-local chains (pair_fused_kernels.hip): the same scaled weights, every unit in the chained k order
        //   k = 16 kc + (e & 3) + 8 (e >> 2) + 4 hh   (the order in which a result tile's registers are the next B operand)
        auto ck = [](int kc, int hh, int e) { return 16 * kc + (e & 3) + 8 * (e >> 2) + 4 * hh; };
        auto emit_O = [&](const TmSave& t) {          // stages W_z{0,1}, W_z{2,3}, W_g{0,1}, W_g{2,3}; unit = 8 (block within the stage) + k-chunk
            for (int which = 0; which < 2; ++which)
                for (int st = 0; st < 2; ++st)
                    for (int obl = 0; obl < 2; ++obl)
                        for (int kc = 0; kc < 8; ++kc) {
                            const std::vector<float>& m = which == 0 ? t.zw : t.gw;
                            const float sc = which == 0 ? t.szo : t.sgo;
                            const size_t ob = 2 * st + obl;
                            hx.unit([&](int i, int hh, int e) { return m[(32 * ob + i) * cp + ck(kc, hh, e)] * sc; });
                        }
        };
        auto emit_T = [&](const PtSave& t) {          // per hidden block: 8 units of W1 (k-chunks), 8 of W2 (chunk c, output block ob) at 8 + 4c + ob
            const size_t nh = (size_t)d.pair_transition_n * cp;
            for (size_t hb = 0; hb < nh / 32; ++hb) {
                for (int kc = 0; kc < 8; ++kc)
                    hx.unit([&](int i, int hh, int e) { return t.w1v[(32 * hb + i) * cp + ck(kc, hh, e)] * t.s1; });
                for (int cc = 0; cc < 2; ++cc)
                    for (int ob = 0; ob < 4; ++ob)
                        hx.unit([&](int j, int hh, int e) { return t.w2[(size_t)(32 * ob + j) * nh + 32 * hb + ck(cc, hh, e)] * t.s2; });
            }
        };
        auto emit_P = [&](const TmSave& t) {          // 8 passes: unit 2kc = the pass's projection rows, 2kc + 1 = its gate rows
            for (size_t pass = 0; pass < 8; ++pass)
                for (int kc = 0; kc < 8; ++kc) {
                    hx.unit([&](int i, int hh, int e) { return t.w[(32 * pass + i) * cp + ck(kc, hh, e)] * t.sp; });
                    hx.unit([&](int i, int hh, int e) { return t.w[(2 * ch + 32 * pass + i) * cp + ck(kc, hh, e)] * t.sg; });
                }
        };
        for (int l = 0; l < d.n_pair_transform_layer; ++l) {
            hxfix.push_back({&h->pair[l].fa.img, hx.begin()});
            emit_O(tm_save[2 * l]); emit_P(tm_save[2 * l + 1]);
            hxfix.push_back({&h->pair[l].fb.img, hx.begin()});
            emit_O(tm_save[2 * l + 1]); emit_T(pt_save[l]);
            if (l + 1 < d.n_pair_transform_layer) emit_P(tm_save[2 * l + 2]);      // (the last block's chain ends with its transition)
        }
    }
    const size_t H = d.n_head_ipa, C = d.c_hidden_ipa, Pq = d.n_qk_point, Pv = d.n_v_point;
    const size_t ncat = ipa_cat_n(d);
    std::vector<float> wb_all, bb_all;
    for (int l = 0; l < d.n_structure_layer; ++l) {
        StructLayerW& S = h->st[l];
        const float* hw = c.take(H);
        const float* q_w = c.take(H * C * cs);            const float* q_b = c.take(H * C);
        const float* kv_w = c.take(2 * H * C * cs);       const float* kv_b = c.take(2 * H * C);
        const float* qp_w = c.take(3 * H * Pq * cs);      const float* qp_b = c.take(3 * H * Pq);
        const float* kp_w = c.take(3 * H * (Pq + Pv) * cs); const float* kp_b = c.take(3 * H * (Pq + Pv));
        const float* b_w = c.take(H * cp);                const float* b_b = c.take(H);
        const float* o_w = c.take(cs * ncat);             const float* o_b = c.take(cs);
        const float* li_g = c.take(cs); const float* li_b = c.take(cs);
        const float* t1w = c.take(cs * cs); const float* t1b = c.take(cs);
        const float* t2w = c.take(cs * cs); const float* t2b = c.take(cs);
        const float* t3w = c.take(cs * cs); const float* t3b = c.take(cs);
        const float* lt_g = c.take(cs); const float* lt_b = c.take(cs);
        const float* bw = c.take(6 * cs); const float* bbs = c.take(6);
        if (!bbs) { SET_ERR(h, "genie_load_weights: blob too short"); return GENIE_E_ARG; }
        slot(&S.head_w, img.raw(hw, H));
        auto w = vcat({{q_w, H * C * cs}, {kv_w, 2 * H * C * cs}, {qp_w, 3 * H * Pq * cs}, {kp_w, 3 * H * (Pq + Pv) * cs}});
        auto bv = vcat({{q_b, H * C}, {kv_b, 2 * H * C}, {qp_b, 3 * H * Pq}, {kp_b, 3 * H * (Pq + Pv)}});
        pack_gemm(&S.proj_w, w.data(), ipa_proj_n(d), (int)cs);
        slot(&S.proj_b, img.raw(bv.data(), bv.size()));
        wb_all.insert(wb_all.end(), b_w, b_w + H * cp);
        bb_all.insert(bb_all.end(), b_b, b_b + H);
        pack_gemm(&S.out_w, o_w, (int)cs, (int)ncat); slot(&S.out_b, img.raw(o_b, cs));
        slot(&S.ln_ipa_g, img.raw(li_g, cs)); slot(&S.ln_ipa_b, img.raw(li_b, cs));
        pack_gemm(&S.t1_w, t1w, (int)cs, (int)cs); slot(&S.t1_b, img.raw(t1b, cs));
        pack_gemm(&S.t2_w, t2w, (int)cs, (int)cs); slot(&S.t2_b, img.raw(t2b, cs));
        pack_gemm(&S.t3_w, t3w, (int)cs, (int)cs); slot(&S.t3_b, img.raw(t3b, cs));
        slot(&S.ln_tr_g, img.raw(lt_g, cs)); slot(&S.ln_tr_b, img.raw(lt_b, cs));
        slot(&S.bb_w, img.raw(bw, 6 * cs)); slot(&S.bb_b, img.raw(bbs, 6));
    }
    pack_gemm(&h->ipa_bias_w, wb_all.data(), (int)(d.n_structure_layer * H), (int)cp);    // f32 pack + hx image (k_ipa_bias_hx)
    slot(&h->ipa_bias_b, img.raw(bb_all.data(), bb_all.size()));
    if (c.left != 0) { SET_ERR(h, "genie_load_weights: %zu floats left over", c.left); return GENIE_E_ARG; }

    if (h->wdev) { (void)hipFree(h->wdev); h->wdev = nullptr; }
    img.add(64);
    HIP_TRY(h, hipMalloc((void**)&h->wdev, img.data.size() * sizeof(float)));
    HIP_TRY(h, hipMemcpy(h->wdev, img.data.data(), img.data.size() * sizeof(float), hipMemcpyHostToDevice));
    h->wdev_floats = img.data.size();
    for (auto& f : fix) *f.first = h->wdev + f.second;
    if (h->hxdev) { (void)hipFree(h->hxdev); h->hxdev = nullptr; }
    hx.begin();
    HIP_TRY(h, hipMalloc((void**)&h->hxdev, hx.d.size() * 2 + 256));
    HIP_TRY(h, hipMemcpy(h->hxdev, hx.d.data(), hx.d.size() * 2, hipMemcpyHostToDevice));
    for (auto& f : hxfix) *f.first = h->hxdev + f.second;
    h->n_hxg = 0;
    for (auto& g : gfix)
        if (h->n_hxg < 64) h->hxg[h->n_hxg++] = HxGemmW{*g.slot, h->hxdev + g.hx_off, g.inv_s};
    h->have_weights = true;
    return GENIE_OK;
}

// ------------------------------------------------------------------ tables
int genie_set_tables(genie_handle_t h, const float* pos_tab, int n_pos, const float* chain_tab, int n_chain,
                     const float* t_tab, const float* sched) {
    if (!h || !pos_tab || !chain_tab || !t_tab || !sched || n_pos < 1 || n_chain < 1) return GENIE_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    const genie_dims_t& d = h->d;
    const size_t T1 = d.n_timestep + 1;
    const size_t a = (size_t)n_pos * d.c_pos_emb, b = (size_t)n_chain * d.c_chain_emb, c = T1 * d.c_timestep_emb, s = 4 * T1;
    if (h->pos_tab) (void)hipFree(h->pos_tab);
    HIP_TRY(h, hipMalloc((void**)&h->pos_tab, (a + b + c + s) * sizeof(float)));
    h->chain_tab = h->pos_tab + a;
    h->t_tab = h->chain_tab + b;
    h->sched = h->t_tab + c;
    HIP_TRY(h, hipMemcpy(h->pos_tab, pos_tab, a * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->chain_tab, chain_tab, b * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->t_tab, t_tab, c * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->sched, sched, s * sizeof(float), hipMemcpyHostToDevice));
    free(h->sched_host);
    h->sched_host = (float*)malloc(s * sizeof(float));
    memcpy(h->sched_host, sched, s * sizeof(float));
    h->n_pos = n_pos;
    h->n_chain = n_chain;
    h->have_tables = true;
    return GENIE_OK;
}

// ------------------------------------------------------------------ batch binding
__global__ void k_mask_to_float(const int32_t* m, float* f, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f[i] = (float)m[i];
}

int genie_prepare_features(genie_handle_t h, genie_stream_t stream, int B, int N, const genie_features_t* f) {
    if (!h || !f || B < 1 || N < 2) return GENIE_E_ARG;
    if (!h->have_weights) { SET_ERR(h, "genie_prepare_features: weights not loaded"); return GENIE_E_STATE; }
    const genie_dims_t& d = h->d;
    if (ipa_attn_lds(d, N) > 160 * 1024) { SET_ERR(h, "N = %d exceeds the attention kernel's LDS budget", N); return GENIE_E_ARG; }
    if ((size_t)24 * N * sizeof(float) > 160 * 1024) { SET_ERR(h, "N = %d exceeds the Frenet kernel's LDS budget (96 B per residue)", N); return GENIE_E_ARG; }
    {   // the pair kernels address [B,N,N,128] f32 tensors with 32-bit buffer offsets (SGPR soffset + VGPR voffset)
        const size_t np = (size_t)((N + 31) / 32 * 32);
        const size_t pair_bytes = (size_t)B * np * np * d.c_p * sizeof(float);
        if (pair_bytes >= ((size_t)1 << 31)) {
            SET_ERR(h, "batch %d x N %d: a pair tensor of %zu bytes exceeds the 2 GiB the kernels address; split the batch", B, N, pair_bytes);
            return GENIE_E_ARG;
        }
    }
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int NP = (N + 31) / 32 * 32;
    const size_t M = (size_t)B * N, P = M * N;
    const size_t cs = d.c_s, cp = d.c_p;
    const int H = d.n_head_ipa, C = d.c_hidden_ipa, Pq = d.n_qk_point, Pv = d.n_v_point;
    const int ldx = (single_in(d) + 7) / 8 * 8;

    // carve plan (bytes, 256-B aligned)
    struct Seg { void** p; size_t bytes; };
    std::vector<Seg> segs;
    auto want = [&](void* pp, size_t bytes) { segs.push_back({(void**)pp, (bytes + 255) & ~(size_t)255}); };
    want(&h->p, P * cp * 4); want(&h->pstatic, P * cp * 4);
    want(&h->acm, (size_t)B * cp * NP * NP * 4); want(&h->bcm, (size_t)B * cp * NP * NP * 4); want(&h->xcm, (size_t)B * cp * NP * NP * 4);
    want(&h->ipa_bias, (size_t)d.n_structure_layer * H * P * 4);
    want(&h->xsingle, M * ldx * 4);
    want(&h->s0, M * cs * 4); want(&h->s, M * cs * 4); want(&h->s1, M * cs * 4); want(&h->s2, M * cs * 4);
    want(&h->h1, M * cs * 4); want(&h->h2, M * cs * 4);
    want(&h->pij, M * 2 * cp * 4); want(&h->proj, M * ipa_proj_n(d) * 4); want(&h->cat, M * ipa_cat_n(d) * 4);
    want(&h->kT, M * H * C * 4); want(&h->v, M * H * C * 4); want(&h->qp, M * H * Pq * 3 * 4);
    want(&h->kpT, M * H * Pq * 3 * 4); want(&h->vp, M * H * Pv * 3 * 4);
    want(&h->vf, ipa_vf_floats(d, B, N) * 4); want(&h->vmax, M * 2 * 4);
    want(&h->rots_w, M * 9 * 4); want(&h->trans_w, M * 3 * 4); want(&h->loop_z, M * 3 * 4);
    want(&h->tsteps, (size_t)B * 4); want(&h->rmaskf, M * 4); want(&h->pmax, 4); want(&h->spart, 3 * M * cs * 4);      // SR_KSPLIT slices
    want(&h->f_aatype, M * 20 * 4); want(&h->f_rmask, M * 4); want(&h->f_ridx, M * 4); want(&h->f_cidx, M * 4);
    want(&h->f_pos, M * 3 * 4); want(&h->f_fsm, M); want(&h->f_fstm, P); want(&h->f_ifm, M);
    size_t total = 0;
    for (auto& s : segs) total += s.bytes;
    if (total > h->ws_bytes) {
        free_batch(h);
        hipError_t e = hipMalloc(&h->ws, total);
        if (e != hipSuccess) { SET_ERR(h, "workspace hipMalloc(%zu MB) failed: %s", total >> 20, hipGetErrorString(e)); return GENIE_E_NOMEM; }
        h->ws_bytes = total;
    }
    size_t off = 0;
    for (auto& s : segs) { *s.p = (char*)h->ws + off; off += s.bytes; }
    h->B = B; h->N = N; h->NP = NP;

    HIP_TRY(h, hipMemcpyAsync(h->f_aatype, f->aatype, M * 20 * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->f_pos, f->atom_positions, M * 3 * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->f_rmask, f->residue_mask, M * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->f_ridx, f->residue_index, M * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->f_cidx, f->chain_index, M * 4, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->f_fsm, f->fixed_sequence_mask, M, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->f_fstm, f->fixed_structure_mask, P, hipMemcpyDeviceToDevice, st));
    HIP_TRY(h, hipMemcpyAsync(h->f_ifm, f->interface_mask, M, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_mask_to_float, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, h->f_rmask, h->rmaskf, (int)M);
    // channel-major TriMul operands rely on zero padding beyond N
    HIP_TRY(h, hipMemsetAsync(h->acm, 0, (size_t)B * cp * NP * NP * 4, st));
    HIP_TRY(h, hipMemsetAsync(h->vf, 0, ipa_vf_floats(d, B, N) * 4, st));          // fragment rows beyond N and the padding columns stay zero
    HIP_TRY(h, hipMemsetAsync(h->bcm, 0, (size_t)B * cp * NP * NP * 4, st));
    single_kernels_init(d, N);
    {   // does this batch condition on structure (any fixed_structure_mask entry set)?  If not, the motif term is identically
        // zero and the step-invariant pair term is a table row per pair: k_pair_init looks it up instead of reading pstatic
        unsigned any = 0;
        HIP_TRY(h, hipMemsetAsync(h->pmax, 0, sizeof(unsigned), st));
        launch_any_nonzero(h, st, h->f_fstm, P, h->pmax);
        HIP_TRY(h, hipMemcpyAsync(&any, h->pmax, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipStreamSynchronize(st));
        h->has_motif = any != 0;
    }
    launch_pair_static(h, st);
    HIP_TRY(h, hipGetLastError());
    h->have_feats = true;
    return GENIE_OK;
}

// ------------------------------------------------------------------ denoiser
static int denoise_internal(genie_ctx* h, hipStream_t st, const float* trans, const float* rots, const int32_t* ts_dev,
                            const int8_t* codes, float* z_out, const genie_taps_t* taps) {
    const genie_dims_t& d = h->d;
    const int B = h->B, N = h->N, M = B * N;
    const size_t P = (size_t)M * N;
    const int cs = d.c_s, cp = d.c_p;
    const int ldx = (single_in(d) + 7) / 8 * 8;
    auto d2d = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st); };

    h->hx_launches = 1;      // tile directions alternate from a fixed start: pair_init writes p forwards, the first projection reads it backwards
    launch_scale_copy(h, st, trans, h->trans_w, M * 3, d.rescale);
    HIP_TRY(h, d2d(h->rots_w, rots, (size_t)M * 9 * 4));

    // single feature net
    launch_single_input(h, st, ts_dev);
    launch_gemm_rows(h, st, h->xsingle, ldx, M, ldx, h->single_w, cs, nullptr, nullptr, 0, h->rmaskf, 0, h->s0, cs);
    if (taps && taps->s) HIP_TRY(h, d2d(taps->s, h->s0, (size_t)M * cs * 4));

    // pair feature net
    launch_gemm_rows(h, st, h->s0, cs, M, cs, h->pij_w, 2 * cp, nullptr, nullptr, 0, nullptr, 0, h->pij, 2 * cp);
    launch_pair_init(h, st, h->trans_w, h->rots_w, codes);
    if (taps && taps->p_init) HIP_TRY(h, d2d(taps->p_init, h->p, P * cp * 4));

    // pair transform net
    const bool fused = launch_pair_stack_fused(h, st, taps ? taps->p_trimul_out0 : nullptr, taps ? taps->p_layer0 : nullptr);
    for (int l = 0; l < (fused ? 0 : d.n_pair_transform_layer); ++l) {
        launch_trimul(h, st, h->pair[l].out, true);
        if (l == 0 && taps && taps->p_trimul_out0) HIP_TRY(h, d2d(taps->p_trimul_out0, h->p, P * cp * 4));
        launch_trimul(h, st, h->pair[l].in, false);
        launch_pair_transition(h, st, h->pair[l]);
        if (l == 0 && taps && taps->p_layer0) HIP_TRY(h, d2d(taps->p_layer0, h->p, P * cp * 4));
    }
    if (taps && taps->p) HIP_TRY(h, d2d(taps->p, h->p, P * cp * 4));

    // structure net
    launch_ipa_bias(h, st);
    HIP_TRY(h, d2d(h->s, h->s0, (size_t)M * cs * 4));
    if (taps && taps->states) HIP_TRY(h, d2d(taps->states, h->s0, (size_t)M * cs * 4));
    int n_state = 1;
    const int nproj = ipa_proj_n(d), ncat = ipa_cat_n(d);
    // The structure layers are chains of small, latency-bound launches (M = B N rows) around one HBM-bound one (the attention's pass
    // over p), and batch entries never meet in them: the two halves of the batch run their layers on two streams, so that one half's
    // small launches sit beside the other half's attention.  Every kernel takes a row / batch range; results are bit-identical to the
    // single-stream order.  (GENIE_NO_STRUCT_SPLIT=1, an odd path -- f32 arithmetic, other widths -- or one structure: one stream.)
    // Measured: +1.1 ... +2 % on the step at B N >= 2048 rows, -4 % at a few hundred (twice the launches): split from 1024 rows up.
    bool split = h->st2 && B >= 2 && M >= 1024 && ipa_attn_splits(h) && !getenv("GENIE_NO_STRUCT_SPLIT");
    for (int l = 0; split && l < d.n_structure_layer; ++l) split = struct_tail_fused(h, h->st[l]);
    const int n_half = split ? 2 : 1;       // (four quarters on four streams measured the same as two halves: +1.2 % vs +1.6 %)
    const int n_layers = d.n_structure_block * d.n_structure_layer;
    hipStream_t hstream[2] = {st, h->st2};
    // (Starting the second half half a layer late -- behind the first half's first attention pass -- measured no better than starting
    //  them together: +0.0 % against +1.1 %; the small launches do not get faster beside the attention's p stream.)
    if (split) {
        HIP_TRY(h, hipEventRecord(h->ev_fork, st));
        HIP_TRY(h, hipStreamWaitEvent(h->st2, h->ev_fork, 0));
    }
    for (int half = 0; half < n_half; ++half) {
        hipStream_t hs = hstream[half];
        const int b0 = (int)((long long)B * half / n_half), nb = (int)((long long)B * (half + 1) / n_half) - b0;
        const size_t r0 = (size_t)b0 * N, nr = (size_t)nb * N;
        int li = 0;
        for (int blk = 0; blk < d.n_structure_block; ++blk) {
            for (int l = 0; l < d.n_structure_layer; ++l, ++li) {
                const StructLayerW& S = h->st[l];
                const bool last = li == n_layers - 1;
                launch_gemm_rows(h, hs, h->s + r0 * cs, cs, (int)nr, cs, S.proj_w, nproj, S.proj_b, nullptr, 0, nullptr, 0, h->proj + r0 * nproj, nproj);
                launch_ipa_prep(h, hs, b0, nb);
                launch_ipa_attn(h, hs, l, S.head_w, b0, nb);
                if (li == 0 && taps && taps->ipa_cat0) HIP_TRY(h, hipMemcpyAsync(taps->ipa_cat0 + r0 * ncat, h->cat + r0 * ncat, nr * ncat * 4, hipMemcpyDeviceToDevice, hs));
                auto state_tap = [&]() -> hipError_t {
                    return (taps && taps->states) ? hipMemcpyAsync(taps->states + (size_t)(1 + li) * M * cs + r0 * cs, h->s + r0 * cs, nr * cs * 4, hipMemcpyDeviceToDevice, hs)
                                                  : hipSuccess;
                };
                if (launch_struct_tail(h, hs, S, last ? trans : nullptr, last ? z_out : nullptr, b0, nb)) { HIP_TRY(h, state_tap()); continue; }
                // (the separate launches: never with a split batch)
                launch_gemm_rows(h, hs, h->cat, ncat, M, ncat, S.out_w, cs, S.out_b, h->s, cs, nullptr, 0, h->s1, cs);
                launch_layernorm_rows(h, hs, h->s1, h->s2, M, cs, S.ln_ipa_g, S.ln_ipa_b);
                launch_gemm_rows(h, hs, h->s2, cs, M, cs, S.t1_w, cs, S.t1_b, nullptr, 0, nullptr, 1, h->h1, cs);
                launch_gemm_rows(h, hs, h->h1, cs, M, cs, S.t2_w, cs, S.t2_b, nullptr, 0, nullptr, 1, h->h2, cs);
                launch_gemm_rows(h, hs, h->h2, cs, M, cs, S.t3_w, cs, S.t3_b, h->s2, cs, nullptr, 0, h->s1, cs);
                launch_layernorm_rows(h, hs, h->s1, h->s, M, cs, S.ln_tr_g, S.ln_tr_b);
                launch_bb_update(h, hs, S, last ? trans : nullptr, last ? z_out : nullptr);
                HIP_TRY(h, state_tap());
            }
        }
    }
    if (split) {
        for (int q = 1; q < n_half; ++q) {
            HIP_TRY(h, hipEventRecord(h->ev_join, hstream[q]));
            HIP_TRY(h, hipStreamWaitEvent(st, h->ev_join, 0));
        }
    }
    (void)n_state;
    if (taps && taps->s_final) HIP_TRY(h, d2d(taps->s_final, h->s, (size_t)M * cs * 4));
    if (taps && taps->rots_out) HIP_TRY(h, d2d(taps->rots_out, h->rots_w, (size_t)M * 9 * 4));
    if (taps && taps->trans_out) launch_scale_copy(h, st, h->trans_w, taps->trans_out, M * 3, 1.0f / d.rescale);
    HIP_TRY(h, hipGetLastError());
    return GENIE_OK;
}

static int check_ready(genie_ctx* h, const char* who) {
    if (!h) return GENIE_E_ARG;
    if (!h->have_weights || !h->have_tables || !h->have_feats) {
        SET_ERR(h, "%s: call genie_load_weights, genie_set_tables and genie_prepare_features first", who);
        return GENIE_E_STATE;
    }
    if (hipSetDevice(h->device) != hipSuccess) { SET_ERR(h, "hipSetDevice failed"); return GENIE_E_HIP; }
    return GENIE_OK;
}

int genie_denoise(genie_handle_t h, genie_stream_t stream, const float* trans, const float* rots, const int32_t* timesteps,
                  const int8_t* quat_codes, float* z_out, const genie_taps_t* taps) {
    if (int rc = check_ready(h, "genie_denoise")) return rc;
    if (!trans || !rots || !timesteps || !z_out) { SET_ERR(h, "genie_denoise: null tensor"); return GENIE_E_ARG; }
    return denoise_internal(h, (hipStream_t)stream, trans, rots, timesteps, quat_codes, z_out, taps);
}

int genie_frenet(genie_handle_t h, genie_stream_t stream, const float* trans, float* rots_out) {
    if (!h || !h->have_feats) { if (h) SET_ERR(h, "genie_frenet: no batch bound"); return h ? GENIE_E_STATE : GENIE_E_ARG; }
    if (!trans || !rots_out) return GENIE_E_ARG;
    HIP_TRY(h, hipSetDevice(h->device));
    launch_frenet(h, (hipStream_t)stream, 0, 0, 0.f, const_cast<float*>(trans), rots_out, nullptr, nullptr);
    HIP_TRY(h, hipGetLastError());
    return GENIE_OK;
}

int genie_adam_step(genie_stream_t stream, size_t n, float* p, const float* g, float* m, float* v, double lr, double beta1, double beta2,
                    double eps, int step) {
    if (!p || !g || !m || !v || step < 1) return GENIE_E_ARG;
    if (n == 0) return GENIE_OK;
    launch_adam((hipStream_t)stream, n, p, g, m, v, lr, beta1, beta2, eps, step);
    return hipGetLastError() == hipSuccess ? GENIE_OK : GENIE_E_HIP;
}

int genie_q_sample(genie_handle_t h, genie_stream_t stream, const float* x0, const float* z, const float* c_x0, const float* c_z,
                   float* trans_out, float* rots_out) {
    if (!h || !h->have_feats) { if (h) SET_ERR(h, "genie_q_sample: no batch bound"); return h ? GENIE_E_STATE : GENIE_E_ARG; }
    if (!x0 || !z || !c_x0 || !c_z || !trans_out || !rots_out) { SET_ERR(h, "genie_q_sample: null tensor"); return GENIE_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    launch_q_sample(h, (hipStream_t)stream, x0, z, c_x0, c_z, trans_out);
    launch_frenet(h, (hipStream_t)stream, 0, 0, 0.f, trans_out, rots_out, nullptr, nullptr);
    HIP_TRY(h, hipGetLastError());
    return GENIE_OK;
}

int genie_training_loss(genie_handle_t h, genie_stream_t stream, const float* z_pred, const float* z, float condition_loss_weight,
                        float* losses_out, float* grad_out) {
    if (!h || !h->have_feats) { if (h) SET_ERR(h, "genie_training_loss: no batch bound"); return h ? GENIE_E_STATE : GENIE_E_ARG; }
    if (!z_pred || !z || !losses_out) { SET_ERR(h, "genie_training_loss: null tensor"); return GENIE_E_ARG; }
    HIP_TRY(h, hipSetDevice(h->device));
    launch_training_loss(h, (hipStream_t)stream, z_pred, z, condition_loss_weight, losses_out, grad_out);
    HIP_TRY(h, hipGetLastError());
    return GENIE_OK;
}

int genie_p_sample(genie_handle_t h, genie_stream_t stream, int step, float scale, float* trans_inout, float* rots_out,
                   const float* z, const float* eps) {
    if (int rc = check_ready(h, "genie_p_sample")) return rc;
    if (step < 1 || step > h->d.n_timestep || !trans_inout || !rots_out || !z) { SET_ERR(h, "genie_p_sample: bad argument"); return GENIE_E_ARG; }
    launch_frenet(h, (hipStream_t)stream, 1, step, scale, trans_inout, rots_out, z, eps);
    HIP_TRY(h, hipGetLastError());
    return GENIE_OK;
}

int genie_sample_loop(genie_handle_t h, genie_stream_t stream, float scale, const float* noise, const int8_t* quat_codes,
                      int first_step, int last_step, float* trans_io, float* rots_io, float* record) {
    if (int rc = check_ready(h, "genie_sample_loop")) return rc;
    const int T = h->d.n_timestep;
    if (!noise || !trans_io || !rots_io || first_step > T || last_step < 1 || first_step < last_step) {
        SET_ERR(h, "genie_sample_loop: bad argument");
        return GENIE_E_ARG;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t n3 = (size_t)h->B * h->N * 3, nn = (size_t)h->B * h->N * h->N;
    if (first_step == T) {
        HIP_TRY(h, hipMemcpyAsync(trans_io, noise, n3 * 4, hipMemcpyDeviceToDevice, st));
        launch_frenet(h, st, 0, 0, 0.f, trans_io, rots_io, nullptr, nullptr);
    }
    int it = 0;
    for (int step = first_step; step >= last_step; --step, ++it) {
        launch_fill_i32(h, st, h->tsteps, h->B, step);
        const int8_t* codes = quat_codes ? quat_codes + (size_t)it * nn : nullptr;
        if (int rc = denoise_internal(h, st, trans_io, rots_io, h->tsteps, codes, h->loop_z, nullptr)) return rc;
        const float* eps = (step == 1) ? nullptr : noise + (size_t)(T - step + 1) * n3;
        launch_frenet(h, st, 1, step, scale, trans_io, rots_io, h->loop_z, eps);
        if (record) HIP_TRY(h, hipMemcpyAsync(record + (size_t)it * n3, trans_io, n3 * 4, hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(h, hipGetLastError());
    return GENIE_OK;
}

// ------------------------------------------------------------------ arithmetic
int genie_set_math(genie_handle_t h, int mode) {
    if (!h || (mode != GENIE_MATH_F32 && mode != GENIE_MATH_HX)) return GENIE_E_ARG;
    h->hx = mode == GENIE_MATH_HX;
    return GENIE_OK;
}
int genie_get_math(genie_handle_t h) { return h && h->hx ? GENIE_MATH_HX : GENIE_MATH_F32; }

// ------------------------------------------------------------------ measurement
int genie_profile_enable(genie_handle_t h, int enable) {
    if (!h) return GENIE_E_ARG;
    h->prof = enable != 0;
    return GENIE_OK;
}

int genie_profile_read(genie_handle_t h, const char** names, double* total_ms, int64_t* launches, int cap) {
    if (!h) return GENIE_E_ARG;
    (void)hipSetDevice(h->device);
    for (int i = 0; i < h->prof_n; ++i) {
        genie_ctx::ProfRec& r = h->prof_recs[i];
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) { h->prof_ms[r.cls] += ms; h->prof_cnt[r.cls] += 1; }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    h->prof_n = 0;
    int n = 0;
    for (int k = 0; k < KC_COUNT && n < cap; ++k) {
        if (names) names[n] = kKernelNames[k];
        if (total_ms) total_ms[n] = h->prof_ms[k];
        if (launches) launches[n] = h->prof_cnt[k];
        ++n;
    }
    for (int k = 0; k < KC_COUNT; ++k) { h->prof_ms[k] = 0; h->prof_cnt[k] = 0; }
    return n;
}
