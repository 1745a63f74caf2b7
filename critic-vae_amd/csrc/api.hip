// api.hip — the C-ABI of include/cvae.h: handle, flat parameter layout, workspace carve and the
// forward / loss / backward / optimizer orchestration.  Host code only; every launch is
// asynchronous on the caller's stream, nothing here allocates or synchronises.
#include "common.h"
#include "../../include/cvae.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";
void cvae_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- in-step kernel probe: ids = kind*9 + layer, kind 0 fwd / 1 dgrad / 2 wgrad (conv kernels; layer 0 = E1: id 0 its
//      forward — the BatchNorm/pool pass in two-pass mode —, id 18 its weight gradient; layer 8 = D4: id 8 forward,
//      id 17 the fused backward);  kind 3: ids 27..30 = the BatchNorm+pool backward apply kernel of encoder block
//      `layer` (HBM-bound), id 31 = the MS-SSIM level-0 tile kernel ----
static constexpr int PROBE_IDS = 32, PROBE_CAP = 128;
struct ProbeSlot { hipEvent_t e0[PROBE_CAP], e1[PROBE_CAP]; int n = 0; bool made = false; };
struct ProbeState { uint32_t mask = 0; ProbeSlot slot[PROBE_IDS]; };
static thread_local ProbeSlot* g_probe_cur = nullptr;
void cvae_probe_begin(hipStream_t st) {
    ProbeSlot* p = g_probe_cur;
    if (p && p->n < PROBE_CAP) (void)hipEventRecord(p->e0[p->n], st);
}
void cvae_probe_end(hipStream_t st) {
    ProbeSlot* p = g_probe_cur;
    if (p && p->n < PROBE_CAP) { (void)hipEventRecord(p->e1[p->n], st); p->n++; }
}

// Compute units of the current device.  Sizes persistent grids and split counts only (never a result), so a failed
// query falls back to the MI355X's 256 — but not silently: the error string says so (the launch that follows still
// succeeds, and cvae_last_error() then explains an unexpected grid).  Cached per device ordinal (ordinals past the
// table are queried every time rather than aliased onto another device's entry).
int cvae_num_cus() {
    constexpr int NCACHE = 64;
    static std::atomic<int> cached[NCACHE];                  // 0 = not yet queried
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) { cvae_set_error("cvae_num_cus: hipGetDevice failed (%s); assuming 256 compute units", hipGetErrorString(e)); return 256; }
    const bool cacheable = dev >= 0 && dev < NCACHE;
    int n = cacheable ? cached[dev].load(std::memory_order_relaxed) : 0;
    if (n > 0) return n;
    e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || n <= 0) {
        cvae_set_error("cvae_num_cus: compute-unit query failed on device %d (%s); assuming 256", dev, hipGetErrorString(e));
        return 256;                                          // not cached: the next call asks again
    }
    if (cacheable) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

thread_local bool g_conv_dry = false;
// include/cvae.h: which kernel family a conv pass of E2..E4 takes at a batch size (host logic only, no device access)
extern "C" int32_t cvae_conv_route(int32_t precision, int32_t width, int32_t layer, int32_t dgrad, int64_t batch) {
    if ((width != 64 && width != 128) || layer < 1 || layer > 3 || precision < 0 || precision > 3 || batch < 1 || batch > 0x7fffffffLL) return CVAE_EINVAL;
    if (precision == 0) return conv_f32_route(layer, width, dgrad != 0, (int)batch);
    if (precision == 1) return conv_bf16_route(layer, width, dgrad != 0, (int)batch);
    return 0;                                                  // fp32 emulation (three operand splits): the per-tile kernels
}

struct SideRed { hipStream_t st; hipEvent_t ev; };
static thread_local const SideRed* g_side_red = nullptr;
hipStream_t cvae_reduce_stream(hipStream_t st) {
    const SideRed* sr = g_side_red;
    if (!sr) return st;
    if (hipEventRecord(sr->ev, st) != hipSuccess || hipStreamWaitEvent(sr->st, sr->ev, 0) != hipSuccess) return st;
    return sr->st;
}

struct ParamEntry { std::string name; int64_t offset, numel; };

struct WsLayout {
    int64_t y[4], a[4], coef[4], bnpart[4];
    int64_t zcat, h, o[4];
    int64_t dout4, d_o[4], d_h, d_zcat, d_a[4], d_y[4];
    int64_t wc[3];             // phase-collapsed weights of D1..D3 (rebuilt every forward)
    int64_t xp;                // bf16 mode: the frame as packed bf16 (r, g, b, 0) pixels, written by E1's statistics pass (B * W * W * 8 bytes)
    int64_t wpack;             // bf16 mode: packed E2..E4 / D0 weights, forward + dgrad orientation (rebuilt every forward)
    int64_t ms, scratch_w, scratch, total;     // scratch_w: wgrad slabs (side stream); scratch: everything else (last)
};

struct cvae_handle_s {
    cvae_config cfg;
    std::vector<ParamEntry> params;
    int64_t param_total;
    int K;                       // bottleneck
    // parameter indices
    int enc_w[4], enc_b[4], enc_g[4], enc_be[4], fc_w, fc_b, dec_w[5], dec_b[5], di_w, di_b;
    // weight-gradient work runs on a lower-priority side stream, off the dgrad critical path
    hipStream_t side = nullptr;
    hipEvent_t ev_ready[8] = {}, ev_side = nullptr, ev_red = nullptr, ev_red_done = nullptr;
    bool streams_ready = false;
    bool e1_two_pass = false;        // bf16 mode: E1 forward as statistics pass + fused BatchNorm/pool pass (CVAE_E1_TWO_PASS=0: conv, then bn_pool_act_fwd)
    const void* xp_ws = nullptr;     // the workspace (and batch) whose packed bf16 frame the last train-mode forward wrote: the backward stages E1's strips from
    int xp_B = 0;                    // it only then (an eval-mode forward, or another workspace, leaves it stale -> the fp32 frame is staged instead)
    bool fuse_e1 = true;             // block 0's BatchNorm backward applied inside E1's weight-gradient kernel (CVAE_FUSE_E1=0: separate apply pass, for A/B runs)
    bool side_reduce = false;        // weight-gradient slab reductions on the side stream: measured -2.8 % (fp32, B=256) and
                                     // -1.6 % (bf16, B=2048) against in-order launches, so OFF; CVAE_SIDE_REDUCE=1 enables it for A/B runs
    ProbeState probe;
};

struct ProbeArm {      // RAII: arm the slot of (kind, layer) for the launches inside the scope
    ProbeArm(cvae_handle_s* h, int kind, int layer) {
        const int id = kind * 9 + layer;
        g_probe_cur = (h->probe.mask >> id) & 1u ? &h->probe.slot[id] : nullptr;
    }
    ~ProbeArm() { g_probe_cur = nullptr; }
};

static int ensure_streams(cvae_handle_s* h) {
    if (h->streams_ready) return 0;
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);      // lo = least priority
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, lo);
    for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->ev_ready[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_side, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_red, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_red_done, hipEventDisableTiming);
    if (e != hipSuccess) { cvae_set_error("side stream setup failed: %s", hipGetErrorString(e)); return (int)e; }
    h->streams_ready = true;
    return 0;
}

static int layer_h(const cvae_handle_s* h, int layer) { return kLayers[layer].h * (h->cfg.width / 64); }

static WsLayout carve(const cvae_handle_s* h, int B) {
    WsLayout w{};
    const int W = h->cfg.width;
    int64_t off = 0;
    auto take = [&](int64_t n) { int64_t o = off; off += align_up(n, 64); return o; };
    // activation / activation-gradient slots hold bf16 elements in precision mode 1: half the floats
    const bool b16 = h->cfg.precision == 1;
    auto act = [&](int64_t elems) { return take(b16 ? (elems + 1) / 2 : elems); };
    for (int l = 0; l < 4; ++l) {
        const int64_t H = layer_h(h, l), C = kLayers[l].cout;
        w.y[l] = act(B * H * H * C);
        w.a[l] = act(B * (H / 2) * (H / 2) * C);
        w.coef[l] = take(C * 4);
        w.bnpart[l] = take((int64_t)2 * bn_num_tiles(l, W, B) * C);
        // block 0's d_y exists only on the CVAE_FUSE_E1=0 path (E1's weight-gradient kernel applies the BatchNorm backward itself);
        // the slot keeps its offset entry (-1 -> "not allocated") for cvae_ws_offset
        w.d_y[l] = (l == 0 && h->fuse_e1) ? -1 : act(B * H * H * C);
        w.d_a[l] = act(B * (H / 2) * (H / 2) * C);
    }
    w.zcat = take((int64_t)B * 33);
    w.d_zcat = take((int64_t)B * 33);
    w.h = act((int64_t)B * h->K);
    w.d_h = act((int64_t)B * h->K);
    for (int i = 0; i < 4; ++i) {
        const int64_t H = layer_h(h, 4 + i), C = kLayers[4 + i].cout;
        w.o[i] = act(B * H * H * C);
        w.d_o[i] = act(B * H * H * C);
    }
    w.dout4 = take(b16 ? 0 : (int64_t)B * 3 * W * W);          // bf16 mode applies the Tanh backward inside d4_bwd: no dOut tensor
    for (int i = 0; i < 3; ++i) w.wc[i] = take(conv_up_wc_floats(5 + i));
    w.xp = take(b16 ? (int64_t)B * W * W * 2 : 0);
    w.wpack = take(h->cfg.precision != 0 ? conv_bf16_pack_floats(h->cfg.precision >= 2 ? 3 : 1) : 0);
    w.ms = take(msssim_ws_floats(W, B));
    int64_t sc = 0, scw = 0;
    auto mx = [&](int64_t v) { if (v > sc) sc = v; };
    for (int l = 1; l <= 4; ++l) { if (wgrad_ws_floats(l, W, B) > scw) scw = wgrad_ws_floats(l, W, B);
        if (h->cfg.precision == 1 && conv_bf16_supported(l, W) && wgrad_bf16_ws_floats(l, W, B) > scw) scw = wgrad_bf16_ws_floats(l, W, B);
        if (h->cfg.precision >= 2 && conv_bf16_supported(l, W) && wgrad_split_ws_floats(l, W, B) > scw) scw = wgrad_split_ws_floats(l, W, B);
        mx(conv_fwd_ws_floats(l, W, B)); mx(conv_dgrad_ws_floats(l, W, B)); }
    for (int l = 5; l <= 7; ++l) { if (conv_up_wgrad_ws_floats(l, W, B) > scw) scw = conv_up_wgrad_ws_floats(l, W, B); mx(conv_up_ws_floats(l, W, B)); }
    if (e1_wgrad_ws_floats(W, B) > scw) scw = e1_wgrad_ws_floats(W, B);
    w.scratch_w = take(scw);
    mx(scw + 36 * 128 * 64);                   // per-op entry points: one scratch = [collapsed W | everything else]
    mx(d4_bwd_ws_floats(W, B));
    for (int l = 0; l < 4; ++l) { mx(bn_bwd_ws_floats(l, W, B)); mx(bn_fwd_ws_floats(l, W)); }
    mx(fc_ws_floats(W, B));
    mx(colsum_ws_floats(0, 256));
    w.scratch = take(sc);
    w.total = off;
    return w;
}

static void add_param(cvae_handle_s* h, int* idx, const char* name, int64_t numel) {
    *idx = (int)h->params.size();
    h->params.push_back(ParamEntry{name, h->param_total, numel});
    h->param_total += align_up(numel, 64);
}

extern "C" {

const char* cvae_version(void) { return "critic-vae_amd 0.3 (gfx950; fp32 MFMA, bf16 MFMA with bf16 storage)"; }
const char* cvae_last_error(void) { return g_err; }

int cvae_create(const cvae_config* cfg, cvae_handle* out) {
    if (!cfg || !out) { cvae_set_error("cvae_create: null argument"); return CVAE_EINVAL; }
    if (cfg->width != 64 && cfg->width != 128) { cvae_set_error("cvae_create: width %d not supported (64 or 128)", cfg->width); return CVAE_EUNSUPPORTED; }
    if (cfg->precision < 0 || cfg->precision > 3) { cvae_set_error("cvae_create: precision %d not supported (0 = fp32, 1 = bf16 MFMA, 2 = fp32 emulated by 3-way bf16 splits)", cfg->precision); return CVAE_EUNSUPPORTED; }
    cvae_handle_s* h = new cvae_handle_s();
    h->cfg = *cfg;
    { const char* e = getenv("CVAE_SIDE_REDUCE"); h->side_reduce = e && e[0] == '1'; }
    { const char* e = getenv("CVAE_FUSE_E1"); h->fuse_e1 = !(e && e[0] == '0'); }
    { const char* e = getenv("CVAE_E1_TWO_PASS"); h->e1_two_pass = cfg->precision == 1 && !(e && e[0] == '0'); }
    h->param_total = 0;
    h->K = 256 * (cfg->width / 16) * (cfg->width / 16);
    char nm[64];
    for (int l = 0; l < 4; ++l) {
        snprintf(nm, sizeof nm, "enc%d.w", l); add_param(h, &h->enc_w[l], nm, (int64_t)25 * kLayers[l].cin * kLayers[l].cout);
        snprintf(nm, sizeof nm, "enc%d.b", l); add_param(h, &h->enc_b[l], nm, kLayers[l].cout);
        snprintf(nm, sizeof nm, "enc%d.gamma", l); add_param(h, &h->enc_g[l], nm, kLayers[l].cout);
        snprintf(nm, sizeof nm, "enc%d.beta", l); add_param(h, &h->enc_be[l], nm, kLayers[l].cout);
    }
    add_param(h, &h->fc_w, "fc.w", (int64_t)h->K * 64);
    add_param(h, &h->fc_b, "fc.b", 64);
    for (int i = 0; i < 5; ++i) {
        snprintf(nm, sizeof nm, "dec%d.w", i); add_param(h, &h->dec_w[i], nm, (int64_t)25 * kLayers[4 + i].cin * kLayers[4 + i].cout);
        snprintf(nm, sizeof nm, "dec%d.b", i); add_param(h, &h->dec_b[i], nm, kLayers[4 + i].cout);
    }
    add_param(h, &h->di_w, "decin.w", (int64_t)33 * h->K);
    add_param(h, &h->di_b, "decin.b", h->K);
    *out = h;
    return CVAE_OK;
}

void cvae_destroy(cvae_handle h) {
    if (!h) return;
    if (h->streams_ready) {
        for (int i = 0; i < 8; ++i) (void)hipEventDestroy(h->ev_ready[i]);
        (void)hipEventDestroy(h->ev_side);
        (void)hipEventDestroy(h->ev_red);
        (void)hipEventDestroy(h->ev_red_done);
        (void)hipStreamDestroy(h->side);
    }
    delete h;
}

int64_t cvae_param_total(cvae_handle h) { return h->param_total; }
int32_t cvae_param_count(cvae_handle h) { return (int32_t)h->params.size(); }
const char* cvae_param_name(cvae_handle h, int32_t i) { return h->params[i].name.c_str(); }
int64_t cvae_param_offset(cvae_handle h, int32_t i) { return h->params[i].offset; }
int64_t cvae_param_numel(cvae_handle h, int32_t i) { return h->params[i].numel; }
int64_t cvae_workspace_bytes(cvae_handle h, int32_t batch) { return carve(h, batch).total * 4; }
int64_t cvae_bn_state_floats(cvae_handle) { return 2 * 480; }

// float offset of a named saved tensor inside the workspace (tests / debugging); -1 if unknown
int64_t cvae_ws_offset(cvae_handle h, int32_t batch, const char* name) {
    const WsLayout w = carve(h, batch);
    char nm[32];
    for (int l = 0; l < 4; ++l) {
        snprintf(nm, sizeof nm, "y%d", l); if (!strcmp(name, nm)) return w.y[l];
        snprintf(nm, sizeof nm, "a%d", l); if (!strcmp(name, nm)) return w.a[l];
        snprintf(nm, sizeof nm, "d_y%d", l); if (!strcmp(name, nm)) return w.d_y[l];
        snprintf(nm, sizeof nm, "d_a%d", l); if (!strcmp(name, nm)) return w.d_a[l];
        snprintf(nm, sizeof nm, "coef%d", l); if (!strcmp(name, nm)) return w.coef[l];
        snprintf(nm, sizeof nm, "o%d", l); if (!strcmp(name, nm)) return w.o[l];
        snprintf(nm, sizeof nm, "d_o%d", l); if (!strcmp(name, nm)) return w.d_o[l];
    }
    if (!strcmp(name, "zcat")) return w.zcat;
    if (!strcmp(name, "d_zcat")) return w.d_zcat;
    if (!strcmp(name, "h")) return w.h;
    if (!strcmp(name, "d_h")) return w.d_h;
    if (!strcmp(name, "dout4")) return w.dout4;
    if (!strcmp(name, "scratch")) return w.scratch;
    if (!strcmp(name, "xp")) return h->cfg.precision == 1 ? w.xp : -1;      // packed bf16 frame (precision mode 1)
    return -1;
}

static const int kBnOff[4] = {0, 32, 96, 224};
int cvae_decode(cvae_handle h, int32_t B, const float* zcat, const float* params, float* recon, void* wsv, void* stream);
int cvae_backward_phases(cvae_handle h, int32_t B, const float* x, const float* pred, const float* eps, const float* params,
                         const float* logvar, const float* recon, const float* d_recon, const float* d_mu,
                         const float* d_logvar, void* wsv, float* grads, int32_t phase_mask, void* stream);

#define P_(idx) (params + h->params[(idx)].offset)
#define G_(idx) (grads + h->params[(idx)].offset)
#define RC(call) do { int rc_ = (call); if (rc_) return rc_; } while (0)

// cvae_config.precision: 1 = bf16-MFMA kernels for every pass of E2..E4 / D0..D3 (conv_bf16.hip); 2 = the same
// forward / input-gradient kernels with 3-way split operands (exact fp32 products, 9 MFMAs each), fp32 wgrad
static bool use_bf16(cvae_handle h, int layer) {
    if (h->cfg.precision >= 2 && layer > 4) return false;     // D1..D3: the fp32 phase-collapsed kernels beat nine bf16 MFMAs per block
    return h->cfg.precision != 0 && conv_bf16_supported(layer, h->cfg.width);
}
// precision 1: activations and activation gradients are bf16 IN HBM (every kernel that touches them is told so)
static bool io_bf16(cvae_handle h) { return h->cfg.precision == 1; }
static bool use_bf16_wgrad(cvae_handle h, int layer) { return h->cfg.precision == 1 && conv_bf16_supported(layer, h->cfg.width); }
// fp32-emulation modes: weight gradients of E2..E4 / D0 on the bf16 MFMA with exact 3-way operand splits (conv_bf16.hip);
// CVAE_SPLIT_WGRAD = bit mask of the layers (bit l-1 = layer l), for A/B runs against the fp32-MFMA kernels
static bool use_split_wgrad(cvae_handle h, int layer) {
    if (h->cfg.precision < 2 || layer < 1 || layer > 4 || !conv_bf16_supported(layer, h->cfg.width)) return false;
    if (!conv_wgrad_split_supported(h->cfg.precision == 3 ? 6 : 9)) return false;
    static const int mask = [] { const char* e = getenv("CVAE_SPLIT_WGRAD"); return e ? atoi(e) : 15; }();
    return ((mask >> (layer - 1)) & 1) != 0;
}
static int bf16_splits(cvae_handle h) { return h->cfg.precision >= 2 ? 3 : 1; }          // packed weight copies
static int bf16_mode(cvae_handle h) { return h->cfg.precision == 2 ? 3 : (h->cfg.precision == 3 ? 6 : 1); }   // launcher code: 1 bf16, 3 x9, 6 x6
static int pack_bf16_weights(cvae_handle h, const float* params, float* ws, const WsLayout& w, hipStream_t st) {
    if (!use_bf16(h, 1)) return 0;
    const float* wl[4] = {P_(h->enc_w[1]), P_(h->enc_w[2]), P_(h->enc_w[3]), P_(h->dec_w[0])};
    return launch_pack_w_bf16(wl, ws + w.wpack, bf16_splits(h), st);
}

static int check(cvae_handle h, int32_t batch, const void* ws) {
    if (!h) { cvae_set_error("null handle"); return CVAE_EINVAL; }
    if (batch < 1 || batch > h->cfg.max_batch) { cvae_set_error("batch %d outside [1, %d]", batch, h->cfg.max_batch); return CVAE_EINVAL; }
    if (!ws) { cvae_set_error("null workspace"); return CVAE_ENOWS; }
    return 0;
}

int cvae_forward(cvae_handle h, int32_t B, const float* x, const float* pred, const float* eps, const float* params,
                 float* bn_state, float* mu, float* logvar, float* recon, void* wsv, int32_t train, void* stream) {
    RC(check(h, B, wsv));
    if (B < 2 && train) { /* BatchNorm with one 1x1... still fine: B*H*W >= 2 */ }
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)wsv;
    const WsLayout w = carve(h, B);
    const int W = h->cfg.width;
    RC(pack_bf16_weights(h, params, ws, w, st));
    for (int l = 0; l < 4; ++l) {
        int tpp = 1;                            // 128-pixel tiles per BatchNorm partial row, as reported by the conv kernel that ran
        if (l == 0 && h->e1_two_pass) {
            // bf16 mode, block 0: conv (statistics only) -> merged statistics -> conv again with BatchNorm/pool/ReLU in its
            // epilogue (writes y0 for the backward and a0); bn_pool_act_fwd's read of y0 is replaced by a second read of x
            if (train) RC(launch_e1_fwd(W, B, x, P_(h->enc_w[0]), P_(h->enc_b[0]), nullptr, ws + w.bnpart[0], st, true, 1, nullptr, nullptr, true, ws + w.xp));     // also writes the packed bf16 frame
            h->xp_ws = train ? wsv : nullptr; h->xp_B = B;
            RC(launch_bn_fwd_finalize(0, W, B, ws + w.bnpart[0], P_(h->enc_g[0]), P_(h->enc_be[0]), bn_state + kBnOff[0],
                                      bn_state + 480 + kBnOff[0], ws + w.coef[0], ws + w.scratch, train, st));
            { ProbeArm pa(h, 0, 0);
              // y0 is written only for the CVAE_FUSE_E1=0 path (or, decided on the device, when a channel's gamma is tiny):
              // the fused weight-gradient kernel recomputes it
              RC(launch_e1_fwd(W, B, x, P_(h->enc_w[0]), P_(h->enc_b[0]), ws + w.y[0], nullptr, st, true, 2, ws + w.coef[0], ws + w.a[0],
                               !h->fuse_e1, train ? ws + w.xp : nullptr)); }          // eval mode: no statistics pass, no packed frame
            continue;
        }
        if (l == 0) { ProbeArm pa(h, 0, 0); RC(launch_e1_fwd(W, B, x, P_(h->enc_w[0]), P_(h->enc_b[0]), ws + w.y[0], ws + w.bnpart[0], st, h->cfg.precision == 1)); }
        else if (use_bf16(h, l)) { ProbeArm pa(h, 0, l); RC(launch_conv_fwd_bf16(l, W, bf16_mode(h), B, ws + w.a[l - 1], ws + w.wpack, P_(h->enc_b[l]), ws + w.y[l], ws + w.bnpart[l], ws + w.scratch, st, &tpp)); }
        else { ProbeArm pa(h, 0, l); RC(launch_conv_fwd(l, W, B, ws + w.a[l - 1], P_(h->enc_w[l]), P_(h->enc_b[l]), ws + w.y[l], ws + w.bnpart[l], ws + w.scratch, st)); }
        RC(launch_bn_fwd_finalize(l, W, B, ws + w.bnpart[l], P_(h->enc_g[l]), P_(h->enc_be[l]), bn_state + kBnOff[l],
                                  bn_state + 480 + kBnOff[l], ws + w.coef[l], ws + w.scratch, train, st,
                                  tpp));
        RC(launch_bn_pool_act_fwd(l, W, B, ws + w.y[l], ws + w.coef[l], ws + w.a[l], st, io_bf16(h)));
    }
    RC(launch_fc_fwd(W, B, ws + w.a[3], P_(h->fc_w), P_(h->fc_b), eps, pred, mu, logvar, ws + w.zcat, ws + w.scratch, st, io_bf16(h)));
    if (!recon) return 0;                       // encode only (VariationalEncoder.forward)
    return cvae_decode(h, B, nullptr, params, recon, wsv, stream);
}

// Decoder.forward (vae_nets.py:139-147) from zcat = [z | pred] (B,33); zcat == NULL uses the one
// cvae_forward left in the workspace.
int cvae_decode(cvae_handle h, int32_t B, const float* zcat, const float* params, float* recon, void* wsv, void* stream) {
    RC(check(h, B, wsv));
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)wsv;
    const WsLayout w = carve(h, B);
    const int W = h->cfg.width;
    if (zcat) {
        hipError_t e = hipMemcpyAsync(ws + w.zcat, zcat, (size_t)B * 33 * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) { cvae_set_error("cvae_decode: copy failed: %s", hipGetErrorString(e)); return (int)e; }
        RC(pack_bf16_weights(h, params, ws, w, st));          // stand-alone decode: cvae_forward did not run
    }
    RC(launch_decin_fwd(W, B, ws + w.zcat, P_(h->di_w), P_(h->di_b), ws + w.h, st, io_bf16(h)));
    // Upsample -> Conv of D1..D3 runs at the low resolution with phase-collapsed weights (conv_up.hip)
    {
        const float* wsrc[3] = {P_(h->dec_w[1]), P_(h->dec_w[2]), P_(h->dec_w[3])};
        float* wdst[3] = {ws + w.wc[0], ws + w.wc[1], ws + w.wc[2]};
        RC(launch_collapse_w3(wsrc, wdst, st));
    }
    if (use_bf16(h, 5)) {
        const float* wcs[3] = {ws + w.wc[0], ws + w.wc[1], ws + w.wc[2]};
        RC(launch_pack_up_bf16(wcs, ws + w.wpack, bf16_splits(h), st));
    }
    for (int i = 0; i < 4; ++i) {
        ProbeArm pa(h, 0, 4 + i);
        if (i == 0) {
            if (use_bf16(h, 4)) RC(launch_conv_fwd_bf16(4, W, bf16_mode(h), B, ws + w.h, ws + w.wpack, P_(h->dec_b[0]), ws + w.o[0], nullptr, ws + w.scratch, st));
            else RC(launch_conv_fwd(4, W, B, ws + w.h, P_(h->dec_w[0]), P_(h->dec_b[0]), ws + w.o[0], nullptr, ws + w.scratch, st));
        } else if (use_bf16(h, 4 + i)) {
            RC(launch_conv_up_fwd_bf16(4 + i, W, bf16_mode(h), B, ws + w.o[i - 1], ws + w.wpack, P_(h->dec_b[i]), ws + w.o[i], st));
        } else {
            RC(launch_conv_up_fwd(4 + i, W, B, ws + w.o[i - 1], ws + w.wc[i - 1], P_(h->dec_b[i]), ws + w.o[i], ws + w.scratch, st));
        }
    }
    { ProbeArm pa(h, 0, 8); RC(launch_d4_fwd(W, B, ws + w.o[3], P_(h->dec_w[4]), P_(h->dec_b[4]), recon, st, io_bf16(h))); }
    return 0;
}

int cvae_loss(cvae_handle h, int32_t B, const float* x, const float* mu, const float* logvar, const float* recon,
              void* wsv, float* scalars, float* d_recon, float* d_mu, float* d_logvar, void* stream) {
    RC(check(h, B, wsv));
    if ((d_recon == nullptr) != (d_mu == nullptr) || (d_mu == nullptr) != (d_logvar == nullptr)) {
        cvae_set_error("cvae_loss: d_recon, d_mu, d_logvar must be all set or all null");
        return CVAE_EINVAL;
    }
    float* ws = (float*)wsv;
    const WsLayout w = carve(h, B);
    ProbeArm pa(h, 3, 4);            // id 31: the level-0 tile kernel
    return launch_msssim(h->cfg.width, B, recon, x, mu, logvar, ws + w.ms, scalars, d_recon, d_mu, d_logvar, (hipStream_t)stream);
}

int cvae_backward(cvae_handle h, int32_t B, const float* x, const float* pred, const float* eps, const float* params,
                  const float* logvar, const float* recon, const float* d_recon, const float* d_mu,
                  const float* d_logvar, void* wsv, float* grads, void* stream) {
    return cvae_backward_phases(h, B, x, pred, eps, params, logvar, recon, d_recon, d_mu, d_logvar, wsv, grads, 7, stream);
}

// Gradient buckets in the order backward completes them (for bucketed all-reduce overlapped with the
// rest of backward): phase 0 = decoder + decoder_input, 1 = fc_mu|fc_var + encoder block 3,
// 2 = encoder blocks 2..0.  Each bucket is one contiguous range of the flat gradient buffer.
int cvae_grad_bucket(cvae_handle h, int32_t phase, int64_t* offset, int64_t* numel) {
    if (!h || !offset || !numel || phase < 0 || phase > 2) { cvae_set_error("cvae_grad_bucket: bad argument"); return CVAE_EINVAL; }
    const int64_t enc3 = h->params[h->enc_w[3]].offset, dec0 = h->params[h->dec_w[0]].offset;
    if (phase == 0) { *offset = dec0; *numel = h->param_total - dec0; }
    else if (phase == 1) { *offset = enc3; *numel = dec0 - enc3; }
    else { *offset = 0; *numel = enc3; }
    return 0;
}

int cvae_backward_phases(cvae_handle h, int32_t B, const float* x, const float* pred, const float* eps, const float* params,
                         const float* logvar, const float* recon, const float* d_recon, const float* d_mu,
                         const float* d_logvar, void* wsv, float* grads, int32_t phase_mask, void* stream) {
    (void)pred;
    RC(check(h, B, wsv));
    if ((phase_mask & ~15) != 0 || (phase_mask & 7) == 0) { cvae_set_error("cvae_backward_phases: phase_mask %d (bits 0..2 = phases, bit 3 = zero the alignment padding)", phase_mask); return CVAE_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (phase_mask & 8) {                               // caller handed an uninitialised gradient buffer
        PadGaps gaps{};                                 // 32 gaps per launch; as many launches as the parameter list needs
        for (const ParamEntry& p : h->params) {
            const int64_t pad = align_up(p.numel, 64) - p.numel;
            if (pad <= 0) continue;
            gaps.off[gaps.n] = p.offset + p.numel; gaps.len[gaps.n] = (int)pad; gaps.n++;
            if (gaps.n == 32) { RC(launch_zero_gaps(grads, gaps, st)); gaps.n = 0; }
        }
        RC(launch_zero_gaps(grads, gaps, st));
    }
    float* ws = (float*)wsv;
    const WsLayout w = carve(h, B);
    const int W = h->cfg.width;
    float* sc = ws + w.scratch;
    float* scw = ws + w.scratch_w;
    // cfg.overlap_wgrad != 0: weight-gradient work on a lower-priority side stream (bit-identical results; round 3: +1.5 % in
    // bf16 mode at B = 2048, -3 % in fp32 mode at B = 256); default: everything in order on the caller's stream.
    const bool overlap = h->cfg.overlap_wgrad != 0;
    const bool side_red = !overlap && h->side_reduce;          // slab reductions only (the wgrad kernels stay on `st`)
    if (overlap || side_red) RC(ensure_streams(h));
    hipStream_t sd = overlap ? h->side : st;
    const SideRed sred{h->side, h->ev_red};
    bool red_pending = false;
    // Arm: the launcher called inside the scope sends its reductions to the side stream; the shared slab scratch is
    // only reused once the previous layer's reductions have read it (main waits for ev_red_done first).
    struct RedArm {
        cvae_handle_s* h; bool on; bool* pending; hipStream_t st;
        RedArm(cvae_handle_s* h_, bool on_, bool* p, hipStream_t st_) : h(h_), on(on_), pending(p), st(st_) {
            if (!on) return;
            if (*pending) (void)hipStreamWaitEvent(st, h->ev_red_done, 0);
            g_side_red = nullptr;
        }
        void arm(const SideRed* sr) { if (on) g_side_red = sr; }
        ~RedArm() {
            if (!on) return;
            g_side_red = nullptr;
            (void)hipEventRecord(h->ev_red_done, h->side);
            *pending = true;
        }
    };
#define HIPRC(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { cvae_set_error("%s: %s", #call, hipGetErrorString(e_)); return (int)e_; } } while (0)
    // `ready k` = the gradient a weight-gradient kernel needs exists on the main stream; the side
    // stream picks it up from there, so dW/db never delay the dgrad chain.
    auto fork = [&](int idx) -> int {
        if (!overlap) return 0;
        HIPRC(hipEventRecord(h->ev_ready[idx], st));
        HIPRC(hipStreamWaitEvent(sd, h->ev_ready[idx], 0));
        return 0;
    };
    auto join = [&]() -> int {                          // side-stream work of this phase is complete on the caller's stream
        if (!overlap && !side_red) return 0;
        HIPRC(hipEventRecord(h->ev_side, h->side));
        HIPRC(hipStreamWaitEvent(st, h->ev_side, 0));
        red_pending = false;
        return 0;
    };
    if (phase_mask & 1) {
    // decoder, last layer first
    { ProbeArm pa(h, 1, 8);
      RC(launch_d4_bwd(W, B, ws + w.o[3], d_recon, recon, P_(h->dec_w[4]), ws + w.dout4, ws + w.d_o[3],
                       G_(h->dec_w[4]), G_(h->dec_b[4]), sc, st, io_bf16(h))); }
    for (int i = 3; i >= 0; --i) {
        const int l = 4 + i;
        const float* in = i == 0 ? ws + w.h : ws + w.o[i - 1];
        RC(fork(3 - i));
        if (i == 0) {
            { ProbeArm pa(h, 2, l); RedArm ra(h, side_red, &red_pending, st); ra.arm(&sred);
              if (use_bf16_wgrad(h, 4)) RC(launch_conv_wgrad_bf16(4, W, B, in, ws + w.d_o[0], G_(h->dec_w[0]), G_(h->dec_b[0]), scw, sd));
              else if (use_split_wgrad(h, 4)) RC(launch_conv_wgrad_split(4, W, bf16_mode(h) == 6 ? 6 : 9, B, in, ws + w.d_o[0], G_(h->dec_w[0]), G_(h->dec_b[0]), scw, sd));
              else RC(launch_conv_wgrad(l, W, B, in, ws + w.d_o[0], G_(h->dec_w[0]), G_(h->dec_b[0]), scw, sd)); }
            { ProbeArm pa(h, 1, l);
              if (use_bf16(h, 4)) RC(launch_conv_dgrad_bf16(4, W, bf16_mode(h), B, ws + w.d_o[0], ws + w.wpack, ws + w.d_h, ws + w.scratch, st));
              else RC(launch_conv_dgrad(l, W, B, ws + w.d_o[0], P_(h->dec_w[0]), nullptr, ws + w.d_h, ws + w.scratch, st)); }
        } else {
            { ProbeArm pa(h, 2, l); RedArm ra(h, side_red, &red_pending, st); ra.arm(&sred);
              RC(launch_conv_up_wgrad(l, W, B, in, ws + w.d_o[i], G_(h->dec_w[i]), G_(h->dec_b[i]), scw, sd, use_bf16_wgrad(h, l))); }
            { ProbeArm pa(h, 1, l);
              if (use_bf16(h, l)) RC(launch_conv_up_dgrad_bf16(l, W, bf16_mode(h), B, ws + w.d_o[i], ws + w.wpack, ws + w.o[i - 1], ws + w.d_o[i - 1], st));
              else RC(launch_conv_up_dgrad(l, W, B, ws + w.d_o[i], ws + w.wc[i - 1], ws + w.o[i - 1], ws + w.d_o[i - 1], sc, st)); }
        }
    }
    // latent
    RC(launch_decin_bwd(W, B, ws + w.zcat, ws + w.d_h, P_(h->di_w), G_(h->di_w), G_(h->di_b), ws + w.d_zcat, sc, st, io_bf16(h)));
    RC(join());
    }
    if (phase_mask & 2)
        RC(launch_fc_bwd(W, B, ws + w.a[3], P_(h->fc_w), ws + w.d_zcat, eps, logvar, d_mu, d_logvar, G_(h->fc_w),
                         G_(h->fc_b), ws + w.d_a[3], sc, st, io_bf16(h)));
    // encoder
    for (int l = 3; l >= 0; --l) {
        if (!(phase_mask & (l == 3 ? 2 : 4))) continue;
        // block 0: only the statistics pass runs here; E1's weight-gradient kernel applies the BatchNorm/pool/ReLU
        // backward while it stages its tiles (d_y[0] is never written: nothing else would read it)
        const bool fuse0 = l == 0 && h->fuse_e1;
        { ProbeArm pa(h, 3, l);
          RC(launch_bn_pool_act_bwd(l, W, B, ws + w.y[l], ws + w.a[l], ws + w.d_a[l], ws + w.coef[l], P_(h->enc_g[l]),
                                    fuse0 ? nullptr : ws + w.d_y[l], G_(h->enc_g[l]), G_(h->enc_be[l]), nullptr, sc, st, io_bf16(h))); }
        RC(fork(7 - l));
        if (l == 0) {
            const float* fu[7] = {ws + w.y[0], ws + w.a[0], ws + w.d_a[0], ws + w.coef[0], bn_bwd_bcoef(0, W, B, sc),
                                  P_(h->enc_w[0]), P_(h->enc_b[0])};
            { ProbeArm pa(h, 2, 0); RedArm ra(h, side_red, &red_pending, st); ra.arm(&sred);
              RC(launch_e1_wgrad(W, B, x, fuse0 ? nullptr : ws + w.d_y[0], G_(h->enc_w[0]), G_(h->enc_b[0]), scw, sd, h->cfg.precision == 1, fuse0 ? fu : nullptr,
                                 (h->e1_two_pass && fuse0 && h->xp_ws == (const void*)ws && h->xp_B == B) ? ws + w.xp : nullptr)); }
        } else {
            { ProbeArm pa(h, 2, l); RedArm ra(h, side_red, &red_pending, st); ra.arm(&sred);
              if (use_bf16_wgrad(h, l)) RC(launch_conv_wgrad_bf16(l, W, B, ws + w.a[l - 1], ws + w.d_y[l], G_(h->enc_w[l]), G_(h->enc_b[l]), scw, sd));
              else if (use_split_wgrad(h, l)) RC(launch_conv_wgrad_split(l, W, bf16_mode(h) == 6 ? 6 : 9, B, ws + w.a[l - 1], ws + w.d_y[l], G_(h->enc_w[l]), G_(h->enc_b[l]), scw, sd));
              else RC(launch_conv_wgrad(l, W, B, ws + w.a[l - 1], ws + w.d_y[l], G_(h->enc_w[l]), G_(h->enc_b[l]), scw, sd)); }
            { ProbeArm pa(h, 1, l);
              if (use_bf16(h, l)) RC(launch_conv_dgrad_bf16(l, W, bf16_mode(h), B, ws + w.d_y[l], ws + w.wpack, ws + w.d_a[l - 1], ws + w.scratch, st));
              else RC(launch_conv_dgrad(l, W, B, ws + w.d_y[l], P_(h->enc_w[l]), nullptr, ws + w.d_a[l - 1], nullptr, st)); }
        }
        if (l == 3 || l == 0) RC(join());               // end of phase 1 / phase 2
    }
#undef HIPRC
    return 0;
}

int cvae_adam_step(cvae_handle h, float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n,
                   int32_t step, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
    if (!h || step < 1) { cvae_set_error("cvae_adam_step: bad handle/step"); return CVAE_EINVAL; }
    return launch_adam(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, grad_scale, (hipStream_t)stream);
}

int cvae_grads_to_bf16(cvae_handle h, const float* grads, void* out_bf16, int64_t n, void* stream) {
    if (h && n == 0) return 0;                       // empty range (its pointers may be null): nothing to do
    if (!h || !grads || !out_bf16 || n < 0) { cvae_set_error("cvae_grads_to_bf16: bad argument"); return CVAE_EINVAL; }
    return launch_grads_bf16(grads, out_bf16, nullptr, n, (hipStream_t)stream);
}
int cvae_grads_from_bf16(cvae_handle h, const void* in_bf16, float* grads, int64_t n, void* stream) {
    if (h && n == 0) return 0;
    if (!h || !grads || !in_bf16 || n < 0) { cvae_set_error("cvae_grads_from_bf16: bad argument"); return CVAE_EINVAL; }
    return launch_grads_bf16(nullptr, const_cast<void*>(in_bf16), grads, n, (hipStream_t)stream);
}

// d_* (out) = d_* (in) * gscale[0]: total_loss.backward()'s incoming factor applied to the three loss
// gradients cvae_loss wrote, one launch, gscale a device scalar (no host read)
int cvae_scale_loss_grads(cvae_handle h, int32_t B, const float* gscale, const float* d_recon, const float* d_mu,
                          const float* d_logvar, float* d_recon_out, float* d_mu_out, float* d_logvar_out, void* stream) {
    if (!h || B < 1 || !gscale) { cvae_set_error("cvae_scale_loss_grads: bad argument"); return CVAE_EINVAL; }
    const int W = h->cfg.width;
    Scale3 a{gscale, {d_recon, d_mu, d_logvar}, {d_recon_out, d_mu_out, d_logvar_out}, {(int64_t)B * 3 * W * W, (int64_t)B * 32, (int64_t)B * 32}};
    return launch_scale3(a, (hipStream_t)stream);
}

// ---- critic inference + uint8 frame pre-processing (vae.py:46-50) ----
int32_t cvae_critic_param_count(void) { return critic_param_count(); }

int cvae_critic_forward(cvae_handle h, int32_t B, const float* x, const float* critic_params, float* pred, void* stream) {
    if (!h || B < 1 || B > h->cfg.max_batch) { cvae_set_error("cvae_critic_forward: bad handle, or batch %d outside [1, max_batch]", B); return CVAE_EINVAL; }
    return launch_critic_fwd(h->cfg.width, B, x, critic_params, pred, (hipStream_t)stream);
}

int cvae_preprocess_u8(cvae_handle h, int32_t B, const uint8_t* frames_hwc, float* x, void* stream) {
    if (!h || B < 1 || !frames_hwc || !x) { cvae_set_error("cvae_preprocess_u8: bad handle/batch/pointer"); return CVAE_EINVAL; }
    if (B > h->cfg.max_batch) { cvae_set_error("cvae_preprocess_u8: batch %d outside [1, %d]", B, h->cfg.max_batch); return CVAE_EINVAL; }
    return launch_preprocess_u8(h->cfg.width, B, frames_hwc, x, (hipStream_t)stream);
}

// |recon_zero - recon_one| -> greyscale difference mask (get_diff_image, vae_utility.py:256-277), batched
int cvae_diff_grey(cvae_handle h, int32_t B, const float* recon_one, const float* recon_zero, float* diff, void* stream) {
    if (!h || B < 1) { cvae_set_error("cvae_diff_grey: bad handle/batch"); return CVAE_EINVAL; }
    return launch_diff_grey(h->cfg.width, B, recon_one, recon_zero, diff, (hipStream_t)stream);
}

// ---- probe API: bracket chosen conv kernels of the real step with HIP events (bench.py roofline) ----
int cvae_probe_config(cvae_handle h, uint32_t mask) {
    if (!h) return CVAE_EINVAL;
    for (int id = 0; id < PROBE_IDS; ++id) {
        ProbeSlot& s = h->probe.slot[id];
        s.n = 0;
        if (((mask >> id) & 1u) && !s.made) {
            for (int i = 0; i < PROBE_CAP; ++i) {
                hipError_t e = hipEventCreate(&s.e0[i]);
                if (e == hipSuccess) e = hipEventCreate(&s.e1[i]);
                if (e != hipSuccess) { cvae_set_error("probe event create: %s", hipGetErrorString(e)); return (int)e; }
            }
            s.made = true;
        }
    }
    h->probe.mask = mask;
    return 0;
}

// elapsed ms of every recorded launch of slot `id` (synchronises on the events); returns the count
int cvae_probe_read(cvae_handle h, int32_t id, float* ms_host, int32_t cap) {
    if (!h || id < 0 || id >= PROBE_IDS) return 0;
    ProbeSlot& s = h->probe.slot[id];
    int n = 0;
    for (int i = 0; i < s.n && n < cap; ++i) {
        if (hipEventSynchronize(s.e1[i]) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.e0[i], s.e1[i]) != hipSuccess) break;
        ms_host[n++] = ms;
    }
    s.n = 0;
    return n;
}

// ------------------------------ per-op entry points ------------------------------
int cvae_op_conv_fwd(cvae_handle h, int32_t layer, int32_t B, const float* in, const float* wt, const float* bias,
                     float* out, float* bn_partials, void* scratch, void* stream) {
    const int W = h->cfg.width;
    if (layer == 0) return launch_e1_fwd(W, B, in, wt, bias, out, bn_partials, (hipStream_t)stream, io_bf16(h));   // y in the handle's storage type
    if (layer == 8) return launch_d4_fwd(W, B, in, wt, bias, out, (hipStream_t)stream);
    if (layer >= 5) {
        float* wc = (float*)scratch;
        RC(launch_collapse_w(layer, wt, wc, (hipStream_t)stream));
        return launch_conv_up_fwd(layer, W, B, in, wc, bias, out, wc + conv_up_wc_floats(layer), (hipStream_t)stream);
    }
    return launch_conv_fwd(layer, W, B, in, wt, bias, out, bn_partials, (float*)scratch, (hipStream_t)stream);
}

int cvae_op_conv_dgrad(cvae_handle h, int32_t layer, int32_t B, const float* dout, const float* wt,
                       const float* mask_src, float* din, void* scratch, void* stream) {
    if (layer >= 5) {
        float* wc = (float*)scratch;
        RC(launch_collapse_w(layer, wt, wc, (hipStream_t)stream));
        return launch_conv_up_dgrad(layer, h->cfg.width, B, dout, wc, mask_src, din, wc + conv_up_wc_floats(layer), (hipStream_t)stream);
    }
    return launch_conv_dgrad(layer, h->cfg.width, B, dout, wt, mask_src, din, nullptr, (hipStream_t)stream);
}

int64_t cvae_op_scratch_floats(cvae_handle h, int32_t B) {
    const WsLayout w = carve(h, B);
    return w.total - w.scratch;
}

int cvae_op_conv_wgrad(cvae_handle h, int32_t layer, int32_t B, const float* in, const float* dout, float* dw,
                       float* dbias, void* scratch, void* stream) {
    const int W = h->cfg.width;
    hipStream_t st = (hipStream_t)stream;
    float* sc = (float*)scratch;
    if (layer == 0) return launch_e1_wgrad(W, B, in, dout, dw, dbias, sc, st);
    if (layer >= 5) return launch_conv_up_wgrad(layer, W, B, in, dout, dw, dbias, sc, st);
    return launch_conv_wgrad(layer, W, B, in, dout, dw, dbias, sc, st);
}

int cvae_op_d4_bwd(cvae_handle h, int32_t B, const float* o3, const float* d_recon, const float* recon, const float* wt,
                   float* dout, float* d_o3, float* dw, float* db, void* scratch, void* stream) {
    return launch_d4_bwd(h->cfg.width, B, o3, d_recon, recon, wt, dout, d_o3, dw, db, (float*)scratch, (hipStream_t)stream);
}

int cvae_op_bn_pool_act_fwd(cvae_handle h, int32_t layer, int32_t B, const float* y, const float* bn_partials,
                            const float* gamma, const float* beta, float* run_mean, float* run_var, float* coef,
                            float* a, void* scratch, int32_t train, void* stream) {
    const int W = h->cfg.width;
    RC(launch_bn_fwd_finalize(layer, W, B, bn_partials, gamma, beta, run_mean, run_var, coef, (float*)scratch, train, (hipStream_t)stream));
    return launch_bn_pool_act_fwd(layer, W, B, y, coef, a, (hipStream_t)stream, io_bf16(h));     // y, a in the handle's storage type
}

int cvae_op_bn_pool_act_bwd(cvae_handle h, int32_t layer, int32_t B, const float* y, const float* a, const float* da,
                            const float* coef, const float* gamma, float* dy, float* dgamma, float* dbeta, float* dbias,
                            void* scratch, void* stream) {
    return launch_bn_pool_act_bwd(layer, h->cfg.width, B, y, a, da, coef, gamma, dy, dgamma, dbeta, dbias,
                                  (float*)scratch, (hipStream_t)stream, io_bf16(h));                 // y, a, da, dy in the handle's storage type
}

int64_t cvae_op_bn_partial_floats(cvae_handle h, int32_t layer, int32_t B) {
    return (int64_t)2 * bn_num_tiles(layer, h->cfg.width, B) * kLayers[layer].cout;
}

int64_t cvae_op_msssim_ws_floats(cvae_handle h, int32_t B) { return msssim_ws_floats(h->cfg.width, B); }

int cvae_op_msssim(cvae_handle h, int32_t B, const float* img1, const float* img2, void* ws, float* scalars,
                   float* d_img1, void* stream) {
    return launch_msssim(h->cfg.width, B, img1, img2, nullptr, nullptr, (float*)ws, scalars, d_img1, nullptr, nullptr,
                         (hipStream_t)stream);
}

}  // extern "C"
