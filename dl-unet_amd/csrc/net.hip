// net.hip — the C ABI of libunet_hip.so: plan, workspace layout, Unet forward / backward
// orchestration and the per-op entry points (see include/unet_hip.h for the contract and the
// reference lines each entry point replaces).
#include "common.hpp"
#include "../../include/unet_hip.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

namespace unet {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static float *g_zero[64] = {nullptr};
static std::mutex g_zero_mu;
const float *zero_page()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("zero_page: bad device"); return nullptr; }
    std::lock_guard<std::mutex> lk(g_zero_mu);
    if (!g_zero[dev]) {
        float *p = nullptr;
        if (hipMalloc((void **)&p, 4096) != hipSuccess || hipMemset(p, 0, 4096) != hipSuccess) {
            set_error("zero_page: hipMalloc failed");
            return nullptr;
        }
        g_zero[dev] = p;
    }
    return g_zero[dev];
}

// layer indices in the reference's declaration order (network.py:23-58)
enum { C11C, C12C, C21C, C22C, C31C, C32C, C41C, C42C, C51C, C52C, UP4, C41E, C42E, UP3, C31E, C32E,
       UP2, C21E, C22E, UP1, C11E, C12E, FINAL };
static const int UP_L[4] = {UP1, UP2, UP3, UP4};
static const int C1E_L[4] = {C11E, C21E, C31E, C41E};
static const int C2E_L[4] = {C12E, C22E, C32E, C42E};

struct Plan {
    int B = 0, S = 0, base = 0, So = 0, training = 0;
    int math = 3;         // arithmetic, fixed when the forward is planned: the backward of that forward uses the same
    int ch[5];
    int ein[5], ea1[5], ea2[5], et[4], eu[4], ed1[4], ed2[4], pad[4];
    size_t a1[5], a2[5], t[4], u[4], d1[4], d2[4];
    size_t wt_fwd[UNET_N_LAYERS], wt_bwd[UNET_N_LAYERS];
    size_t wu_fwd[UNET_N_LAYERS], wu_bwd[UNET_N_LAYERS];     // Winograd-transformed filters (math mode 3), 16/9 of the 3x3 layers
    size_t g_a1[5], g_a2[5], g_t[4], g_ts[4], g_u[4], g_d1[4], g_d2[4];
    size_t slab = 0, slab_bytes = 0, small = 0, small_bytes = 0, xin = 0;
    size_t total = 0;
};

}  // namespace unet

using namespace unet;

struct unet_dp;       // dp.hip: RCCL communicator + its stream (null until unet_dp_init)
int unet_dp_free(unet_dp *d);

struct unet_handle {
    int base_ch;
    int device;
    int math;             // arithmetic of this handle's forwards (unet_config::math); -1 = the process default at each forward
    unet_dp *dp = nullptr;
    float grad_scale = 1.f;   // unet_set_grad_scale: the backward reads dlogits * grad_scale (data parallel: 1/world)
    // opt-in (unet_set_overlap / UNET_OVERLAP=1): the weight gradients of a backward stage run on an auxiliary stream next to
    // the dgrad chain (they only share dz); the streams re-join at the end of every stage
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // plans of the training forwards still awaiting their backward, keyed by workspace pointer
    std::mutex mu;
    std::vector<std::pair<void *, Plan>> live;
    void remember(void *ws, const Plan &pl)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : live) if (e.first == ws) { e.second = pl; return; }
        if (live.size() >= 16) live.erase(live.begin());
        live.emplace_back(ws, pl);
    }
    bool lookup(void *ws, Plan &pl)
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &e : live) if (e.first == ws) { pl = e.second; return true; }
        return false;
    }
};

unet_dp **unet_handle_dp_slot(unet_handle *h) { return &h->dp; }
int unet_handle_device(const unet_handle *h) { return h->device; }

namespace unet {

// Arithmetic mode of the C-ABI call this thread is in; the descriptor builders copy it into every launch descriptor
// (handle entry points: the plan's mode; per-op entry points: the process default).
static thread_local int t_math = 3;
static thread_local int t_es = 4;            // element size of activations / activation gradients / packed filters: 2 (bf16) in mode 2
struct MathScope {
    int prev;
    explicit MathScope(int m) : prev(t_math) { t_math = m; t_es = m == 2 ? 2 : 4; }
    ~MathScope() { t_math = prev; t_es = prev == 2 ? 2 : 4; }
};
// pointer arithmetic on tensors whose element size depends on the mode (the descriptors carry them as float*)
static inline const float *adv(const float *p, size_t elems) { return (const float *)((const char *)p + elems * t_es); }
static inline float *adv(float *p, size_t elems) { return (float *)((char *)p + elems * t_es); }

static size_t layer_numel(int base, int layer, bool bias)
{
    const int c[5] = {base, base * 2, base * 4, base * 8, base * 16};
    int ci, co, k;
    bool up = false;
    switch (layer) {
    case C11C: ci = 1; co = c[0]; k = 3; break;
    case C12C: ci = c[0]; co = c[0]; k = 3; break;
    case C21C: ci = c[0]; co = c[1]; k = 3; break;
    case C22C: ci = c[1]; co = c[1]; k = 3; break;
    case C31C: ci = c[1]; co = c[2]; k = 3; break;
    case C32C: ci = c[2]; co = c[2]; k = 3; break;
    case C41C: ci = c[2]; co = c[3]; k = 3; break;
    case C42C: ci = c[3]; co = c[3]; k = 3; break;
    case C51C: ci = c[3]; co = c[4]; k = 3; break;
    case C52C: ci = c[4]; co = c[4]; k = 3; break;
    case UP4: ci = c[4]; co = c[3]; k = 2; up = true; break;
    case C41E: ci = c[4]; co = c[3]; k = 3; break;
    case C42E: ci = c[3]; co = c[3]; k = 3; break;
    case UP3: ci = c[3]; co = c[2]; k = 2; up = true; break;
    case C31E: ci = c[3]; co = c[2]; k = 3; break;
    case C32E: ci = c[2]; co = c[2]; k = 3; break;
    case UP2: ci = c[2]; co = c[1]; k = 2; up = true; break;
    case C21E: ci = c[2]; co = c[1]; k = 3; break;
    case C22E: ci = c[1]; co = c[1]; k = 3; break;
    case UP1: ci = c[1]; co = c[0]; k = 2; up = true; break;
    case C11E: ci = c[1]; co = c[0]; k = 3; break;
    case C12E: ci = c[0]; co = c[0]; k = 3; break;
    default: ci = c[0]; co = 2; k = 1; break;
    }
    (void)up;
    return bias ? (size_t)co : (size_t)ci * co * k * k;
}

// floats of the Winograd U matrix of a layer (0 for the layers wino.hip does not serve: conv11c, up-convs, head)
static size_t layer_wino_floats(int base, int layer)
{
    if (layer == C11C || layer == UP4 || layer == UP3 || layer == UP2 || layer == UP1 || layer == FINAL) return 0;
    // U is kept in whole 64-row blocks (wino_u_floats); forward: rows = output channels, one sub-matrix per source of the
    // virtual concat; dgrad: rows = input channels, one launch (and sub-matrix) per source.  The larger of the two directions.
    const size_t co = layer_numel(base, layer, true);
    const size_t ci = layer_numel(base, layer, false) / 9 / co;
    const bool two = layer == C11E || layer == C21E || layer == C31E || layer == C41E;
    const size_t fwd = wino_u_floats((int)ci, (int)co);
    const size_t bwd = two ? 2 * wino_u_floats((int)co, (int)(ci / 2)) : wino_u_floats((int)co, (int)ci);
    return fwd > bwd ? fwd : bwd;
}

// One 3x3 layer's reference (OIHW) weights and their packed igemm copy.  The copy is made on first need: in math mode 3
// most launches take the Winograd path, whose filter transform reads the reference weights directly.
struct WLayer {
    const float *w; int O, I;      // reference tensor [O][I][3][3]
    bool dgrad;                    // packed copy: forward [O][9*(C1|C2)] or dgrad [I][9*O]
    float *wt; int C1, C2;
    bool packed = false;
    int ensure_packed(hipStream_t st)
    {
        if (packed) return 0;
        packed = true;
        return dgrad ? pack_conv_dgrad(w, wt, O, I, t_es, st) : pack_conv_fwd(w, wt, O, C1, C2, t_es, st);
    }
};
static WLayer wl_fwd(const float *w, int K, int C1, int C2, float *wt) { return WLayer{w, K, C1 + C2, false, wt, C1, C2}; }
static WLayer wl_dgrad(const float *w, int K, int C, float *wt) { return WLayer{w, K, C, true, wt, 0, 0}; }

// Route a 3x3 launch: math mode 3 and a shape wino.hip takes -> transform the filters (rows n0.., columns k0.. of the
// launch's filter matrix within the layer) into `wu`; otherwise make sure the packed igemm weights exist.
static int with_wino(IgemmP &p, float *wu, WLayer &L, int n0, int k0, hipStream_t st)
{
    if (p.math != 3 || !wu || !wino_applicable(p)) return L.ensure_packed(st);
    int kc = 0;
    for (int i = 0; i < p.nsrc; ++i) kc += p.src[i].nch;
    int rc = wino_transform_ref(L.w, L.I, L.dgrad ? 1 : 0, n0, p.Nn, k0, kc, wu, st);
    if (rc) return rc;
    p.wino_u = wu;
    return 0;
}

static int check_size(int S)
{
    if (S < 60 + 16 * 8 || (S - 60) % 16 != 0 || ((S - 60) / 16) % 2 != 0) {
        set_error("input size %d is not 16L+60 with L even >= 8: the reference's crop_and_concat/torch.cat "
                  "raises on it (network.py:124-127)", S);
        return UNET_E_BADSIZE;
    }
    return 0;
}

static const char *const LAYER_NAME[UNET_N_LAYERS] = {
    "conv11c", "conv12c", "conv21c", "conv22c", "conv31c", "conv32c", "conv41c", "conv42c", "conv51c", "conv52c",
    "upconv4", "conv41e", "conv42e", "upconv3", "conv31e", "conv32e", "upconv2", "conv21e", "conv22e",
    "upconv1", "conv11e", "conv12e", "finalconv"};
// profile row of a launch group: "<layer>.<fwd|dgrad|wgrad>" (bench.py maps them onto SURVEY 8a rows)
struct RowScope : ProfScope {
    static const char *make(char (&buf)[40], int layer, const char *what) { snprintf(buf, sizeof(buf), "%s.%s", LAYER_NAME[layer], what); return buf; }
    char buf[40];
    RowScope(int layer, const char *what) : ProfScope(make(buf, layer, what)) {}
};

// ---- descriptor builders (shared by the plan sizing and the launches) ----------------------------
static IgemmP conv_fwd_desc(const float *x1, int H1, int W1, int C1, int pad1, const float *x2, int C2,
                            int B, int H, int W, const float *wt, const float *bias, int K, int relu, float *y)
{
    IgemmP p{};
    p.nsrc = x2 ? 2 : 1;
    p.src[0] = GSrc{x1, H1, W1, C1, 0, C1, pad1};
    if (x2) p.src[1] = GSrc{x2, H, W, C2, 0, C2, 0};
    p.wt = wt; p.Kd = 9 * (C1 + C2);
    p.T = 9; p.TX = 3; p.stride = 1; p.oy0 = 0; p.ox0 = 0;
    p.NB = B; p.OH = H - 2; p.OW = W - 2; p.M = B * p.OH * p.OW; p.Nn = K;
    p.dst = y; p.DH = p.OH; p.DW = p.OW; p.DC = K; p.dn0 = 0;
    p.bias = bias; p.relu = relu;
    p.math = t_math;
    return p;
}

// Forward 3x3 conv over the virtual concat.  When the skip source is zero-padded (pad > 0) its taps only
// reach the output window [pad-2, pad+H1): running it as part of one GEMM would spend 30-50 % of that
// layer's MFMA work on zeros.  So: launch 1 = up-conv source over the full domain (+bias, ReLU outside the
// window), launch 2 = skip source over the window only, accumulating in place (+ReLU).  Same math, same
// K order per source; the two partial sums are added in fp32.
static int conv_fwd_launch(const float *x1, int H1, int C1, int pad1, const float *x2, int C2, int B, int H,
                           WLayer &L, const float *bias, int K, int relu, float *y, hipStream_t st, float *wu = nullptr)
{
    const int Ho = H - 2;
    const float *wt = L.wt;
    int rc;
    int w0 = pad1 - 2; if (w0 < 0) w0 = 0;
    int w1 = pad1 + H1; if (w1 > Ho) w1 = Ho;
    static const double split_thr = [] { const char *e = getenv("UNET_SPLIT_THR"); return e ? atof(e) : 0.85; }();
    // (not with bf16 tensors: the partial sum between the two launches would be rounded to bf16 — a second rounding of
    //  the layer's output — and the zero-padded taps cost that mode no memory traffic, only cheap MFMA time)
    const bool split = x2 && pad1 > 0 && t_math != 2 && (double)(w1 - w0) * (w1 - w0) < split_thr * (double)Ho * Ho;
    if (!split) {
        IgemmP p = conv_fwd_desc(x1, H1, H1, C1, pad1, x2, x2 ? C2 : 0, B, H, H, wt, bias, K, relu, y);
        if ((rc = with_wino(p, wu, L, 0, 0, st))) return rc;
        return launch_igemm(p, st);
    }
    const int ldw = 9 * (C1 + C2);
    IgemmP a = conv_fwd_desc(x2, H, H, C2, 0, nullptr, 0, B, H, H, adv(wt, 9 * C1), bias, K, relu, y);
    a.ldw = ldw;
    if (relu) { a.rw0 = w0; a.rw1 = w1; }
    if ((rc = with_wino(a, wu, L, 0, C1, st))) return rc;
    if ((rc = launch_igemm(a, st))) return rc;
    IgemmP b = conv_fwd_desc(x1, H1, H1, C1, pad1, nullptr, 0, B, H, H, wt, nullptr, K, relu, y);
    b.ldw = ldw;
    b.OH = b.OW = w1 - w0; b.M = B * b.OH * b.OW; b.oy0 = b.ox0 = w0;
    b.scatter = 2; b.dwy0 = b.dwx0 = w0; b.DH = b.DW = Ho;
    b.add = y;
    if ((rc = with_wino(b, wu ? wu + wino_u_floats(C2, K) : nullptr, L, 0, 0, st))) return rc;
    return launch_igemm(b, st);
}

// dgrad of a 3x3 valid conv: dx over the window [oy0, oy0+OHW) of the conv's (virtual) input
static IgemmP conv_dgrad_desc(const float *dz, int Ho, int Wo, int K, int B, int OHW, int o0,
                              const float *wt_rows, int Nn, float *dx, const float *mask, const float *add)
{
    IgemmP p{};
    p.nsrc = 1;
    p.src[0] = GSrc{dz, Ho, Wo, K, 0, K, 2};
    p.wt = wt_rows; p.Kd = 9 * K;
    p.T = 9; p.TX = 3; p.stride = 1; p.oy0 = o0; p.ox0 = o0;
    p.NB = B; p.OH = OHW; p.OW = OHW; p.M = B * OHW * OHW; p.Nn = Nn;
    p.dst = dx; p.DH = OHW; p.DW = OHW; p.DC = Nn; p.dn0 = 0;
    p.mask = mask; p.add = add;
    p.math = t_math;
    return p;
}

static WgradP conv_wgrad_desc(const float *X, int XH, int XC, int xpad, const float *dz, int Ho, int K, int B,
                              float *dw, int Ctot, int c_off, float *slab, size_t slab_bytes, float *db = nullptr)
{
    WgradP p{};
    p.X = X; p.XH = XH; p.XW = XH; p.XC = XC; p.xc0 = 0; p.xpad = xpad;
    p.Y = dz; p.YH = Ho; p.YW = Ho; p.YC = K; p.yc0 = 0;
    p.NB = B; p.stride = 1; p.TY = 3; p.TX = 3; p.oy0 = 0; p.ox0 = 0;
    int w0 = xpad - 2; if (w0 < 0) w0 = 0;
    int w1 = xpad + XH; if (w1 > Ho) w1 = Ho;
    p.ywin0 = w0; p.ywin1 = w1; p.xwin0 = w0; p.xwin1 = w1;
    p.Ci = XC; p.Cj = K;
    p.out = dw ? dw + (size_t)c_off * 9 : nullptr; p.si = 9; p.sj = (long)Ctot * 9; p.st = 1;
    p.slab = slab; p.slab_bytes = slab_bytes; p.db = db;
    p.math = t_math;
    return p;
}

static WgradP upconv_wgrad_desc(const float *x, int H, int Ci, const float *dy, int Co, int B, float *dw,
                                float *slab, size_t slab_bytes, float *db = nullptr)
{
    WgradP p{};
    p.X = dy; p.XH = 2 * H; p.XW = 2 * H; p.XC = Co; p.xc0 = 0; p.xpad = 0;
    p.Y = x; p.YH = H; p.YW = H; p.YC = Ci; p.yc0 = 0;
    p.NB = B; p.stride = 2; p.TY = 2; p.TX = 2; p.oy0 = 0; p.ox0 = 0;
    p.ywin0 = 0; p.ywin1 = H; p.xwin0 = 0; p.xwin1 = H;
    p.Ci = Co; p.Cj = Ci;
    p.out = dw; p.si = 4; p.sj = (long)Co * 4; p.st = 1;
    p.slab = slab; p.slab_bytes = slab_bytes;
    p.db = db; p.db_on_x = db ? 1 : 0;        // db[co] = sum dOut: X is dOut here
    p.math = t_math;
    return p;
}

static int make_plan(Plan &pl, int base, int B, int S, int training, int math)
{
    int rc = check_size(S);
    if (rc) return rc;
    if (B <= 0) { set_error("batch must be positive"); return UNET_E_BADARG; }
    pl = Plan();
    pl.B = B; pl.S = S; pl.base = base; pl.training = training; pl.math = math;
    MathScope ms(math);                      // the slab sizing below builds weight-gradient descriptors
    for (int l = 0; l < 5; ++l) pl.ch[l] = base << l;
    int cur = S;
    for (int l = 0; l < 5; ++l) {
        pl.ein[l] = cur; pl.ea1[l] = cur - 2; pl.ea2[l] = cur - 4;
        if (l < 4) { pl.et[l] = pl.ea2[l] / 2; cur = pl.et[l]; }
    }
    int d = pl.ea2[4];
    for (int l = 3; l >= 0; --l) {
        pl.eu[l] = 2 * d; pl.pad[l] = (pl.eu[l] - pl.et[l]) / 2;
        // pad > 0: the skip is zero-padded (every S >= 380); pad < 0: it is cropped (188 <= S < 380).
        // Both are the reference's F.pad(A, (-c,)*4) with c = int((A-B)/2) (network.py:124-126).
        if ((pl.eu[l] - pl.et[l]) % 2) { set_error("internal: odd skip difference"); return UNET_E_BADSIZE; }
        pl.ed1[l] = pl.eu[l] - 2; pl.ed2[l] = pl.eu[l] - 4; d = pl.ed2[l];
    }
    pl.So = d;
    if (math == 2 && base % 64 != 0) { set_error("arithmetic mode 2 (bf16 tensors) needs base_ch %% 64 == 0 (got %d)", base); return UNET_E_UNSUPPORTED; }
    // element size of activations, their gradients and the packed filters: bf16 in mode 2, else fp32
    const size_t es = math == 2 ? 2 : 4;
    size_t off = 0;
    auto take_b = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    auto take = [&](size_t elems) { return take_b(elems * es); };
    auto sq = [&](int e, int c) { return (size_t)B * e * e * c; };
    for (int l = 0; l < 5; ++l) { pl.a1[l] = take(sq(pl.ea1[l], pl.ch[l])); pl.a2[l] = take(sq(pl.ea2[l], pl.ch[l])); }
    for (int l = 0; l < 4; ++l) {
        pl.t[l] = take(sq(pl.et[l], pl.ch[l])); pl.u[l] = take(sq(pl.eu[l], pl.ch[l]));
        pl.d1[l] = take(sq(pl.ed1[l], pl.ch[l])); pl.d2[l] = take(sq(pl.ed2[l], pl.ch[l]));
    }
    for (int i = 0; i < UNET_N_LAYERS; ++i) pl.wt_fwd[i] = take(layer_numel(base, i, false));
    for (int i = 0; i < UNET_N_LAYERS; ++i) pl.wu_fwd[i] = take_b(math == 3 ? layer_wino_floats(base, i) * 4 : 0);
    if (training) {
        pl.xin = take_b((size_t)B * S * S * 4);
        for (int i = 0; i < UNET_N_LAYERS; ++i) pl.wt_bwd[i] = take(layer_numel(base, i, false));
        for (int i = 0; i < UNET_N_LAYERS; ++i) pl.wu_bwd[i] = take_b(math == 3 ? layer_wino_floats(base, i) * 4 : 0);
        for (int l = 0; l < 5; ++l) { pl.g_a1[l] = take(sq(pl.ea1[l], pl.ch[l])); pl.g_a2[l] = take(sq(pl.ea2[l], pl.ch[l])); }
        for (int l = 0; l < 4; ++l) {
            pl.g_t[l] = take(sq(pl.et[l], pl.ch[l])); pl.g_ts[l] = take(sq(pl.et[l], pl.ch[l]));
            pl.g_u[l] = take(sq(pl.eu[l], pl.ch[l]));
            pl.g_d1[l] = take(sq(pl.ed1[l], pl.ch[l])); pl.g_d2[l] = take(sq(pl.ed2[l], pl.ch[l]));
        }
        // split-K slab: the largest need over all weight-gradient launches
        size_t need = 0;
        auto upd = [&](const WgradP &w) { size_t n = wgrad_slab_need(w); if (n > need) need = n; };
        {
            for (int l = 0; l < 5; ++l) {
                if (l > 0) upd(conv_wgrad_desc(nullptr, pl.ein[l], pl.ch[l - 1], 0, nullptr, pl.ea1[l], pl.ch[l], B, nullptr, pl.ch[l - 1], 0, nullptr, 0));
                upd(conv_wgrad_desc(nullptr, pl.ea1[l], pl.ch[l], 0, nullptr, pl.ea2[l], pl.ch[l], B, nullptr, pl.ch[l], 0, nullptr, 0));
            }
            for (int l = 0; l < 4; ++l) {
                upd(conv_wgrad_desc(nullptr, pl.et[l], pl.ch[l], pl.pad[l], nullptr, pl.ed1[l], pl.ch[l], B, nullptr, 2 * pl.ch[l], 0, nullptr, 0));
                upd(conv_wgrad_desc(nullptr, pl.eu[l], pl.ch[l], 0, nullptr, pl.ed1[l], pl.ch[l], B, nullptr, 2 * pl.ch[l], 0, nullptr, 0));
                upd(conv_wgrad_desc(nullptr, pl.ed1[l], pl.ch[l], 0, nullptr, pl.ed2[l], pl.ch[l], B, nullptr, pl.ch[l], 0, nullptr, 0));
                upd(upconv_wgrad_desc(nullptr, pl.eu[l] / 2, pl.ch[l + 1], nullptr, pl.ch[l], B, nullptr, nullptr, 0));
            }
        }
        pl.slab_bytes = need;
        pl.slab = take_b(need);
        // small scratch: bias-grad partials, conv11c / head partials
        size_t sm = 0;
        auto upds = [&](size_t n) { if (n > sm) sm = n; };
        for (int l = 0; l < 5; ++l) { upds(bias_grad_scratch_bytes(sq(pl.ea1[l], 1), pl.ch[l])); upds(bias_grad_scratch_bytes(sq(pl.ea2[l], 1), pl.ch[l])); }
        for (int l = 0; l < 4; ++l) { upds(bias_grad_scratch_bytes(sq(pl.eu[l], 1), pl.ch[l])); upds(bias_grad_scratch_bytes(sq(pl.ed1[l], 1), pl.ch[l])); }
        upds(unet_conv1ch_bwd_scratch_bytes(B, S, base));
        upds(unet_head1x1_bwd_scratch_bytes(B, pl.So, pl.So, base));
        pl.small_bytes = sm;
        pl.small = take_b(sm + 4);
    }
    pl.total = off;
    return 0;
}

}  // namespace unet

// =================================================================================================
extern "C" {

const char *unet_last_error(void) { return g_err; }
int unet_abi_version(void) { return 4; }

int unet_set_math(int mode)
{
    ARG_CHECK(mode >= 0 && mode <= 3, "unet_set_math: mode must be 0 (fp32 MFMA), 1 (bf16x3), 2 (bf16) or 3 (fp32 Winograd)");
    set_math_mode(mode);
    return 0;
}
int unet_get_math(void) { return get_math_mode(); }

int unet_set_lds_dma(int mode)
{
    ARG_CHECK(mode == 0 || mode == 1, "unet_set_lds_dma: mode must be 0 (global_load_lds) or 1 (buffer descriptors when tensors fit)");
    set_lds_dma_mode(mode);
    return 0;
}

// the device a call must run on: the handle's.  Everything the library keeps per device (zero page, LDS attributes) is
// keyed on hipGetDevice(), so a call made while another device is current would enqueue on the wrong device.
#define CHECK_DEVICE(h, what)                                                                                          \
    do {                                                                                                               \
        int cur_ = -1;                                                                                                 \
        HIP_TRY(hipGetDevice(&cur_));                                                                                  \
        ARG_CHECK(cur_ == (h)->device, what ": the handle belongs to device %d but device %d is current "               \
                  "(one process per GPU: hipSetDevice / torch.cuda.set_device first)", (h)->device, cur_);             \
    } while (0)

int unet_create(unet_handle **out, const unet_config *cfg)
{
    ARG_CHECK(out && cfg, "unet_create: null argument");
    ARG_CHECK(cfg->base_ch == 64 || cfg->base_ch == 32, "unet_create: base_ch %d unsupported (32 or 64)", cfg->base_ch);
    ARG_CHECK(cfg->math >= -1 && cfg->math <= 3, "unet_create: math %d unsupported (-1 = process default, 0..3)", cfg->math);
    // the zero page is made on the handle's device; the caller's current device is left as it was
    int prev = -1;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(cfg->device));
    const bool ok = zero_page() != nullptr;
    HIP_TRY(hipSetDevice(prev));
    if (!ok) return UNET_E_BADARG;
    unet_handle *h = new unet_handle();
    h->base_ch = cfg->base_ch;
    h->device = cfg->device;
    h->math = cfg->math;
    *out = h;
    return 0;
}

int unet_destroy(unet_handle *h)
{
    if (h && h->dp) (void)unet_dp_free(h->dp);
    if (h) {
        if (h->aux) { (void)hipStreamSynchronize(h->aux); (void)hipStreamDestroy(h->aux); }
        if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
        if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    }
    delete h;
    return 0;
}

int unet_set_grad_scale(unet_handle *h, float scale)
{
    ARG_CHECK(h, "unet_set_grad_scale: null handle");
    ARG_CHECK(scale > 0.f && scale <= 1.f, "unet_set_grad_scale: scale must be in (0, 1] (1/world), got %g", (double)scale);
    h->grad_scale = scale;
    return 0;
}

// -1 (default): by what was measured - on with bf16 tensors (+2.6 ... +3 % per step: the weight gradients fill the dgrad chain's
// partial rounds) and in fp32 at batches of <= 4 tiles (+0.2 ... +1.2 %), off in fp32 at larger batches (-0.6 ... -1 % at B = 8: the
// co-running fp32 MFMA kernels slow each other by more)
static int g_overlap = [] { const char *e = getenv("UNET_OVERLAP"); return e ? atoi(e) : -1; }();
int unet_set_overlap(int on)
{
    ARG_CHECK(on >= -1 && on <= 1, "unet_set_overlap: -1 (per arithmetic mode), 0 or 1");
    g_overlap = on;
    return 0;
}

static int handle_math(const unet_handle *h) { return h->math >= 0 ? h->math : get_math_mode(); }

int unet_output_size(int S, int *out_size)
{
    int rc = check_size(S);
    if (rc) return rc;
    if (out_size) *out_size = S - 184;
    return 0;
}

int unet_param_count(const unet_handle *h, int idx, size_t *numel)
{
    ARG_CHECK(h && numel && idx >= 0 && idx < UNET_N_PARAMS, "unet_param_count: bad argument");
    *numel = layer_numel(h->base_ch, idx / 2, idx & 1);
    return 0;
}

size_t unet_workspace_bytes(const unet_handle *h, int B, int S, int training)
{
    if (!h) { set_error("null handle"); return 0; }
    Plan pl;
    if (make_plan(pl, h->base_ch, B, S, training, handle_math(h))) return 0;
    return pl.total;
}

double unet_flops(const unet_handle *h, int B, int S, int backward)
{
    if (!h) return 0.0;
    Plan pl;
    if (make_plan(pl, h->base_ch, B, S, 0, handle_math(h))) return 0.0;
    double f = 0.0, f11c = 0.0;
    auto conv = [&](int eo, int ci, int co, int k) { return 2.0 * B * (double)eo * eo * ci * co * k * k; };
    for (int l = 0; l < 5; ++l) {
        const int ci = l ? pl.ch[l - 1] : 1;
        const double a = conv(pl.ea1[l], ci, pl.ch[l], 3);
        if (l == 0) f11c = a;
        f += a + conv(pl.ea2[l], pl.ch[l], pl.ch[l], 3);
    }
    for (int l = 0; l < 4; ++l) {
        f += 2.0 * B * (double)(pl.eu[l] / 2) * (pl.eu[l] / 2) * pl.ch[l + 1] * pl.ch[l] * 4;   // up-conv
        f += conv(pl.ed1[l], 2 * pl.ch[l], pl.ch[l], 3) + conv(pl.ed2[l], pl.ch[l], pl.ch[l], 3);
    }
    f += conv(pl.So, pl.ch[0], 2, 1);
    return backward ? 3.0 * f - f11c : f;      // bwd = dgrad + wgrad, conv11c needs no dgrad (A23)
}

#define WS(off) ((float *)((char *)workspace + (off)))
#define PARAM(i) ((const float *)params[(i)])

int unet_forward(unet_handle *h, const void *const *params, const void *x, void *logits, int B, int S,
                 void *workspace, size_t workspace_bytes, int training, void *stream)
{
    ARG_CHECK(h && params && x && logits && workspace, "unet_forward: null argument");
    CHECK_DEVICE(h, "unet_forward");
    Plan pl;
    int rc = make_plan(pl, h->base_ch, B, S, training, handle_math(h));
    if (rc) return rc;
    MathScope ms(pl.math);
    ARG_CHECK(workspace_bytes >= pl.total, "unet_forward: workspace too small (%zu < %zu)", workspace_bytes, pl.total);
    ARG_CHECK(((uintptr_t)workspace & 255) == 0, "unet_forward: workspace must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int *ch = pl.ch;

    // repack the parameters (reference layout, owned by the caller and updated by its optimizer)
    // (the 3x3 layers' weights are packed lazily, only if a launch of the layer takes the implicit-GEMM path: with_wino)
    for (int l = 0; l < 4; ++l) {
        RowScope rs(UP_L[l], "fwd");
        if ((rc = pack_upconv_fwd(PARAM(2 * UP_L[l]), WS(pl.wt_fwd[UP_L[l]]), ch[l + 1], ch[l], t_es, st))) return rc;
    }

    // encoder (network.py:131-156); the input is kept for conv11c's weight gradient
    {
        RowScope rs(C11C, "fwd");
        if (training) HIP_TRY(hipMemcpyAsync(WS(pl.xin), x, (size_t)B * S * S * sizeof(float), hipMemcpyDeviceToDevice, st));
        if ((rc = conv1ch_fwd((const float *)x, B, S, PARAM(0), PARAM(1), ch[0], WS(pl.a1[0]), t_es, st))) return rc;
    }
    for (int l = 0; l < 5; ++l) {
        if (l > 0) {
            RowScope rs(2 * l, "fwd");
            IgemmP p = conv_fwd_desc(WS(pl.t[l - 1]), pl.ein[l], pl.ein[l], ch[l - 1], 0, nullptr, 0, B, pl.ein[l], pl.ein[l],
                                     WS(pl.wt_fwd[2 * l]), PARAM(2 * (2 * l) + 1), ch[l], 1, WS(pl.a1[l]));
            WLayer L = wl_fwd(PARAM(2 * (2 * l)), ch[l], ch[l - 1], 0, WS(pl.wt_fwd[2 * l]));
            if ((rc = with_wino(p, WS(pl.wu_fwd[2 * l]), L, 0, 0, st))) return rc;
            if ((rc = launch_igemm(p, st))) return rc;
        }
        bool pool_fused = false;
        {
            RowScope rs(2 * l + 1, "fwd");
            IgemmP p = conv_fwd_desc(WS(pl.a1[l]), pl.ea1[l], pl.ea1[l], ch[l], 0, nullptr, 0, B, pl.ea1[l], pl.ea1[l],
                                     WS(pl.wt_fwd[2 * l + 1]), PARAM(2 * (2 * l + 1) + 1), ch[l], 1, WS(pl.a2[l]));
            WLayer L2 = wl_fwd(PARAM(2 * (2 * l + 1)), ch[l], ch[l], 0, WS(pl.wt_fwd[2 * l + 1]));
            if ((rc = with_wino(p, WS(pl.wu_fwd[2 * l + 1]), L2, 0, 0, st))) return rc;
            pool_fused = l < 4 && p.wino_u && wino_fuses_pool(p);       // the Winograd epilogue writes t[l] too
            if (pool_fused) p.pool_dst = WS(pl.t[l]);
            if ((rc = launch_igemm(p, st))) return rc;
        }
        if (l < 4 && !pool_fused) {
            char nm[40]; snprintf(nm, sizeof(nm), "pool%d.fwd", l + 1);
            ProfScope ps(nm);
            if ((rc = maxpool2_fwd(WS(pl.a2[l]), WS(pl.t[l]), B, pl.ea2[l], pl.ea2[l], ch[l], t_es, st))) return rc;
        }
    }
    // decoder (network.py:159-188): up-conv, virtual zero-pad-concat, two convs
    const float *dsrc = WS(pl.a2[4]);
    for (int l = 3; l >= 0; --l) {
        const int hin = pl.eu[l] / 2;
        {
            RowScope rs(UP_L[l], "fwd");
            IgemmP u{};
            u.nsrc = 1; u.src[0] = GSrc{dsrc, hin, hin, ch[l + 1], 0, ch[l + 1], 0};
            u.wt = WS(pl.wt_fwd[UP_L[l]]); u.Kd = ch[l + 1];
            u.T = 1; u.TX = 1; u.stride = 1;
            u.NB = B; u.OH = hin; u.OW = hin; u.M = B * hin * hin; u.Nn = 4 * ch[l];
            u.dst = WS(pl.u[l]); u.DH = pl.eu[l]; u.DW = pl.eu[l]; u.DC = ch[l]; u.scatter = 1; u.cout = ch[l];
            u.bias = PARAM(2 * UP_L[l] + 1);
            u.math = pl.math;
            if ((rc = launch_igemm(u, st))) return rc;
        }
        {
            RowScope rs(C1E_L[l], "fwd");
            WLayer L1 = wl_fwd(PARAM(2 * C1E_L[l]), ch[l], ch[l], ch[l], WS(pl.wt_fwd[C1E_L[l]]));
            if ((rc = conv_fwd_launch(WS(pl.t[l]), pl.et[l], ch[l], pl.pad[l], WS(pl.u[l]), ch[l], B, pl.eu[l],
                                      L1, PARAM(2 * C1E_L[l] + 1), ch[l], 1, WS(pl.d1[l]), st, WS(pl.wu_fwd[C1E_L[l]])))) return rc;
        }
        {
            RowScope rs(C2E_L[l], "fwd");
            IgemmP c2 = conv_fwd_desc(WS(pl.d1[l]), pl.ed1[l], pl.ed1[l], ch[l], 0, nullptr, 0, B, pl.ed1[l], pl.ed1[l],
                                      WS(pl.wt_fwd[C2E_L[l]]), PARAM(2 * C2E_L[l] + 1), ch[l], 1, WS(pl.d2[l]));
            WLayer L2e = wl_fwd(PARAM(2 * C2E_L[l]), ch[l], ch[l], 0, WS(pl.wt_fwd[C2E_L[l]]));
            if ((rc = with_wino(c2, WS(pl.wu_fwd[C2E_L[l]]), L2e, 0, 0, st))) return rc;
            if ((rc = launch_igemm(c2, st))) return rc;
        }
        dsrc = WS(pl.d2[l]);
    }
    {
        RowScope rs(FINAL, "fwd");
        if ((rc = head1x1_fwd(WS(pl.d2[0]), B, pl.So, pl.So, ch[0], PARAM(2 * FINAL), PARAM(2 * FINAL + 1), (float *)logits, t_es, st))) return rc;
    }
    if (training) h->remember(workspace, pl);
    return 0;
}

int unet_activation_bytes(const unet_handle *h)
{
    if (!h) return 0;
    return handle_math(h) == 2 ? 2 : 4;
}

/* Debug/introspection: byte offset and element count of a named workspace buffer, e.g. "a1_0",
 * "a2_4", "t_2", "u_3", "d1_0", "d2_3", and with training=1 "g_a1_0", "g_a2_4", "g_t_1", "g_ts_1",
 * "g_u_2", "g_d1_3", "g_d2_3", "xin".  Tensors are NHWC [B,e,e,C]; extent and channels describe them. */
int unet_debug_buffer(const unet_handle *h, int B, int S, int training, const char *name,
                      size_t *offset, int *extent, int *channels)
{
    ARG_CHECK(h && name && offset && extent && channels, "unet_debug_buffer: null argument");
    Plan pl;
    int rc = make_plan(pl, h->base_ch, B, S, training, handle_math(h));
    if (rc) return rc;
    char kind[16];
    int l = 0;
    const char *us = strrchr(name, '_');
    if (!strcmp(name, "xin")) { ARG_CHECK(training, "xin exists only with training=1"); *offset = pl.xin; *extent = S; *channels = 1; return 0; }
    ARG_CHECK(us && (size_t)(us - name) < sizeof(kind), "unet_debug_buffer: bad name %s", name);
    memcpy(kind, name, us - name); kind[us - name] = 0;
    l = atoi(us + 1);
    ARG_CHECK(l >= 0 && l < 5, "unet_debug_buffer: bad level in %s", name);
    const bool g = !strncmp(kind, "g_", 2);
    ARG_CHECK(!g || training, "gradient buffers exist only with training=1");
    const char *k = g ? kind + 2 : kind;
    *channels = pl.ch[l];
    if (!strcmp(k, "a1")) { *offset = g ? pl.g_a1[l] : pl.a1[l]; *extent = pl.ea1[l]; return 0; }
    if (!strcmp(k, "a2")) { *offset = g ? pl.g_a2[l] : pl.a2[l]; *extent = pl.ea2[l]; return 0; }
    ARG_CHECK(l < 4, "unet_debug_buffer: bad level in %s", name);
    if (!strcmp(k, "t")) { *offset = g ? pl.g_t[l] : pl.t[l]; *extent = pl.et[l]; return 0; }
    if (!strcmp(k, "ts") && g) { *offset = pl.g_ts[l]; *extent = pl.et[l]; return 0; }
    if (!strcmp(k, "u")) { *offset = g ? pl.g_u[l] : pl.u[l]; *extent = pl.eu[l]; return 0; }
    if (!strcmp(k, "d1")) { *offset = g ? pl.g_d1[l] : pl.d1[l]; *extent = pl.ed1[l]; return 0; }
    if (!strcmp(k, "d2")) { *offset = g ? pl.g_d2[l] : pl.d2[l]; *extent = pl.ed2[l]; return 0; }
    set_error("unet_debug_buffer: unknown buffer %s", name);
    return UNET_E_BADARG;
}

// ---- backward ------------------------------------------------------------------------------------
// Stage order = reverse layer order, so gradient buckets complete early for the all-reduce:
//   0: finalconv, conv12e, conv11e, upconv1      1: conv22e, conv21e, upconv2
//   2: conv32e, conv31e, upconv3                 3: conv42e, conv41e, upconv4
//   4: conv52c, conv51c                          5: conv42c ... conv11c
static const int N_STAGES = 6;
int unet_backward_stages(void) { return N_STAGES; }

int unet_backward_stage_params(int stage, int *idx, int cap)
{
    int layers[12], n = 0;
    if (stage >= 0 && stage < 4) {
        const int l = stage;
        if (l == 0) layers[n++] = FINAL;
        layers[n++] = C2E_L[l]; layers[n++] = C1E_L[l]; layers[n++] = UP_L[l];
    } else if (stage == 4) {
        layers[n++] = C52C; layers[n++] = C51C;
    } else if (stage == 5) {
        for (int l = C42C; l >= C11C; --l) layers[n++] = l;
    }
    int k = 0;
    for (int i = 0; i < n; ++i)
        for (int b = 0; b < 2; ++b)
            if (k < cap && idx) idx[k++] = 2 * layers[i] + b; else if (!idx || k >= cap) ++k;
    return 2 * n;
}

#define GRAD(i) ((float *)grads[(i)])

// Stream of the weight-gradient launches of the backward stage this thread is in: the caller's stream, or (overlap on) the
// handle's auxiliary stream after it has been made to wait for everything enqueued on the caller's stream so far - so a
// weight gradient must be enqueued BEFORE the dgrad it is to run next to.
struct WgradStream {
    unet_handle *h; hipStream_t main; bool overlap; bool used;
    hipError_t err = hipSuccess;     // sticky: a failed fork / join means the two streams are not ordered - the stage must not report success
    hipStream_t get()
    {
        if (!overlap) return main;
        // fork failed: the weight gradient stays on the caller's stream (ordered by construction)
        hipError_t e = hipEventRecord(h->ev_fork, main);
        if (e == hipSuccess) e = hipStreamWaitEvent(h->aux, h->ev_fork, 0);
        if (e != hipSuccess) { if (err == hipSuccess) err = e; overlap = false; return main; }
        used = true;
        return h->aux;
    }
    void join()
    {
        if (!used) return;
        hipError_t e = hipEventRecord(h->ev_join, h->aux);
        if (e == hipSuccess) e = hipStreamWaitEvent(main, h->ev_join, 0);
        if (e != hipSuccess) {
            // last resort: the caller's stream cannot be made to wait, so the host waits for the auxiliary stream
            (void)hipStreamSynchronize(h->aux);
            if (err == hipSuccess) err = e;
        }
        used = false;
    }
};
static thread_local WgradStream *t_wst = nullptr;
static inline hipStream_t wgrad_stream(hipStream_t st) { return t_wst ? t_wst->get() : st; }
static inline hipStream_t wst_same(hipStream_t st) { return (t_wst && t_wst->overlap && t_wst->used) ? t_wst->h->aux : st; }   // right after a wgrad_stream() call

static int conv_backward(const Plan &pl, void *workspace, hipStream_t st, const void *const *params, void *const *grads,
                         int layer, const float *X, int XH, int C, const float *dz, int Ho, int K,
                         float *dx, const float *mask, const float *add)
{
    // single-source 3x3 conv: wgrad + bias grad, dgrad (optional)
    const int B = pl.B;
    int rc;
    {
        RowScope rs(layer, "wgrad");
        WgradP w = conv_wgrad_desc(X, XH, C, 0, dz, Ho, K, B, GRAD(2 * layer), C, 0, WS(pl.slab), pl.slab_bytes, GRAD(2 * layer + 1));
        if ((rc = launch_wgrad(w, wgrad_stream(st)))) return rc;
    }
    if (dx) {
        RowScope rs(layer, "dgrad");
        WLayer L = wl_dgrad(PARAM(2 * layer), K, C, WS(pl.wt_bwd[layer]));
        IgemmP d = conv_dgrad_desc(dz, Ho, Ho, K, B, XH, 0, WS(pl.wt_bwd[layer]), C, dx, mask, add);
        if ((rc = with_wino(d, WS(pl.wu_bwd[layer]), L, 0, 0, st))) return rc;
        if ((rc = launch_igemm(d, st))) return rc;
    }
    return 0;
}

static int pool_backward(const Plan &pl, void *workspace, int l, void *stream)
{
    char nm[40]; snprintf(nm, sizeof(nm), "pool%d.bwd", l + 1);
    ProfScope ps(nm);
    return maxpool2_bwd(WS(pl.a2[l]), WS(pl.g_t[l]), WS(pl.g_a2[l]), pl.B, pl.ea2[l], pl.ea2[l], pl.ch[l], t_es, (hipStream_t)stream);
}

static int backward_stage_body(unet_handle *h, const Plan &pl, int stage, const void *const *params, const void *dlogits, void *const *grads,
                               void *workspace, void *stream);

int unet_backward_stage(unet_handle *h, int stage, const void *const *params, const void *dlogits, void *const *grads,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    ARG_CHECK(h && params && grads && workspace, "unet_backward: null argument");
    CHECK_DEVICE(h, "unet_backward");
    Plan pl;
    if (!h->lookup(workspace, pl)) {
        set_error("unet_backward: no training forward has been run on this workspace");
        return UNET_E_NOTREADY;
    }
    ARG_CHECK(workspace_bytes >= pl.total, "unet_backward: workspace too small");
    ARG_CHECK(stage >= 0 && stage < N_STAGES, "unet_backward: bad stage %d", stage);
    MathScope ms(pl.math);                   // the arithmetic the forward was planned with
    // (no overlap while per-launch events are recorded: launches of two streams would interleave their begin / end events)
    // (fp32 at batches of <= 4 tiles: the deep layers' launches leave CUs idle that the weight gradients can take: +0.7 % at B = 2,
    //  +1.2 % at B = 1; at B = 8 the two families only get in each other's way)
    const bool overlap = (g_overlap < 0 ? (pl.math == 2 || pl.B <= 4) : g_overlap != 0) && !prof_active();
    if (overlap && !h->aux) {
        HIP_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
    }
    WgradStream wst{h, (hipStream_t)stream, overlap, false};
    t_wst = &wst;
    int rc = backward_stage_body(h, pl, stage, params, dlogits, grads, workspace, stream);
    wst.join();                              // every path re-joins the streams before the caller sees the stage as enqueued
    t_wst = nullptr;
    if (rc == 0 && wst.err != hipSuccess) {
        set_error("unet_backward: ordering the weight-gradient stream against the caller's failed: %s", hipGetErrorString(wst.err));
        rc = (int)wst.err;
    }
    return rc;
}

static int backward_stage_body(unet_handle *h, const Plan &pl, int stage, const void *const *params, const void *dlogits, void *const *grads,
                               void *workspace, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const int B = pl.B;
    const int *ch = pl.ch;
    int rc;

    if (stage < 4) {
        const int l = stage;
        if (l == 0) {
            ARG_CHECK(dlogits, "unet_backward: null dlogits");
            // finalconv backward, fused with the ReLU backward of conv12e -> dz of conv12e
            RowScope rs(FINAL, "bwd");
            if ((rc = head1x1_bwd(WS(pl.d2[0]), B, pl.So, pl.So, ch[0], PARAM(2 * FINAL), (const float *)dlogits, h->grad_scale, WS(pl.g_d2[0]),
                                  GRAD(2 * FINAL), GRAD(2 * FINAL + 1), WS(pl.small), t_es, st))) return rc;
        }
        // conv_l2e: input d1[l] (ReLU output of conv_l1e)
        if ((rc = conv_backward(pl, workspace, st, params, grads, C2E_L[l], WS(pl.d1[l]), pl.ed1[l], ch[l], WS(pl.g_d2[l]), pl.ed2[l], ch[l],
                                WS(pl.g_d1[l]), WS(pl.d1[l]), nullptr))) return rc;
        // conv_l1e: virtual concat input.  dgrad per source half (skip half only over the crop window)
        const int lay = C1E_L[l];
        {
            RowScope rs(lay, "wgrad");
            WgradP ws_ = conv_wgrad_desc(WS(pl.t[l]), pl.et[l], ch[l], pl.pad[l], WS(pl.g_d1[l]), pl.ed1[l], ch[l], B,
                                         GRAD(2 * lay), 2 * ch[l], 0, WS(pl.slab), pl.slab_bytes);
            if ((rc = launch_wgrad(ws_, wgrad_stream(st)))) return rc;
            WgradP wu = conv_wgrad_desc(WS(pl.u[l]), pl.eu[l], ch[l], 0, WS(pl.g_d1[l]), pl.ed1[l], ch[l], B,
                                        GRAD(2 * lay), 2 * ch[l], ch[l], WS(pl.slab), pl.slab_bytes, GRAD(2 * lay + 1));
            if ((rc = launch_wgrad(wu, wst_same(st)))) return rc;
        }
        WLayer Ld = wl_dgrad(PARAM(2 * lay), ch[l], 2 * ch[l], WS(pl.wt_bwd[lay]));
        {
            RowScope rs(lay, "dgrad");
            IgemmP ds = conv_dgrad_desc(WS(pl.g_d1[l]), pl.ed1[l], pl.ed1[l], ch[l], B, pl.et[l], pl.pad[l],
                                        WS(pl.wt_bwd[lay]), ch[l], WS(pl.g_ts[l]), nullptr, nullptr);
            if ((rc = with_wino(ds, WS(pl.wu_bwd[lay]), Ld, 0, 0, st))) return rc;
            if ((rc = launch_igemm(ds, st))) return rc;
            IgemmP du = conv_dgrad_desc(WS(pl.g_d1[l]), pl.ed1[l], pl.ed1[l], ch[l], B, pl.eu[l], 0,
                                        adv(WS(pl.wt_bwd[lay]), (size_t)ch[l] * 9 * ch[l]), ch[l], WS(pl.g_u[l]), nullptr, nullptr);
            if ((rc = with_wino(du, WS(pl.wu_bwd[lay]) + wino_u_floats(ch[l], ch[l]), Ld, ch[l], 0, st))) return rc;
            if ((rc = launch_igemm(du, st))) return rc;
        }
        // upconv_l: input is d2[l+1] (or a2[4]); its dgrad is masked by that ReLU output
        {
            const int ul = UP_L[l];
            const int hin = pl.eu[l] / 2;
            const float *uin = l == 3 ? WS(pl.a2[4]) : WS(pl.d2[l + 1]);
            float *dzin = l == 3 ? WS(pl.g_a2[4]) : WS(pl.g_d2[l + 1]);
            {
                RowScope rs(ul, "wgrad");
                WgradP w = upconv_wgrad_desc(uin, hin, ch[l + 1], WS(pl.g_u[l]), ch[l], B, GRAD(2 * ul), WS(pl.slab), pl.slab_bytes, GRAD(2 * ul + 1));
                if ((rc = launch_wgrad(w, wgrad_stream(st)))) return rc;
            }
            {
                RowScope rs(ul, "dgrad");
                if ((rc = pack_upconv_dgrad(PARAM(2 * ul), WS(pl.wt_bwd[ul]), ch[l + 1], ch[l], t_es, st))) return rc;
                IgemmP d{};
                d.nsrc = 1; d.src[0] = GSrc{WS(pl.g_u[l]), pl.eu[l], pl.eu[l], ch[l], 0, ch[l], 0};
                d.wt = WS(pl.wt_bwd[ul]); d.Kd = 4 * ch[l];
                d.T = 4; d.TX = 2; d.stride = 2;
                d.NB = B; d.OH = hin; d.OW = hin; d.M = B * hin * hin; d.Nn = ch[l + 1];
                d.dst = dzin; d.DH = hin; d.DW = hin; d.DC = ch[l + 1];
                d.mask = uin;
                d.math = pl.math;
                if ((rc = launch_igemm(d, st))) return rc;
            }
        }
        return 0;
    }

    if (stage == 4) {
        if ((rc = conv_backward(pl, workspace, st, params, grads, C52C, WS(pl.a1[4]), pl.ea1[4], ch[4], WS(pl.g_a2[4]), pl.ea2[4], ch[4],
                                WS(pl.g_a1[4]), WS(pl.a1[4]), nullptr))) return rc;
        if ((rc = conv_backward(pl, workspace, st, params, grads, C51C, WS(pl.t[3]), pl.ein[4], ch[3], WS(pl.g_a1[4]), pl.ea1[4], ch[4],
                                WS(pl.g_t[3]), nullptr, WS(pl.g_ts[3])))) return rc;
        return pool_backward(pl, workspace, 3, stream);
    }

    // stage 5: encoder levels 3..0
    for (int l = 3; l >= 0; --l) {
        if ((rc = conv_backward(pl, workspace, st, params, grads, 2 * l + 1, WS(pl.a1[l]), pl.ea1[l], ch[l], WS(pl.g_a2[l]), pl.ea2[l], ch[l],
                                WS(pl.g_a1[l]), WS(pl.a1[l]), nullptr))) return rc;
        if (l == 0) {
            // conv11c: weight/bias gradient only (A1 needs no dgrad)
            RowScope rs(C11C, "wgrad");
            return conv1ch_bwd(WS(pl.xin), B, pl.S, ch[0], WS(pl.g_a1[0]), GRAD(0), GRAD(1), WS(pl.small), t_es, st);
        }
        if ((rc = conv_backward(pl, workspace, st, params, grads, 2 * l, WS(pl.t[l - 1]), pl.ein[l], ch[l - 1], WS(pl.g_a1[l]), pl.ea1[l], ch[l],
                                WS(pl.g_t[l - 1]), nullptr, WS(pl.g_ts[l - 1])))) return rc;
        if ((rc = pool_backward(pl, workspace, l - 1, stream))) return rc;
    }
    return 0;
}

int unet_backward_input(unet_handle *h, const void *const *params, void *dx, void *workspace, size_t workspace_bytes, void *stream)
{
    ARG_CHECK(h && params && dx && workspace, "unet_backward_input: null argument");
    CHECK_DEVICE(h, "unet_backward_input");
    Plan pl;
    if (!h->lookup(workspace, pl)) {
        set_error("unet_backward_input: no training forward has been run on this workspace");
        return UNET_E_NOTREADY;
    }
    ARG_CHECK(workspace_bytes >= pl.total, "unet_backward_input: workspace too small");
    MathScope ms(pl.math);
    RowScope rs(C11C, "dgrad");
    // g_a1[0] = d loss / d conv11c's output, masked by its ReLU: final once the last backward stage has been enqueued
    return conv1ch_dgrad(WS(pl.g_a1[0]), pl.B, pl.S, pl.ch[0], PARAM(0), (float *)dx, t_es, (hipStream_t)stream);
}

int unet_backward(unet_handle *h, const void *const *params, const void *dlogits, void *const *grads,
                  void *workspace, size_t workspace_bytes, void *stream)
{
    for (int s = 0; s < N_STAGES; ++s) {
        int rc = unet_backward_stage(h, s, params, dlogits, grads, workspace, workspace_bytes, stream);
        if (rc) return rc;
    }
    return 0;
}

// Upper bound of the split-K slab for the per-op entry point (the split of C into two sources and the
// skip window are not known when the caller sizes its scratch): twice the larger full-window need.
static size_t conv_bwd_slab_bound(int B, int H, int C, int K)
{
    size_t need = 0;
    for (int math = 0; math <= 2; math += 2) {          // the decomposition differs between the fp32 and the bf16 kernels
        MathScope ms(math);
        for (int ci = 64; ci <= (C + 63) / 64 * 64; ci *= 2) {
            WgradP w = conv_wgrad_desc(nullptr, H, ci, 0, nullptr, H - 2, K, B, nullptr, ci, 0, nullptr, 0);
            const size_t n = wgrad_slab_need(w);
            if (n > need) need = n;
        }
    }
    return align_up(2 * need, 256);
}

static size_t upconv_slab_bound(int B, int H, int Ci, int Co)
{
    size_t need = 0;
    for (int math = 0; math <= 2; math += 2) {
        MathScope ms(math);
        WgradP w = upconv_wgrad_desc(nullptr, H, Ci, nullptr, Co, B, nullptr, nullptr, 0);
        const size_t n = wgrad_slab_need(w);
        if (n > need) need = n;
    }
    return align_up(need, 256);
}

// ---- per-op entry points (unit tests) --------------------------------------------------------------
static size_t wino_scratch_bytes(int C, int K)
{
    // U of a per-op call, whole 64-row blocks (wino_u_floats): forward 16 C rup64(K) (the sources' sub-matrices add up to it);
    // dgrad one sub-matrix per source, 16 K (rup64(C1) + rup64(C2)) <= 16 K (rup64(C) + 64)
    const size_t fwd = wino_u_floats(C, K), bwd = wino_u_floats(K, C) + wino_u_floats(K, 64);
    return align_up((fwd > bwd ? fwd : bwd) * sizeof(float), 256);
}
size_t unet_conv3x3_scratch_bytes(int C, int K) { return align_up((size_t)K * C * 9 * sizeof(float), 256) + wino_scratch_bytes(C, K); }

int unet_conv3x3_fwd(const void *x1, int H1, int W1, int C1, int pad1, const void *x2, int C2, int B, int H, int W,
                     const void *w_oihw, const void *bias, int K, int relu, void *y, void *scratch, void *stream)
{
    ARG_CHECK(x1 && w_oihw && y && scratch, "conv3x3_fwd: null argument");
    MathScope ms(get_math_mode());
    ProfScope ps("op.conv3x3_fwd");
    ARG_CHECK(x2 || (H1 + 2 * pad1 == H && W1 + 2 * pad1 == W), "conv3x3_fwd: single source must match the input extent");
    hipStream_t st = (hipStream_t)stream;
    ARG_CHECK(H == W && H1 == W1, "conv3x3_fwd: square tiles only");
    WLayer L = wl_fwd((const float *)w_oihw, K, C1, x2 ? C2 : 0, (float *)scratch);
    float *wu = (float *)((char *)scratch + align_up((size_t)K * (C1 + (x2 ? C2 : 0)) * 9 * sizeof(float), 256));
    return conv_fwd_launch((const float *)x1, H1, C1, pad1, (const float *)x2, x2 ? C2 : 0, B, H, L,
                           (const float *)bias, K, relu, (float *)y, st, wu);
}

size_t unet_conv3x3_bwd_scratch_bytes(int B, int H, int W, int C, int K)
{
    return align_up((size_t)K * C * 9 * sizeof(float), 256) + conv_bwd_slab_bound(B, H, C, K) +
           align_up(bias_grad_scratch_bytes((size_t)B * (H - 2) * (W - 2), K), 256) + wino_scratch_bytes(C, K);
}

int unet_conv3x3_bwd(const void *x1, int H1, int W1, int C1, int pad1, const void *x2, int C2, int B, int H, int W,
                     const void *w_oihw, int K, const void *dz, void *dx1, const void *mask1, const void *add1,
                     void *dx2, const void *mask2, void *dw, void *db, void *scratch, void *stream)
{
    ARG_CHECK(x1 && w_oihw && dz && scratch, "conv3x3_bwd: null argument");
    MathScope ms(get_math_mode());
    ProfScope ps("op.conv3x3_bwd");
    ARG_CHECK(H == W && H1 == W1, "conv3x3_bwd: square tiles only");
    hipStream_t st = (hipStream_t)stream;
    const int C = C1 + (x2 ? C2 : 0);
    const int Ho = H - 2;
    float *wt = (float *)scratch;
    const size_t wt_bytes = align_up((size_t)K * C * 9 * sizeof(float), 256);
    const size_t slab_bytes = conv_bwd_slab_bound(B, H, C, K);
    float *slab = (float *)((char *)scratch + wt_bytes);
    float *small = (float *)((char *)scratch + wt_bytes + slab_bytes);
    float *wu = (float *)((char *)scratch + wt_bytes + slab_bytes + align_up(bias_grad_scratch_bytes((size_t)B * Ho * Ho, K), 256));
    int rc;
    if (dx1 || dx2) {
        WLayer L = wl_dgrad((const float *)w_oihw, K, C, wt);
        if (dx1) {
            IgemmP d = conv_dgrad_desc((const float *)dz, Ho, Ho, K, B, H1, pad1, wt, C1, (float *)dx1, (const float *)mask1, (const float *)add1);
            if ((rc = with_wino(d, wu, L, 0, 0, st))) return rc;
            if ((rc = launch_igemm(d, st))) return rc;
        }
        if (dx2 && x2) {
            IgemmP d = conv_dgrad_desc((const float *)dz, Ho, Ho, K, B, H, 0, adv(wt, (size_t)C1 * 9 * K), C2, (float *)dx2, (const float *)mask2, nullptr);
            if ((rc = with_wino(d, wu + wino_u_floats(K, C1), L, C1, 0, st))) return rc;
            if ((rc = launch_igemm(d, st))) return rc;
        }
    }
    bool db_done = false;
    if (dw) {
        // the bias gradient rides on whichever weight-gradient launch covers the full dz window
        const bool full1 = !x2 && pad1 == 0;
        WgradP w1 = conv_wgrad_desc((const float *)x1, H1, C1, pad1, (const float *)dz, Ho, K, B, (float *)dw, C, 0, slab, slab_bytes,
                                    full1 ? (float *)db : nullptr);
        if ((rc = launch_wgrad(w1, st))) return rc;
        db_done = full1 && db;
        if (x2) {
            WgradP w2 = conv_wgrad_desc((const float *)x2, H, C2, 0, (const float *)dz, Ho, K, B, (float *)dw, C, C1, slab, slab_bytes, (float *)db);
            if ((rc = launch_wgrad(w2, st))) return rc;
            db_done = db != nullptr;
        }
    }
    if (db && !db_done && (rc = bias_grad(dz, (size_t)B * Ho * Ho, K, (float *)db, small, t_es, st))) return rc;
    return 0;
}

size_t unet_upconv2_scratch_bytes(int B, int H, int W, int Ci, int Co)
{
    (void)W;
    return align_up((size_t)Ci * Co * 4 * sizeof(float), 256) + upconv_slab_bound(B, H, Ci, Co) +
           align_up(bias_grad_scratch_bytes((size_t)B * 4 * H * W, Co), 256);
}

int unet_upconv2_fwd(const void *x, int B, int H, int W, int Ci, const void *w_iohw, const void *bias, int Co,
                     void *y, void *scratch, void *stream)
{
    ARG_CHECK(x && w_iohw && y && scratch, "upconv2_fwd: null argument");
    MathScope ms(get_math_mode());
    ProfScope ps("op.upconv2_fwd");
    hipStream_t st = (hipStream_t)stream;
    int rc = pack_upconv_fwd((const float *)w_iohw, scratch, Ci, Co, t_es, st);
    if (rc) return rc;
    IgemmP u{};
    u.nsrc = 1; u.src[0] = GSrc{(const float *)x, H, W, Ci, 0, Ci, 0};
    u.wt = (const float *)scratch; u.Kd = Ci;
    u.T = 1; u.TX = 1; u.stride = 1;
    u.NB = B; u.OH = H; u.OW = W; u.M = B * H * W; u.Nn = 4 * Co;
    u.dst = (float *)y; u.DH = 2 * H; u.DW = 2 * W; u.DC = Co; u.scatter = 1; u.cout = Co;
    u.bias = (const float *)bias;
    u.math = t_math;
    return launch_igemm(u, st);
}

int unet_upconv2_bwd(const void *x, int B, int H, int W, int Ci, const void *w_iohw, int Co, const void *dy,
                     void *dx, const void *mask, void *dw, void *db, void *scratch, void *stream)
{
    ARG_CHECK(x && w_iohw && dy && scratch, "upconv2_bwd: null argument");
    ARG_CHECK(H == W, "upconv2_bwd: square tiles only");
    MathScope ms(get_math_mode());
    ProfScope ps("op.upconv2_bwd");
    hipStream_t st = (hipStream_t)stream;
    float *wt = (float *)scratch;
    const size_t wt_bytes = align_up((size_t)Ci * Co * 4 * sizeof(float), 256);
    const size_t slab_bytes = upconv_slab_bound(B, H, Ci, Co);
    float *slab = (float *)((char *)scratch + wt_bytes);
    float *small = (float *)((char *)scratch + wt_bytes + slab_bytes);
    int rc;
    if (dx) {
        if ((rc = pack_upconv_dgrad((const float *)w_iohw, wt, Ci, Co, t_es, st))) return rc;
        IgemmP d{};
        d.nsrc = 1; d.src[0] = GSrc{(const float *)dy, 2 * H, 2 * W, Co, 0, Co, 0};
        d.wt = wt; d.Kd = 4 * Co;
        d.T = 4; d.TX = 2; d.stride = 2;
        d.NB = B; d.OH = H; d.OW = W; d.M = B * H * W; d.Nn = Ci;
        d.dst = (float *)dx; d.DH = H; d.DW = W; d.DC = Ci;
        d.mask = (const float *)mask;
        d.math = t_math;
        if ((rc = launch_igemm(d, st))) return rc;
    }
    if (dw) {
        WgradP w = upconv_wgrad_desc((const float *)x, H, Ci, (const float *)dy, Co, B, (float *)dw, slab, slab_bytes, (float *)db);
        if ((rc = launch_wgrad(w, st))) return rc;
    } else if (db && (rc = bias_grad(dy, (size_t)B * 4 * H * W, Co, (float *)db, small, t_es, st))) return rc;
    return 0;
}

}  // extern "C"
