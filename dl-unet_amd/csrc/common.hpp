// common.hpp — shared host/device declarations for libunet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace unet {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void set_error(const char *fmt, ...);
const float *zero_page();          // 4 KiB of device zeros on the current device (lazy, one-time)

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) {                                                    \
            unet::set_error("%s failed: %s", #expr, hipGetErrorString(e_));        \
            return (int)e_;                                                        \
        }                                                                          \
    } while (0)

#define ARG_CHECK(cond, ...)                                                       \
    do {                                                                           \
        if (!(cond)) { unet::set_error(__VA_ARGS__); return -2; }                  \
    } while (0)

// per-launch event timing (prof.hip).  kind: which kernel family; prof_scope names the SURVEY 8a row the following
// launches of this thread belong to ("conv12c.fwd", "pool1.bwd", ...; nullptr clears).
enum { PK_IGEMM = 0, PK_WGRAD = 1, PK_REDUCE = 2, PK_WINO = 3, PK_STENCIL = 4, PK_ELEMWISE = 5, PK_COMM = 6, PK_KINDS = 7 };
void prof_scope(const char *row);
void prof_begin(int kind, const char *tag, hipStream_t st, double alg_flops, double exec_flops, double alg_bytes);
void prof_end(hipStream_t st);
bool prof_active();          // per-launch events are being recorded (concurrent streams would blur them: the overlap knob stays off)
struct ProfScope {          // RAII: names the row for the launches of a block
    explicit ProfScope(const char *row) { prof_scope(row); }
    ~ProfScope() { prof_scope(nullptr); }
};

// raise a kernel's dynamic-LDS limit once per (kernel, device)
static inline int ensure_dynamic_lds(const void *kern, int bytes, bool (&done)[64])
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("bad device"); return -2; }
    if (!done[dev]) {
        HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        done[dev] = true;
    }
    return 0;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// Implicit-GEMM descriptor (igemm.hip):  D[m][n] = epi( sum_{src,tap,c} A[m][src,tap,c] * Wt[n][kd] )
//   m <-> (img, oy, ox) over the output domain [NB, OH, OW]
//   A row for (src, tap=(ty,tx)) = src.p[img][(oy+oy0)*stride + ty - pad][(ox+ox0)*stride + tx - pad][c0 .. c0+nch)
//   (zero outside the tensor: virtual zero padding == the reference's crop_and_concat pad, and the
//    "full" padding of the dgrad correlation)
//   kd order: source-major, then tap, then channel — the weight packers produce exactly this.
// ---------------------------------------------------------------------------------------------
// exact unsigned division by a launch-time constant: q = (n * mul) >> shift for n < 2^31 (64-bit product)
struct FastDiv { unsigned long long mul; int shift; };
static inline FastDiv make_fastdiv(unsigned d)
{
    int s = 0;
    while ((1ull << s) < d) ++s;
    FastDiv f;
    f.shift = 32 + s;
    f.mul = ((1ull << f.shift) + d - 1) / d;
    return f;
}

struct GSrc {
    const float *p;
    int H, W, C;      // tensor extent per image and channel pitch
    int c0, nch;      // channel window contributing to K (nch % 32 == 0)
    int pad;          // virtual zero padding per side
};

struct IgemmP {
    GSrc src[2];
    int nsrc;
    const float *wt;  // packed weights [Nn][ldw], this launch contracts over columns [0, Kd)
    int Kd, ldw;
    int T, TX;        // taps, taps per row
    int stride, oy0, ox0;
    int NB, OH, OW, M, Nn;
    float *dst;
    int DH, DW, DC, dn0;
    int scatter;      // 0: dst pixel == m (DH==OH, DW==OW); 1: up-conv scatter n -> (a,b,co);
                      // 2: window: dst pixel = (oy + dwy0, ox + dwx0) of a DH x DW tensor
    int dwy0, dwx0;
    int rw0, rw1;     // if rw1 > rw0: ReLU is applied only OUTSIDE the output window [rw0,rw1)^2 (a later launch finishes it)
    int cout;         // scatter: channels per (a,b) group; also bias index = n % cout
    const float *bias;
    int relu;
    const float *mask;   // same geometry as dst: v = mask>0 ? v : 0  (ReLU backward)
    const float *add;    // same geometry as dst: v += add
    const float *zeros;
    int buf_bytes[3];      // set by launch_igemm: buffer-descriptor sizes of src[0], src[1] and wt (bytes)
    float *pool_dst;       // optional: 2x2 max-pool of the output [NB, OH/2, OW/2, Nn], written by the Winograd epilogue (wino_fuses_pool)
    const float *wino_u;   // math mode 3: Winograd-transformed filters of this launch (wino.hip), else null
    int math;              // arithmetic of this launch (unet_set_math codes); the plan fixes it at forward time
    int mtiles, ntiles;
    FastDiv d_ohw, d_ow;   // set by launch_igemm: division by OH*OW and by OW (pixel index -> image, row, column)
};
int launch_igemm(IgemmP p, hipStream_t st);
// Winograd F(2x2,3x3) path (wino.hip): filter transform into U (wino_u_floats(channels, Nn) floats) and applicability
bool wino_applicable(const IgemmP &p);
bool wino_fuses_pool(const IgemmP &p);
size_t wino_u_floats(int Kc, int Nn);
int wino_transform_ref(const float *w_oihw, int I, int dgrad, int n0, int Nn, int k0, int Kc, float *U, hipStream_t st);
int get_math_mode();          // process default (UNET_MATH / unet_set_math); plans and per-op calls copy it into their descriptors
void set_math_mode(int m);
int get_lds_dma_mode();       // 1 (default): buffer-descriptor LDS-DMA when every tensor of a launch is below 2 GiB; 0: always global_load_lds
void set_lds_dma_mode(int m);
double igemm_alg_flops(const IgemmP &p);
double igemm_alg_bytes(const IgemmP &p);

// ---------------------------------------------------------------------------------------------
// Weight-gradient descriptor (wgrad.hip):  D[t][i][j] = sum_{img,y,x} X[img][(y+oy0)*s+ty-xpad][(x+ox0)*s+tx-xpad][xc0+i]
//                                                                   * Y[img][y][x][yc0+j]
// computed as split-K partial slabs + a deterministic reduce into out[base + i*si + j*sj + t*st].
// ---------------------------------------------------------------------------------------------
struct WgradP {
    const float *X; int XH, XW, XC, xc0, xpad;
    const float *Y; int YH, YW, YC, yc0;
    int NB, stride, TY, TX, oy0, ox0;
    int ywin0, ywin1, xwin0, xwin1;    // Y-domain window whose taps can touch X
    int Ci, Cj;                        // multiples of 64
    float *out; long si, sj, st;       // final gradient tensor strides (elements)
    float *db;                         // optional fused bias gradient db[yc0 + j] = sum Y (needs the full window)
    int db_on_x;                       // 1: db[xc0 + i] = sum X instead (up-conv: X is dOut, stride == taps so every X pixel is staged once)
    float *slab; size_t slab_bytes;    // scratch for partials
    const float *zeros;
    int math;                          // arithmetic of this launch (unet_set_math codes)
};
size_t wgrad_slab_need(const WgradP &p);   // slab bytes launch_wgrad will use for this descriptor (p.math decides the kernel)
double wgrad_alg_flops(const WgradP &p);
double wgrad_alg_bytes(const WgradP &p);
int launch_wgrad(WgradP p, hipStream_t st);

// Packers from the reference's parameter layouts into the igemm weight matrices (direct.hip); es = element size of the
// packed matrix: 4 (fp32) or 2 (bf16, arithmetic mode 2)
int pack_conv_fwd(const float *w_oihw, void *wt, int K, int C1, int C2, int es, hipStream_t st);   // [K][9*C1 | 9*C2]
int pack_conv_dgrad(const float *w_oihw, void *wt, int K, int C, int es, hipStream_t st);          // [C][9*K], taps flipped
int pack_upconv_fwd(const float *w_iohw, void *wt, int Ci, int Co, int es, hipStream_t st);        // [4*Co][Ci]
int pack_upconv_dgrad(const float *w_iohw, void *wt, int Ci, int Co, int es, hipStream_t st);      // [Ci][4*Co]
int bias_grad(const void *dz, size_t M, int K, float *db, float *scratch, int es, hipStream_t st); // db[k] = sum_m dz[m][k]
size_t bias_grad_scratch_bytes(size_t M, int K);

// HBM-bound layers (direct.hip); es = element size of the activation tensors (4: fp32, 2: bf16)
int conv1ch_fwd(const float *x, int B, int S, const float *w, const float *bias, int K, void *y, int es, hipStream_t st);
int conv1ch_bwd(const float *x, int B, int S, int K, const void *dz, float *dw, float *db, float *scratch, int es, hipStream_t st);
int conv1ch_dgrad(const void *dz, int B, int S, int K, const float *w, float *dx, int es, hipStream_t st);   // d loss / d image
int head1x1_fwd(const void *x, int B, int H, int W, int C, const float *w, const float *bias, float *logits, int es, hipStream_t st);
int head1x1_bwd(const void *x, int B, int H, int W, int C, const float *w, const float *dlogits, float dl_scale, void *dz, float *dw,
                float *db, float *scratch, int es, hipStream_t st);   // every use of dlogits is dlogits * dl_scale (data parallel: 1/world)
int maxpool2_fwd(const void *x, void *y, int B, int H, int W, int C, int es, hipStream_t st);
int maxpool2_bwd(const void *pre, const void *dy, void *dpre, int B, int H, int W, int C, int es, hipStream_t st);

}  // namespace unet
