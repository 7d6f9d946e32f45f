/*
 * TEST INFRASTRUCTURE — CPU restatement of the reference's U-Net hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.  It is the checker, never the product.
 *
 * This body is included twice by unet_oracle.c, once with REAL=float (suffix
 * _f32) and once with REAL=double (suffix _f64).  Layout is the reference's:
 * NCHW activations, OIHW conv weights, IOHW transposed-conv weights.
 *
 * The arithmetic of the path lives in PyTorch ATen (network.py:131-190 are the
 * call sites; the reference pins no torch version).  The published semantics of
 * those ops are restated here; the restatement is pinned against the imported
 * reference by tests/golden/ (see tests/golden/make_golden.py).
 */

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

/* nn.Conv2d(k=R, stride 1, no padding) [+ F.relu]  — network.py:23-36,131-156.
 * y[n,k,oy,ox] = b[k] + sum_{c,r,s} x[n,c,oy+r,ox+s] * w[k,c,r,s]              */
void FN(oracle_conv_valid_fwd)(const REAL *x, const REAL *w, const REAL *b, REAL *y,
                               int N, int C, int H, int W, int K, int R, int relu)
{
    const int Ho = H - R + 1, Wo = W - R + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            REAL *yp = y + ((size_t)n * K + k) * Ho * Wo;
            const REAL bk = b ? b[k] : (REAL)0;
            for (int i = 0; i < Ho * Wo; ++i) yp[i] = bk;
            for (int c = 0; c < C; ++c) {
                const REAL *xp = x + ((size_t)n * C + c) * H * W;
                const REAL *wp = w + ((size_t)k * C + c) * R * R;
                for (int r = 0; r < R; ++r)
                    for (int s = 0; s < R; ++s) {
                        const REAL wv = wp[r * R + s];
                        for (int oy = 0; oy < Ho; ++oy) {
                            const REAL *xr = xp + (size_t)(oy + r) * W + s;
                            REAL *yr = yp + (size_t)oy * Wo;
                            for (int ox = 0; ox < Wo; ++ox) yr[ox] += wv * xr[ox];
                        }
                    }
            }
            if (relu)
                for (int i = 0; i < Ho * Wo; ++i) yp[i] = yp[i] > 0 ? yp[i] : (REAL)0;
        }
}

/* autograd backward of the conv above (A23): dx = full correlation of dy with the
 * flipped filter, dw = sum_{n,oy,ox} x*dy, db = sum dy.  dx may be NULL (conv11c). */
void FN(oracle_conv_valid_bwd)(const REAL *x, const REAL *w, const REAL *dy,
                               REAL *dx, REAL *dw, REAL *db,
                               int N, int C, int H, int W, int K, int R)
{
    const int Ho = H - R + 1, Wo = W - R + 1;
    if (dx) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int c = 0; c < C; ++c) {
                REAL *dxp = dx + ((size_t)n * C + c) * H * W;
                for (int i = 0; i < H * W; ++i) dxp[i] = 0;
                for (int k = 0; k < K; ++k) {
                    const REAL *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                    const REAL *wp = w + ((size_t)k * C + c) * R * R;
                    for (int r = 0; r < R; ++r)
                        for (int s = 0; s < R; ++s) {
                            const REAL wv = wp[r * R + s];
                            for (int oy = 0; oy < Ho; ++oy) {
                                REAL *dxr = dxp + (size_t)(oy + r) * W + s;
                                const REAL *dyr = dyp + (size_t)oy * Wo;
                                for (int ox = 0; ox < Wo; ++ox) dxr[ox] += wv * dyr[ox];
                            }
                        }
                }
            }
    }
    if (dw) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < C; ++c)
                for (int r = 0; r < R; ++r)
                    for (int s = 0; s < R; ++s) {
                        REAL acc = 0;
                        for (int n = 0; n < N; ++n) {
                            const REAL *xp = x + ((size_t)n * C + c) * H * W;
                            const REAL *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                            for (int oy = 0; oy < Ho; ++oy) {
                                const REAL *xr = xp + (size_t)(oy + r) * W + s;
                                const REAL *dyr = dyp + (size_t)oy * Wo;
                                REAL a = 0;
                                for (int ox = 0; ox < Wo; ++ox) a += xr[ox] * dyr[ox];
                                acc += a;
                            }
                        }
                        dw[(((size_t)k * C + c) * R + r) * R + s] = acc;
                    }
    }
    if (db) {
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; ++k) {
            REAL acc = 0;
            for (int n = 0; n < N; ++n) {
                const REAL *dyp = dy + ((size_t)n * K + k) * Ho * Wo;
                REAL a = 0;
                for (int i = 0; i < Ho * Wo; ++i) a += dyp[i];
                acc += a;
            }
            db[k] = acc;
        }
    }
}

/* F.relu backward: dz = dy * (y > 0), y is the ReLU OUTPUT.  In place on dy. */
void FN(oracle_relu_bwd)(const REAL *y, REAL *dy, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) dy[i] = y[i] > 0 ? dy[i] : (REAL)0;
}

/* F.max_pool2d(k=2,s=2) — network.py:133,139,145,151.  Window scan is row-major,
 * a later element replaces the max only if strictly greater (first max wins).
 * idx holds the winning position 0..3 for the backward. H, W even. */
void FN(oracle_maxpool2_fwd)(const REAL *x, REAL *y, unsigned char *idx,
                             int N, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc) {
        const REAL *xp = x + (size_t)nc * H * W;
        REAL *yp = y + (size_t)nc * Ho * Wo;
        unsigned char *ip = idx ? idx + (size_t)nc * Ho * Wo : 0;
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                REAL m = xp[(size_t)(2 * oy) * W + 2 * ox];
                int mi = 0;
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        const REAL v = xp[(size_t)(2 * oy + a) * W + 2 * ox + b];
                        if (v > m) { m = v; mi = a * 2 + b; }
                    }
                yp[(size_t)oy * Wo + ox] = m;
                if (ip) ip[(size_t)oy * Wo + ox] = (unsigned char)mi;
            }
    }
}

void FN(oracle_maxpool2_bwd)(const REAL *dy, const unsigned char *idx, REAL *dx,
                             int N, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc) {
        const REAL *dyp = dy + (size_t)nc * Ho * Wo;
        const unsigned char *ip = idx + (size_t)nc * Ho * Wo;
        REAL *dxp = dx + (size_t)nc * H * W;
        for (int i = 0; i < H * W; ++i) dxp[i] = 0;
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const int mi = ip[(size_t)oy * Wo + ox];
                dxp[(size_t)(2 * oy + (mi >> 1)) * W + 2 * ox + (mi & 1)] =
                    dyp[(size_t)oy * Wo + ox];
            }
    }
}

/* nn.ConvTranspose2d(k=2, s=2) — network.py:38,43,48,53; calls :159,167,175,183.
 * y[n,co,2i+a,2j+b] = b[co] + sum_ci x[n,ci,i,j] * w[ci,co,a,b]; no activation. */
void FN(oracle_upconv2_fwd)(const REAL *x, const REAL *w, const REAL *b, REAL *y,
                            int N, int Ci, int H, int W, int Co)
{
    const int Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Co; ++co) {
            REAL *yp = y + ((size_t)n * Co + co) * Ho * Wo;
            const REAL bv = b ? b[co] : (REAL)0;
            for (int i = 0; i < Ho * Wo; ++i) yp[i] = bv;
            for (int ci = 0; ci < Ci; ++ci) {
                const REAL *xp = x + ((size_t)n * Ci + ci) * H * W;
                const REAL *wp = w + ((size_t)ci * Co + co) * 4;
                for (int i = 0; i < H; ++i)
                    for (int a = 0; a < 2; ++a) {
                        REAL *yr = yp + (size_t)(2 * i + a) * Wo;
                        const REAL *xr = xp + (size_t)i * W;
                        const REAL w0 = wp[a * 2 + 0], w1 = wp[a * 2 + 1];
                        for (int j = 0; j < W; ++j) {
                            yr[2 * j] += xr[j] * w0;
                            yr[2 * j + 1] += xr[j] * w1;
                        }
                    }
            }
        }
}

void FN(oracle_upconv2_bwd)(const REAL *x, const REAL *w, const REAL *dy,
                            REAL *dx, REAL *dw, REAL *db,
                            int N, int Ci, int H, int W, int Co)
{
    const int Ho = 2 * H, Wo = 2 * W;
    if (dx) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int ci = 0; ci < Ci; ++ci) {
                REAL *dxp = dx + ((size_t)n * Ci + ci) * H * W;
                for (int i = 0; i < H * W; ++i) dxp[i] = 0;
                for (int co = 0; co < Co; ++co) {
                    const REAL *dyp = dy + ((size_t)n * Co + co) * Ho * Wo;
                    const REAL *wp = w + ((size_t)ci * Co + co) * 4;
                    for (int i = 0; i < H; ++i)
                        for (int a = 0; a < 2; ++a) {
                            const REAL *dyr = dyp + (size_t)(2 * i + a) * Wo;
                            REAL *dxr = dxp + (size_t)i * W;
                            const REAL w0 = wp[a * 2 + 0], w1 = wp[a * 2 + 1];
                            for (int j = 0; j < W; ++j)
                                dxr[j] += dyr[2 * j] * w0 + dyr[2 * j + 1] * w1;
                        }
                }
            }
    }
    if (dw) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int ci = 0; ci < Ci; ++ci)
            for (int co = 0; co < Co; ++co)
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        REAL acc = 0;
                        for (int n = 0; n < N; ++n) {
                            const REAL *xp = x + ((size_t)n * Ci + ci) * H * W;
                            const REAL *dyp = dy + ((size_t)n * Co + co) * Ho * Wo;
                            for (int i = 0; i < H; ++i) {
                                REAL s = 0;
                                for (int j = 0; j < W; ++j)
                                    s += xp[(size_t)i * W + j] *
                                         dyp[(size_t)(2 * i + a) * Wo + 2 * j + b];
                                acc += s;
                            }
                        }
                        dw[((size_t)ci * Co + co) * 4 + a * 2 + b] = acc;
                    }
    }
    if (db) {
#pragma omp parallel for schedule(static)
        for (int co = 0; co < Co; ++co) {
            REAL acc = 0;
            for (int n = 0; n < N; ++n) {
                const REAL *dyp = dy + ((size_t)n * Co + co) * Ho * Wo;
                REAL s = 0;
                for (int i = 0; i < Ho * Wo; ++i) s += dyp[i];
                acc += s;
            }
            db[co] = acc;
        }
    }
}

/* Unet.crop_and_concat — network.py:108-127.  c = int((Ha-Hb)*0.5) truncates toward
 * zero; F.pad(A,(-c,-c,-c,-c)) crops for c>0 and ZERO-PADS for c<0 (the case every
 * valid full-net input takes); then cat((A', B), dim=1).  Returns -1 when
 * the padded/cropped A does not match B (the reference's torch.cat raises, Q7).     */
int FN(oracle_crop_and_concat_fwd)(const REAL *A, const REAL *B, REAL *out,
                                   int N, int Ca, int Ha, int Cb, int Hb)
{
    const int c = (int)((Ha - Hb) * 0.5);
    if (Ha - 2 * c != Hb) return -1;
    const int Ct = Ca + Cb;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ch = 0; ch < Ct; ++ch) {
            REAL *op = out + ((size_t)n * Ct + ch) * Hb * Hb;
            if (ch >= Ca) {
                const REAL *bp = B + ((size_t)n * Cb + (ch - Ca)) * Hb * Hb;
                for (int i = 0; i < Hb * Hb; ++i) op[i] = bp[i];
            } else {
                const REAL *ap = A + ((size_t)n * Ca + ch) * Ha * Ha;
                for (int y = 0; y < Hb; ++y)
                    for (int x = 0; x < Hb; ++x) {
                        const int ya = y + c, xa = x + c;
                        op[(size_t)y * Hb + x] =
                            (ya >= 0 && ya < Ha && xa >= 0 && xa < Ha)
                                ? ap[(size_t)ya * Ha + xa] : (REAL)0;
                    }
            }
        }
    return 0;
}

int FN(oracle_crop_and_concat_bwd)(const REAL *dout, REAL *dA, REAL *dB,
                                   int N, int Ca, int Ha, int Cb, int Hb)
{
    const int c = (int)((Ha - Hb) * 0.5);
    if (Ha - 2 * c != Hb) return -1;
    const int Ct = Ca + Cb;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ch = 0; ch < Ct; ++ch) {
            const REAL *op = dout + ((size_t)n * Ct + ch) * Hb * Hb;
            if (ch >= Ca) {
                REAL *bp = dB + ((size_t)n * Cb + (ch - Ca)) * Hb * Hb;
                for (int i = 0; i < Hb * Hb; ++i) bp[i] = op[i];
            } else {
                REAL *ap = dA + ((size_t)n * Ca + ch) * Ha * Ha;
                for (int ya = 0; ya < Ha; ++ya)
                    for (int xa = 0; xa < Ha; ++xa) {
                        const int y = ya - c, x = xa - c;
                        ap[(size_t)ya * Ha + xa] =
                            (y >= 0 && y < Hb && x >= 0 && x < Hb)
                                ? op[(size_t)y * Hb + x] : (REAL)0;
                    }
            }
        }
    return 0;
}

/* nn.BCEWithLogitsLoss(weight=w)(x, z), reduction='mean' — trainer.py:72-75 (L1).
 * l = w*(max(x,0) - x*z + log1p(exp(-|x|))); grad = w*(sigmoid(x)-z)/n.
 * w may be NULL (unweighted) and is already expanded to x's shape otherwise.    */
double FN(oracle_bce_logits)(const REAL *x, const REAL *z, const REAL *w, REAL *dx, size_t n)
{
    double acc = 0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const double xv = x[i], zv = z[i], wv = w ? (double)w[i] : 1.0;
        const double l = (xv > 0 ? xv : 0) - xv * zv + log1p(exp(-fabs(xv)));
        acc += wv * l;
        if (dx) {
            const double sg = 1.0 / (1.0 + exp(-xv));
            dx[i] = (REAL)(wv * (sg - zv) / (double)n);
        }
    }
    return acc / (double)n;
}

/* preds.argmax(dim=1) over 2 classes — trainer.py:82, tester.py:30 (L2).
 * First maximum wins on ties -> class 0.                                        */
void FN(oracle_argmax2)(const REAL *x, long long *out, int N, size_t HW)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (size_t i = 0; i < HW; ++i)
            out[(size_t)n * HW + i] =
                x[((size_t)n * 2 + 1) * HW + i] > x[((size_t)n * 2) * HW + i] ? 1 : 0;
}

/* optim.SGD(lr, momentum) step — trainer.py:30,78 (L3): no dampening, nesterov or
 * weight decay.  First step: buf = g; afterwards buf = mu*buf + g; p -= lr*buf.    */
void FN(oracle_sgd_momentum)(REAL *p, const REAL *g, REAL *buf, size_t n,
                             REAL lr, REAL mu, int first_step)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        const REAL b = first_step ? g[i] : mu * buf[i] + g[i];
        buf[i] = b;
        p[i] -= lr * b;
    }
}

/* "Same branch" evaluation: ReLU masks / pool selections taken from ANOTHER run (the HIP path's
 * forward) instead of this run's own signs.  A ReLU or max-pool is piecewise linear; two fp32/fp64
 * evaluations that take a different piece at a near-zero activation or a near-tie differ by O(1) in
 * that element's gradient, so gradient parity is asserted on the SAME piece.        */
void FN(oracle_apply_mask)(REAL *y, const unsigned char *mask, size_t n)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) y[i] = mask[i] ? y[i] : (REAL)0;
}

void FN(oracle_maxpool2_select)(const REAL *x, REAL *y, const unsigned char *idx, int N, int C, int H, int W)
{
    const int Ho = H / 2, Wo = W / 2;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc)
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                const int mi = idx[((size_t)nc * Ho + oy) * Wo + ox];
                y[((size_t)nc * Ho + oy) * Wo + ox] =
                    x[(size_t)nc * H * W + (size_t)(2 * oy + (mi >> 1)) * W + 2 * ox + (mi & 1)];
            }
}

/* ------------------------------------------------------------------------------
 * Whole net: Unet.forward (network.py:129-192) and its autograd backward (A23).
 * params: the 46 state-dict tensors in declaration order (network.py:23-58),
 * weight then bias per layer.  base = 64 in the reference (hard-coded widths).
 * grads (46 tensors, same shapes) and dlogits may be NULL for forward only.
 * Returns 0, or -1 on a size the reference would raise on.
 * ------------------------------------------------------------------------------ */
#ifndef ORACLE_LAYER_ENUM
#define ORACLE_LAYER_ENUM
enum { L_C11C, L_C12C, L_C21C, L_C22C, L_C31C, L_C32C, L_C41C, L_C42C, L_C51C, L_C52C,
       L_UP4, L_C41E, L_C42E, L_UP3, L_C31E, L_C32E, L_UP2, L_C21E, L_C22E,
       L_UP1, L_C11E, L_C12E, L_FINAL, L_COUNT };
#endif

int FN(oracle_unet_fwd_bwd)(const REAL *const *params, const REAL *x, int N, int S,
                            int base, REAL *logits, const REAL *dlogits,
                            REAL *const *grads,
                            const unsigned char *const *relu_mask,   /* NULL, or 18 NCHW masks: a1[l],a2[l] at 2l,2l+1; d1[l],d2[l] at 10+2l,11+2l */
                            const unsigned char *const *pool_sel)    /* NULL, or 4 NCHW selections (0..3) for pool l */
{
#define MASKED(buf, idx, n) do { if (relu_mask) FN(oracle_apply_mask)((buf), relu_mask[(idx)], (n)); } while (0)
#define RELU_BWD(act, g, idx, n) do { if (relu_mask) FN(oracle_apply_mask)((g), relu_mask[(idx)], (n)); else FN(oracle_relu_bwd)((act), (g), (n)); } while (0)
#define PW(l) params[2 * (l)]
#define PB(l) params[2 * (l) + 1]
#define GW(l) (grads ? grads[2 * (l)] : 0)
#define GB(l) (grads ? grads[2 * (l) + 1] : 0)
#define ALLOC(n) ((REAL *)malloc(sizeof(REAL) * (size_t)(n)))
    const int ch[5] = { base, base * 2, base * 4, base * 8, base * 16 };
    /* encoder */
    REAL *a1[5], *a2[5], *t[4];        /* conv_k1c out, conv_k2c out, pooled */
    unsigned char *pidx[4];
    int e_in[5], e_a1[5], e_a2[5], e_t[4];
    int cur = S, cin = 1;
    const REAL *src = x;
    int rc = 0;
    for (int l = 0; l < 5; ++l) {
        e_in[l] = cur;
        e_a1[l] = cur - 2; e_a2[l] = cur - 4;
        if (e_a2[l] <= 0) return -1;
        a1[l] = ALLOC((size_t)N * ch[l] * e_a1[l] * e_a1[l]);
        a2[l] = ALLOC((size_t)N * ch[l] * e_a2[l] * e_a2[l]);
        FN(oracle_conv_valid_fwd)(src, PW(2 * l), PB(2 * l), a1[l], N, cin, cur, cur, ch[l], 3, !relu_mask);
        MASKED(a1[l], 2 * l, (size_t)N * ch[l] * e_a1[l] * e_a1[l]);
        FN(oracle_conv_valid_fwd)(a1[l], PW(2 * l + 1), PB(2 * l + 1), a2[l], N, ch[l], e_a1[l], e_a1[l], ch[l], 3, !relu_mask);
        MASKED(a2[l], 2 * l + 1, (size_t)N * ch[l] * e_a2[l] * e_a2[l]);
        if (l < 4) {
            if (e_a2[l] & 1) return -1;
            e_t[l] = e_a2[l] / 2;
            t[l] = ALLOC((size_t)N * ch[l] * e_t[l] * e_t[l]);
            pidx[l] = (unsigned char *)malloc((size_t)N * ch[l] * e_t[l] * e_t[l]);
            if (pool_sel) {
                memcpy(pidx[l], pool_sel[l], (size_t)N * ch[l] * e_t[l] * e_t[l]);
                FN(oracle_maxpool2_select)(a2[l], t[l], pidx[l], N, ch[l], e_a2[l], e_a2[l]);
            } else
                FN(oracle_maxpool2_fwd)(a2[l], t[l], pidx[l], N, ch[l], e_a2[l], e_a2[l]);
            src = t[l]; cur = e_t[l]; cin = ch[l];
        }
    }
    /* decoder, level 3 (dec4) .. 0 (dec1) */
    static const int up_l[4]  = { L_UP1, L_UP2, L_UP3, L_UP4 };
    static const int c1e_l[4] = { L_C11E, L_C21E, L_C31E, L_C41E };
    static const int c2e_l[4] = { L_C12E, L_C22E, L_C32E, L_C42E };
    REAL *u[4], *cat[4], *d1[4], *d2[4];
    int e_u[4], e_d1[4], e_d2[4];
    const REAL *dsrc = a2[4];
    int dcur = e_a2[4];
    for (int l = 3; l >= 0; --l) {
        e_u[l] = 2 * dcur;
        u[l] = ALLOC((size_t)N * ch[l] * e_u[l] * e_u[l]);
        FN(oracle_upconv2_fwd)(dsrc, PW(up_l[l]), PB(up_l[l]), u[l], N, ch[l + 1], dcur, dcur, ch[l]);
        cat[l] = ALLOC((size_t)N * 2 * ch[l] * e_u[l] * e_u[l]);
        rc = FN(oracle_crop_and_concat_fwd)(t[l], u[l], cat[l], N, ch[l], e_t[l], ch[l], e_u[l]);
        if (rc) return rc;
        e_d1[l] = e_u[l] - 2; e_d2[l] = e_u[l] - 4;
        d1[l] = ALLOC((size_t)N * ch[l] * e_d1[l] * e_d1[l]);
        d2[l] = ALLOC((size_t)N * ch[l] * e_d2[l] * e_d2[l]);
        FN(oracle_conv_valid_fwd)(cat[l], PW(c1e_l[l]), PB(c1e_l[l]), d1[l], N, 2 * ch[l], e_u[l], e_u[l], ch[l], 3, !relu_mask);
        MASKED(d1[l], 10 + 2 * l, (size_t)N * ch[l] * e_d1[l] * e_d1[l]);
        FN(oracle_conv_valid_fwd)(d1[l], PW(c2e_l[l]), PB(c2e_l[l]), d2[l], N, ch[l], e_d1[l], e_d1[l], ch[l], 3, !relu_mask);
        MASKED(d2[l], 11 + 2 * l, (size_t)N * ch[l] * e_d2[l] * e_d2[l]);
        dsrc = d2[l]; dcur = e_d2[l];
    }
    const int So = dcur;           /* = S - 184 */
    FN(oracle_conv_valid_fwd)(d2[0], PW(L_FINAL), PB(L_FINAL), logits, N, ch[0], So, So, 2, 1, 0);

    if (dlogits && grads) {
        /* head */
        REAL *g = ALLOC((size_t)N * ch[0] * So * So);      /* grad wrt d2[0] (post-ReLU) */
        FN(oracle_conv_valid_bwd)(d2[0], PW(L_FINAL), dlogits, g, GW(L_FINAL), GB(L_FINAL), N, ch[0], So, So, 2, 1);
        for (int l = 0; l < 4; ++l) {
            /* conv_l2e */
            RELU_BWD(d2[l], g, 11 + 2 * l, (size_t)N * ch[l] * e_d2[l] * e_d2[l]);
            REAL *g1 = ALLOC((size_t)N * ch[l] * e_d1[l] * e_d1[l]);
            FN(oracle_conv_valid_bwd)(d1[l], PW(c2e_l[l]), g, g1, GW(c2e_l[l]), GB(c2e_l[l]), N, ch[l], e_d1[l], e_d1[l], ch[l], 3);
            free(g);
            /* conv_l1e */
            RELU_BWD(d1[l], g1, 10 + 2 * l, (size_t)N * ch[l] * e_d1[l] * e_d1[l]);
            REAL *gc = ALLOC((size_t)N * 2 * ch[l] * e_u[l] * e_u[l]);
            FN(oracle_conv_valid_bwd)(cat[l], PW(c1e_l[l]), g1, gc, GW(c1e_l[l]), GB(c1e_l[l]), N, 2 * ch[l], e_u[l], e_u[l], ch[l], 3);
            free(g1);
            /* split: skip grad is kept in cat[l]'s place for the encoder, upconv grad -> gu */
            REAL *gt = ALLOC((size_t)N * ch[l] * e_t[l] * e_t[l]);
            REAL *gu = ALLOC((size_t)N * ch[l] * e_u[l] * e_u[l]);
            FN(oracle_crop_and_concat_bwd)(gc, gt, gu, N, ch[l], e_t[l], ch[l], e_u[l]);
            free(gc);
            free(cat[l]); cat[l] = gt;               /* reuse slot: skip gradient of t[l] */
            /* upconv_l : input is d2[l+1] (or a2[4] at the bottleneck) */
            const REAL *uin = (l == 3) ? a2[4] : d2[l + 1];
            const int uh = e_u[l] / 2;
            g = ALLOC((size_t)N * ch[l + 1] * uh * uh);
            FN(oracle_upconv2_bwd)(uin, PW(up_l[l]), gu, g, GW(up_l[l]), GB(up_l[l]), N, ch[l + 1], uh, uh, ch[l]);
            free(gu);
        }
        /* encoder, level 4 .. 0; g is grad wrt a2[4] (post-ReLU) */
        for (int l = 4; l >= 0; --l) {
            RELU_BWD(a2[l], g, 2 * l + 1, (size_t)N * ch[l] * e_a2[l] * e_a2[l]);
            REAL *g1 = ALLOC((size_t)N * ch[l] * e_a1[l] * e_a1[l]);
            FN(oracle_conv_valid_bwd)(a1[l], PW(2 * l + 1), g, g1, GW(2 * l + 1), GB(2 * l + 1), N, ch[l], e_a1[l], e_a1[l], ch[l], 3);
            free(g);
            RELU_BWD(a1[l], g1, 2 * l, (size_t)N * ch[l] * e_a1[l] * e_a1[l]);
            if (l == 0) {
                FN(oracle_conv_valid_bwd)(x, PW(0), g1, 0, GW(0), GB(0), N, 1, S, S, ch[0], 3);
                free(g1);
                g = 0;
            } else {
                const size_t nt = (size_t)N * ch[l - 1] * e_t[l - 1] * e_t[l - 1];
                REAL *gt = ALLOC(nt);
                FN(oracle_conv_valid_bwd)(t[l - 1], PW(2 * l), g1, gt, GW(2 * l), GB(2 * l), N, ch[l - 1], e_in[l], e_in[l], ch[l], 3);
                free(g1);
                for (size_t i = 0; i < nt; ++i) gt[i] += cat[l - 1][i];   /* + skip gradient */
                g = ALLOC((size_t)N * ch[l - 1] * e_a2[l - 1] * e_a2[l - 1]);
                FN(oracle_maxpool2_bwd)(gt, pidx[l - 1], g, N, ch[l - 1], e_a2[l - 1], e_a2[l - 1]);
                free(gt);
            }
        }
    }
    for (int l = 0; l < 5; ++l) { free(a1[l]); free(a2[l]); }
    for (int l = 0; l < 4; ++l) { free(t[l]); free(pidx[l]); free(u[l]); free(cat[l]); free(d1[l]); free(d2[l]); }
    return 0;
#undef MASKED
#undef RELU_BWD
#undef PW
#undef PB
#undef GW
#undef GB
#undef ALLOC
}

#undef FN
#undef CAT
#undef CAT_
