/*
 * unet_hip.h — C ABI of libunet_hip.so: the MI355X (gfx950) U-Net forward+backward hot path.
 *
 * Drop-in boundary for nsirons/DL-unet's `Unet.forward` (+ its autograd backward) and the
 * step either side of it.  Each entry point names the reference interface it replaces
 * (file:line in the reference repo).  Plain C: pointers are DEVICE pointers unless stated,
 * sizes are ints/size_t, `stream` is a hipStream_t passed as void* (0 = default stream).
 *
 * Conventions
 *   - return 0 = ok; <0 = library error (UNET_E_*); >0 = hipError_t.  unet_last_error()
 *     returns a thread-local message.  No entry point synchronises the host or allocates
 *     device memory, except unet_create()/unet_destroy() (a 4 KiB zero page per handle).
 *   - Public tensors keep the reference's layouts: images/logits NCHW fp32, conv weights
 *     OIHW, transposed-conv weights IOHW, int64 labels/masks.  Internally activations are
 *     NHWC fp32 (per-op entry points take NHWC).
 *   - The caller owns every buffer, including the workspace (size: unet_workspace_bytes).
 *   - All work is enqueued on the caller's stream; a handle is re-entrant per stream: nothing a call needs lives in
 *     process-wide state (the arithmetic mode is fixed per forward and kept with its plan for the backward).
 *   - One process per GPU: a handle's calls must be made while its device is the current HIP device (checked).
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_N_PARAMS 46          /* 23 layers x (weight, bias), network.py:23-58 order */
#define UNET_N_LAYERS 23

enum {
    UNET_E_BADSIZE  = -1,         /* S must be 16L+60 with L even >= 8 (network.py:124-127, Q7) */
    UNET_E_BADARG   = -2,
    UNET_E_NOTREADY = -3,         /* backward without a training forward on this workspace */
    UNET_E_UNSUPPORTED = -4,
    UNET_E_COMM     = -5          /* RCCL error (data parallel) */
};

typedef struct unet_handle unet_handle;

typedef struct unet_config {
    int base_ch;                  /* 64 in the reference (network.py:23); 32 for config #5 */
    int device;                   /* HIP device ordinal */
    int math;                     /* arithmetic of this handle's forwards/backwards (codes below); -1 = follow unet_set_math */
} unet_config;

const char *unet_last_error(void);
int unet_abi_version(void);     /* 4.  (2: unet_config::math, unet_dp_*;  3: unet_bce_step, unet_set_grad_scale, unet_set_overlap;  4: unet_backward_input) */

/* Arithmetic of the dense contractions — the process default, used by the per-op entry points and by handles created
 * with math = -1 (read when a forward is planned; its backward keeps that forward's mode):
 * 3 = fp32 on the fp32 MFMA, the stride-1 3x3 layers' forward and dgrad as Winograd F(2x2,3x3) (default: 16 of the 36
 *     multiplies of the direct correlation, all arithmetic fp32, same parity tolerances as mode 0),
 * 0 = fp32 on the fp32 MFMA, direct correlation everywhere (fmaf-chain numerics),
 * 1 = bf16x3: fp32 operands split into two bf16 terms, three bf16 MFMAs per product, fp32 accumulation
 *     (~16-bit products; logits stay within ~1e-5 of fp32),
 * 2 = bf16 (BASELINE config #3): activations, their gradients and the packed filters are bf16 IN HBM (NHWC, 2 B/element;
 *     the workspace halves), every contraction runs on the bf16 matrix cores with fp32 accumulation; parameters, their
 *     gradients, biases, the input image, logits and dlogits stay fp32 (fp32 master weights).  In this mode the per-op
 *     entry points below take bf16 activation / activation-gradient tensors where they take fp32 in the other modes.
 * Also settable with UNET_MATH.                                                                                       */
int unet_set_math(int mode);
int unet_get_math(void);
/* Operand staging of the MFMA kernels: 1 (default) = LDS-DMA through buffer descriptors whenever every tensor of a launch is
 * below 2 GiB, else global_load_lds; 0 = always global_load_lds (what tensors >= 2 GiB take, e.g. config #5 at batch 16).
 * A tuning / test knob: results are identical.  Also settable with UNET_LDS_DMA.                                       */
int unet_set_lds_dma(int mode);

/* ---- handle ------------------------------------------------------------------------------
 * replaces: Unet.__init__ bookkeeping that is not parameters (network.py:20-58).            */
int unet_create(unet_handle **out, const unet_config *cfg);
int unet_destroy(unet_handle *h);

/* Size contract of the valid-conv net (functions.py:121-146 input_size_compute):
 * returns 0 and writes out_size = S-184 when S = 16L+60, L even >= 8; UNET_E_BADSIZE else. */
int unet_output_size(int S, int *out_size);
int unet_param_count(const unet_handle *h, int idx, size_t *numel);   /* idx in [0,46) */

/* Bytes of caller-provided workspace for a batch of B tiles of S x S.
 * training=1 also reserves gradient/activation-gradient storage for unet_backward.        */
size_t unet_workspace_bytes(const unet_handle *h, int B, int S, int training);

/* ---- whole path ---------------------------------------------------------------------------
 * replaces: Unet.forward (network.py:129-192), called at trainer.py:58,100 and tester.py:27.
 *   params : host array of 46 device pointers, reference state-dict order/layout (fp32)
 *   x      : [B,1,S,S] fp32;  logits : [B,2,S-184,S-184] fp32 NCHW
 *   training=1 keeps the activation stash in `workspace` for unet_backward.               */
int unet_forward(unet_handle *h, const void *const *params, const void *x, void *logits,
                 int B, int S, void *workspace, size_t workspace_bytes, int training,
                 void *stream);

/* replaces: the autograd backward of every op in Unet.forward, triggered by
 * loss.backward() at trainer.py:77.
 *   dlogits : [B,2,So,So] fp32 NCHW (contiguous)
 *   grads   : host array of 46 device pointers, same shapes/layouts as params; OVERWRITTEN
 * Stages let the caller overlap the gradient all-reduce with the rest of the backward:
 * stage s in [0, unet_backward_stages()) must be run in increasing order; after stage s
 * returns, every gradient tensor listed by unet_backward_stage_params(s) is final (in
 * stream order).  unet_backward == all stages.                                              */
int unet_backward(unet_handle *h, const void *const *params, const void *dlogits,
                  void *const *grads, void *workspace, size_t workspace_bytes, void *stream);
/* Weight gradients of each backward stage on an auxiliary stream of the handle, next to the dgrad chain (they only share dz);
 * the streams re-join before unet_backward_stage returns control of `stream`'s order, so callers see no difference (results
 * are bit-identical).  Process-wide: -1 (default; also UNET_OVERLAP) = by what was measured: on with bf16 tensors (+3 % per
 * step) and in fp32 at batches of <= 4 tiles (+0.2 ... +1.2 %), off in fp32 at larger batches (-1 % at B = 8); 0 / 1 force it.  Never active while unet_profile_enable(1) records per-launch events
 * (two streams would interleave them); an event / wait failure fails the stage. */
int unet_set_overlap(int on);
int unet_backward_stages(void);
int unet_backward_stage(unet_handle *h, int stage, const void *const *params,
                        const void *dlogits, void *const *grads, void *workspace,
                        size_t workspace_bytes, void *stream);
/* replaces: the gradient autograd returns for the input image when a caller asks for it (t.requires_grad; the reference's
 * trainer / tester never do: trainer.py:58, tester.py:27 - conv11c's dgrad is the one backward op of network.py:131 they skip).
 *   dx : [B,1,S,S] fp32, OVERWRITTEN.  Call after the LAST backward stage of the same forward has been enqueued on `stream`. */
int unet_backward_input(unet_handle *h, const void *const *params, void *dx, void *workspace, size_t workspace_bytes, void *stream);
/* writes up to cap parameter indices completed by `stage`; returns how many */
int unet_backward_stage_params(int stage, int *idx, int cap);

/* Algorithmic FLOPs (2*MAC) of one forward / forward+backward for B tiles of S. */
double unet_flops(const unet_handle *h, int B, int S, int backward);

/* Element size of the workspace's activation tensors for this handle's arithmetic: 2 (bf16, mode 2) or 4 (fp32). */
int unet_activation_bytes(const unet_handle *h);

/* Debug/introspection: byte offset into the workspace, extent e and channel count of a named NHWC
 * [B,e,e,C] buffer of the plan for (B,S,training): activations "a1_l","a2_l" (l=0..4), "t_l","u_l",
 * "d1_l","d2_l" (l=0..3); with training=1 also their gradients "g_*", "g_ts_l" and "xin". */
int unet_debug_buffer(const unet_handle *h, int B, int S, int training, const char *name,
                      size_t *offset, int *extent, int *channels);

/* ---- data parallel (the reference is single-device, main_main.py:157-158) -------------------------
 * Batch-sharded replicas, one process per GPU; the only exchange is the gradient all-reduce, done by RCCL over xGMI on
 * a communicator stream owned by the handle.  Rendezvous: rank 0 calls unet_dp_unique_id and the HOST carries the
 * UNET_DP_ID_BYTES to every rank (any channel), then every rank calls unet_dp_init.
 *   unet_dp_allreduce : in-place SUM all-reduce of `count` fp32 at `buf`, ordered after everything enqueued on `stream`
 *                       so far, running on the communicator stream (the caller's stream is not blocked: issue it after
 *                       each unet_backward_stage to overlap the bucket with the next stage)
 *   unet_dp_join      : `stream` waits for every collective issued so far (call before the optimizer step)
 *   unet_dp_broadcast : `count` fp32 from `root` to all ranks (initial parameters); same ordering as allreduce
 * Gradients are linear in dlogits, so SUM-reducing the gradients of dlogits / world gives the global-batch mean:
 *   unet_set_grad_scale : the backward of this handle reads dlogits * scale (applied inside the head's backward kernel, no
 *                         extra pass); 1.0 at creation.  A replica of a world-N job sets 1/N once. */
#define UNET_DP_ID_BYTES 128
int unet_set_grad_scale(unet_handle *h, float scale);
int unet_dp_unique_id(void *id_out_host);
int unet_dp_init(unet_handle *h, int rank, int world, const void *id_host);
int unet_dp_destroy(unet_handle *h);
int unet_dp_world(unet_handle *h);                /* ranks of the handle's communicator, 0 if none */
int unet_dp_rccl_version(void);                   /* ncclGetVersion of the bound librccl, 0 if it cannot be loaded */
int unet_dp_allreduce(unet_handle *h, void *buf, size_t count, void *stream);
int unet_dp_broadcast(unet_handle *h, void *buf, size_t count, int root, void *stream);
int unet_dp_join(unet_handle *h, void *stream);

/* ---- measurement -----------------------------------------------------------------------------
 * Optional HIP-event timing around kernel launches, recorded on the launch stream.  A pair of events per launch costs ~1 ms
 * per training step when every launch carries one, so bench.py selects only the dominant kernel kind inside its timed region
 * (unet_profile_select) and builds the per-layer table in a separate, fully instrumented pass.  Every launch records its kernel kind, the SURVEY 8a
 * row it belongs to, its algorithmic FLOPs (2*MAC, in-bounds taps only), the FLOPs the matrix cores execute for it
 * (Winograd: 16/36 of the direct count; tile padding included) and its algorithmic HBM bytes.
 * kind: 0 implicit GEMM (igemm*.hip), 1 weight gradient, 2 its split-K reduce, 3 Winograd 3x3 (wino.hip),
 *       4 conv11c stencil, 5 element-wise / reductions (pool, head, loss, SGD, packers), 6 RCCL collectives.
 * unet_profile_read synchronises on the recorded events and returns totals of one kind since the last reset.       */
int unet_profile_enable(int on);
/* launch kinds that get events while profiling is on: bit k = kind k (default: all) */
int unet_profile_select(unsigned kinds);
int unet_profile_reset(void);
int unet_profile_read(int kind, double *ms_total, long *launches, double *flops_total, double *exec_flops_total,
                      double *bytes_total);
/* one CSV line per recorded launch: kind,ms,gflop,exec_gflop,mbytes,row,tag */
int unet_profile_dump(const char *path);

/* ---- step-side kernels (L1-L3) -------------------------------------------------------------
 * L1 replaces nn.BCEWithLogitsLoss(weight=w)(preds, ll) + its backward (trainer.py:63-77).
 *   logits/target/dlogits : [B,2,H,W] fp32 contiguous; target is the one-hot `ll`
 *   weight : NULL, or fp32 with element strides (wsB,wsC,wsH,wsW) (0 = broadcast dim) —
 *            the caller decides the broadcast (reference: Q4 aligns B with the class axis)
 *   loss_out : 1 fp32 (mean over B*2*H*W); dlogits may be NULL; grad_scale multiplies the
 *            gradient (1/world_size for data parallel).  scratch: >= unet_bce_scratch_bytes */
size_t unet_bce_scratch_bytes(size_t numel);
int unet_bce_logits(const void *logits, const void *target, const void *weight,
                    long wsB, long wsC, long wsH, long wsW, int B, int H, int W,
                    void *loss_out, void *dlogits, float grad_scale, void *scratch, void *stream);
/* L1 + L2 of a training step in one pass (trainer.py:60-82): logits [B,2,H,W] fp32 addressed through element strides
 * (batch, class plane, row; unit pixel stride - the trainer's centre crop of preds is such a view), int64 labels [B,1,H,W]
 * in {0,1} instead of the one-hot target [1-y, y], the optional weight as in unet_bce_logits.  Writes the mean loss, the
 * dense dlogits [B,2,H,W] (x grad_scale; may be NULL) and the argmax mask [B,H,W] int64, ties -> class 0 (may be NULL).
 * scratch: >= unet_bce_step_scratch_bytes(B*H*W). */
size_t unet_bce_step_scratch_bytes(size_t npix);
int unet_bce_step(const void *logits, long xsB, long xsC, long xsH, const void *labels_i64, const void *weight,
                  long wsB, long wsC, long wsH, long wsW, int B, int H, int W, void *loss_out, void *dlogits,
                  float grad_scale, void *mask_i64, void *scratch, void *stream);
/* builds ll from integer labels on device: ll[:,0]=1-y, ll[:,1]=y (trainer.py:63-66) */
int unet_onehot2(const void *labels_i64, void *target, int B, int H, int W, void *stream);

/* L2 replaces preds.argmax(dim=1) (trainer.py:82, tester.py:30): [B,2,H,W] fp32 with row
 * stride ld (elements) and plane stride ps -> [B,H,W] int64; ties -> class 0.               */
int unet_argmax2(const void *logits, long batch_stride, long plane_stride, long row_stride,
                 void *out_i64, int B, int H, int W, void *stream);

/* L3 replaces optim.SGD(lr, momentum).step() (trainer.py:30,78): for each of n tensors
 * buf = first ? g : mu*buf + g ; p -= lr*buf.  Pointer tables are HOST arrays.             */
int unet_sgd_momentum(void *const *params, const void *const *grads, void *const *bufs,
                      const size_t *numel, int n, float lr, float mu, int first_step,
                      void *stream);

/* ---- callers either side of the path; fp32 images [B,H,W], int64 labels ----------
 * N2 overlap-tile front end, replaces mirror_transform + the [0,1] normalisation (data.py:184-188,
 * :249-277): out[b,0,Y,X] = x[b, r(Y), r(X)] with the reference's asymmetric reflection (top/left band
 * without the edge pixel, bottom/right band with it); minmax (from unet_minmax, [B][2]) may be NULL.   */
int unet_minmax(const void *x, int B, size_t n_per_image, void *out_minmax, void *stream);
int unet_mirror_pad(const void *x, int B, int n, int S, const void *minmax, void *out, void *stream);
/* N2 back end, replaces pred[:, :, pad:pad+n, pad:pad+n].argmax(dim=1) + IoU / Pixel_error counting
 * (tester.py:29-42, functions.py:174-213): mask int64 [B,n,n]; with labels int64 [B,n,n]:
 * stats u64 [B][3] = {sum(pred&label), sum(pred|label), sum|pred-label|} (exact integer atomics).      */
int unet_eval_masks(const void *logits, long batch_stride, long plane_stride, long row_stride, int pad,
                    const void *labels_i64, void *mask_i64, int B, int n, void *stats_u64, void *stream);
/* N3, replaces functions.class_balance (functions.py:82-117) for {0,1} labels: w = 1 on cells,
 * count(1)/count(0) on background; counts_u64 [B] receives count(1) (caller checks the degenerate case). */
int unet_class_balance(const void *labels_i64, int B, int H, int W, void *weights, void *counts_u64, void *stream);
/* N1, replaces elastic_transform's two steps (data.py:225-245): scipy.ndimage.gaussian_filter(field,
 * sigma, mode="constant") * scale as two 1-D passes with the caller's normalised taps [2*radius+1], and
 * map_coordinates(img, (row+dy, col+dx), order=1) (bilinear, 0 outside [0,n-1]).                       */
int unet_gaussian_filter(const void *field, int B, int H, int W, const void *weights, int radius, float scale,
                         void *tmp, void *out, void *stream);
int unet_warp_bilinear(const void *img, const void *dy, const void *dx, int B, int H, int W, void *out, void *stream);
/* N1, replaces the reflect pad + random rotation + centre crop of ImageDataset.__getitem__ (data.py:108-125):
 *   np.pad(image, pad, 'reflect') -> scipy.ndimage.rotate(deg) (cubic spline, reshape=True, 'constant') -> [t:b, l:r] (S x S),
 * fused so that only the pixels the crop samples are formed.  img fp32 [B,n,n] (n = the random crop, e.g. 388), pad = the
 * np.pad width (the reference passes input_size = S), angles: HOST array of B degrees, out fp32 [B,S,S].
 * levels: 0 keeps the float result; 255 / 65535 reproduce scipy's conversion for the uint8 / uint16 images the reference
 * loads: (type)min(t > 0 ? t + 0.5 : 0, levels) (scipy 1.15).  scratch >= unet_rotate_scratch_bytes(B, S).             */
size_t unet_rotate_scratch_bytes(int B, int S);
int unet_reflect_rotate_crop(const void *img, int B, int n, int pad, int S, const float *angles_deg_host, int levels,
                             void *out, void *scratch, void *stream);

/* ---- per-op entry points (NHWC fp32), used by the unit tests ---------------------------------
 * Each replaces the ATen op dispatched at the cited line.  w_* are in reference layout.    */
/* nn.Conv2d(3x3, valid)+ReLU, network.py:131-188.  Second source (x2) is the virtual
 * crop_and_concat (network.py:108-127): x1 = skip [B,H1,W1,C1] zero-padded by pad1 per side,
 * x2 = up-conv output [B,H,W,C2]; pass x2=NULL,C2=0,pad1=0 for a plain conv.  H,W = extent of
 * the (virtual) input; y : [B,H-2,W-2,K].  scratch >= unet_conv3x3_scratch_bytes(C1+C2,K).  */
size_t unet_conv3x3_scratch_bytes(int C, int K);
int unet_conv3x3_fwd(const void *x1, int H1, int W1, int C1, int pad1, const void *x2, int C2,
                     int B, int H, int W, const void *w_oihw, const void *bias, int K, int relu,
                     void *y, void *scratch, void *stream);
/* backward of the same: dz is the gradient w.r.t. the PRE-activation of this conv
 * ([B,H-2,W-2,K]).  Outputs (any may be NULL): dx1 [B,H1,W1,C1] (crop of the padded region),
 * dx2 [B,H,W,C2], dw OIHW, db.  mask1/mask2: if non-NULL, dx is multiplied by (mask>0)
 * (ReLU backward of the producer).  add1: if non-NULL it is added to dx1 (skip gradient).
 * scratch >= unet_conv3x3_bwd_scratch_bytes(B,H,W,C1+C2,K).                                  */
size_t unet_conv3x3_bwd_scratch_bytes(int B, int H, int W, int C, int K);
int unet_conv3x3_bwd(const void *x1, int H1, int W1, int C1, int pad1, const void *x2, int C2,
                     int B, int H, int W, const void *w_oihw, int K, const void *dz,
                     void *dx1, const void *mask1, const void *add1, void *dx2, const void *mask2,
                     void *dw, void *db, void *scratch, void *stream);
/* F.max_pool2d(2,2), network.py:133-151 and its backward fused with the ReLU backward of
 * the pooled tensor's producer: dpre = route(dy) * (pre > 0).                              */
int unet_maxpool2_fwd(const void *x, void *y, int B, int H, int W, int C, void *stream);
int unet_maxpool2_bwd(const void *pre, const void *dy, void *dpre, int B, int H, int W, int C,
                      void *stream);
/* nn.ConvTranspose2d(k2,s2), network.py:159-183: x [B,H,W,Ci] -> y [B,2H,2W,Co]; w IOHW. */
size_t unet_upconv2_scratch_bytes(int B, int H, int W, int Ci, int Co);
int unet_upconv2_fwd(const void *x, int B, int H, int W, int Ci, const void *w_iohw,
                     const void *bias, int Co, void *y, void *scratch, void *stream);
int unet_upconv2_bwd(const void *x, int B, int H, int W, int Ci, const void *w_iohw, int Co,
                     const void *dy, void *dx, const void *mask, void *dw, void *db,
                     void *scratch, void *stream);
/* finalconv 1x1 (network.py:190): x NHWC [B,H,W,C] -> logits NCHW [B,2,H,W]; and backward:
 * dz = (dlogits . W) * (x > 0) (x is the ReLU output of conv12e), dw [2,C,1,1], db [2].     */
int unet_head1x1_fwd(const void *x, int B, int H, int W, int C, const void *w, const void *bias,
                     void *logits, void *stream);
size_t unet_head1x1_bwd_scratch_bytes(int B, int H, int W, int C);
int unet_head1x1_bwd(const void *x, int B, int H, int W, int C, const void *w,
                     const void *dlogits, void *dz, void *dw, void *db, void *scratch,
                     void *stream);
/* conv11c (network.py:23,131): the 1->K stencil layer, direct HBM-bound kernel.
 * x [B,S,S] -> y [B,S-2,S-2,K] (+ReLU); backward gives dw [K,1,3,3], db [K] only.          */
int unet_conv1ch_fwd(const void *x, int B, int S, const void *w, const void *bias, int K,
                     void *y, void *stream);
size_t unet_conv1ch_bwd_scratch_bytes(int B, int S, int K);
int unet_conv1ch_bwd(const void *x, int B, int S, int K, const void *dz, void *dw, void *db,
                     void *scratch, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H */
