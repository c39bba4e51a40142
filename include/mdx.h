/*
 * mdx.h -- C-ABI of the MI355X (gfx950) photometric hot path: libmdx_hip.so
 *
 * Drop-in boundary for the self-supervised depth training step of
 * russellgeum/Digging-into-Self-Supervised-Monocular-Depth-Estimation:
 *   model_tool/processor.py:139-163  compute.image2warping
 *   model_tool/processor.py:166-218  compute.compute_loss
 *   model_layer/warp.py:12-39,193-269, model_loss/model_loss.py:11-116 (the ops those two call)
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory (hipMalloc / a torch CUDA tensor's
 *     data_ptr) unless the parameter is named host_*;
 *   - float32, contiguous, NCHW planar; indices uint8;
 *   - the caller allocates every input, output and workspace; the library never allocates, frees or
 *     keeps a pointer after return; outputs are fully overwritten;
 *   - kernels are enqueued on `stream` (a hipStream_t, NULL = default stream) and the call returns
 *     without synchronising; re-entrant, no global mutable state;
 *   - return 0 on success, a negative mdx_status otherwise (never throws, never aborts).
 *
 * Numerics: every per-pixel result reproduces the reference's CPU path (PyTorch ATen, CPU) bit for
 * bit -- see DESIGN.md "pinned operation orders"; reductions and gradients agree to <= 1e-4 rel.
 */
#ifndef MDX_H
#define MDX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_VERSION 510
#define MDX_MAX_SRC 4
#define MDX_MAX_SCALES 4

typedef enum mdx_status {
    MDX_OK = 0,
    MDX_ERR_BAD_SHAPE = -1,       /* non-positive size, S out of range, h/w not H>>k, W or H < 4 */
    MDX_ERR_NULL_POINTER = -2,    /* a required pointer is NULL */
    MDX_ERR_WORKSPACE = -3,       /* workspace missing or too small */
    MDX_ERR_LAUNCH = -4,          /* hipGetLastError() after launch was not hipSuccess */
    MDX_ERR_UNSUPPORTED = -5,     /* flag/mode combination not implemented */
    MDX_ERR_MISALIGNED = -6       /* pointer not 4-byte (float) / 8-byte (workspace) aligned */
} mdx_status;

/* flags */
#define MDX_FLAG_AUTOMASK 1u       /* channels [ident(0..S-1), reproj(0..S-1)] (processor.py:186-196) */
#define MDX_FLAG_UPSAMPLE_PREMUL 2u /* ATen's small-output bilinear kernel (H+W <= 128); set by mdx_desc_init */
#define MDX_FLAG_FASTDIV_W 4u      /* W-1 is in the verified constant-division table (set by mdx_desc_init) */
#define MDX_FLAG_FASTDIV_H 8u      /* H-1 likewise */

/* Problem descriptor of one scale of the step. */
typedef struct mdx_desc {
    int32_t B, H, W;      /* images, full resolution (opt.height, opt.width) */
    int32_t h, w;         /* disparity resolution of this scale: outputs[("disp", s)] is [B,1,h,w] */
    int32_t S;            /* number of source frames: len(opt.frame_ids) - 1, 1..MDX_MAX_SRC */
    uint32_t flags;
    float disp_a, disp_b; /* scaled_disp = disp_a + disp_b*disp (warp.py:34-37), f32-rounded */
} mdx_desc;

/* S source images, each [B,3,H,W] (inputs[("color", frame_id, 0)], order of opt.frame_ids[1:]) */
typedef struct mdx_sources {
    const float *img[MDX_MAX_SRC];
} mdx_sources;

/* Fills a descriptor; computes disp_a/disp_b from (min_depth, max_depth) exactly as
 * warp.py:34-37 does (Python doubles, rounded to f32), and sets MDX_FLAG_UPSAMPLE_PREMUL. */
int mdx_desc_init(mdx_desc *d, int B, int H, int W, int h, int w, int S, int automask,
                  double min_depth, double max_depth);

int mdx_version(void);
const char *mdx_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * Fused path (what the training step runs)
 * ---------------------------------------------------------------------------------------- */

/* P[b] = (K[b] @ T[b])[:3,:]   replaces warp.py:260.  K,T [B,4,4] -> P [B,3,4]. */
int mdx_compose_projection(const float *K, const float *T, int B, float *P, void *stream);

/* Identity (auto-mask) losses, hoisted out of the scale loop because they do not depend on the
 * scale: ident[b,f] = ReprojectionLoss(color_f, target)   replaces processor.py:187-191.
 * target [B,3,H,W], ident out [B,S,H,W] (noise NOT added here). */
int mdx_identity_loss(const mdx_desc *d, const float *target, const mdx_sources *src, float *ident,
                      void *stream);

size_t mdx_photometric_workspace_bytes(const mdx_desc *d);

/* One scale of image2warping + compute_loss, forward   replaces processor.py:141-162,172-204,212.
 *   disp [B,1,h,w]; target [B,3,H,W]; invK [B,4,4]; P [S,B,3,4]; ident [B,S,H,W]; noise [B,S,H,W]
 *   (ident/noise only with MDX_FLAG_AUTOMASK; noise is the N(0,1) draw of processor.py:195).
 * Outputs: idx [B,H,W] uint8 (arg-min channel, torch.min's first-minimum rule);
 *          loss_sum [1] float = sum over B,H,W of to_optimise (divide by B*H*W for .mean()); NULL skips
 *          the finishing pass and leaves one double per tile at the start of the workspace;
 * optional (NULL to skip): to_opt [B,H,W]; depth [B,1,H,W]; warp [S,B,3,H,W]; reproj [B,S,H,W];
 *          coef [B,3,H,W,3] (9*B*H*W floats) = the SSIM coefficient triplets (alpha,beta,gamma) per colour
 *          channel of each pixel's
 *          arg-min frame, zero where an identity channel won -- hand coef (or, without it, warp) to
 *          mdx_photometric_bwd. */
int mdx_photometric_fwd(const mdx_desc *d, const float *disp, const float *target,
                        const mdx_sources *src, const float *invK, const float *P,
                        const float *ident, const float *noise, uint8_t *idx, float *loss_sum,
                        float *to_opt, float *depth, float *warp, float *reproj, float *coef,
                        void *workspace, size_t workspace_bytes, void *stream);

/* Measurement hook: a pair of hipEvent_t handles that the *_timed entry points record on `stream` immediately
 * before and immediately after the FUSED kernel of the call (not around its small finishing passes), so that a
 * caller can time that kernel inside a real training step.  mdx_event_* are thin wrappers over hipEventCreate /
 * hipEventDestroy / hipEventSynchronize + hipEventElapsedTime for callers without a HIP binding. */
typedef struct mdx_timing { void *start, *stop; } mdx_timing;
void *mdx_event_create(void);
void mdx_event_destroy(void *event);
int mdx_event_elapsed_us(void *start, void *stop, float *us);   /* waits for `stop` */

/* mdx_photometric_fwd / mdx_photometric_bwd with the hook (t == NULL: exactly the plain call). */
int mdx_photometric_fwd_timed(const mdx_desc *d, const float *disp, const float *target,
                              const mdx_sources *src, const float *invK, const float *P,
                              const float *ident, const float *noise, uint8_t *idx, float *loss_sum,
                              float *to_opt, float *depth, float *warp, float *reproj, float *coef,
                              void *workspace, size_t workspace_bytes, void *stream, const mdx_timing *t);

/* Backward of the above for d(loss)/d(to_optimise[b,y,x]) = g_const * (*g_dev) on every pixel
 * (g_dev may be NULL = 1).  Needs only the inputs and idx; `warp` (optional, [S,B,3,H,W]) is the forward's
 * warped-colour output -- when given the kernel reads it instead of re-warping the 2-pixel halo; `coef`
 * (optional, [B,3,H,W,3]) is the forward's coefficient output -- with it the backward skips the window
 * statistics altogether and needs no `warp` (it re-samples a pixel's own warped colour from the corners it
 * gathers for the gradient; `warp` is ignored when `coef` is given).
 * Outputs: gdisp [B,1,h,w]; gP [S,B,3,4] (d loss / d P; chain to T with K^T outside).  Passing NULL for BOTH
 * runs the fused kernel alone and leaves its raw per-tile outputs in the workspace (timing aid). */
int mdx_photometric_bwd(const mdx_desc *d, const float *disp, const float *target,
                        const mdx_sources *src, const float *invK, const float *P,
                        const uint8_t *idx, const float *warp, const float *coef, float g_const,
                        const float *g_dev, float *gdisp, float *gP, void *workspace, size_t workspace_bytes,
                        void *stream);
int mdx_photometric_bwd_timed(const mdx_desc *d, const float *disp, const float *target,
                        const mdx_sources *src, const float *invK, const float *P,
                        const uint8_t *idx, const float *warp, const float *coef, float g_const,
                        const float *g_dev, float *gdisp, float *gP, void *workspace, size_t workspace_bytes,
                        void *stream, const mdx_timing *t);

/* ------------------------------------------------------------------------------------------
 * Training step: every scale, forward AND gradient, in one launch
 *   replaces the scale loops of processor.py:140-163 and :167-204,212 plus their autograd backward.
 * Everything downstream of sum(to_optimise) is linear in the upstream gradient, so the kernel returns the
 * gradients for a UNIT upstream: d loss / d disp_s = g_s * gdisp[s], d loss / d P = sum_s g_s * gP[s]
 * where g_s = d loss / d loss_sum[s] (the caller's autograd multiplies; nothing else is kept for backward).
 * ---------------------------------------------------------------------------------------- */
typedef struct mdx_train_desc {
    int32_t B, H, W;                 /* images, full resolution */
    int32_t S;                       /* source frames, 1..MDX_MAX_SRC */
    int32_t nscales;                 /* len(opt.scales), 1..MDX_MAX_SCALES */
    uint32_t flags;                  /* as mdx_desc.flags */
    float disp_a, disp_b;            /* as mdx_desc */
    int32_t h[MDX_MAX_SCALES], w[MDX_MAX_SCALES];   /* outputs[("disp", s)] is [B,1,h[s],w[s]] */
    int32_t rows_per_chunk;          /* rows one wave walks; 0 = chosen by the library */
} mdx_train_desc;

int mdx_train_desc_init(mdx_train_desc *d, int B, int H, int W, int S, int nscales, const int32_t *h,
                        const int32_t *w, int automask, double min_depth, double max_depth, int rows_per_chunk);
size_t mdx_photometric_train_workspace_bytes(const mdx_train_desc *d);

/* disp, P, noise, idx, gdisp, to_opt: HOST arrays of nscales DEVICE pointers.
 *   disp[s] [B,1,h[s],w[s]]; P[s] [S,B,3,4] (the same pointer for every scale unless the pose depends on the
 *   scale); noise[s], ident [B,S,H,W] (MDX_FLAG_AUTOMASK only); target [B,3,H,W]; invK [B,4,4].
 * Outputs: idx[s] [B,H,W] uint8; loss_sum [nscales]; gdisp[s] [B,1,h[s],w[s]]; gP [nscales,S,B,3,4];
 *   optional depth0 [B,1,H,W] (depth of scale 0, outputs[("depth",0,0)]); optional to_opt (array may be NULL,
 *   entries may be NULL) [B,H,W].
 * gdisp == NULL and gP == NULL (both): every scale's FORWARD alone -- loss sums, indices, depth0, to_opt -- in one
 *   launch: what the validation loop (model_train.py:75-79, under torch.no_grad()) and model_test.py need.
 * t (optional): events recorded right before / after the fused kernel. */
int mdx_photometric_train(const mdx_train_desc *d, const float *const *disp, const float *target,
                          const mdx_sources *src, const float *invK, const float *const *P, const float *ident,
                          const float *const *noise, uint8_t *const *idx, float *loss_sum, float *const *gdisp,
                          float *gP, float *depth0, float *const *to_opt, void *workspace, size_t workspace_bytes,
                          void *stream, const mdx_timing *t);

/* ------------------------------------------------------------------------------------------
 * What the scales of one step share, once per step (csrc/photo_prologue.hip)
 *   processor.py:187-191 (identity losses), model_loss.py:28-33 (the target's SSIM window statistics, evaluated
 *   nscales * S times per step by the reference), processor.py:194-204 (identity + 1e-5 * randn and, since the identity
 *   channels come first and torch.min keeps the first minimum, the best identity channel of each pixel).
 * Outputs: tstat [B,H,W,6] = (mu_y[3], sigma_y[3]) per pixel; with MDX_FLAG_AUTOMASK per scale bidfi[s] [B,H,W,2] =
 *   (best identity value as float32, its channel index as int32); optional ident [B,S,H,W].
 * noise: HOST array of nscales device pointers [B,S,H,W] (injected draws: parity), or NULL: N(0,1) drawn in the kernel
 *   (Philox4x32-10 + Box-Muller) from rng_state = DEVICE {uint64 seed, uint64 offset}; advance_rng != 0 adds one to the
 *   offset afterwards (mdx_photometric_train_pre does so itself when it is handed rng_state). */
int mdx_photometric_prologue(const mdx_train_desc *d, const float *target, const mdx_sources *src,
                             const float *const *noise, unsigned long long *rng_state, int advance_rng,
                             float *ident, float *tstat, float *const *bidfi, void *stream);

/* mdx_photometric_train with the prologue's outputs in place of ident / noise: the kernel loads the target statistics
 * and the best identity channel (32 B per pixel and scale) instead of re-deriving / re-reading them.  Same outputs, bit
 * for bit.  rng_state (optional): its offset is advanced by the finishing kernel -- the next step (also the next replay
 * of a captured graph) draws new noise. */
int mdx_photometric_train_pre(const mdx_train_desc *d, const float *const *disp, const float *target,
                              const mdx_sources *src, const float *invK, const float *const *P,
                              const float *tstat, const float *const *bidfi, unsigned long long *rng_state,
                              uint8_t *const *idx, float *loss_sum, float *const *gdisp, float *gP, float *depth0,
                              float *const *to_opt, void *workspace, size_t workspace_bytes, void *stream,
                              const mdx_timing *t);

/* Edge-aware smoothness   replaces model_loss.py:77-88,112-115 (processor.py:208).
 * disp [B,1,h,w], color [B,3,h,w] -> loss [1]; gdisp (optional) = d loss / d disp for unit upstream.
 * normalize = 1: SmoothLoss (disp / (mean_HW(disp) + 1e-7) first); 0: EdgeAwareSmooth on disp as given.
 * Two launches (MDX_VERSION 400; four before): a main pass on the disparity as given + a finishing pass that forms the
 * per-image mean from the main pass's partials. */
size_t mdx_smooth_workspace_bytes(int B, int h, int w);
int mdx_smooth_loss(int B, int h, int w, const float *disp, const float *color, int normalize, float *loss,
                    float *gdisp, void *workspace, size_t workspace_bytes, void *stream);

/* The same for every scale of a step, both passes launched once for all scales (the scale loop of processor.py:208):
 * h, w, disp, color, gdisp: HOST arrays of nscales entries (gdisp NULL or all entries given); loss [nscales]. */
size_t mdx_smooth_multi_workspace_bytes(int nscales, int B, const int32_t *h, const int32_t *w);
int mdx_smooth_loss_multi(int nscales, int B, const int32_t *h, const int32_t *w, const float *const *disp,
                          const float *const *color, int normalize, float *loss, float *const *gdisp,
                          void *workspace, size_t workspace_bytes, void *stream);

/* The scalar tail of the loss   processor.py:208-217 (scale_loss = mean + disp_smoothness * smooth / 2**scale; total = sum / len)
 * and its backward fan-out, one launch each way instead of ~20 / ~30 scalar launches + three whole-map passes per scale.
 * sums [nscales] = sum over the pixels of to_optimise (mdx_photometric_train), smooth [nscales] (mdx_smooth_loss_multi),
 * scale: HOST array opt.scales, pixels = B*H*W -> total [1].  Rounded op by op as ATen-GPU rounds the reference's expression.
 * bwd: g_total [1] (device), gd_photo / gd_smooth: HOST arrays of nscales device pointers (the unit-upstream gradients the two
 * kernels left), count[s] = elements of scale s -> gdisp[s] = gd_photo[s] * c1 + gd_smooth[s] * c2[s];
 * gP [nscales][nP] (optional) -> gP_out [nscales][nP] * c1 (per_scale_P) or their sum over the scales [nP]. */
int mdx_loss_total_fwd(int nscales, const float *sums, const float *smooth, const int32_t *scale, int64_t pixels,
                       double disp_smoothness, float *total, void *stream);
int mdx_loss_total_bwd(int nscales, const float *g_total, const int32_t *scale, int64_t pixels, double disp_smoothness,
                       const float *const *gd_photo, const float *const *gd_smooth, const int64_t *count,
                       float *const *gdisp, const float *gP, int nP, int per_scale_P, float *gP_out, void *stream);

/* The optimiser step   reference model_tool/loader.py:93-97 (torch.optim.Adam, no weight decay / amsgrad / maximize) in ONE launch
 * for every parameter of a group.  table: DEVICE array of { float *param, *exp_avg, *exp_avg_sq; const float *step; int64_t numel }
 * (mdx_adam_table_entry_bytes() each; step = the tensor's float32 step count, already incremented); one call takes entries
 * first .. first + count - 1, count <= mdx_adam_max_tensors(); grads: HOST array of count device pointers (they change from step
 * to step); blockmap: DEVICE array of nblocks { int32 tensor (0-based within the call), int32 chunk } covering every tensor in
 * chunks of mdx_adam_chunk() elements; lr_ptr: device float32 (a captured step's learning rate) or NULL (then lr).
 * Arithmetic: ATen's fused Adam, expression by expression (ATen/native/cuda/fused_adam_utils.cuh). */
int mdx_adam_max_tensors(void);
int mdx_adam_chunk(void);
size_t mdx_adam_table_entry_bytes(void);
int mdx_adam_step(const void *table, int first, int count, const float *const *grads, const void *blockmap, int nblocks,
                  const float *lr_ptr, double lr, double beta1, double beta2, double eps, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fine-grained ops behind the reference's model_layer / model_loss API (each differentiable)
 * ---------------------------------------------------------------------------------------- */

/* interpolate(x, H, W, "bilinear", align_corners=False)   warp.py:18-20.  x [BC,h,w] -> [BC,H,W] */
int mdx_interpolate_bilinear_fwd(const float *x, int BC, int h, int w, float *out, int H, int W,
                                 void *stream);
int mdx_interpolate_bilinear_bwd(const float *gout, int BC, int H, int W, float *gin, int h, int w,
                                 void *stream);

/* disparity2depth   warp.py:29-39.  sd/depth may be NULL.  bwd: gdisp = gsd*b + gdepth*(-b*depth^2) */
int mdx_disparity2depth_fwd(const float *disp, size_t n, double min_depth, double max_depth,
                            float *sd, float *depth, void *stream);
int mdx_disparity2depth_bwd(const float *disp, const float *gsd, const float *gdepth, size_t n,
                            double min_depth, double max_depth, float *gdisp, void *stream);

/* Depth2PointCloud.forward   warp.py:237-246.  depth [B,1,H,W], invK [B,4,4] -> cam [B,4,HW] */
int mdx_backproject_fwd(const float *depth, const float *invK, int B, int H, int W, float *cam,
                        void *stream);
int mdx_backproject_bwd(const float *gcam, const float *invK, int B, int H, int W, float *gdepth,
                        void *stream);

/* PointCloud2Pixel.forward   warp.py:259-269, with P = (K@T)[:3].  cam [B,4,HW] -> grid [B,H,W,2]
 * bwd: ggrid -> gcam [B,4,HW] (row 3 zero) and gP [B,3,4]. workspace: mdx_project_workspace_bytes */
size_t mdx_project_workspace_bytes(int B, int H, int W);
int mdx_project_fwd(const float *cam, const float *P, int B, int H, int W, float eps, float *grid,
                    void *stream);
int mdx_project_bwd(const float *cam, const float *P, const float *ggrid, int B, int H, int W,
                    float eps, float *gcam, float *gP, void *workspace, size_t workspace_bytes,
                    void *stream);

/* grid_sample(img, grid, "border", align_corners=True), bilinear   warp.py:12-14.
 * img [B,C,Hi,Wi], grid [B,Ho,Wo,2] -> out [B,C,Ho,Wo];  bwd -> ggrid (gimg optional, atomics) */
int mdx_grid_sample_border_fwd(const float *img, const float *grid, int B, int C, int Hi, int Wi,
                               int Ho, int Wo, float *out, void *stream);
int mdx_grid_sample_border_bwd(const float *img, const float *grid, const float *gout, int B, int C,
                               int Hi, int Wi, int Ho, int Wo, float *ggrid, float *gimg,
                               void *stream);

/* ReprojectionLoss.forward   model_loss.py:97-103 (SSIM 28-41).  pred,target [B,3,H,W] -> [B,1,H,W]
 * bwd: gout [B,1,H,W] -> gpred [B,3,H,W] (gtarget optional) */
int mdx_reprojection_loss_fwd(const float *pred, const float *target, int B, int H, int W,
                              float *out, void *stream);
int mdx_reprojection_loss_bwd(const float *pred, const float *target, const float *gout, int B,
                              int H, int W, float *gpred, float *gtarget, void *stream);
/* SSIM.forward alone   model_loss.py:28-41.  x,y [BC,H,W] -> [BC,H,W] */
int mdx_ssim_fwd(const float *x, const float *y, int BC, int H, int W, float *out, void *stream);
/* its backward (the reference's SSIM module is differentiable through autograd): gout [BC,H,W] -> gx and / or gy [BC,H,W] */
int mdx_ssim_bwd(const float *x, const float *y, const float *gout, int BC, int H, int W, float *gx, float *gy,
                 void *stream);

/* identity+noise, concat, per-pixel min   processor.py:194-204.  ident/noise/reproj [B,S,H,W]
 * -> combined [B,C,H,W] (optional), to_opt [B,H,W], idx [B,H,W] uint8 */
int mdx_min_automask_fwd(const float *ident, const float *noise, const float *reproj, int B, int S,
                         int H, int W, int automask, float *combined, float *to_opt, uint8_t *idx,
                         void *stream);

/* param2matrix   model_layer/warp.py:126-153 (with vector2translation :43-61, angle2rotation :65-122).
 * axisangle, translation [N,3] (the reference's [N,1,3]) -> M [N,4,4]; invert as in the reference.
 * bwd: gM [N,4,4] -> gaxisangle, gtranslation [N,3]. */
int mdx_param2matrix_fwd(const float *axisangle, const float *translation, int N, int invert, float *M, void *stream);
int mdx_param2matrix_bwd(const float *axisangle, const float *translation, const float *gM, int N, int invert,
                         float *gaxisangle, float *gtranslation, void *stream);

/* The pose network's output -> camera-to-camera matrices and projections for every source frame, one launch each way
 * (processor.py:61-83 slices + param2matrix, :143-160 K @ T): raw [M,F,6] = the pose head's 0.01-scaled output
 * (axis-angle | translation; pose_decoder.py:51-53); source s reads rows row0[s] .. row0[s]+B-1, entry frame[s], inverted if
 * invert[s] (HOST arrays of S entries); K [B,4,4] -> T [S,B,4,4], P [S,B,3,4] = (K @ T)[:, :3].
 * bwd: gP [S,B,3,4] and / or gT [S,B,4,4] (either may be NULL) -> graw [M,F,6] (entries no source reads: zeros). */
int mdx_pose_projection_fwd(const float *raw, int M, int F, const float *K, int B, int S, const int32_t *row0,
                            const int32_t *frame, const int32_t *invert, float *T, float *P, void *stream);
int mdx_pose_projection_bwd(const float *raw, int M, int F, const float *K, int B, int S, const int32_t *row0,
                            const int32_t *frame, const int32_t *invert, const float *gP, const float *gT, float *graw,
                            void *stream);

/* ---- network glue around the convolutions (no reference FFI: these replace torch op sequences of
 * model_layer/depth_decoder.py:44-47,96-106 and the ResNet stem max-pool, model_layer/depth_encoder.py) ----
 * dtype codes: 0 = float32, 1 = bfloat16 (storage; arithmetic is float32). */

/* out [B,C1+C2,u*h+2,u*w+2] = ReflectionPad2d(1)( cat( nearest_up_u( act(raw [B,C1,h,w] + bias [C1]) ), skip [B,C2,u*h,u*w] ) ),
 * u = upsample ? 2 : 1, act = elu ? ELU : identity; skip may be NULL when C2 == 0; bias (float32, may be NULL) is
 * the bias of the convolution that produced raw, when that convolution was run without it.
 * (in_dtype, out_dtype) in {(0,0), (1,1), (1,0)}; skip has in_dtype. */
int mdx_decoder_glue_fwd(const void *raw, const void *skip, const float *bias, void *out, int B, int C1, int C2, int h,
                         int w, int upsample, int elu, int in_dtype, int out_dtype, void *stream);
/* gout has out's shape / out_dtype; graw, gskip have the inputs' shapes / in_dtype (gskip NULL when C2 == 0);
 * dbias [C1] float32 = sum of graw over (b, y, x) (NULL to skip; needs the workspace, fixed summation order). */
size_t mdx_decoder_glue_workspace_bytes(int B, int C1, int h, int w);
int mdx_decoder_glue_bwd(const void *gout, const void *raw, const float *bias, void *graw, void *gskip, float *dbias,
                         int B, int C1, int C2, int h, int w, int upsample, int elu, int in_dtype, int out_dtype,
                         void *workspace, size_t workspace_bytes, void *stream);

/* MaxPool2d(3, stride 2, padding 1) on [BC,H,W] -> [BC,Ho,Wo], Ho = (H-1)/2+1; arg [BC,Ho,Wo] uint8 = window tap
 * (0..8, row-major) of the first maximum, consumed by the backward (a gather, no atomics). */
int mdx_maxpool3s2_fwd(const void *in, void *out, uint8_t *arg, int BC, int H, int W, int dtype, void *stream);
int mdx_maxpool3s2_bwd(const void *gout, const uint8_t *arg, void *gin, int BC, int H, int W, int dtype, void *stream);

/* Training-mode BatchNorm2d + residual add + ReLU of the ResNet blocks (model_layer/depth_encoder.py; the reference
 * gets them from torchvision): y = act(bn(x) [+ res]) with batch statistics over (B,H,W), running statistics updated
 * in place as torch.nn.functional.batch_norm(training=True) does (run_mean/run_var may both be NULL).
 * The tensors hold `groups` consecutive sub-batches of B images: x, res, y, dy, dx, dres [groups*B,C,H,W] float32
 * (dtype 0) or bfloat16 (dtype 1); each sub-batch is normalised with its own statistics and updates the running
 * statistics in turn -- what `groups` calls on the sub-batches do.  gamma, beta, statistics float32;
 * save_mean / save_invstd [groups,C] go from forward to backward.  One launch each way for maps up to 24 K elements
 * per channel and sub-batch, two per sub-batch above. */
size_t mdx_bn_workspace_bytes(int B, int C, int H, int W);
int mdx_bn_act_fwd(const void *x, const void *res, const float *gamma, const float *beta, float *run_mean,
                   float *run_var, void *y, float *save_mean, float *save_invstd, int B, int C, int H, int W,
                   int groups, float eps, float momentum, int relu, int dtype, void *workspace, size_t workspace_bytes,
                   void *stream);
/* dz = dy * (y > 0) when relu; dx, d(res) = dz (dres NULL when there was no residual); dgamma, dbeta [C] summed
 * over the sub-batches. */
int mdx_bn_act_bwd(const void *dy, const void *y, const void *x, const float *gamma, const float *save_mean,
                   const float *save_invstd, void *dx, void *dres, float *dgamma, float *dbeta, int B, int C, int H,
                   int W, int groups, int relu, int dtype, void *workspace, size_t workspace_bytes, void *stream);

/* ---- the same network glue for CHANNELS-LAST maps (memory [B][H][W][C]; csrc/norm_nhwc.hip, csrc/glue_nhwc.hip) ----
 * What MIOpen's implicit-GEMM convolutions read and write without a layout transpose on either side.  Same operators,
 * same arithmetic and summation order per element as the planar entry points above; a thread owns one 16-byte channel
 * vector, so every channel count must be a multiple of 4 (float32) / 8 (bfloat16) and every map pointer 16-byte aligned
 * (else MDX_ERR_BAD_SHAPE / MDX_ERR_MISALIGNED: the caller then takes the planar entry points).
 *
 * mdx_bn_act_nhwc_*: model_layer/depth_encoder.py:27,95 (BatchNorm2d + ReLU + residual of every ResNet block).
 * x, res, y, dy, dx, dres [groups*B][H][W][C]; three launches each way (block partials, a float64 finalize pass, apply),
 * no atomics.  dy2 (may be NULL): a second upstream gradient of y, added on the way in -- a block's output feeds the next
 * block's first convolution AND its identity path, and autograd would otherwise spend one more pass over the map on
 * adding the two.  bwd with y NULL (relu, no residual: dres NULL; beta given): the ReLU mask is re-derived from x with the forward
 * pass's own expressions instead of being read -- one map less in each of the two passes (beta is read in that case only). */
size_t mdx_bn_nhwc_workspace_bytes(int B, int C, int H, int W, int groups, int dtype);
int mdx_bn_act_nhwc_fwd(const void *x, const void *res, const float *gamma, const float *beta, float *run_mean,
                        float *run_var, void *y, float *save_mean, float *save_invstd, int B, int C, int H, int W,
                        int groups, float eps, float momentum, int relu, int dtype, void *workspace,
                        size_t workspace_bytes, void *stream);
int mdx_bn_act_nhwc_bwd(const void *dy, const void *dy2, const void *y, const void *x, const float *gamma,
                        const float *beta, const float *save_mean, const float *save_invstd, void *dx, void *dres, float *dgamma,
                        float *dbeta, int B, int C, int H, int W, int groups, int relu, int dtype, void *workspace,
                        size_t workspace_bytes, void *stream);
/* Between the pose decoder's convolutions   model_layer/pose_decoder.py:24-53 (the convolution runs without its bias).
 * bias_act: y = act(x + bias), x, y [B][H][W][C] channels-last (dtype 0 float32 / 1 bfloat16, C a multiple of the 16-byte vector),
 * bias [C] float32, relu 0 / 1.  bwd: dy, y -> dx, dbias [C]: one launch + a finishing pass over the block partials (workspace).
 * mean_bias: out [B][C] float32 = scale * (mean over H*W of x + bias) -- the head's spatial mean and 0.01 (pose_decoder.py:51-53);
 * any C.  bwd: gout [B][C] -> dx (x's dtype), dbias [C] (may be NULL). */
size_t mdx_bias_act_nhwc_workspace_bytes(int B, int C, int H, int W, int dtype);
int mdx_bias_act_nhwc_fwd(const void *x, const float *bias, void *y, int B, int C, int H, int W, int relu, int dtype, void *stream);
int mdx_bias_act_nhwc_bwd(const void *dy, const void *y, void *dx, float *dbias, int B, int C, int H, int W, int relu, int dtype,
                          void *workspace, size_t workspace_bytes, void *stream);
int mdx_mean_bias_nhwc_fwd(const void *x, const float *bias, float *out, int B, int C, int H, int W, float scale, int dtype,
                           void *stream);
int mdx_mean_bias_nhwc_bwd(const float *gout, void *dx, float *dbias, int B, int C, int H, int W, float scale, int dtype,
                           void *stream);

/* The networks' input   depth_encoder.py:89 ((x - 0.45) / 0.225) + processor.py:61-75 (the pose network's frame pairs concatenated
 * along the channels; this package: the pairs along the batch) written channels-last in one pass.  src: HOST array of
 * blocks * groups (1..2 each) device pointers, src[k * groups + g] a planar float32 [n][3][H][W] frame;
 * out [blocks * n][H][W][3 * groups] (dtype 0 float32 / 1 bfloat16) = (src - mean) * inv_std. */
int mdx_encoder_input_nhwc(const float *const *src, int blocks, int groups, int n, int H, int W, float mean, float inv_std,
                           void *out, int dtype, void *stream);

/* Weight gradient of the decoder's thin 3x3 convolutions   model_layer/depth_decoder.py:96-106 (16 output channels, 16 or 32 input
 * channels, on the two biggest maps): x [B][h+2][w+2][Cin] the reflection-padded input, gy [B][h][w][16], float32 channels-last,
 * w a multiple of 4 -> gweight, element (co, ci, ky, kx) at gweight[co * s_o + ci * s_c + ky * s_y + kx * s_x].  One MFMA launch
 * (v_mfma_f32_16x16x4_f32; operands loaded as they lie in memory) + a finishing pass over the block partials (fixed order). */
size_t mdx_thin_conv3x3_wgrad_workspace_bytes(int B, int Cin, int Cout, int h, int w);
int mdx_thin_conv3x3_wgrad(const float *x, const float *gy, float *gweight, int64_t s_o, int64_t s_c, int64_t s_y, int64_t s_x, int B,
                           int Cin, int Cout, int h, int w, void *workspace, size_t workspace_bytes, void *stream);

/* The decoder's disparity heads   model_layer/depth_decoder.py:73-74,108-110: sigmoid(Conv3x3(C -> 1)(x)) on a channels-last map.
 * x [B][h+2][w+2][C] = the reflection-padded input (mdx_decoder_glue_nhwc_fwd's output), dtype 0 float32 / 1 bfloat16, C a
 * power of two from 4 to 256; weight: float32, element (c, ky, kx) at
 * weight[c * w_stride_c + ky * w_stride_ky + kx * w_stride_kx] (planar [1,C,3,3]: 9, 3, 1; channels-last: 1, 3C, C);
 * bias [1] (may be NULL) -> disp [B][h][w] float32.  One launch.
 * bwd: gdisp, disp [B][h][w] -> gx (x's dtype and shape), gweight (the weight's strides), gbias [1] (may be NULL): one launch
 * that reads x once and writes gx once + a finishing pass over the block partials (fixed order, no atomics). */
size_t mdx_disp_head_nhwc_workspace_bytes(int B, int C, int h, int w, int dtype);
int mdx_disp_head_nhwc_fwd(const void *x, const float *weight, int64_t w_stride_c, int64_t w_stride_ky, int64_t w_stride_kx,
                           const float *bias, float *disp, int B, int C, int h, int w, int dtype, void *stream);
int mdx_disp_head_nhwc_bwd(const void *x, const float *weight, int64_t w_stride_c, int64_t w_stride_ky, int64_t w_stride_kx,
                           const float *gdisp, const float *disp, void *gx, float *gweight, float *gbias, int B, int C, int h,
                           int w, int dtype, void *workspace, size_t workspace_bytes, void *stream);
/* mdx_decoder_glue_nhwc_*: model_layer/depth_decoder.py:44-47,96-106.  raw [B][h][w][C1], skip [B][u*h][u*w][C2],
 * out / gout [B][u*h+2][u*w+2][C1+C2]; dtype pairs as mdx_decoder_glue_fwd. */
int mdx_decoder_glue_nhwc_fwd(const void *raw, const void *skip, const float *bias, void *out, int B, int C1, int C2,
                              int h, int w, int upsample, int elu, int in_dtype, int out_dtype, void *stream);
size_t mdx_decoder_glue_nhwc_workspace_bytes(int B, int C1, int h, int w, int in_dtype);
int mdx_decoder_glue_nhwc_bwd(const void *gout, const void *raw, const float *bias, void *graw, void *gskip,
                              float *dbias, int B, int C1, int C2, int h, int w, int upsample, int elu, int in_dtype,
                              int out_dtype, void *workspace, size_t workspace_bytes, void *stream);
/* mdx_maxpool3s2_nhwc_*: the ResNet stem's MaxPool2d(3, 2, 1).  in / gin [B][H][W][C]; out, arg, gout [B][Ho][Wo][C];
 * gout2 (may be NULL): a second upstream gradient, added on the way in. */
int mdx_maxpool3s2_nhwc_fwd(const void *in, void *out, uint8_t *arg, int B, int C, int H, int W, int dtype,
                            void *stream);
int mdx_maxpool3s2_nhwc_bwd(const void *gout, const void *gout2, const uint8_t *arg, void *gin, int B, int C, int H,
                            int W, int dtype, void *stream);

/* Train-time depth monitor   replaces model_loss/model_metric.py:70-105 (called every step, model_train.py:69).
 * pred [B,1,h,w] (outputs[("depth",0,0)]), gt [B,1,gh,gw] (0 = no return); window rows r0:r1, cols c0:c1 (the Garg crop).
 * out [8] = abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3, number of valid pixels.  Bilinear resize to the ground truth's
 * size, clamp, batch-level median scaling (torch.median's lower median, exact), clamp, compute_depth_error -- without
 * compacting or sorting: two 16-bit radix-histogram passes find the medians. */
size_t mdx_depth_monitor_workspace_bytes(int B, int r0, int r1, int c0, int c1);
int mdx_depth_monitor(const float *pred, int B, int h, int w, const float *gt, int gh, int gw, int r0, int r1, int c0,
                      int c1, float min_depth, float max_depth, float *out, void *workspace, size_t workspace_bytes,
                      void *stream);

/* ---- image preparation on the GPU   replaces, per sample and frame, the Pillow / torchvision calls of
 * model_loader/kitti_mono.py:288-291 + 349-353 (transforms.Resize((H>>s, W>>s), Image.ANTIALIAS) of the original image
 * for every scale, ColorJitter, ToTensor) and kitti_mono.py:302-303 (FLIP_LEFT_RIGHT); same in kitti_stereo.py.
 * Bit-exact with Pillow (Resample.c, Blend.c, Convert.c, ImageEnhance.py) -- uint8 results, float32 = u8 / 255.
 * Jobs are HOST arrays of plain structs whose pointers are DEVICE pointers (except where noted); they are passed to the
 * kernels by value (resampling: packed, MDX_IMG_JOBS jobs sharing at most 16 plans per axis per launch; jitter:
 * MDX_JITTER_JOBS per launch), so the caller may free or reuse the array on return.  A call may hold any number of jobs. */
#define MDX_IMG_JOBS 72
#define MDX_JITTER_JOBS 56
#define MDX_JITTER_PARTIALS 128

/* taps per output sample of the Lanczos-3 plan in_size -> out_size (2*ceil(3*max(in/out,1)) + 1) */
int mdx_resample_ksize(int in_size, int out_size);
/* Resample.c precompute_coeffs + normalize_coeffs_8bpc: bounds [out_size][2] = (first source sample, number of taps),
 * kk [ksize][out_size] = 22-bit fixed-point weights, TAP-MAJOR (the transpose of Pillow's table, so that consecutive
 * outputs read consecutive weights).  HOST arrays (the caller uploads and caches them per size pair). */
int mdx_resample_plan(int in_size, int out_size, int *bounds, int *kk);
/* The same weights COLUMN-MAJOR for the rows form of the horizontal pass (round 4), which fetches a column's taps with scalar
 * loads and evaluates two neighbouring columns over the union of their windows: table [2][out_size][row] ints -- direction 0:
 * table[0][x][lead + t] = weight t of output x; direction 1 (flipped images): the weights of x in reverse order,
 * table[1][x][lead + t] = weight (n_x - 1 - t); everything else 0.  lead >= the largest offset between the windows of two
 * neighbouring outputs (either direction); row = lead + ksize + lead rounded up to 16, + 16: every 16-entry read of a pair's
 * union stays inside a row.  Call with table == NULL for the sizes (*lead, *row); HOST arrays, uploaded and cached by the
 * caller like the plan itself. */
int mdx_resample_plan_cols(int in_size, int out_size, int *lead, int *row, int *table);

typedef struct mdx_resample_job {
    const uint8_t *src;          /* interleaved RGB, rows of in_stride bytes (>= 3*in_w): the decoder's layout */
    const int *xbounds, *xkk;    /* plan in_w -> out_w (device copies) */
    const int *ybounds, *ykk;    /* plan in_h -> out_h */
    uint8_t *inter;              /* scratch, planar [3][in_h][out_w] */
    uint8_t *dst_u8;             /* planar [3][out_h][out_w], or NULL */
    float *dst_f32;              /* planar [3][out_h][out_w] = u8 / 255 (ToTensor), or NULL */
    int in_h, in_w, in_stride, flip;   /* flip: resize image.transpose(FLIP_LEFT_RIGHT) */
    int out_h, out_w, xksize, yksize;
    const int *xkc;              /* mdx_resample_plan_cols table of in_w -> out_w (device copy) or NULL: without it the */
    int xkc_lead, xkc_row;       /* horizontal pass runs in its general gather form (slower, same bytes)              */
} mdx_resample_job;
/* Image.resize((out_w, out_h), Image.LANCZOS): horizontal pass to uint8, then vertical pass. */
int mdx_resample_lanczos_u8(const mdx_resample_job *jobs, int njobs, void *stream);

typedef struct mdx_jitter_job {
    const uint8_t *src;          /* planar [3][h][w] */
    float *dst_f32;              /* planar [3][h][w] = u8 / 255, or NULL */
    uint8_t *dst_u8;             /* planar, or NULL (may alias src: each pixel is read before it is written) */
    unsigned long long *lsum;    /* scratch, MDX_JITTER_PARTIALS 8-byte words per job (partial sums of L for Contrast) */
    int h, w;
    int order[4];                /* 0 brightness, 1 contrast, 2 saturation, 3 hue, 4 = empty slot; each at most once */
    int hue_shift;               /* int(hue_factor * 255), added to the H byte modulo 256 */
    float brightness, contrast, saturation;   /* ImageEnhance factors */
} mdx_jitter_job;
/* torchvision ColorJitter on PIL images: the adjustments in `order`, uint8 after each. */
int mdx_color_jitter_u8(const mdx_jitter_job *jobs, int njobs, void *stream);

/* Unfused parity op: Pillow's convert() maps on planar uint8 [3][npix]: mode 0 RGB->HSV, 1 HSV->RGB, 2 RGB->L ([npix] out). */
int mdx_color_convert_u8(int mode, const uint8_t *src, uint8_t *dst, size_t npix, void *stream);

/* torchvision ToTensor on a uint8 image (kitti_mono.py:283,351; kitti_stereo.py:280-281): dst[i] = float32(src[i]) / 255
 * with the IEEE divide (NOT a multiplication by 1/255: that differs in the last bit for 126 of the 256 byte values).
 * src: n bytes, any layout; dst: n floats, 16-byte aligned. */
int mdx_to_tensor_u8(const uint8_t *src, float *dst, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MDX_H */
