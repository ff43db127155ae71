/*
 * nfs_hip.h -- C ABI of libnfs_hip.so: the MI355X (gfx950) operator library for
 * the stylisation hot path of byungsook/neural-flow-style.
 *
 * The reference has no FFI (it is 100% Python on TensorFlow-1.15 stock ops), so
 * the boundary is the set of module-level operators its Styler graph is built
 * from.  Each entry point below names the reference code it replaces
 * (file:line relative to the reference tree).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add on the reference side.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless it says host;
 *     kernels never allocate; no hidden synchronisation; all work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream);
 *   - all tensors are float32, dense, channels-last, in the reference's layouts
 *     ([B,D,H,W,C] volumes, [B,H,W,C] images, particles [N,3] ordered (z,y,x));
 *   - return value 0 = ok, <0 = error (NFS_E*), message via nfs_last_error()
 *     (thread-local); no exceptions cross the ABI;
 *   - "accumulate" outputs (g_d of the scatter adjoints) are += targets: the
 *     caller zeroes them once per iteration, which is what lets views accumulate.
 */
#ifndef NFS_HIP_H
#define NFS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFS_OK 0
#define NFS_EINVAL (-1)  /* bad argument (null pointer, non-positive size, unsupported shape) */
#define NFS_ELAUNCH (-2) /* HIP launch/runtime error */

typedef void* nfs_stream_t;

int nfs_version(void);
const char* nfs_last_error(void);
/* number of CUs of the current device (used by host-side tile heuristics) */
int nfs_device_cus(void);
/* Measurement aid for bench.py's roofline: while enabled, every launch of the batched f32-MFMA GEMM kernel
 * (winograd_gemm_kernel: the Winograd products of the conv layers and the Gram gradient) is bracketed by a
 * HIP event pair on the stream it is launched on.  nfs_gemm_timer_read synchronises the device, returns the
 * summed kernel time [ms], the summed executed MFMA flops (2*Z*T*K*N per launch) and the launch count, and
 * clears the record.  Disabled by default; no effect on results. */
int nfs_gemm_timer(int enable);

/* Arithmetic of the batched Winograd GEMMs (process-wide; returns the previous mode; any other value only queries):
 *   1  split-limb form (the default): every float32 operand is written exactly as three bf16 limbs (round to nearest
 *      at each level) and the six leading limb products run on v_mfma_f32_16x16x32_bf16 with float32 accumulation --
 *      inputs, outputs and accumulation stay float32, each product is carried to within 2^-26 (float32-equivalent
 *      accuracy: measured against float64 it is no worse than mode 0 at every tested shape and at the 200^3 headline
 *      size, tests/test_ops_gpu.py::test_split_limb_gemm_is_float32_accurate, bench.py parity.full_size), at 2.67x the
 *      matrix-pipe rate of the f32-input MFMA.  A (transformed activations) is split while a block stages it into LDS,
 *      B (the filters) once, by nfs_conv3x3_pack: the packed buffer holds its limb planes (6 bytes per value);
 *   0  float32-input MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2).
 * The Gram gradient (mask / scale / symmetric operand) runs on the f32-input MFMA in both modes.
 * NFS_GEMM_MODE=0 presets 0. */
int nfs_gemm_mode(int mode);
int nfs_gemm_timer_read(double* ms_total, double* flops_total, long long* launches);
/* the same for one kind of launch only (split_limb != 0: the bf16-MFMA split-limb launches; 0: the f32-input ones);
 * records of the other kind stay for a later read.  bytes_total (nullable): the summed algorithmic operand bytes of the
 * split-limb launches, Z (4 T K + 6 K N + 4 T N) each -- V read, the filters' limb planes read, M written */
int nfs_gemm_timer_read_kind(int split_limb, double* ms_total, double* flops_total, long long* launches,
                             double* bytes_total);

/* ---- A2: batch_warp3d / _interpolate3d (transform.py:238-269, 343-433) -------------
 * imgs [B,X,Y,Z,C], coords [B,3,X,Y,Z] normalised [-1,1] (axis order = array order),
 * out [B,X,Y,Z,C].  Border-replicating trilinear gather.  bwd: g_imgs += scatter,
 * g_coords (nullable) overwritten. */
int nfs_warp3d_fwd(const float* imgs, const float* coords, float* out,
                   int B, int X, int Y, int Z, int C, nfs_stream_t stream);
int nfs_warp3d_bwd(const float* imgs, const float* coords, const float* g_out,
                   float* g_imgs_acc, float* g_coords,
                   int B, int X, int Y, int Z, int C, nfs_stream_t stream);

/* ---- A3: rotate (transform.py:611-628) ----------------------------------------------
 * d [D,H,W,C] (one volume, tiled V times by the reference), rot [V,9] row-major 3x3
 * acting on (D,H,W)-ordered normalised coords, out [V,D,H,W,C].  Coordinates are
 * computed in-register (no mgrid tensor).  bwd: g_d [D,H,W,C] += over all views.
 * `workspace` (device, >= 64 bytes, nullable): with C == 1 and a workspace the adjoint runs
 * output-stationary (a block owns a tile of g_d, accumulates in 64-bit fixed point in LDS, no
 * global atomics, bit-reproducible); otherwise it scatters with global float atomics.
 * `g_max` (device, nullable): max |g_out| as produced by nfs_render_bwd(gmax_out) -- the fixed-point scale
 * is derived from it; when NULL a streaming pre-pass over g_out computes it into the workspace.
 * `overwrite` != 0 (tiled adjoint only): g_d = sum over the views instead of +=; the tiles partition the volume, so
 * the caller needs no zero fill and the kernel no read of g_d. */
int nfs_rotate_fwd(const float* d, const float* rot, float* out,
                   int V, int D, int H, int W, int C, nfs_stream_t stream);
int nfs_rotate_bwd(const float* g_out, const float* rot, float* g_d_acc,
                   int V, int D, int H, int W, int C, float* workspace, const float* g_max,
                   int overwrite, nfs_stream_t stream);

/* ---- A11: advect, order 1 (transform.py:557-569) ------------------------------------
 * d [D,H,W,C], vel [D,H,W,3] normalised units (component k moves along array axis k),
 * out [D,H,W,C] = trilinear(d, mgrid - vel).  bwd: g_d_acc (nullable) +=, g_vel
 * (nullable) overwritten. */
int nfs_advect_fwd(const float* d, const float* vel, float* out,
                   int D, int H, int W, int C, nfs_stream_t stream);
int nfs_advect_bwd(const float* d, const float* vel, const float* g_out,
                   float* g_d_acc, float* g_vel,
                   int D, int H, int W, int C, nfs_stream_t stream);

/* advect velocity gradient fused with the TF ApplyAdam update of the velocity (styler_3p.py:320-323 on the
 * gradient of transform.py:557-569): vel, m, v [D,H,W,3] are updated in place, the gradient is never stored.
 * Scalar field only (C = 1), D,H,W >= 2, D*H*W % 4 == 0; same arithmetic as nfs_advect_bwd + nfs_adam_tf_step. */
int nfs_advect_bwd_adam(const float* d, float* vel, const float* g_out, float* m, float* v,
                        int D, int H, int W, float lr_t, float beta1, float beta2, float eps,
                        nfs_stream_t stream);

/* ... and, in the same pass, the NEXT iteration's forward sample adv_next [D,H,W] = nfs_advect_fwd(d, UPDATED vel): advect
 * (transform.py:557-569) reads the velocity of its own voxel only and the density it gathers from is constant, so the
 * sample is formed while the new velocity is still in registers -- the forward advect launch of the next iteration
 * (styler_3p.py:112-125 runs it at the top of every sess.run) and its 96 MB velocity read disappear from the steady-state
 * step.  Bit-identical to calling nfs_advect_fwd on the updated velocity.  adv_next must not alias d or g_out. */
int nfs_advect_bwd_adam_fwd(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next,
                            int D, int H, int W, float lr_t, float beta1, float beta2, float eps,
                            nfs_stream_t stream);

/* Slab forms for the view-sharded (strong-scaling) run, where the replicated field work -- advect, smooth, their
 * adjoints, ApplyAdam -- is sharded over D-slabs between a reduce-scatter of the density-field gradient and an
 * all-gather of the smoothed density (engine.GridStylizer): d is the WHOLE [D,H,W] density (back-traced
 * points leave the slab), vel / out / g_out / m / v hold the nz planes [z0, z0 + nz) only.  Same arithmetic per voxel
 * as nfs_advect_fwd (C = 1) / nfs_advect_bwd_adam; nz*H*W % 4 == 0. */
int nfs_advect_fwd_slab(const float* d, const float* vel, float* out, int D, int H, int W, int z0, int nz,
                        nfs_stream_t stream);
int nfs_advect_bwd_adam_slab(const float* d, float* vel, const float* g_out, float* m, float* v,
                             int D, int H, int W, int z0, int nz, float lr_t, float beta1, float beta2, float eps,
                             nfs_stream_t stream);
/* nfs_advect_bwd_adam_fwd on a slab: adv_next [nz,H,W] = the slab's planes of advect(d, updated vel) */
int nfs_advect_bwd_adam_fwd_slab(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next,
                                 int D, int H, int W, int z0, int nz, float lr_t, float beta1, float beta2,
                                 float eps, nfs_stream_t stream);

/* ---- histogram loss (styler_base.py:187-209 + util.histogram_match_tf, util.py:317-399) -------------
 * feat [B,HW,C] (a layer of the loss network, or d_img for hist_layer 'input'), templ [Bt,HWt,C] (the same layer of
 * the style image; image b uses template min(b, Bt-1)).  Per (image, channel): 255-bin histogram matching of feat to
 * templ over their joint range, matched values = bin centres; loss_acc[b] += weight * sum((feat - matched)^2) and
 * g_acc (nullable) += 2 weight (feat - matched) [only where feat > 0 when relu_mask: gradient wrt the pre-activation
 * of a post-ReLU layer].  `matched` is piecewise constant in feat (no gradient through it, as in the TF graph). */
int nfs_hist_loss(const float* feat, const float* templ, float* loss_acc, float* g_acc,
                  int B, int Bt, int HW, int HWt, int C, float weight, int relu_mask, nfs_stream_t stream);
/* The masked branch (styler_base.py:104-125, 196-201; style_mask = True): mask [B,HW] (nullable = nfs_hist_loss) is the
 * density mask d_gray resized to the layer (nfs_resize_bicubic_tf1); pixels where it is 0 are removed from the source
 * (tf.boolean_mask: out of the value range, the source histogram and the loss), the template stays whole.
 * Both entry points: a channel with nothing to match -- max == min over source and template (the reference's
 * tf.range(min, max, 0) has no defined result there), or every source pixel masked out -- adds loss 0, gradient 0. */
int nfs_hist_loss_masked(const float* feat, const float* templ, const float* mask, float* loss_acc, float* g_acc,
                         int B, int Bt, int HW, int HWt, int C, float weight, int relu_mask, nfs_stream_t stream);
/* The same loss (mask nullable) for images of 1..4 channels -- the default hist layer is the 3-channel loss-net INPUT
 * (config.py:97 hist_layer ['input']): the per-(image, channel) kernels above would run 3 blocks for a whole image, so
 * here the pixels are spread over the chip and the per-(image, channel) state (value range, the two histograms, the
 * lookup table) lives in `workspace` (nfs_hist_loss_wide_workspace_floats(B, C, HW, HWt) floats, 16-byte aligned;
 * -1 for unsupported arguments).  Integer atomics and fixed-order sums only: deterministic; the same bins, table and
 * matched values as nfs_hist_loss_masked. */
int64_t nfs_hist_loss_wide_workspace_floats(int B, int C, int HW, int HWt);
int nfs_hist_loss_wide(const float* feat, const float* templ, const float* mask, float* loss_acc, float* g_acc,
                       float* workspace, int64_t workspace_floats, int B, int Bt, int HW, int HWt, int C, float weight,
                       int relu_mask, nfs_stream_t stream);

/* ---- A7 mask variant (styler_base.py:165-173, style_mask = True) -------------------------------------------------
 * nfs_resize_bicubic_tf1: tf.compat.v1.image.resize(BICUBIC) (legacy kernel: align_corners False, no half-pixel
 *   centres, Keys A = -0.75 through the 1024-entry table), x [B,H,W,C] -> out [B,oh,ow,C]; forward only (the mask
 *   d_gray is a constant of the colour stylizer, styler_2p.py:91-97).
 * nfs_style_mask_apply: Fm = F * mask (mask [B,HW] broadcast over C), scale[b] = 1 / (2 * sum(mask[b]) * C) -- the
 *   Gram denominator 2 * area * C of the masked branch (feed scale as nfs_gram_fwd's per-image scale).
 * nfs_style_mask_bwd: dF = dFm * mask * (F > 0). */
int nfs_resize_bicubic_tf1(const float* x, float* out, int B, int H, int W, int C, int oh, int ow,
                           nfs_stream_t stream);
int nfs_style_mask_apply(const float* F, const float* mask, float* Fm, float* scale, int B, int HW, int C,
                         nfs_stream_t stream);
int nfs_style_mask_bwd(const float* dFm, const float* mask, const float* F, float* dF, int B, int HW, int C,
                       nfs_stream_t stream);

/* ---- A2 (2-D twin): batch_warp2d / _interpolate2d (transform.py:206-236, 280-341) ---------------------------
 * imgs [B,X,Y,C], coords [B,2,X,Y] normalised [-1,1], out [B,X,Y,C]; border-replicating bilinear gather (the
 * reference's only known-answer vector, transform.py:1859-1885, pins this stencil).  bwd: g_imgs_acc (nullable) +=
 * scatter, g_coords (nullable) overwritten. */
int nfs_warp2d_fwd(const float* imgs, const float* coords, float* out, int B, int X, int Y, int C,
                   nfs_stream_t stream);
int nfs_warp2d_bwd(const float* imgs, const float* coords, const float* g_out, float* g_imgs_acc,
                   float* g_coords, int B, int X, int Y, int C, nfs_stream_t stream);

/* ---- A11 (2-D branch): advect order 1 (transform.py:583-588) ---------------------------------------------------
 * d [H,W,C], vel [H,W,2] normalised units (component k moves along array axis k), out = bilinear(d, mgrid - vel). */
int nfs_advect2d_fwd(const float* d, const float* vel, float* out, int H, int W, int C, nfs_stream_t stream);
int nfs_advect2d_bwd(const float* d, const float* vel, const float* g_out, float* g_d_acc, float* g_vel,
                     int H, int W, int C, nfs_stream_t stream);

/* ---- advect order 2, MacCormack (transform.py:570-582 3-D, 590-607 2-D) ---------------------------
 * Second pass of the scheme on top of d_fwd = advect(d, vel) (nfs_advect_fwd / nfs_advect2d_fwd):
 *   d_bwd = advect(d_fwd, -vel);  d_adv = d_fwd + (d - d_bwd)/2;  where d_adv leaves the range of d over the corners
 *   of the back-traced interpolation cell, d_fwd is kept (the reference's "soft clamp").  The reference's own limiter
 *   is broken (tf.to_int32 of [-1,1] coordinates; d_max[grids]); this is the scheme it transcribes, done as intended.
 * d, d_fwd, out [D,H,W,C] (2-D: D = 1), vel [D,H,W,nd], nd = 2 | 3.  Forward only (as its uses: transport of results). */
int nfs_advect_maccormack(const float* d, const float* vel, const float* d_fwd, float* out,
                          int D, int H, int W, int C, int nd, nfs_stream_t stream);

/* ---- curl of a stream function (transform.py:517-555) -------------------------------------------
 * nd = 2: s [H,W] -> out [H,W,2] = (ds/dy, -ds/dx);  nd = 3: s [D,H,W,3] -> out [D,H,W,3] (forward differences, last
 * slice replicated).  bwd: g_s = curl^T g_out (gather form, deterministic).  2-D: D = 1. */
int nfs_curl_fwd(const float* s, float* out, int D, int H, int W, int nd, nfs_stream_t stream);
int nfs_curl_bwd(const float* g_out, float* g_s, int D, int H, int W, int nd, nfs_stream_t stream);

/* ---- Laplacian-pyramid gradient normalisation (util.py:57-110) -------------------------------------
 * nfs_lap_down: out [ceil(D/2),ceil(H/2),ceil(W/2),C] = conv(x [D,H,W,C], k, stride 2, 'SAME') (tf.nn.conv3d / conv2d of
 *   lap_split, 60-66); k = k5x5x5 [5][5][5] (nd = 3) or k5x5 [5][5] (nd = 2, D = 1), the same for every channel.
 * nfs_lap_up: out [D,H,W,C] = scale * conv_transpose(lo, k, output shape [D,H,W], stride 2) + addend (nullable):
 *   lo2 of lap_split (scale 5 in 3-D, 4 in 2-D, hi = img - lo2 via scale < 0) and the merge step (80-83).
 * nfs_normalize_mean: out = x / max(m, eps), m = sqrt(mean(x^2)) (normalize_std, 86-90) or mean|x| (use_abs: the
 *   scale_n = 0 branch, 97-99); workspace >= 64 floats; deterministic (fixed partial-sum order). */
int nfs_lap_down(const float* x, const float* k, float* out, int D, int H, int W, int C, int nd, nfs_stream_t stream);
int nfs_lap_up(const float* lo, const float* k, float scale, const float* addend, float* out,
               int D, int H, int W, int C, int nd, nfs_stream_t stream);
/* nfs_lap_up with the RMS normalisation of pyramid levels (normalize_std, 86-90) riding in it -- a level of the 200^3 x 3
 *   pyramid is 96 MB, and a normalisation pass of its own reads it twice and writes it once:
 *   out_part (nullable) [nfs_lap_up_rms_parts()]: per-block sums of out^2, written in the same pass (the level's RMS);
 *   addend_part (nullable) [addend_nparts] + addend_n + eps: the addend enters as addend / max(sqrt(sum / addend_n), eps),
 *   the sum formed by every block from the partial sums in a fixed order (deterministic).
 * nfs_lap_up_rms_parts: the number of partial sums (blocks) for an output volume, 0 where only nfs_lap_up applies
 *   (2-D, other channel counts, volumes under 2^21 cells). */
int nfs_lap_up_rms_parts(int D, int H, int W, int C, int nd);
int nfs_lap_up_rms(const float* lo, const float* k, float scale, const float* addend, const float* addend_part,
                   int addend_nparts, int64_t addend_n, float eps, float* out, float* out_part, int D, int H, int W, int C,
                   int nd, nfs_stream_t stream);
int nfs_normalize_mean(const float* x, float* out, int64_t n, int use_abs, float eps, float* workspace, int ws_floats,
                       nfs_stream_t stream);

/* ---- A11': one step of StylerBase._transport (styler_base.py:59-89: g <- advect(g, +-v[i]) per frame crossed, or
 * advect(g, +-v[a]*|b-a|) in its one-step form) for a C-channel grid field, with the weighted accumulation of the
 * temporal filter (styler_3p.py:380-386 applied to grid fields, see styler_grid.py) fused in:
 *   out [D,H,W,C] = w_g * advect(g, scale * u) + w_addend * addend   (addend nullable; out must not alias g)
 * g [D,H,W,C], u [D,H,W,3] in advect units, same border-replicating trilinear stencil as nfs_advect_fwd. */
int nfs_transport_step(const float* g, const float* u, float scale, float w_g, const float* addend,
                       float w_addend, float* out, int D, int H, int W, int C, nfs_stream_t stream);

/* ---- A9: smoothing conv + clamp (styler_3p.py:112-125) ------------------------------
 * out = max(conv3d_SAME(d, [1,k,1]^3/(k+2)^3), 0), d/out [D,H,W].  k<=0 skips the conv.
 * out stores -0.0f where the pre-activation was negative, so the TF Maximum gradient
 * mask (pre >= 0) is recoverable from the sign bit; numerically -0.0f == 0.0f.
 * bwd: g_d = conv(g_out * (pre >= 0)), overwritten. */
int nfs_smooth3d_relu_fwd(const float* d, float* out, int D, int H, int W, float k,
                          nfs_stream_t stream);
int nfs_smooth3d_relu_bwd(const float* out, const float* g_out, float* g_d,
                          int D, int H, int W, float k, nfs_stream_t stream);

/* ---- A4: render block (styler_3p.py:147-158) -----------------------------------------
 * d [V,D,H,W] (C=1), img [V,H,W]: smoke (liquid=0) I = sum_z d[z]*exp(-tau*sum_{z'>=z} d[z'])
 * (un-normalised; the global-max division is nfs_maxnorm_*), liquid=1: 1-exp(-tau*sum_z d).
 * raysum [V,H,W] = sum_z d is saved for the adjoint.  bwd overwrites g_d [V,D,H,W] (may alias d);
 * gmax_out (device, nullable, 4 bytes) receives max |g_d| for nfs_rotate_bwd's fixed-point scale.
 * `liquid` doubles as the ray mode: 2 = reduce_max along the ray (the line the reference keeps commented out,
 * styler_3p.py:149; raysum then carries the maximum, its gradient is split equally among ties like TF's), 3 =
 * reduce_mean (raysum = the sum); tau is ignored for both.  The fused rotate + render entry points take 0 / 1 only. */
int nfs_render_fwd(const float* d, float* img, float* raysum,
                   int V, int D, int H, int W, float tau, int liquid, nfs_stream_t stream);
int nfs_render_bwd(const float* d, const float* raysum, const float* g_img, float* g_d,
                   int V, int D, int H, int W, float tau, int liquid, float* gmax_out,
                   nfs_stream_t stream);

/* fused A3+A4: d [D,H,W], rot [V,9] -> img/raysum [V,H,W] in one pass over the rays.  d_rot
 * (nullable, [V,D,H,W]) receives the rotated samples so that the adjoint can run as
 * nfs_render_bwd(d_rot) + nfs_rotate_bwd (LDS-tiled, no global atomics); with d_rot = NULL
 * nothing of size [V,D,H,W] is materialised and the adjoint is nfs_rotate_render_bwd
 * (g_d_acc [D,H,W] += over all views and samples with global float atomics). */
int nfs_rotate_render_fwd(const float* d, const float* rot, float* img, float* raysum, float* d_rot,
                          int V, int D, int H, int W, float tau, int liquid,
                          nfs_stream_t stream);
int nfs_rotate_render_bwd(const float* d, const float* rot, const float* raysum,
                          const float* g_img, float* g_d_acc,
                          int V, int D, int H, int W, float tau, int liquid,
                          nfs_stream_t stream);

/* The same adjoint (transform.py:611-628 + styler_3p.py:147-158, transmittance mode) without a pass over the rotated
 * volume: the image gradient of a sample is affine in a per-sample quantity the forward march already holds,
 *     dI/ds_z = T_z - tau * sum_{z' <= z} s_z' T_z' = E u_z - tau (I - F),
 * u_z = t_z + tau * i_z (t_z: the transmittance factor inside the sample's depth segment, i_z: the segment's image sum
 * before the sample), E / F per (ray, depth segment).  nfs_rotate_render_fwd_coef writes u [V,D,H,W] instead of the
 * samples and seg [3][V][nseg][H][W] (segment ray sum, segment image sum, max |u|); nfs_render_ray_coef turns seg and
 * the image gradient g_img [V,H,W] into ab [V][nseg][H][W][2] = (A, B) with sample gradient = A u - B, and writes
 * bounds [nfs_render_ray_coef_bounds(V,H,W)]: every launch block's bound on max |sample gradient| (their maximum sets the
 * adjoint's fixed-point scale; no zero-initialised scalar, no atomics); nfs_rotate_bwd_coef is nfs_rotate_bwd (tiled
 * form, overwrite as there) reading u, ab and the bounds.  nfs_render_coef_layout: NFS_OK and (nseg, seg_len) when the shape is supported (D >= 16,
 * H, W >= 2, V*D*H*W < 2^30), else NFS_EINVAL -- the caller then takes nfs_rotate_render_fwd(d_rot) + nfs_render_bwd +
 * nfs_rotate_bwd.  Results agree with that path to float32 rounding. */
int nfs_render_coef_layout(int V, int D, int H, int W, int* nseg, int* seg_len);
int nfs_rotate_render_fwd_coef(const float* d, const float* rot, float* img, float* raysum, float* u_rot, float* seg,
                               int V, int D, int H, int W, float tau, nfs_stream_t stream);
int nfs_render_ray_coef_bounds(int V, int H, int W);
int nfs_render_ray_coef(const float* g_img, const float* seg, float* ab, float* bounds,
                        int V, int H, int W, float tau, nfs_stream_t stream);
int nfs_rotate_bwd_coef(const float* u_rot, const float* ab, const float* rot, float* g_d_acc,
                        int V, int D, int H, int W, int nseg, int seg_len, const float* bounds, int nbounds,
                        int overwrite, nfs_stream_t stream);

/* Dead-region skipping for the velocity variable (the adjoint of transform.py:557-569 is g(x) * grad d(x - v), an exact
 * zero wherever the eight density corners the back-traced point interpolates are equal -- empty space and plateaus --
 * WHATEVER g(x) is; so the chain above it need not produce g there).
 *   live [nfs_live_mask_words(D,H,W)] 64-bit words, bit (z H + y) W + x = "the corners of voxel (z,y,x)'s back-traced
 *   stencil differ".  nfs_advect_fwd_live = nfs_advect_fwd (C = 1; D,H,W >= 2, D*H*W % 4 == 0, else NFS_EINVAL) + the
 *   mask; nfs_advect_bwd_adam_fwd_live = nfs_advect_bwd_adam_fwd + the mask of the UPDATED velocity (the one the next
 *   iteration's adjoint needs).
 *   nfs_rotate_bwd_coef_live = nfs_rotate_bwd_coef restricted to the voxels within `dilate` cells of a live voxel
 *   (`dilate` = reach of the linear stencil between g_d and the advect adjoint: 1 for the 3x3x3 smoothing of
 *   styler_3p.py:112-125, 0 without it; RT tile + 2 dilate <= 63): a tile without such voxels returns before its sample
 *   loop, the others accumulate only the bounding box of theirs, longest first (two small launches ahead of the adjoint
 *   find the boxes and sort the tiles by work into `workspace`, nfs_rotate_live_workspace_ints ints, no initialisation
 *   needed; one workspace per stream that runs this concurrently).  g_d there is bit-identical to nfs_rotate_bwd_coef;
 *   elsewhere it holds zeros, which only ever meet the zero factor.  The resulting velocity gradient / Adam update is
 *   bit-identical with and without the mask. */
int nfs_live_mask_words(int D, int H, int W);
int nfs_advect_fwd_live(const float* d, const float* vel, float* out, unsigned long long* live,
                        int D, int H, int W, nfs_stream_t stream);
int nfs_advect_bwd_adam_fwd_live(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next,
                                 unsigned long long* live_next, int D, int H, int W, float lr_t, float beta1,
                                 float beta2, float eps, nfs_stream_t stream);
/* ... and without touching the voxels that have NEVER been live: `ever` [nfs_live_mask_words] = OR of every mask since the
 * Adam moments were zeroed (zero it with them; do not use it once anything else has written m or v).  Where a voxel's bit
 * is clear in `ever` and in the current mask (`live` on entry), m = v = +0 and this gradient is +-0: ApplyAdam is an
 * exact no-op there and the next sample and mask bit are what they were, so a wave of 256 such voxels returns before its
 * first load.  `live`: the current mask on entry, the next one on return.  Bit-identical to the call above. */
int nfs_advect_bwd_adam_fwd_live_ever(const float* d, float* vel, const float* g_out, float* m, float* v, float* adv_next,
                                      unsigned long long* live, unsigned long long* ever, int D, int H, int W,
                                      float lr_t, float beta1, float beta2, float eps, nfs_stream_t stream);
int nfs_rotate_bwd_coef_live(const float* u_rot, const float* ab, const float* rot, float* g_d_acc,
                             int V, int D, int H, int W, int nseg, int seg_len, const float* bounds, int nbounds,
                             int overwrite, const unsigned long long* live, int dilate, int* workspace,
                             nfs_stream_t stream);
int nfs_rotate_live_workspace_ints(int D, int H, int W);

/* d /= reduce_max(d) (styler_3p.py:158): G groups of n contiguous floats, one max per
 * group (v_batch views form one group; v_batch=1 => per view).  gmax [G] is written by
 * fwd and read by bwd; the max gradient is split equally among ties like TF's.  bwd `workspace`
 * (device, >= 64*G floats, nullable): groups of >= 16384 elements are reduced by 32 blocks each with the
 * partial sums combined in a fixed order; without it one block per group does everything. */
int nfs_maxnorm_fwd(const float* img, float* out, float* gmax, int G, int n,
                    nfs_stream_t stream);
int nfs_maxnorm_bwd(const float* img, const float* gmax, const float* g_out, float* g_img,
                    int G, int n, float* workspace, nfs_stream_t stream);
/* The same normalisation fused with the loss-net input (styler_base.py:41-45 + vgg.py:50-53) for the grey render at the loss
 * net's own size (resize_scale 1): x [.,3] = (img / max) * 255 - mean[c]; adjoint g_img = adjoint_maxnorm(255 * (g_x[.,0] +
 * g_x[.,1] + g_x[.,2])).  To float32 rounding the arithmetic of nfs_maxnorm_fwd + nfs_loss_net_input_fwd and of
 * nfs_loss_net_input_bwd + nfs_maxnorm_bwd, without the [V,H,W] intermediates and two launches less per direction.
 * img [G*n], x / g_x [G*n,3], gmax [G] (written by fwd), workspace >= 64*G floats. */
int nfs_maxnorm_input_fwd(const float* img, float* x, float* gmax, int G, int n, nfs_stream_t stream);
int nfs_maxnorm_input_bwd(const float* img, const float* gmax, const float* g_x, float* g_img, int G, int n,
                          float* workspace, nfs_stream_t stream);


/* ---- A5: _plugin_to_loss_net + vgg.preprocess (styler_base.py:33-45, vgg.py:50-53) ---
 * img [B,H,W,Cin] in [0,1] (Cin=1 grey or 3 colour) -> d_img [B,H2,W2,3] in 0..255
 * (optional TF1 legacy bilinear resize, *255, grey->3ch) and x = d_img - mean.
 * Either output may be NULL.  bwd takes g wrt x (== g wrt d_img) and overwrites g_img. */
int nfs_loss_net_input_fwd(const float* img, float* d_img, float* x,
                           int B, int H, int W, int Cin, int H2, int W2, nfs_stream_t stream);
int nfs_loss_net_input_bwd(const float* g_x, float* g_img,
                           int B, int H, int W, int Cin, int H2, int W2, nfs_stream_t stream);

/* ---- A6: VGG-19 conv / pool (vgg.py:44-48, 89-108) ------------------------------------
 * 3x3 SAME stride-1 conv on NHWC in float32 on the f32 MFMA.  Layers with >= 64 channels on both
 * sides run as Winograd F(4x4,3x3) (input transform -> 36 batched GEMMs -> output transform, all
 * f32; differs from the direct form by f32 rounding, ~3e-6 relative L2 per layer), conv1_1 (3
 * input channels) and shapes the Winograd path does not take run as a direct implicit GEMM.
 * Weights are frozen: pack them once.  kind 0 = forward (HWIO [3,3,Ci,Co] -> packed, GEMM N=Co,
 * K=Ci per tap), kind 1 = data-gradient (taps flipped, N=Ci, K=Co).  The packed buffer holds the
 * direct packing (9*Ci*Co floats) followed, for Winograd-eligible layers, by the transformed
 * weights G g G^T (36*Ci*Co floats); nfs_conv3x3_packed_floats gives the total in floats. */
int64_t nfs_conv3x3_packed_floats(int Ci, int Co, int kind);
int nfs_conv3x3_pack(const float* w_hwio, float* packed, int Ci, int Co, int kind,
                     nfs_stream_t stream);
/* Workspace (device floats): the Winograd path keeps the transformed activations V [36][T][K] and
 * products M [36][T][N] there (T = B*ceil(H/4)*ceil(W/4) tiles); the direct path uses it for split-K
 * partial sums when a layer gives too few M x N tiles to fill 256 CUs.  Without a workspace (NULL)
 * or with one smaller than nfs_conv3x3_workspace_floats the layer runs direct and unsplit. */
int64_t nfs_conv3x3_workspace_floats(int B, int H, int W, int Ci, int Co);
/* ReLU bit cache of a layer (optional, Winograd F(4x4) path only).  The forward pass can record the two masks
 * its data gradient needs -- (x > 0) of its input and, for a pooled layer, (y > 0) of its output -- as one bit
 * per element (a uint32 word = one 4x4 tile x one channel pair), so that the backward pass never re-reads the
 * activations for them: pass the same `relu_bits` buffer of nfs_conv3x3_relu_bits_words(...) words to the
 * layer's fwd / fwd_pool and later to its dgrad / dgrad_pool.  Layout [in: T*Ci/2][out (pooled): T*Co/2],
 * T = B*ceil(H/4)*ceil(W/4).  The query returns 0 for a layer that does not keep the cache (pass NULL then;
 * x_in / x_out are still required arguments and are what a NULL cache falls back to). */
int64_t nfs_conv3x3_relu_bits_words(int B, int H, int W, int Ci, int Co, int pooled);
/* MFMA flops the conv call for K input / N output channels of its kernel EXECUTES (forward: K = Ci, N = Co; data
 * gradient: K = Co, N = Ci; pooled != 0: the fused-pool forms): 2*36*T4*K*N for F(4x4,3x3), 2*49*T5*K*N for F(5x5,3x3),
 * the direct 2*B*H*W*9*K*N otherwise.  The path is a function of the shapes alone, so this is too (measurement aid). */
double nfs_conv3x3_executed_flops(int B, int H, int W, int K, int N, int pooled);
/* y = relu?(conv(x) + bias); x [B,H,W,Ci], y [B,H,W,Co]; bias nullable; relu_bits nullable (written) */
int nfs_conv3x3_fwd(const float* x, const float* packed_fwd, const float* bias, float* y,
                    int B, int H, int W, int Ci, int Co, int relu,
                    float* workspace, int64_t workspace_floats, uint32_t* relu_bits,
                    nfs_stream_t stream);
/* gx = dgrad(gy) * (x_in > 0 if x_in) + (addend if addend); gy [B,H,W,Co] is the gradient
 * wrt the conv's pre-activation, gx [B,H,W,Ci]; relu_bits nullable (read instead of x_in).
 * addend_unmasked != 0 (F(4x4) Winograd path with x_in only): the addend is a gradient wrt the OUTPUT of the
 * layer below that has not been through that layer's ReLU mask yet: gx = (dgrad(gy) + addend) * (x_in > 0). */
int nfs_conv3x3_dgrad(const float* gy, const float* packed_dgrad, const float* x_in,
                      const float* addend, float* gx,
                      int B, int H, int W, int Ci, int Co,
                      float* workspace, int64_t workspace_floats, const uint32_t* relu_bits,
                      int addend_unmasked, nfs_stream_t stream);
/* Fused forms for a conv that is followed by the 2x2 average pool (conv1_2, conv2_2, conv3_4, conv4_4):
 * fwd_pool also writes y_pool [B,H/2,W/2,Co] = avg_pool2d(y); dgrad_pool takes the gradient at the POOLED
 * resolution gy_pool [B,H/2,W/2,Co] plus the conv's own output x_out [B,H,W,Co] and forms
 * 0.25*gy_pool[h/2,w/2]*(x_out>0) on the fly (x_in / addend as in nfs_conv3x3_dgrad).  On the Winograd path
 * both are folded into the transforms (no separate pool kernels, no full-resolution round trip); otherwise
 * they run the separate kernels (dgrad_pool then needs >= B*H*W*Co workspace floats).  With a ReLU bit cache on the
 * fused path the full-resolution tensor is optional on both sides: fwd_pool accepts y = NULL (only the pool and the
 * cache are written), dgrad_pool accepts x_out = NULL. */
int nfs_conv3x3_fwd_pool(const float* x, const float* packed_fwd, const float* bias, float* y,
                         float* y_pool, int B, int H, int W, int Ci, int Co, int relu,
                         float* workspace, int64_t workspace_floats, uint32_t* relu_bits,
                         nfs_stream_t stream);
int nfs_conv3x3_dgrad_pool(const float* gy_pool, const float* x_out, const float* packed_dgrad,
                           const float* x_in, const float* addend, float* gx,
                           int B, int H, int W, int Ci, int Co,
                           float* workspace, int64_t workspace_floats, const uint32_t* relu_bits,
                           int addend_unmasked, nfs_stream_t stream);
/* slim.avg_pool2d [2,2]: stride 2, VALID (odd sizes floor).  x [B,H,W,C] -> y [B,H/2,W/2,C].
 * bwd: gx = 0.25*gy[h/2,w/2] (0 outside the pooled area) * (x > 0 if x) + (addend if addend) */
int nfs_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, nfs_stream_t stream);
int nfs_avgpool2_bwd(const float* gy, const float* x, const float* addend, float* gx,
                     int B, int H, int W, int C, nfs_stream_t stream);

/* ---- A7: Gram matrix + style loss (styler_base.py:96-102, 152-185) --------------------
 * F [B,HW,C] -> G [B,C,C] = F^T F * scale[b] (f32 MFMA, split over pixel slabs).  With a
 * workspace (>= nfs_gram_workspace_floats) the slab partials are summed by a second pass in a
 * fixed order (deterministic, G overwritten); without one they are accumulated with float
 * atomics and G must be zeroed by the caller.  The factor applied to image b is
 * scale * (scale_dev ? scale_dev[b] : 1): `scale` = 1/(2*HW*C) in the plain case; scale_dev is a
 * DEVICE array [B] for the style_mask variant whose denominator 2*area_b*C lives on the GPU. */
int64_t nfs_gram_workspace_floats(int B, int HW, int C);
int nfs_gram_fwd(const float* F, float* G, int B, int HW, int C, const float* scale_dev, float scale,
                 float* workspace, int64_t workspace_floats, nfs_stream_t stream);
/* loss_acc[b] += weight * sum((G[b]-Gs[bs])^2) where bs = b % Bs; Dmat [B,C,C] = 2*weight*(G-Gs)
 * (the symmetric matrix nfs_gram_bwd multiplies by). */
int nfs_style_loss_fwd(const float* G, const float* Gs, float* loss_acc, float* Dmat,
                       int B, int Bs, int C, float weight, nfs_stream_t stream);
/* dF [B,HW,C] = scale[b] * 2 * F @ Dmat[b], optionally masked by (F > 0) (F is a post-ReLU
 * activation: this folds the ReLU gradient of the style layer). */
int nfs_gram_bwd(const float* F, const float* Dmat, float* dF, int B, int HW, int C,
                 const float* scale_dev, float scale, int relu_mask, nfs_stream_t stream);

/* ---- A7, all style layers of a step at once (the loop of styler_base.py:152-185 as three launches) ---------------
 * One descriptor per style layer; every layer has the same batch B (the local views).
 *   nfs_gram_style_group_fwd: per layer G = F^T F * scale (written only when G != NULL), Dmat = 2 weight (G - Gs[b % Bs]),
 *     and the layer's share of the style loss as PARTIAL SUMS: loss_parts is [P][B] floats, P =
 *     nfs_gram_style_group_parts(...); every entry is written (plain stores, no atomics, fixed order: deterministic) and
 *     the style loss of image b is sum_p loss_parts[p][b].  Launches: the tile pairs of every layer (big units first) +
 *     one slab reduction; workspace >= nfs_gram_style_group_workspace_floats(...) floats.
 *   nfs_gram_group_bwd: per layer dF = 2 scale F @ Dmat, masked by (F > 0) where relu_mask != 0 -- the five GEMMs of
 *     nfs_gram_bwd as ONE launch of the batched f32-MFMA GEMM (tile list over all layers, deep K first). */
typedef struct {
  const float* F;     /* [B,HW,C] post-ReLU activation of the layer */
  const float* Gs;    /* [Bs,C,C] style Gram (scaled like G) */
  float* G;           /* [B,C,C] out, nullable */
  float* Dmat;        /* [B,C,C] out (fwd) / in (bwd) */
  float* dF;          /* [B,HW,C] out of nfs_gram_group_bwd (unused by fwd) */
  int B, Bs, HW, C;
  float scale;        /* 1 / (2 HW C), styler_base.py:157,176 */
  float weight;       /* w_style_layer * w_style */
  int relu_mask;      /* bwd: fold the ReLU gradient of the style layer */
} nfs_gram_layer_t;
int64_t nfs_gram_style_group_workspace_floats(const nfs_gram_layer_t* layers, int n);
int nfs_gram_style_group_parts(const nfs_gram_layer_t* layers, int n);
int nfs_gram_style_group_fwd(const nfs_gram_layer_t* layers, int n, float* loss_parts, float* workspace,
                             int64_t workspace_floats, nfs_stream_t stream);
int nfs_gram_group_bwd(const nfs_gram_layer_t* layers, int n, nfs_stream_t stream);

/* ---- content loss on a layer of the loss network (styler_base.py:135-150) ---------------
 * F [B,HW,C] is a post-ReLU activation.  loss_acc[b] += image b's share of weight * L, g_acc [B,HW,C] +=
 * dL/d(pre-activation) = weight * dL/dF * (F > 0), with L (means over the whole batch, as reduce_mean):
 *   mode 0: -mean(F[...,channel]) + mean|F[...,:channel]| + mean|F[...,channel+1:]|   (0 < channel < C;
 *           an empty upper slice contributes nothing, where TF's reduce_mean of an empty tensor is NaN)
 *   mode 1: -mean(F)                                   (content_channel == 0 in the reference)
 *   mode 2: mean((F - amp * target[b % Bt])^2)         (content image; target [Bt,HW,C], amp = w_content_amp) */
int nfs_content_loss(const float* F, const float* target, float* loss_acc, float* g_acc, int B, int Bt,
                     int HW, int C, int channel, int mode, float weight, float amp, nfs_stream_t stream);
/* the same term on a tensor that is NOT a ReLU output (the Inception graph's '*_pre_relu' content layers of
 * run.bat:14-20): |F| with d|F| = sign(F), and the gradient is wrt F itself (no ReLU mask) */
int nfs_content_loss_signed(const float* F, const float* target, float* loss_acc, float* g_acc, int B, int Bt,
                            int HW, int C, int channel, int mode, float weight, float amp, nfs_stream_t stream);

/* ---- the Inception-v1 loss network (styler_base.py:17-23, 51-57, 91-94) ------------
 * The reference imports ``tensorflow_inception_graph.pb`` with tf.import_graph_def and reads feature tensors by
 * node name ('conv2d2', 'mixed3b', 'mixed4b_pool_reduce_pre_relu' ...: test_smokegun.py:141, run.bat:14-20).  These
 * are the node types of that graph between its input and its feature tensors; neural-flow-style_amd/inception.py
 * assembles them under the graph's own node names.  All tensors NHWC float32, TF SAME padding
 * (out = ceil(in / stride), pad_before = max((out - 1) stride + k - in, 0) / 2).  "ld*" = floats per pixel of the
 * buffer a pointer points INTO: operands may be channel ranges of wider rows, so that the branches of an inception
 * module write straight into the module's concatenated output (ConcatV2 never runs).
 *
 * nfs_conv2d_pack: HWIO filters [kh,kw,Ci,Co] -> the layout nfs_conv2d_fwd reads (opaque,
 *   nfs_conv2d_packed_floats floats).  transpose = 1 packs the DATA-GRADIENT filters of a stride-1 convolution (taps
 *   flipped, channels swapped): its data gradient is nfs_conv2d_fwd(gy, ..., Cin = Co, Cout = Ci).
 * nfs_conv2d_fwd: Conv2D (+ BiasAdd, + Relu).  y[b,oy,ox,n] (+)= relu(bias[n] + sum x'[b,oy s+dy-pt,ox s+dx-pl,c] w),
 *   x' = x * (x_mask > 0) when x_mask is given (the ReLU adjoint applied while a gradient is consumed; x_mask has the
 *   indexing of x with row stride ldm).  y_pre (nullable) receives the value before the ReLU (the graph's
 *   '*_pre_relu' tensors).  accumulate: y += (branch gradients meeting at a module input).  Cin <= 4 (the image) or
 *   a multiple of 4; Cout, ld* multiples of 4; filters up to 7x7; stride 1 or 2.  workspace (nullable):
 *   nfs_conv2d_workspace_floats floats of scratch; with it, calls that would leave most CUs idle (a few hundred
 *   pixels, K in the thousands) split K over blocks and finish with a fixed-order reduction (deterministic).
 * nfs_conv2d_dgrad_small: data gradient of a convolution with <= 4 input channels (the 7x7 stride-2 first layer, down
 *   to the image): gx [B,H,W,Ci] = sum gy (y_act > 0) w, w_hwio unpacked; y_act nullable.
 * nfs_maxpool3_fwd/bwd: MaxPool 3x3, stride 1 or 2, SAME (padding taps do not take part).  arg [B,Ho,Wo,C] bytes =
 *   window position (0..8 row-major) of the FIRST maximum (TF's CPU kernel; ties only matter at exact equality, and
 *   zero ties after a ReLU carry no gradient past that ReLU).  C = floats per pixel (multiple of 4).  relu_of
 *   (nullable, shaped like gx): the result is multiplied by (relu_of > 0) -- the pooled tensor's own ReLU adjoint.
 * nfs_lrn_fwd/bwd: tf.nn.lrn, y = x / (bias + alpha sum_{|j-c| <= radius} x_j^2)^beta; scale = the bracket, kept
 *   for the adjoint.  C channels in rows of ld floats.
 * nfs_relu_mask_add: out = g (act > 0) + addend (each of g / act / addend nullable): a gradient injected at a
 *   '*_pre_relu' tensor joins the chain after the ReLU adjoint. */
int64_t nfs_conv2d_packed_floats(int kh, int kw, int Ci, int Co, int transpose);
int nfs_conv2d_pack(const float* w_hwio, float* packed, int kh, int kw, int Ci, int Co, int transpose,
                    nfs_stream_t stream);
int64_t nfs_conv2d_workspace_floats(int B, int H, int W, int Cin, int Cout, int kh, int kw, int stride);
int nfs_conv2d_fwd(const float* x, int ldx, const float* x_mask, int ldm, const float* packed, const float* bias,
                   float* y, int ldy, float* y_pre, int ldp, int B, int H, int W, int Cin, int Cout, int kh, int kw,
                   int stride, int relu, int accumulate, float* workspace, int64_t workspace_floats,
                   nfs_stream_t stream);
/* Several stride-1 convolutions of one batch in ONE launch (+ one reduction launch): the branches of an inception
 * module are a few hundred pixels each and would leave most of the chip idle one by one.  Fields as the arguments of
 * nfs_conv2d_fwd.  sum_with_prev: the result is added to the previous problem's (same pixels, Cout and y) -- the data
 * gradients of a module's three 1x1 branches meet in one tensor; they leave as partial sums and one fixed-order
 * reduction writes y (no race, deterministic).  K is split over blocks until the launch fills the chip.  n <= 6. */
typedef struct {
  const float* x;
  const float* x_mask;
  const float* packed;
  const float* bias;
  float* y;
  float* y_pre;
  int ldx, ldm, ldy, ldp;
  int H, W, Cin, Cout, kh, kw;
  int relu, accumulate, sum_with_prev;
} nfs_conv2d_desc_t;
int64_t nfs_conv2d_group_workspace_floats(const nfs_conv2d_desc_t* descs, int n, int B);
int nfs_conv2d_group(const nfs_conv2d_desc_t* descs, int n, int B, float* workspace, int64_t workspace_floats,
                     nfs_stream_t stream);
int nfs_conv2d_dgrad_small(const float* gy, int ldg, const float* y_act, int lda, const float* w_hwio, float* gx,
                           int B, int H, int W, int Ci, int Co, int kh, int kw, int stride, nfs_stream_t stream);
int nfs_maxpool3_fwd(const float* x, float* y, uint8_t* arg, int B, int H, int W, int C, int stride,
                     nfs_stream_t stream);
int nfs_maxpool3_bwd(const float* gy, const uint8_t* arg, float* gx, int B, int H, int W, int C, int stride,
                     int accumulate, const float* relu_of, nfs_stream_t stream);
int nfs_lrn_fwd(const float* x, float* y, float* scale, int64_t npix, int C, int ld, int radius, float bias,
                float alpha, float beta, nfs_stream_t stream);
int nfs_lrn_bwd(const float* x, const float* y, const float* scale, const float* gy, float* gx, int64_t npix, int C,
                int ld, int radius, float alpha, float beta, int accumulate, nfs_stream_t stream);
/* AvgPool k x k, stride 1, VALID on rows of C floats (the graph's avgpool0, 7 x 7, in front of the classifier whose
 * logits 'softmax2_pre_activation' the reference's top_k content target reads: styler_base.py:240-245); the fully
 * connected layer behind it is nfs_conv2d_fwd with a 1 x 1 filter */
int nfs_avgpool_valid_fwd(const float* x, float* y, int B, int H, int W, int C, int k, nfs_stream_t stream);
int nfs_avgpool_valid_bwd(const float* gy, float* gx, int B, int H, int W, int C, int k, int accumulate,
                          nfs_stream_t stream);
int nfs_relu_mask_add(const float* g, int ldg, const float* act, int lda, const float* addend, int ldadd, float* out,
                      int ldo, int64_t npix, int C, nfs_stream_t stream);

/* ---- A12: TV loss (styler_base.py:211-213) --------------------------------------------
 * loss_acc[0] += weight * mean_b(sum|dh| + sum|dw|) on d_img [B,H,W,3]; g_acc (nullable) +=. */
int nfs_tv_loss(const float* d_img, float* loss_acc, float* g_acc, int B, int H, int W, int C,
                float weight, nfs_stream_t stream);

/* ---- A8: SPH splat (transform.py:1233-1267, 1310-1453, 1577-1704) ---------------------
 * p [N,nd] in [0,1] ordered (z,y,x)/(y,x); grid [res...,C] H-flipped like the reference.
 * mode 0: density, grid += mass*W                      (p2g, pc=None)
 * mode 1: colour,  grid += mass*W*attr/pd              (p2g, pc=attr[N,C], pd[N] or rest_density)
 * mode 2: weighted-average accumulation: grid += W*attr, wsum += W (p2g_wavg; finish with
 *         nfs_p2g_wavg_finish).  grid/wsum must be zeroed by the caller. */
typedef struct {
  int nd;            /* 2 or 3 */
  int res[3];        /* grid resolution (array order) */
  float domain[3];   /* domain size (array order) */
  float radius, support, rest_density;
  int nsize, clip, mode;
} nfs_splat_cfg;
int nfs_p2g_fwd(const float* p, const float* attr, const float* pd, float* grid, float* wsum,
                int N, int C, const nfs_splat_cfg* cfg_host, nfs_stream_t stream);
/* g_p [N,nd] / g_attr [N,C] / g_pd [N] overwritten (each nullable).  For mode 2 pass the
 * gradients wrt the raw accumulators (from nfs_p2g_wavg_finish_bwd) as g_grid / g_wsum. */
int nfs_p2g_bwd(const float* p, const float* attr, const float* pd, const float* g_grid,
                const float* g_wsum, float* g_p, float* g_attr, float* g_pd,
                int N, int C, const nfs_splat_cfg* cfg_host, nfs_stream_t stream);
/* out = wsum > eps ? xsum/wsum : xsum (transform.py:1701-1703); n cells x C channels */
int nfs_p2g_wavg_finish(const float* xsum, const float* wsum, float* out, int64_t n, int C,
                        float eps, nfs_stream_t stream);
/* nfs_p2g_wavg_finish_bwd + nfs_p2g_bwd (mode 2) in ONE launch (round 4): the adjoint of p2g_wavg (transform.py:1577-1704)
 * from g_out [cells,C] = dL/d(finished average), the raw accumulators xsum [cells,C] / wsum [cells] of the forward and
 * the particles: the gradients wrt the accumulators are formed per cell while a block stages its box of the grid, not
 * in a five-array pass over the whole grid.  g_p [N,nd] / g_attr [N,C] overwritten (each nullable).  NFS_EINVAL for
 * (nd, nsize) without a compile-time neighbourhood: use the two-step path then. */
int nfs_p2g_wavg_bwd(const float* p, const float* attr, const float* xsum, const float* wsum, const float* g_out,
                     float* g_p, float* g_attr, int N, int C, float eps, const nfs_splat_cfg* cfg_host,
                     nfs_stream_t stream);
int nfs_p2g_wavg_finish_bwd(const float* xsum, const float* wsum, const float* g_out,
                            float* g_xsum, float* g_wsum, int64_t n, int C, float eps,
                            nfs_stream_t stream);

/* ---- g2p_linear / g2p_cubic (transform.py:771-1231) ----------------------
 * g [X,Y,(Z),C] cell-centred grid, p [N,nd] in [0,1] (axis order = array order), out [N,C].
 * x = p*n, base = floor(x-0.5); linear: cells base, base+1 clipped, dx = x-(clipped base+0.5);
 * cubic: cells base-1..base+2 clipped, Catmull-Rom (_hermite) in t = x-(clipped base+0.5).
 * Forward only: the reference's resampler (test_smokegun_resim.py) never differentiates it. */
int nfs_g2p_fwd(const float* g, const float* p, float* out, int nd, int X, int Y, int Z, int C,
                int64_t N, int cubic, nfs_stream_t stream);

/* ---- A10: TF ApplyAdam (styler_3p.py:320-323) -----------------------------------------
 * m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; x -= lr_t*m/(sqrt(v)+eps), lr_t =
 * lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller.  NaN gradients are NOT sanitised. */
int nfs_adam_tf_step(float* x, float* m, float* v, const float* g, int64_t n,
                     float lr_t, float beta1, float beta2, float eps, nfs_stream_t stream);

/* small helpers used by the host loop */
int nfs_fill(float* x, float value, int64_t n, nfs_stream_t stream);
int nfs_axpy(float* y, const float* x, float a, int64_t n, nfs_stream_t stream); /* y += a*x */

/* ---- 2-D colour stylizer: the elementwise links of its chain (styler_2p.py:68-102, 259-262) -------------------
 * The reference's graph for one frame is c_ = clip(c, 0, 1) -> p2g(p, pc = c_, pd = r) -> clip(., 0, 1) -> loss net, and
 * per iteration g_opt += nan_to_num(c_new) - g_opt.  As TF ops (or torch autograd nodes) these are a dozen launches of a
 * few thousand elements around the splat; here:
 *   nfs_colour_clamp_gather       out[i] = clip(var[order[i]], 0, 1)  ([N,C]; order nullable = identity: the frame's grid
 *                                 order, a permutation held as int64)
 *   nfs_clamp01_bwd               out = g where 0 <= x <= 1, else 0   (adjoint of the clip of the splatted image)
 *   nfs_colour_clamp_scatter_bwd  g_var[order[i]] = g_cc[i] where 0 <= var[order[i]] <= 1, else 0
 *   nfs_iterate_update            r = g_opt + (nan_to_num(x) - g_opt) (the reference's two roundings); g_opt <- r, x <- r
 *                                 (x: the variable ApplyAdam has just updated: the next iteration's start) */
int nfs_colour_clamp_gather(const float* var, const long long* order, float* out, int64_t N, int C, nfs_stream_t stream);
int nfs_clamp01_bwd(const float* g, const float* x, float* out, int64_t n, nfs_stream_t stream);
int nfs_colour_clamp_scatter_bwd(const float* g_cc, const long long* order, const float* var, float* g_var, int64_t N,
                                 int C, nfs_stream_t stream);
int nfs_iterate_update(float* x, float* g_opt, int64_t n, nfs_stream_t stream);

/* ---- (e) multi-GPU: send buffer of the D-slab reduce-scatter ----------------------------------
 * The reference has no collective; the view-sharded step exchanges the density-field gradient as a
 * reduce-scatter over D-slabs whose chunks OVERLAP by a two-plane halo either side (engine.GridStylizer._slab_setup).
 * gpad [D+5][plane]: planes [2, D+2) = the local gradient, planes 0,1,D+2,D+3 zero, plane D+4 carries the local loss in
 * its first word.  pack [world][cs+5][plane]: chunk k = planes [k*cs, k*cs+cs+4) of gpad (zero where that runs past
 * plane D+3: the short / empty slabs of a ragged split) followed by plane D+4.  plane % 4 == 0, 16-byte aligned
 * pointers.  One copy kernel (4*(world*(cs+5))*plane bytes written) in place of a gather by an index table. */
int nfs_slab_pack(const float* gpad, float* pack, int D, int64_t plane, int world, int cs, nfs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NFS_HIP_H */
