/*
 * nlt_hip.h -- C ABI of libnlt_hip.so: the MI355X (gfx950) hot path of
 * google/neural-light-transport's UV-texture-space renderer.
 *
 * The reference has NO native boundary: every op below is executed today by a
 * TensorFlow-2.2 / TF-Addons kernel reached from Python.  Each
 * entry point names the reference call site it replaces (paths relative to the
 * reference tree).  A reference-side binding is therefore a Python ctypes stub; see
 * INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; tensors are NHWC,
 *     fp32, densely packed unless a per-texel stride (`ld*`, in floats) says otherwise;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL =
 *     the default stream), re-entrant per stream, keeps no global state and never
 *     throws: it returns NLT_OK or a negative nlt_status;
 *   - sizes are ints; tensors hold < 2^31 elements.
 */
#ifndef NLT_HIP_H_
#define NLT_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  NLT_OK = 0,
  NLT_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, misaligned pointer/stride */
  NLT_ERR_UNSUPPORTED = -2,  /* shape/algorithm combination this build does not implement */
  NLT_ERR_LAUNCH = -3        /* hipGetLastError() != hipSuccess after the launch */
} nlt_status;

/* Conv families of nlt/networks/elements.py:26-39 as Keras/TF execute them with
 * padding='same' (SURVEY.md 8a rows a-C1..a-D3).  (h, w) below are INPUT dims. */
typedef enum {
  NLT_CONV1X1 = 0,     /* Conv2D k1 s1: out (h,w);   W keras (1,1,Cin,Cout)                       */
  NLT_CONV_K2S2 = 1,   /* Conv2D k2 s2: out (h/2,w/2), h,w even; W keras (2,2,Cin,Cout)            */
  NLT_CONV_K2S1 = 2,   /* Conv2D k2 s1: out (h,w), zero pad bottom/right; W keras (2,2,Cin,Cout)   */
  NLT_DECONV_K2S2 = 3, /* Conv2DTranspose k2 s2: out (2h,2w); W keras (2,2,Cout,Cin)               */
  NLT_DECONV_K2S1 = 4  /* Conv2DTranspose k2 s1: out (h,w), zero pad top/left; W (2,2,Cout,Cin)    */
} nlt_conv_mode;

typedef enum {
  NLT_ALGO_AUTO = 0,
  NLT_ALGO_DIRECT = 1, /* one thread per texel x 4 outputs, fp32 FMA; reads Keras-layout weights  */
  NLT_ALGO_MFMA = 2    /* implicit GEMM on v_mfma_f32_16x16x4_f32; reads nlt_pack_conv_weights()  */
} nlt_conv_algo;

const char* nlt_version(void);
const char* nlt_status_string(int status);

/* Number of floats nlt_pack_conv_weights() writes for (mode, c0, c1, cout). */
long nlt_packed_weight_floats(int mode, int c0, int c1, int cout);

/* Re-lays a Keras kernel for the MFMA path: [K/16][N/16][64 lanes][4] fragments, each
 * K segment (tap x source) zero-padded to a multiple of 16.  Replaces nothing in the
 * reference (Keras keeps HWIO / HWOI); call it whenever the weights change. */
int nlt_pack_conv_weights(int mode, const float* w_keras, int c0, int c1, int cout,
                          float* w_packed, void* stream);

/*
 * One conv / transposed-conv layer with fused bias, LeakyReLU and "virtual concat".
 *   replaces: tf.keras Conv2D / Conv2DTranspose / LeakyReLU as called through
 *             nlt/networks/elements.py:26-39,72-73 (layer stacks of convnet.py:44-85) and
 *             the tf.concat of nlt/models/nlt.py:174,190 (the two sources ARE the concat).
 * Input channels [0,c0) come from src0 (per-texel stride ld0 floats), [c0,c0+c1) from src1
 * (stride ld1; c1 = 0 and src1 = NULL for a single source); both are [n,h,w,*].  The
 * output goes to out[texel*ldo + 0..cout) so a layer can write a channel slice of a wider
 * interleaved tensor.  act != 0 applies LeakyReLU(alpha) (elements.py:72-73: alpha = 0.3).
 * Backward-data reuse: mask_src (stride ldm) != NULL multiplies the result by
 * (mask_src > 0 ? 1 : alpha) instead of activating it, and accumulate != 0 adds the old
 * contents of `out` before masking.
 * algo DIRECT needs w_keras, MFMA needs w_packed; AUTO uses MFMA when it can.
 * tile_hint: 0 = auto, else 16*RT + CT (wave tile = 16*RT texels x 16*CT outputs).
 */
int nlt_conv_forward(int mode, int algo, int tile_hint,
                     const float* src0, int ld0, int c0,
                     const float* src1, int ld1, int c1,
                     int n, int h, int w,
                     const float* w_keras, const float* w_packed, const float* bias,
                     int cout, float* out, int ldo,
                     int act, float alpha,
                     const float* mask_src, int ldm, int accumulate,
                     void* stream);

/*
 * L0 of both paths, fused (the original-resolution 1x1 convs, no activation).
 *   replaces: nlt/models/nlt.py:95-96 (tf.concat of base|cvis|lvis; nn_rgb - nn_base),
 *             layer 0 of net['query'] and net['obs'] (convnet.py:44), and the first
 *             observation mean + concat (nlt.py:161-164,174).
 * nn_rgb / nn_base: [n,k,h,w,3] (k observations per frame); obs_weights: [n,k] or NULL.
 * fm0 [n,h,w,2*c]: channels [0,c) = query L0, [c,2c) = mean_k(obs L0 (* weight)).
 * obs0 [n,k,h,w,c]: per-observation L0 outputs (input of obs layer 1).
 * Weights Keras layout: wq (1,1,5,c), wo (1,1,3,c).
 */
int nlt_stem_forward(const float* base, const float* cvis, const float* lvis,
                     const float* nn_rgb, const float* nn_base, const float* obs_weights,
                     int n, int k, int h, int w, int c,
                     const float* wq, const float* bq, const float* wo, const float* bo,
                     float* fm0, float* obs0, void* stream);

/*
 * Mean over the k observation feature maps of a level, written into a channel slice.
 *   replaces: nlt/models/nlt.py:161-164 (expand_dims, concat, obs_weights product, reduce_mean).
 * obs [n,k,h,w,c]; obs_weights [n,k] or NULL; out[texel*ldo + 0..c) = sum_i w_i*obs_i / k.
 */
int nlt_obs_mean_forward(const float* obs, const float* obs_weights, int n, int k, int hw, int c,
                         float* out, int ldo, void* stream);

/*
 * Output head: 1x1 conv over the virtual concat [dec | skip] -> 3 channels, + base,
 * texel (0,0) of every frame forced to 0.
 *   replaces: last layer of net['query'] (convnet.py:85), nlt/models/nlt.py:99-102
 *             (pred += base) and :110 (imgutil.set_left_top_corner(pred, 0)).
 * dec [n,h,w,*] stride ldd, cd channels; skip stride lds, cs channels; w keras (1,1,cd+cs,3).
 * base may be NULL (skip_connect_base = False).  pred [n,h,w,3].
 */
int nlt_head_forward(const float* dec, int ldd, int cd, const float* skip, int lds, int cs,
                     const float* w_keras, const float* bias, const float* base,
                     int n, int h, int w, float* pred, void* stream);

/*
 * UV -> camera bilinear gather of the three buffers the model warps, in one pass.
 *   replaces: nlt/models/nlt.py:104-114: warp * (uvw, uvh); fg = ones with texel (0,0)
 *             zeroed; set_left_top_corner(base, 0); three tfa.image.resampler calls
 *             (TF-Addons 0.10.0 resampler op).
 * pred, base [n,uvh,uvw,3] (base's texel (0,0) is treated as 0; pred must already have it
 * zeroed, as nlt_head_forward leaves it; base may be NULL); warp [n,hc,wc,2] in [0,1] UV
 * units, x first.  Outputs [n,hc,wc,3] each (any may be NULL).  idx_out (optional)
 * [n,hc,wc,4] int32 = (fx, fy, inside, 0): the integer UV indices of the gather, for the
 * bit-exact parity check.
 */
int nlt_warp_forward(const float* pred, const float* base, const float* warp,
                     int n, int uvh, int uvw, int hc, int wc,
                     float* pred_cam, float* base_cam, float* fg_cam, int* idx_out, void* stream);

/* nlt_warp_forward on a STORE-RESIDENT batch (Dataset.load_batch(ids, resident=True)): base comes from the uint8 diffuse
 * store [F,uvh,uvw,3] and the map from the fp16 uv2cam store [F,hc,wc,2] (data_gen/util.py:67-70), frame ids[f] of each,
 * converted in registers exactly as `_load_data` does (uint8 -> float64 / 255 -> float32, nlt/datasets/nlt.py:131-136;
 * fp16 -> fp32, :125,176).  Bit-identical outputs to nlt_warp_forward on the assembled float batch.
 *   replaces: the same reference lines as nlt_warp_forward + the `base` / `warp` entries of `_load_data`'s 11-tuple. */
int nlt_warp_forward_store(const float* pred, const unsigned char* diffuse_store, const unsigned short* uv2cam_store,
                           const int* ids, int n, int uvh, int uvw, int hc, int wc,
                           float* pred_cam, float* base_cam, float* fg_cam, int* idx_out, void* stream);

/* The raw resampler op on a map of c channels (c % 4 == 0): data [n,h,w,c], warp_px [n,hc,wc,2] in PIXEL units (x first)
 * -> out [n,hc,wc,c]; tfa.image.resampler's tap rule and order (no corner mask, no scaling).  The channel-width stress
 * point of the per-texel kernels ("1024^2 x 64-ch"); the model itself only warps 3-channel maps.
 *   replaces: tfa.image.resampler(data, warp) (nlt/models/nlt.py:112-114) for a wider `data`. */
int nlt_resample_forward(const float* data, const float* warp_px, int n, int h, int w, int c, int hc, int wc,
                         float* out, void* stream);

/*
 * Bilinear resize, half-pixel centres, no antialias.
 *   replaces: tf.image.resize via nlt/util/img.py:92-120 (nlt/models/nlt.py:116-120).
 */
int nlt_resize_bilinear_forward(const float* x, int n, int h, int w, int c, int oh, int ow,
                                float* out, void* stream);

/* out = a * b elementwise (imgutil.alpha_blend with tensor2=None, nlt/util/img.py:74-89;
 * nlt/models/nlt.py:132-133). */
int nlt_mul_forward(const float* a, const float* b, long count, float* out, void* stream);

/* ======================= train step (nlt/trainvali.py:272-281) ======================= */

/*
 * Weight and bias gradients of one conv layer: dW += X^T dPre, db += sum dPre, accumulated into
 * Keras-layout buffers (zero them first).  X is given exactly as to nlt_conv_forward (sources,
 * strides, input dims); dpre [n,oh,ow,*] (stride ldp) is the gradient w.r.t. the layer's
 * PRE-activation output.
 *   replaces: tape.gradient(loss, layer.kernel / layer.bias) for tf.keras Conv2D / Conv2DTranspose
 *             (nlt/trainvali.py:279; layers built in nlt/networks/elements.py:26-39).
 * Backward-DATA needs no extra entry point: it is nlt_conv_forward in the adjoint mode
 * (CONV_K2Sx <-> DECONV_K2Sx) on the SAME Keras kernel array, with mask_src / accumulate.
 */
int nlt_conv_backward_weights(int mode, int algo,
                              const float* src0, int ld0, int c0,
                              const float* src1, int ld1, int c1,
                              int n, int h, int w,
                              const float* dpre, int ldp, int cout,
                              float* dw_keras, float* dbias, void* stream);

/* nlt_conv_backward_weights, second generation (csrc/wgrad_tile.hip): 16-byte operand loads feeding 64 x 64 blocks of
 * dW on the MFMA, row slices reduced through `workspace` in a fixed order (deterministic, no atomics).  Same
 * arguments and accumulate-into semantics; channel counts and strides must be multiples of 4.
 * workspace: nlt_wgrad_workspace_floats() floats. */
long nlt_wgrad_workspace_floats(int mode, int c0, int c1, int n, int h, int w, int cout);
int nlt_conv_backward_weights_tiled(int mode,
                                    const float* src0, int ld0, int c0, const float* src1, int ld1, int c1,
                                    int n, int h, int w, const float* dpre, int ldp, int cout,
                                    float* dw_keras, float* dbias, float* workspace, long workspace_floats,
                                    void* stream);

/* out = g * (y > 0 ? 1 : alpha): LeakyReLU backward from the saved OUTPUT y (elements.py:72-73). */
int nlt_lrelu_backward(const float* g, int ldg, const float* y, int ldy, int c, long texels, float alpha,
                       float* out, int ldo, void* stream);

/* Backward of nlt_obs_mean_forward fused with the observation layer's LeakyReLU backward:
 * dpre_obs[f,i] = (dobs_partial[f,i] + dmean[f] * w_i / k) * lrelu'(obs_y[f,i]); obs_y / obs_weights /
 * dobs_partial may be NULL (no activation / unit weights / no other consumer).  nlt/models/nlt.py:161-166. */
int nlt_obs_mean_backward(const float* dmean, int ldm, const float* obs_y, const float* obs_weights,
                          const float* dobs_partial, int n, int k, int hw, int c, float alpha,
                          float* dpre_obs, void* stream);

/* nlt_lrelu_backward on the query half and nlt_obs_mean_backward on the observation half of one level's
 * dfm [n,hw,ld >= 2c] = [query c | observation mean c] in ONE launch (nlt/models/nlt.py:155-167 backward):
 *   dfm[.., 0:c] *= lrelu'(fm_y[.., 0:c]) in place;  dpre_obs[f,i] = (dobs_partial[f,i] + dfm[f][.., c:2c] w_i / k) * lrelu'(obs_y[f,i]).
 * fm_y = the saved fm[l] (same ld); obs_weights / dobs_partial may be NULL. */
int nlt_level_split_backward(float* dfm, const float* fm_y, int ld, const float* obs_y, const float* obs_weights,
                             const float* dobs_partial, int n, int k, int hw, int c, float alpha_q, float alpha_o,
                             float* dpre_obs, void* stream);

/* Backward of nlt_stem_forward: accumulates dwq (5,c), dbq (c), dwo (3,c), dbo (c) from
 * dfm0 [n,h,w,2c] and the per-observation partial dobs0 [n,k,h,w,c] (may be NULL). */
int nlt_stem_backward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                      const float* nn_base, const float* obs_weights, int n, int k, int h, int w, int c,
                      const float* dfm0, const float* dobs0_partial,
                      float* dwq, float* dbq, float* dwo, float* dbo, void* stream);

/* Backward of nlt_head_forward: d_dec / d_skip (written), dw (cd+cs,3) and db (3) accumulated;
 * the gradient of texel (0,0) is dropped (set_left_top_corner). */
int nlt_head_backward(const float* dec, int ldd, int cd, const float* skip, int lds, int cs,
                      const float* w_keras, const float* dpred, int n, int h, int w,
                      float* d_dec, int ldgd, float* d_skip, int ldgs, float* dw, float* db, void* stream);

/* Gradient of nlt_warp_forward w.r.t. pred: 4-corner scatter-add (the TFA resampler gradient,
 * nlt/models/nlt.py:114 under the tape); dpred [n,uvh,uvw,3] is zero-filled first; texel (0,0) gets none. */
int nlt_warp_backward(const float* dpred_cam, const float* warp, int n, int uvh, int uvw, int hc, int wc,
                      float* dpred, void* stream);

/* Gradient of nlt_resize_bilinear_forward w.r.t. x (zero-filled first). */
int nlt_resize_bilinear_backward(const float* dout, int n, int h, int w, int c, int oh, int ow, float* dx,
                                 void* stream);

/* losses.L2 with keep_batch=True (nlt/losses.py:39-53): loss[f] = mean over H,W,C of (gt-pred)^2. */
int nlt_l2_loss_forward(const float* pred, const float* gt, int n, long per_example, float* loss, void* stream);
int nlt_l2_loss_backward(const float* pred, const float* gt, const float* gloss, int n, long per_example,
                         float* dpred, void* stream);

/* losses.L2 with `weights=` (nlt/losses.py:42-43: Keras `sample_weight` on MeanSquaredError(reduction='none')): the per-texel
 * loss map [n,H,W] (mean over the c channels) is multiplied by weights [n,H,W] before the mean over H,W:
 * loss[f] = sum_px weights[f,px] * mean_c (gt-pred)^2 / hw.  The host mirror broadcasts the shapes Keras accepts
 * (scalar, [N,1,1], [N,H,W], [N,H,W,1]) to [n,H,W] first. */
int nlt_l2_loss_weighted_forward(const float* pred, const float* gt, const float* weights, int n, long hw, int c,
                                 float* loss, void* stream);
int nlt_l2_loss_weighted_backward(const float* pred, const float* gt, const float* weights, const float* gloss, int n,
                                  long hw, int c, float* dpred, void* stream);

/* The l2 train step's loss in one launch (nlt/models/nlt.py:238-245 `gt = rgb_camspc * fg`; losses.L2 keep_batch;
 * trainvali.py:277-278 `sum / global batch`; and the gradient w.r.t. pred): gt = rgb * fg, loss[0] = sum_f mean((pred_f - gt_f)^2)
 * / global_bs, dpred = 2 (pred - gt) / per_example / global_bs.  Same arithmetic as nlt_mul_forward + nlt_l2_loss_forward +
 * nlt_l2_loss_backward with gloss = 1 / global_bs. */
int nlt_l2_train_loss(const float* pred, const float* rgb, const float* fg, int n, long per_example, float inv_global_bs,
                      float* gt, float* dpred, float* loss, void* stream);

/* losses.Barron with keep_batch=True (nlt/losses.py:90-118; robust_loss adaptive.py:453-538 with
 * alpha = 1, scale = 0.01, CDF9/7, 5 levels, sYUV): loss[f]; if dpred_unit != NULL also
 * d loss[f] / d pred (multiply by the upstream per-example gradient with nlt_scale_rows).
 * workspace: nlt_barron_workspace_floats(n,h,w) floats.  min(h,w) >= 17. */
long nlt_barron_workspace_floats(int n, int h, int w);
int nlt_barron_loss(const float* pred, const float* gt, int n, int h, int w, float* workspace,
                    float* loss, float* dpred_unit, void* stream);

/* out[f,:] = x[f,:] * scale[f] */
int nlt_scale_rows(const float* x, const float* scale, int n, long per_row, float* out, void* stream);

/*
 * bf16 middle of the network (BASELINE config 5): the conv family on v_mfma_f32_16x16x32_bf16 with fp32 accumulation,
 * bf16-STORED activations between layers, fp32 bias + LeakyReLU on the accumulator.  Same modes, geometry, virtual
 * concat (src0 | src1) and Keras weight layouts as nlt_conv_forward; each source and the output may independently be
 * fp32 (the region's boundaries: rounded to bf16, nearest even, on load / kept fp32 on store) or bf16.
 * c0, c1 multiples of 8, cout multiple of 4, ld* multiples of 8 (bf16) / 8 (fp32 source), 16-byte aligned pointers.
 *   replaces: Conv2D / Conv2DTranspose + LeakyReLU of the encoder levels >= 3 and the expanding blocks mirroring them
 *             (nlt/networks/convnet.py:50-59,67-76) when the model's `precision` is bf16.
 * nlt_conv_bf16_pack: Keras fp32 kernel -> bf16 MFMA fragments (nlt_conv_bf16_packed_elems uint16 elements).
 * nlt_obs_mean_bf16: mean over the k stored-bf16 observation maps of a level into a channel slice of the bf16 fm[l]
 *   (tf.reduce_mean, nlt/models/nlt.py:161-164): fp32 sum in observation order, * (1/k), rounded to bf16.
 */
long nlt_conv_bf16_packed_elems(int mode, int c0, int c1, int cout);
int nlt_conv_bf16_pack(int mode, const float* w_keras, int c0, int c1, int cout, unsigned short* packed, void* stream);
int nlt_conv_bf16_forward(int mode, int tile_hint,
                          const void* src0, int ld0, int c0, int src0_is_f32,
                          const void* src1, int ld1, int c1, int src1_is_f32,
                          int n, int h, int w, const unsigned short* w_packed, const float* bias,
                          int cout, void* out, int ldo, int out_is_f32, int act, float alpha, void* stream);
int nlt_obs_mean_bf16(const unsigned short* obs, int n, int k, long hw, int c, unsigned short* out, int ldo, void* stream);

/* Layers of the config branches the released .ini files leave off (executed layer by layer, nlt_amd/generic.py):
 *   nlt_act_forward / _backward       kind 0: LeakyReLU(alpha) / ReLU (alpha = 0), kind 1: tf.keras.layers.ELU(alpha)
 *                                     (nlt/networks/elements.py:69-78); backward takes the layer's OUTPUT y
 *   nlt_pixelnorm_forward / _backward y = x * rsqrt(mean_c(x^2) + eps), per texel over c channels (elements.py:103-121)
 *   nlt_pool2x2_forward / _backward   MaxPooling2D / AveragePooling2D(pool 2, strides 2, 'same') on even sizes, kind 0 max /
 *                                     1 average (elements.py:81-94); max backward: the first maximal tap gets the gradient
 * All NHWC fp32, any channel count. */
int nlt_sub_forward(const float* a, const float* b, long count, float* out, void* stream);      /* y_obs = nn_rgb - nn_base (nlt.py:96) */
/* pred = y (+ base when base != NULL), texel (0,0) of every frame zeroed (nlt/models/nlt.py:99-110); y, base, pred [n,h,w,3] */
int nlt_finish_pred(const float* y, const float* base, int n, int h, int w, float* pred, void* stream);
int nlt_act_forward(const float* x, long count, int kind, float alpha, float* y, void* stream);
int nlt_act_backward(const float* g, const float* y, long count, int kind, float alpha, float* dx, void* stream);
int nlt_pixelnorm_forward(const float* x, long texels, int c, float eps, float* y, void* stream);
int nlt_pixelnorm_backward(const float* g, const float* x, long texels, int c, float eps, float* dx, void* stream);
int nlt_pool2x2_forward(const float* x, int n, int h, int w, int c, int kind, float* y, void* stream);
int nlt_pool2x2_backward(const float* g, const float* x, int n, int h, int w, int c, int kind, float* dx, void* stream);

/* norm = layer / batch (nlt/networks/elements.py:51-56; between every conv and its activation, convnet.py:50-59,67-76),
 * per texel over its c channels (NHWC fp32, c <= 1024):  y[ch] = (x[ch] - m) * r * gamma[ch] + beta[ch]
 *   kind 0  tf.keras.layers.LayerNormalization(epsilon, center, scale): m = mean_ch(x), r = rsqrt(biased var_ch(x) + eps);
 *   kind 1  tf.keras.layers.BatchNormalization(momentum=0.99, epsilon) in INFERENCE mode -- the mode the reference's loop
 *           runs it in (no `training=True` anywhere under nlt/: networks/seq.py:36-41, models/nlt.py:154-195), so
 *           m = mean[ch] (moving_mean), r = rsqrt(var[ch] + eps) (moving_variance), both never updated.
 * backward: dx, and dgamma / dbeta ACCUMULATED (+=) in a fixed order (per-workgroup partial rows in `workspace`,
 * nlt_norm_workspace_floats(texels, c) floats; -1: unsupported c).  mean / var may be NULL for kind 0.
 *   replaces: elements.norm('layer' | 'batch') layers and their gradients under nlt/trainvali.py:272-280. */
long nlt_norm_workspace_floats(long texels, int c);
int nlt_norm_forward(int kind, const float* x, long texels, int c, const float* gamma, const float* beta,
                     const float* mean, const float* var, float eps, float* y, void* stream);
int nlt_norm_backward(int kind, const float* g, const float* x, long texels, int c, const float* gamma,
                      const float* mean, const float* var, float eps, float* dx, float* dgamma, float* dbeta,
                      float* workspace, void* stream);

/* Keras `clipnorm` (config key mgm > 0): every variable's gradient g -> g * clip / max(||g||_2, clip)
 * (tf.clip_by_norm, multiply then divide), in place, over the slots of the flat gradient bucket.
 * slots: device int64 [n_slots][2] = (first element, element count) of each kernel / bias.
 *   replaces: tf.keras.optimizers.Adam(clipnorm=mgm) (nlt/trainvali.py:122-127). */
int nlt_clip_by_norm_slots(float* grad, const long* slots, int n_slots, float clipnorm, void* stream);

/* One fused Keras Adam(amsgrad=True) step over a flat parameter bucket (TF 2.2 OptimizerV2:
 * p -= lr_t * m / (sqrt(vhat) + eps), lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller).
 *   replaces: optimizer.apply_gradients (nlt/trainvali.py:124-127,280). */
int nlt_adam_amsgrad_step(float* param, const float* grad, float* m, float* v, float* vhat, long count,
                          float lr_t, float beta1, float beta2, float eps, void* stream);

/* ============ texel-buffer assembly (data_gen/ + nlt/datasets/nlt.py; SURVEY.md 8a a-B1..a-B6) ============
 * Byte / integer work on float64 geometry: every result below is bit-exact against oracle/buffers.py
 * (the float64 stages follow the reference's NumPy operation order; no fused multiply-add). */

/* dtype of a [0,1] coordinate map handed to the remap entry points */
typedef enum { NLT_MAP_F64 = 0, NLT_MAP_F32 = 1, NLT_MAP_F16 = 2 } nlt_map_dtype;

/*
 * View / light cosine map of one camera: cos = <normalize(src - p), normalize(n)> at pixels with valid != 0
 * and (occluded == NULL or occluded == 0), 0 elsewhere.
 *   replaces: data_gen/render.py:209-228 calc_view_cosines (src = camera location, occluded = NULL) and
 *             :231-276 calc_light_cosines (src = light location; `occluded` is the result of the reference's
 *             BVH shadow rays, which stay in Blender), plus :164,170 np.clip + xm.img.denormalize_float.
 * locs, normals [pixels,3] float64; valid / occluded [pixels] uint8.  cos_out float64 and/or u8_out (clipped to
 * [0,1], TRUNCATED x255) -- either may be NULL.
 */
int nlt_cosine_map(const double* locs, const double* normals, const unsigned char* valid,
                   const unsigned char* occluded, double sx, double sy, double sz, long pixels,
                   double* cos_out, unsigned char* u8_out, void* stream);

/*
 * UV albedo = (sum over frames of rgb/255, in frame order) / max of that sum.
 *   replaces: data_gen/postproc.py:53-64.   rgb_frames [frames, elems] uint8 (elems = H*W*3); albedo [elems]
 *   float64; workspace8: 8 bytes of device scratch.
 */
int nlt_albedo(const unsigned char* rgb_frames, int frames, long elems, double* albedo, void* workspace8, void* stream);

/* diffuse base = clip(albedo * lvis/255, 0, 1) truncated to uint8, for `frames` light-visibility maps at once.
 *   replaces: data_gen/postproc.py:66-76.   lvis [frames,texels] uint8 -> diffuse [frames,texels,3] uint8. */
int nlt_diffuse_base(const double* albedo, const unsigned char* lvis, int frames, long texels,
                     unsigned char* diffuse, void* stream);

/*
 * Bilinear gather of src at mapping * (w, h) with cv2.remap(INTER_LINEAR, BORDER_CONSTANT 0) semantics --
 * coordinates quantised to 1/32 texel (round half to even), 15-bit fixed-point weights for 8-bit sources --
 * and source texel (0,0) read as 0 when force_kbg.
 *   replaces: data_gen/util.py:45-58 remap (camera->UV: data_gen/render.py:174-176; UV->camera:
 *             data_gen/postproc.py:78-82).
 * src [h,w,c]; mapping [oh,ow,ldm] of map_dtype, channel 0 = x, 1 = y, in [0,1]; out [oh,ow,c].
 */
int nlt_remap_bilinear_u8(const unsigned char* src, int h, int w, int c, const void* mapping, int map_dtype,
                          int ldm, int oh, int ow, int force_kbg, unsigned char* out, void* stream);
int nlt_remap_bilinear_f32(const float* src, int h, int w, int c, const void* mapping, int map_dtype,
                           int ldm, int oh, int ow, int force_kbg, float* out, void* stream);

/*
 * UV-index map: paints `samples` scattered (u,v) -> value samples onto an h x w grid.  A texel is TRUSTED when
 * an occupied texel -- integer indices ri = int((1-v)(h-1)), ci = int(u(w-1)) of some sample -- lies within L1
 * distance max_l1; trusted texels take the value of the nearest sample (Euclidean in uv; ties -> lowest sample
 * index), the others `fill`.
 *   replaces: xiuminglib/img.py:289-431 grid_query_unstruct(method griddata/nearest, max_l1_interp) as called by
 *             data_gen/render.py:279-351 calc_bidir_mapping (scipy KD-tree + cv2.distanceTransform(DIST_L1)).
 * uvs [samples,2], values [samples,m] float64 -> out [h,w,m] float64; index_out [h,w] int32 (chosen sample, -1 =
 * filled) or NULL.  workspace: nlt_uv_index_map_workspace_bytes() bytes.  h, w >= 2; max_l1 <= 64.
 */
long nlt_uv_index_map_workspace_bytes(int h, int w, long samples);
int nlt_uv_index_map(const double* uvs, const double* values, long samples, int m, int h, int w,
                     int max_l1, double fill, void* workspace, double* out, int* index_out, void* stream);

/*
 * k nearest candidates (non-zero distance, nearest first, equal distances in candidate order) of every reference
 * position; -1 where fewer than k qualify.
 *   replaces: data_gen/get_neighbors.py:52-71 get_neighbors (k = 1), whose result picks the observation maps
 *             (data_gen/render.py:200-206, nlt/datasets/nlt.py:88-100,150-171).
 * ref_pos [np,3], cand_pos [nq,3] float64 -> out [np,k] int32.
 */
int nlt_knn_indices(const double* ref_pos, int np, const double* cand_pos, int nq, int k, int* out, void* stream);

/* The two sums behind PSNR on luma (xiuminglib/metric.py:105-151; luma = 0.2126 r + 0.7152 g + 0.0722 b, img.py:600-611), in
 * float64 like the reference's `im.astype(float)`: out2[0] = sum over masked pixels of (lum(im1) - lum(im2))^2,
 * out2[1] = number of masked pixels (mask NULL = all).  im1 / im2 [pixels, channels] float32, channels 1 or 3;
 * workspace: 512 doubles.  PSNR = 10 log10(drange^2 / (out2[0] / out2[1])).
 *   replaces: xm.metric.PSNR(np.float32)(gt, pred) of Model.vis_batch (nlt/models/nlt.py:64,259-269). */
int nlt_psnr_sums(const float* im1, const float* im2, const unsigned char* mask, long pixels, int channels,
                  double* workspace, double* out2, void* stream);

/* cv2.resize(img, (ow, oh)) -- default INTER_LINEAR -- on the NORMALISED image, then astype(float32): what `_load_data`
 * does to every texel buffer of a capture stored at another resolution than uvh / (imh, imw)
 * (nlt/datasets/nlt.py:138-146,162-170 through xm.img.resize, third_party/xiuminglib/xiuminglib/img.py:88-118).
 * src [n,h,w,c]: src_kind 0 = uint8 (normalised by 255, nlt.py:131-136), 1 = int32 holding 16-bit PNG samples
 * (normalised by 65535), 2 = float32 already in [0,1].  out [n,oh,ow,c] float32.  OpenCV's arithmetic restated
 * (tap weights in float32, horizontal pass first, float64 sums); cv2 itself cannot be installed here: parity unpinned. */
int nlt_resize_cv_linear(const void* src, int src_kind, int n, int h, int w, int c, int oh, int ow, float* out,
                         void* stream);

/* out[i] = float32(float64(store[ids[i]]) / 255) for whole frames of per_frame bytes (per_frame % 4 == 0);
 * ids == NULL means frames 0..n-1, id -1 a frame of zeros.  The primitive behind nlt_assemble_batch; also serves
 * the camera-space images (rgb_camspc, nn_rgb_camspc; nlt/datasets/nlt.py:129-136,162-171). */
int nlt_gather_frames_u8(const unsigned char* store, const int* ids, int n, long per_frame, float* out, void* stream);

/*
 * Batch assembly from a uint8 frame store resident in HBM: gathers frames `ids` [n] and their neighbours
 * `nn_ids` [n,k] (-1 = missing neighbour -> zeros) and converts uint8 -> float64/255 -> float32.
 *   replaces: nlt/datasets/nlt.py:115-184 _load_data (PNG decode itself stays on the host).
 * stores: diffuse/rgb [F,texels,3], cvis/lvis [F,texels] uint8.  Outputs float32: base, rgb [n,texels,3];
 * cvis, lvis [n,texels]; nn_base, nn_rgb [n,k,texels,3].  test_mode != 0: rgb = 0 (nlt.py:126-128).
 */
int nlt_assemble_batch(const unsigned char* diffuse_store, const unsigned char* rgb_store,
                       const unsigned char* cvis_store, const unsigned char* lvis_store,
                       const int* ids, const int* nn_ids, int n, int k, long texels, int test_mode,
                       float* base, float* cvis, float* lvis, float* rgb, float* nn_base, float* nn_rgb,
                       void* stream);

/* ======================= fused inference ends (csrc/fused.hip) =======================
 * Forward-only fusions of everything that touches full-resolution texels; algebraically identical to the
 * layer-by-layer entry points above (L0 is linear, so it folds into its consumers), fp32 re-association only. */

/* Floats nlt_front_pack_weights() writes. */
long nlt_front_packed_floats(void);

/* Folds L0 of both nets into L1's stride-2 convs and into the head, and lays L1's four kernels out as MFMA
 * fragments.  All inputs Keras layout: wq0 (1,1,5,16) wo0 (1,1,3,16) [convnet.py:44]; query L1 wqa (2,2,32,16),
 * wqb (2,2,16,16); obs L1 woa, wob (2,2,16,16) [convnet.py:50-59]; head wh (1,1,36,3) [convnet.py:85].
 * Call again whenever any of them changes. */
int nlt_front_pack_weights(const float* wq0, const float* bq0, const float* wo0, const float* bo0,
                           const float* wqa, const float* bqa, const float* wqb, const float* bqb,
                           const float* woa, const float* boa, const float* wob, const float* bob,
                           const float* wh, const float* bh, float* packed, void* stream);

/*
 * Layers 0 and 1 of both paths + both observation means, from the raw texel buffers, in one pass.
 *   replaces: nlt/models/nlt.py:95-96 (input assembly) and the i = 0, 1 iterations of Model._call's layer loop
 *             (nlt.py:153-180: obs layers, reduce_mean, query layers, concat), i.e. nlt_stem_forward +
 *             4x nlt_conv_forward + nlt_obs_mean_forward of the unfused plan.
 * base [n,h,w,3], cvis/lvis [n,h,w,1], nn_rgb/nn_base [n,k,h,w,3] (h, w even).  Writes
 *   fm1   [n,h/2,w/2,32]    = [query L1 | mean_k obs L1]
 *   obs1  [n,k,h/2,w/2,16]  per-observation L1 maps
 *   skip3 [n,h,w,3]         = head weights applied to the (never materialised) L0 features + head bias
 *                             (+ base when add_base): what nlt_back_forward adds to finish pred.
 */
int nlt_front_forward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                      const float* nn_base, int n, int k, int h, int w, const float* packed,
                      int add_base, float alpha, float* fm1, float* obs1, float* skip3, void* stream);

/* Front kernel that also runs LEVEL 2's stride-2 convs (k <= 4, h and w multiples of 4): the workgroup's 8 x 16 tile
 * of level 1 is exactly a 4 x 8 tile of level 2 (k2s2 needs no halo), so the per-observation level-1 maps never leave
 * the chip.  packed_l2 = nlt_front_pack_l2_weights(query level-2 strided kernel (2,2,32,32), obs (2,2,16,32)).
 * Outputs: fm1, skip3 as nlt_front_forward; qtmp2 [n,h/4,w/4,32], otmp2 [n,k,h/4,w/4,32] = LeakyReLU(Conv2D k2s2)
 * of fm1 / of each observation's level-1 map (the inputs of level 2's stride-1 convs).
 *   replaces, on top of nlt_front_forward: the first Conv2D + LeakyReLU of the third entries of net['query'].layers /
 *   net['obs'].layers (convnet.py:50-53). */
long nlt_front_l2_packed_floats(void);
int nlt_front_pack_l2_weights(const float* wq, const float* bq, const float* wo, const float* bo, float* packed, void* stream);
int nlt_front2_forward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                       const float* nn_base, int n, int k, int h, int w, const float* packed,
                       const float* packed_l2, int add_base, float alpha, float* fm1, float* skip3,
                       float* qtmp2, float* otmp2, void* stream);

/* Second generation of nlt_front2_forward (csrc/front4.hip): the same launch with NO workgroup barrier -- one wave =
 * one workgroup = one 4 x 16 strip of level-1 texels = one 16-texel column tile of level 2; raw rows staged in
 * wave-private LDS by row-contiguous 16-byte loads (instead of 29 twelve-byte gathers per texel), observations
 * streamed (any k), so that the staging / LDS / store phases of one wave run under the MFMAs of the other waves of
 * its SIMD.  Same inputs, outputs and arithmetic as nlt_front2_forward (bit-identical results); the five input buffers
 * must be 16-byte aligned (NLT_ERR_UNSUPPORTED otherwise: use nlt_front2_forward); 0 <= alpha <= 1 (LeakyReLU
 * evaluated as max(v, alpha v)).  waves_per_simd: reserved (0).
 *   replaces: what nlt_front2_forward replaces (nlt/models/nlt.py:95-96,153-180).
 *
 * nlt_front4_forward_u8 is the same launch reading the RESIDENT uint8 capture store instead of float batch buffers:
 * `_load_data`'s uint8 -> float64 / 255 -> float32 (nlt/datasets/nlt.py:131-136,173-181; xiuminglib
 * img.normalize_uint) happens in registers, so nlt_assemble_batch's float buffers are neither written nor read back
 * (29 B per texel instead of 116 at k = 4).  Stores as nlt_assemble_batch takes them: diffuse_store / rgb_store
 * [F,h,w,3], cvis_store / lvis_store [F,h,w] uint8; ids [n] = frame of each sample; nn_ids [n,k] = frame of each
 * observation (nn_rgb = rgb_store[nn], nn_base = diffuse_store[nn]; -1 = missing neighbour = zeros,
 * datasets/nlt.py:152-157).  w must be a multiple of 8.  Results are bit-identical to nlt_front4_forward on
 * nlt_assemble_batch's output. */
int nlt_front4_forward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                       const float* nn_base, int n, int k, int h, int w, const float* packed,
                       const float* packed_l2, int add_base, float alpha, float* fm1, float* skip3,
                       float* qtmp2, float* otmp2, int waves_per_simd, void* stream);
int nlt_front4_forward_u8(const unsigned char* diffuse_store, const unsigned char* rgb_store,
                          const unsigned char* cvis_store, const unsigned char* lvis_store,
                          const int* ids, const int* nn_ids, int n, int k, int h, int w,
                          const float* packed, const float* packed_l2, int add_base, float alpha,
                          float* fm1, float* skip3, float* qtmp2, float* otmp2, int waves_per_simd, void* stream);

/* TRAINING form of nlt_front4_forward: the same launch, which additionally keeps the maps the backward pass reads -- exactly what
 * nlt_front_forward_train keeps (obs1 [n,k,h/2,w/2,16], qtmp1 [n,h/2,w/2,16], otmp1 [n,k,h/2,w/2,16]) -- while STILL running level
 * 2's stride-2 convs (qtmp2 / otmp2, which the backward needs as stored activations anyway): the train forward then launches neither
 * the first-generation front kernel nor L2.q.s2 / L2.o.s2.
 *   replaces: the same reference lines as nlt_front4_forward, run under the GradientTape of nlt/trainvali.py:272-274. */
int nlt_front4_forward_train(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                             const float* nn_base, int n, int k, int h, int w, const float* packed,
                             const float* packed_l2, int add_base, float alpha, float* fm1, float* skip3,
                             float* qtmp2, float* otmp2, float* obs1, float* qtmp1, float* otmp1, void* stream);

/*
 * One expanding block in one launch (inference): Conv2DTranspose k2s2 (cx + cs -> c) + LeakyReLU on the virtual concat
 * [x | skip], Conv2DTranspose k2s1 (c -> c) + LeakyReLU; the intermediate map stays in LDS.  c = 8 or 16 (the blocks at
 * 1/2 and 1/4 resolution, where the intermediate's HBM round trip costs more than the block's output), cx and cs
 * multiples of 4.
 *   replaces: one `Sequential[iden, deconv(2,n,s2), iden, lrelu, deconv(2,n,s1), iden, lrelu]` entry of
 *             net['query'].layers (convnet.py:67-76) as Model._call runs it (nlt.py:182-195), i.e. 2x nlt_conv_forward.
 * x [n,h,w,cx], skip [n,h,w,cs]; Keras weights w_s2 (2,2,c,cx+cs), w_s1 (2,2,c,c); out [n,2h,2w,c].
 */
int nlt_dec_block_forward(const float* x, int cx, const float* skip, int cs, int n, int h, int w,
                          const float* w_s2, const float* b_s2, const float* w_s1, const float* b_s1,
                          int c, float alpha, float* out, void* stream);

/*
 * Last expanding block + output head: Conv2DTranspose k2s2 (8 + 32 -> 4) + LeakyReLU, Conv2DTranspose k2s1
 * (4 -> 4) + LeakyReLU, 1x1 head on those 4 channels + skip3, texel (0,0) of every frame forced to 0.
 *   replaces: the last two entries of net['query'].layers (convnet.py:67-76,85) as Model._call runs them
 *             (nlt.py:182-195) and nlt.py:99-102,110, i.e. 2x nlt_conv_forward + nlt_head_forward.
 * x [n,h2,w2,8] (previous decoder output), fm1 [n,h2,w2,32] (the popped skip), skip3 [n,2h2,2w2,3];
 * Keras weights w_s2 (2,2,4,40), w_s1 (2,2,4,4); w_head = first 4 input rows of the (1,1,36,3) head kernel.
 * pred [n,2h2,2w2,3].
 */
int nlt_back_forward(const float* x, const float* fm1, const float* skip3, int n, int h2, int w2,
                     const float* w_s2, const float* b_s2, const float* w_s1, const float* b_s1,
                     const float* w_head, float alpha, float* pred, void* stream);

/*
 * Refresh every packed weight buffer of a model with ONE launch (csrc/repack.hip).  After an optimizer step the train
 * loop needs the fragment arrays of every conv again (nlt_pack_conv_weights for the forward family, the same for the
 * adjoint family of each backward-data launch, nlt_pack_conv_tile_weights); the buffers keep their addresses, and a
 * table of descriptors in DEVICE memory tells the kernel how to refill each from the Keras-layout arrays.
 *   replaces: nothing in the reference (Keras reads its variables in place); it is the packed-layout bookkeeping of
 *             this library, per step instead of per layer.
 * kind NLT_REPACK_MFMA: dst = what nlt_pack_conv_weights(mode, src, c0, c1, cout) writes, with `src` read as the slice
 *   [lo, lo + cout) (conv families: of the output-channel axis; transposed families: of their (kh,kw,Cout,Cin) array's
 *   Cout axis) of an array whose sliced axis has `full` entries -- lo = 0, full = cout for a whole kernel.  This is how
 *   backward-data w.r.t. input channels [lo, hi) of a forward layer reads that layer's own array as the adjoint family.
 * kind NLT_REPACK_TILE: dst = what nlt_pack_conv_tile_weights(mode, src, cin = c0, cout, tn) writes.
 * kind NLT_REPACK_WINO: dst = what nlt_pack_conv_wino_weights[_adjoint](mode, src, cin = c0, cout, tn, full, lo) writes.
 * first_block: exclusive prefix sum of ceil(total / 256) over the table; total_blocks = its grand total.
 */
typedef enum { NLT_REPACK_MFMA = 0, NLT_REPACK_TILE = 1, NLT_REPACK_WINO = 2 } nlt_repack_kind;
typedef struct {
  const float* src;
  float* dst;
  long total;          /* floats in dst */
  long first_block;
  int kind, mode;
  int c0, c1, cout;
  int tn;
  int lo, full;
} nlt_repack_desc;
int nlt_repack_weights(const nlt_repack_desc* descs_device, int n_desc, long total_blocks, void* stream);

/*
 * Native replay of a recorded launch tape.  A plan issues the same C calls with the same arguments every step; the host mirror
 * records them once (entry point + resolved arguments) and replays the array with ONE call instead of one ctypes call per launch
 * (~7 us each from Python: 1.3-1.7 ms per train step, which made the released 512^2 training shape host-bound).
 * nlt_tape_call: fn = an entry point of this library whose parameters are all integer-class (int, long, pointer) or float, with at
 * most 32 of the former and 4 of the latter; iargs / fargs = its integer-class / float arguments, each in declaration order (the
 * x86-64 System V convention assigns the two classes to registers independently).  Calls run in order; the first non-zero status
 * stops the replay and is returned (failed_index = its position).  nlt_event_record / nlt_stream_wait_event are hipEventRecord /
 * hipStreamWaitEvent as tape-able entry points (the plan's second-stream hand-overs).
 *   replaces: nothing in the reference (TF eager re-dispatches every op every step); host-side launch cost only.
 */
typedef struct {
  void* fn;
  int n_float, reserved;
  long iargs[32];
  float fargs[4];
} nlt_tape_call;
int nlt_tape_play(const nlt_tape_call* calls, int n, int* failed_index);
int nlt_event_create(int no_system_fence, void** event);   /* hipEventDisableTiming [| hipEventDisableSystemFence]: same-device stream ordering */
int nlt_event_destroy(void* event);
int nlt_event_record(void* event, void* stream);
int nlt_stream_wait_event(void* stream, void* event);

/* Weight / bias gradient for the NARROW layers (csrc/wgrad_narrow.hip): N = cout (4 * cout for Conv2DTranspose k2s2)
 * <= 32 output columns and K = taps * (c0 + c1) <= 128, modes NLT_CONV_K2S2 / K2S1 / NLT_DECONV_K2S2 / K2S1, any channel
 * count (4-byte operand loads; the MFMA tile is [K index 16] x [output channel 16], so 8 / 16 / 32-channel layers waste
 * no matrix-core work).  Same arguments, accumulation and determinism as nlt_conv_backward_weights_tiled.
 *   replaces: the same tape.gradient terms (nlt/trainvali.py:279) as nlt_conv_backward_weights.
 * nlt_wgrad_narrow_workspace_floats(): workspace size in floats, -1 when the layer is not narrow / unsupported. */
long nlt_wgrad_narrow_workspace_floats(int mode, int c0, int c1, int n, int h, int w, int cout);
int nlt_conv_backward_weights_narrow(int mode,
                                     const float* src0, int ld0, int c0, const float* src1, int ld1, int c1,
                                     int n, int h, int w, const float* dpre, int ldp, int cout,
                                     float* dw_keras, float* dbias, float* workspace, long workspace_floats,
                                     void* stream);

/*
 * TRAINING forms of the fused ends.  The forward passes are the same launches as nlt_front_forward / nlt_back_forward
 * and additionally keep what the backward pass needs:
 *   qtmp1 [n,h/2,w/2,16], otmp1 [n,k,h/2,w/2,16] = LeakyReLU(Conv2D k2s2) of level 1 (inputs of its stride-1 convs);
 *   u [n,2h2,2w2,4] = LeakyReLU(Conv2DTranspose k2s2), v [n,2h2,2w2,4] = LeakyReLU(Conv2DTranspose k2s1) of the last block.
 *   replaces (train mode): the same reference lines as the inference forms, run under the GradientTape of
 *             nlt/trainvali.py:272-274.
 */
int nlt_front_forward_train(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                            const float* nn_base, int n, int k, int h, int w, const float* packed,
                            int add_base, float alpha, float* fm1, float* obs1, float* skip3,
                            float* qtmp1, float* otmp1, void* stream);
int nlt_back_forward_train(const float* x, const float* fm1, const float* skip3, int n, int h2, int w2,
                           const float* w_s2, const float* b_s2, const float* w_s1, const float* b_s1,
                           const float* w_head, float alpha, float* pred, float* u, float* v, void* stream);

/*
 * Backward of layers 0-1's linear part without any full-resolution feature tensor (csrc/train_fused.hip):
 * L0 is a 1x1 conv with no activation, so the weight gradients of L0 (both paths), of level 1's two stride-2 convs
 * and of the head's 32 skip rows are small matrix products with texel sums of (raw channels x gradients), which one
 * pass over the raw buffers forms on the matrix cores (deterministic two-pass reduction).
 *   replaces: tape.gradient (nlt/trainvali.py:279) through nlt/models/nlt.py:95-96 and the i = 0, 1 layer iterations
 *             (nlt.py:153-180) for those weights, i.e. of the unfused plan: nlt_stem_backward, the skip half of
 *             nlt_head_backward, nlt_conv_backward_weights + backward-data of L1.{q,o}.s2.
 * dy1q [n,h/2,w/2,16] / dy1o [n,k,h/2,w/2,16]: gradients w.r.t. the PRE-activation outputs of level 1's stride-2
 * convs; dpred [n,h,w,3] (texel (0,0) ignored: set_left_top_corner).  h even, w a multiple of 8.
 * Weights in Keras layouts: wq0 (1,1,5,16), wo0 (1,1,3,16), wqa (2,2,32,16), woa (2,2,16,16), wh (1,1,36,3).
 * Gradients are ACCUMULATED (+=): dwq0, dbq0, dwo0, dbo0, dwqa, dbqa, dwoa, dboa and rows 4..35 of dwh.
 * workspace: nlt_front_backward_workspace_floats(n, h, w) floats (-1: unsupported shape).
 */
long nlt_front_backward_workspace_floats(int n, int h, int w);
int nlt_front_backward(const float* base, const float* cvis, const float* lvis, const float* nn_rgb,
                       const float* nn_base, int n, int k, int h, int w, const float* dy1q, const float* dy1o,
                       const float* dpred, const float* wq0, const float* bq0, const float* wo0,
                       const float* bo0, const float* wqa, const float* woa, const float* wh,
                       float* dwq0, float* dbq0, float* dwo0, float* dbo0, float* dwqa, float* dbqa,
                       float* dwoa, float* dboa, float* dwh, float* workspace, void* stream);

/*
 * Backward of the last expanding block + head in one pass (csrc/train_back.hip), from the maps nlt_back_forward_train kept.
 *   replaces: tape.gradient (nlt/trainvali.py:279) through the last two entries of net['query'].layers
 *             (convnet.py:67-76,85; nlt.py:182-195) and nlt.py:99-102,110, i.e. of the unfused plan: nlt_head_backward (decoder
 *             rows), nlt_lrelu_backward, 2x nlt_conv_backward_weights and 3x backward-data nlt_conv_forward.
 * x [n,h2,w2,8], fm1 [n,h2,w2,32] (the block's two inputs), u / v [n,2h2,2w2,4] (its two LeakyReLU outputs), dpred
 * [n,2h2,2w2,3] (texel (0,0) ignored).  Keras weights w_s2 (2,2,4,40), w_s1 (2,2,4,4), w_head (1,1,36,3) (rows 0..3 used).
 * Writes dx [n,h2,w2,8] and dfm1 [n,h2,w2,32]: gradients w.r.t. the two inputs; x is the previous block's LeakyReLU(alpha)
 * output and dx is handed back w.r.t. that block's PRE-activation (times 1 or alpha by the sign of x); dfm1 is plain.
 * ACCUMULATES (+=) dw_s2, db_s2, dw_s1, db_s1, rows 0..3 of dw_head, db_head.
 * workspace: nlt_back_backward_workspace_floats(n, h2, w2) floats.
 */
long nlt_back_backward_workspace_floats(int n, int h2, int w2);
int nlt_back_backward(const float* x, const float* fm1, const float* u, const float* v, const float* dpred,
                      int n, int h2, int w2, const float* w_s2, const float* w_s1, const float* w_head,
                      float alpha, float* dx, float* dfm1, float* dw_s2, float* db_s2, float* dw_s1,
                      float* db_s1, float* dw_head, float* db_head, float* workspace, void* stream);

/* nlt_conv_forward on the MFMA path with the K loop split into `ksplit` slices run by different waves (for
 * the deep levels: a few hundred texels x thousands of input channels would otherwise occupy a fraction of the
 * chip).  ONE launch (r06): a workgroup adds four slices in LDS; with more than four slices the groups' partial tiles go to
 * `workspace` and the workgroup that arrives last at a tile adds them in group order and applies the usual epilogue -- the
 * order of additions depends on the launch shape only (bit-reproducible).  `workspace` = nlt_conv_splitk_workspace_floats()
 * floats, ZEROED ONCE by the caller when it is allocated (its first 16384 words are the tiles' ticket counters; every launch
 * leaves them zero) and used by one stream at a time.  ksplit = 1 is nlt_conv_forward(algo = MFMA).
 * ksplit < 0: |ksplit| slices in the TWO-launch form of rounds 2-5 (every slice a wave of its own, a second launch adds the
 * slices in slice order and applies the epilogue; same workspace contract, nlt_conv_splitk_workspace_floats(.., ksplit) floats):
 * the faster form where a handful of GEMM rows meet 32-128 slices; callers choose by timing.  ksplit = 0 is an error. */
long nlt_conv_splitk_workspace_floats(int mode, int n, int h, int w, int cout, int ksplit);
int nlt_conv_forward_splitk(int mode, int tile_hint, int ksplit, float* workspace,
                            const float* src0, int ld0, int c0, const float* src1, int ld1, int c1,
                            int n, int h, int w, const float* w_packed, const float* bias,
                            int cout, float* out, int ldo, int act, float alpha,
                            const float* mask_src, int ldm, int accumulate, void* stream);

/*
 * The reference's INFERENCE mode: Model.call(batch, 'test', obs_override=feat_agg) (nlt/nlt_test.py:78-94,
 * nlt/models/nlt.py:154-155,172-174).  Every level's aggregated observation map is GIVEN -- one [1,h,w,C] map shared by all
 * frames -- and concatenated behind the query features, so what it adds to the next conv's pre-activation,
 * W[o rows] * ovr + b, is linear and frame-independent: the host evaluates it once per feat_agg (an "override map") and the
 * per-frame convs read it where a bias would be added.
 *
 * nlt_conv_forward_map = nlt_conv_forward_splitk (MFMA path, optional split-K) with the per-output-texel bias map
 * bias_map [map_frames, oh, ow, cout] (dense; map_frames = 1: shared by all n frames, or n) added next to `bias` before
 * the activation:  out = act(conv(src0 | src1) + bias + bias_map).
 */
int nlt_conv_forward_map(int mode, int tile_hint, int ksplit, float* workspace,
                         const float* src0, int ld0, int c0, const float* src1, int ld1, int c1,
                         int n, int h, int w, const float* w_packed, const float* bias,
                         int cout, float* out, int ldo, int act, float alpha,
                         const float* bias_map, int map_frames, void* stream);

/* The two fused tail launches of that mode, reading the QUERY half of the interleaved encoder map only:
 *   nlt_dec_block_forward_map = nlt_dec_block_forward for the U-Net's widths (x [n,h,w,2c], skip = [query 4c | given 4c] with
 *     per-texel stride lds), w_s2q = the Keras (2,2,c,6c) slice of the first conv over [x | query]; bias_map [1,2h,2w,c] = its
 *     given-half rows * the given map + its bias.  c = 8 or 16.
 *   nlt_back_forward_map = nlt_back_forward with q1 = the 16 query channels of the level-1 map (stride ldq), w_s2q = the Keras
 *     (2,2,4,24) slice over [x 8 | query 16], bias_map [1,2 h2,2 w2,4].  The head's share of L0 arrives through skip3 as before. */
int nlt_dec_block_forward_map(const float* x, const float* skip, int lds, int n, int h, int w,
                              const float* w_s2q, const float* w_s1, const float* b_s1, int c, float alpha,
                              const float* bias_map, float* out, void* stream);
int nlt_back_forward_map(const float* x, const float* q1, int ldq, const float* skip3, int n, int h2, int w2,
                         const float* w_s2q, const float* w_s1, const float* b_s1, const float* w_head, float alpha,
                         const float* bias_map, float* pred, void* stream);

/* Query-only fused front launch of that mode (csrc/front_ovr.hip): layers 0-1 of the query path and level 2's stride-2
 * conv from the raw texel buffers base [n,h,w,3], cvis / lvis [n,h,w,1] (nlt/models/nlt.py:95) --
 *   y1 = lrelu(fold(L0, L1.s2)[query rows] * raw5 + p1)          p1 [h/2,w/2,16]: L1.s2's observation rows * ovr0 + folded bias
 *   q1 = lrelu(L1.s1 * y1 + b)                                   -> q1 [n,h/2,w/2,16], per-texel stride ldq
 *   qtmp2 = lrelu(L2.s2[query rows] * q1 + p2)                   p2 [h/4,w/4,32]: L2.s2's observation rows * ovr1 + bias
 *   skip3 = head[L0 query rows, folded] * raw5 + s0 (+ base)     s0 [h,w,4] (3 used): head's observation rows * ovr0 + folded bias
 * `packed` = nlt_front_pack_weights with a ZERO observation L0 (its observation rows and bias terms then vanish),
 * `packed_l2` = nlt_front_pack_l2_weights.  h, w multiples of 4; 0 <= alpha <= 1; 16-byte aligned arrays. */
int nlt_front_ovr_forward(const float* base, const float* cvis, const float* lvis, int n, int h, int w,
                          const float* packed, const float* packed_l2, const float* p1, const float* s0,
                          const float* p2, int add_base, float alpha, float* q1, int ldq, float* skip3,
                          float* qtmp2, void* stream);
/* The same launch on a STORE-RESIDENT batch (nlt/datasets/nlt.py:131-136,173-181 keep the capture as uint8 images; this run's
 * Dataset keeps them in HBM): frame ids[i] of the uint8 stores diffuse [F,h,w,3], cvis / lvis [F,h,w] instead of float
 * buffers -- `_load_data`'s `/ 255` happens in registers, as in nlt_front4_forward_u8.  w a multiple of 8.  <= 3e-7 relative
 * from nlt_front_ovr_forward on the assembled batch (fl(W / 255) . u against W . fl(u / 255)). */
int nlt_front_ovr_forward_u8(const unsigned char* diffuse_store, const unsigned char* cvis_store,
                             const unsigned char* lvis_store, const int* ids, int n, int h, int w,
                             const float* packed, const float* packed_l2, const float* p1, const float* s0,
                             const float* p2, int add_base, float alpha, float* q1, int ldq, float* skip3,
                             float* qtmp2, void* stream);

/*
 * Backward-data of one conv: the gradient w.r.t. the layer's input channels from the gradient w.r.t. its pre-activation
 * output dpre [n,h,w,cpre] (stride ldp), as the ADJOINT conv family on the same Keras array (CONV_K2Sx <-> DECONV_K2Sx;
 * w_packed = nlt_pack_conv_weights(adj_mode, slice of the forward kernel), zero_bias = cout zeros), MFMA path, optional
 * split-K (ksplit > 1: workspace of nlt_conv_splitk_workspace_floats(adj_mode, n, h, w, cout, ksplit) floats).
 * Epilogue, per output element:  v (+= out when accumulate);
 *   mask_src != NULL: v *= LeakyReLU'(mask_src) -- the result is then the gradient w.r.t. the PRODUCER's pre-activation;
 *   split_c > 0 (the target is dfm[l] = [query c | observation-mean c], ONE observation per frame): channels >= split_c are
 *     not stored to out; (v + (split_partial ? split_d : 0)) * LeakyReLU'(split_y) goes to split_d [rows, split_c] -- the
 *     gradient w.r.t. the observation path's pre-activation.  The tf.reduce_mean adjoint of nlt/models/nlt.py:161-164 and
 *     both activations' derivatives thus cost no pass of their own (nlt_level_split_backward's work).
 *   replaces: the input-gradient half of tf.GradientTape.gradient through Conv2D / Conv2DTranspose (nlt/trainvali.py:279).
 */
int nlt_conv_backward_data(int adj_mode, int tile_hint, int ksplit, float* workspace,
                           const float* dpre, int ldp, int cpre, int n, int h, int w,
                           const float* w_packed, const float* zero_bias, int cout, float* out, int ldo,
                           const float* mask_src, int ldm, float mask_alpha, int accumulate,
                           int split_c, const float* split_y, float* split_d, float split_alpha, int split_partial,
                           void* stream);

/* ======================= Winograd F(2x2, 2x2) stride-1 k2 convs (csrc/conv_wino.hip) =======================
 * The same results as nlt_conv_forward(NLT_CONV_K2S1 / NLT_DECONV_K2S1) / nlt_conv_tile_forward up to fp32 re-association
 * (nlt/networks/elements.py:26-39: Conv2D / Conv2DTranspose(kernel 2, stride 1, 'same') + bias [+ LeakyReLU]): a 2 x 2 block of
 * outputs from 9 instead of 16 products per channel pair (Y = A^T[(G g G^T) (.) (B^T d B)]A), exact fp32 products on
 * v_mfma_f32_16x16x4_f32, fp32 accumulation.  cin % 8 == 0, cout % tn == 0, tn = 32 | 64.
 *   replaces: the stride-1 conv of every `Sequential[conv(2,n,s2), ..., conv(2,n,s1), ...]` / `[..., deconv(2,n,s1), ...]` block
 *             of net['query'] / net['obs'].layers (convnet.py:50-76) as Model._call runs it (nlt.py:154-195).
 * src [frames*kobs, h, w, ld >= cin]; packed = nlt_pack_conv_wino_weights(mode, Keras array, cin, cout, tn)
 * (nlt_conv_wino_packed_floats floats); out [frames*kobs, h, w, ldo] (may be NULL when only the mean is wanted);
 * mean_out [frames, h, w, ldm]: mean over the kobs observation frames of a frame, kept in registers (tn = 32, NLT_CONV_K2S1 only:
 * kobs > 1 or mean_out with tn = 64 returns NLT_ERR_UNSUPPORTED). */
long nlt_conv_wino_packed_floats(int mode, int cin, int cout, int tn);
int nlt_pack_conv_wino_weights(int mode, const float* w_keras, int cin, int cout, int tn, float* packed, void* stream);
int nlt_conv_wino_forward(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                          const float* packed, const float* bias, int cout, int tn,
                          float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream);
/* Backward-data on the Winograd kernel: gradient w.r.t. input channels [lo, lo + cout) of a stride-1 k2 layer from the gradient
 * dpre [n,h,w,ldp >= cpre] w.r.t. its pre-activation output (GradientTape through Conv2D / Conv2DTranspose, nlt/trainvali.py:279);
 * adj_mode = the adjoint family (NLT_DECONV_K2S1 for a forward Conv2D, NLT_CONV_K2S1 for a forward Conv2DTranspose), packed =
 * nlt_pack_conv_wino_weights_adjoint(adj_mode, the layer's own Keras array, cpre, cout, tn, full = its input-channel extent, lo).
 * Epilogue as nlt_conv_tile_backward_data: out (+= when accumulate), then x LeakyReLU'(mask_src) with slope mask_alpha. */
int nlt_pack_conv_wino_weights_adjoint(int adj_mode, const float* w_keras, int cpre, int cout, int tn, int full, int lo,
                                       float* packed, void* stream);
int nlt_conv_wino_backward_data(int adj_mode, const float* dpre, int ldp, int cpre, int n, int h, int w,
                                const float* packed, int cout, int tn, float* out, int ldo,
                                const float* mask_src, int ldm, float mask_alpha, int accumulate, void* stream);

/* Conv2D k2s1 'same' + bias [+ LeakyReLU] [+ observation mean] of the narrow levels (cin = 16 | 32, cout = 32: level 2 of both
 * paths) with the layer's weights resident in LDS and a whole frame per stage (csrc/conv_c32.hip); arguments and results as
 * nlt_conv_tile_forward with tn = 32, packed = nlt_pack_conv_tile_weights(NLT_CONV_K2S1, w, cin, 32, 32).
 *   replaces: the stride-1 conv of `Sequential[conv(2,n,s2), ..., conv(2,n,s1), ...]` at n = 32 (convnet.py:50-59). */
int nlt_conv_c32_supported(int mode, int cin, int cout);
int nlt_conv_c32_forward(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                         const float* packed, const float* bias, int cout,
                         float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream);

/* ======================= LDS-tiled encoder convs (csrc/conv_tile.hip) =======================
 * Same arithmetic as nlt_conv_forward for mode NLT_CONV_K2S2 / NLT_CONV_K2S1 with a single source (bias +
 * optional LeakyReLU), laid out for the MFMA-bound levels: 8 x 16 output tile x tn output channels per
 * workgroup, input slab + weight fragments staged through double-buffered LDS.
 *   replaces: the Conv2D + LeakyReLU pairs of the contracting blocks (nlt/networks/convnet.py:50-59) and,
 *             with kobs > 1 and mean_out, the observation mean of nlt/models/nlt.py:161-164.
 * cin % 16 == 0, tn in {32, 64}, cout % tn == 0.  src holds frames*kobs frames [h,w,ld]; out (may be NULL)
 * gets frames*kobs frames [oh,ow,ldo]; mean_out (may be NULL) gets, per frame, the mean over its kobs
 * consecutive source frames at stride ldm. */
long nlt_conv_tile_packed_floats(int mode, int cin, int cout, int tn);
int nlt_pack_conv_tile_weights(int mode, const float* w_keras, int cin, int cout, int tn, float* packed, void* stream);
int nlt_conv_tile_forward(int mode, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                          const float* packed, const float* bias, int cout, int tn,
                          float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream);

/*
 * Backward-data on the LDS-tiled kernel (csrc/conv_tile.hip): the gradient w.r.t. a conv's input channels [lo, lo + cout) from the
 * gradient w.r.t. its pre-activation output dpre [n,h,w,cpre], as the adjoint conv family adj_mode on the layer's own Keras array:
 *   NLT_CONV_K2S1 / NLT_CONV_K2S2  (the layer is a Conv2DTranspose: its (kh,kw,Cout,Cin) array read as a conv Cout -> Cin slice);
 *   NLT_DECONV_K2S1                (the layer is a Conv2D k2s1: its (kh,kw,Cin,Cout) array read as the transposed conv, halo on
 *                                   the top / left of the tile);
 *   NLT_DECONV_K2S2                (the layer is a Conv2D k2s2: a 1x1-conv-shaped GEMM over the tile of dpre texels with N = 4 * cout
 *                                   columns (a, b, channel) and a scatter store; cpre % 32 == 0, cout % 16 == 0, 4 * cout % tn == 0;
 *                                   this mode also takes the level-split epilogue arguments of nlt_conv_backward_data).
 * packed = nlt_pack_conv_tile_weights_adjoint(adj_mode, layer array, cpre, cout, tn, full = the array's input-channel extent, lo)
 * (nlt_conv_tile_packed_floats(adj_mode, cpre, cout, tn) floats; cpre % 16 == 0, cout % tn == 0).  Epilogue as
 * nlt_conv_backward_data (split_c = 0 unless adj_mode is NLT_DECONV_K2S2): v (+= out when accumulate), times LeakyReLU'(mask_src) when mask_src != NULL.
 *   replaces: the input-gradient half of tf.GradientTape.gradient through Conv2D / Conv2DTranspose (nlt/trainvali.py:279).
 */
int nlt_pack_conv_tile_weights_adjoint(int adj_mode, const float* w_keras, int cpre, int cout, int tn, int full, int lo,
                                       float* packed, void* stream);
int nlt_conv_tile_backward_data(int adj_mode, const float* dpre, int ldp, int cpre, int n, int h, int w,
                                const float* packed, int cout, int tn, float* out, int ldo,
                                const float* mask_src, int ldm, float mask_alpha, int accumulate,
                                int split_c, const float* split_y, float* split_d, float split_alpha, int split_partial,
                                void* stream);

/*
 * `precision = f32x3` form of nlt_conv_tile_forward (csrc/conv_tile3.hip): the same convs with every fp32 operand split exactly
 * into three bf16 terms (hi + mid + lo) and the term products -- each exact in fp32 -- accumulated in fp32 on
 * v_mfma_f32_16x16x32_bf16.  nprod = 9: all nine term products (error = fp32 accumulation rounding, as the native fp32 MFMA);
 * nprod = 6: the three products of relative order 2^-24 dropped (nprod = 3 / 1: only the products down to 2^-16 / hi * hi alone --
 * the precision ladder's lower rungs, for measurement).  Same arguments and semantics as nlt_conv_tile_forward;
 * `packed` = nlt_pack_conv_tile3_weights (nlt_conv_tile3_packed_elems() bf16 elements: the kernel's three terms in fragment order).
 * An explicit inference / forward mode reported beside the native fp32 path, never instead of it.
 *   replaces: the same Conv2D (+ LeakyReLU, + tf.reduce_mean over observations) lines as nlt_conv_tile_forward.
 */
long nlt_conv_tile3_packed_elems(int mode, int cin, int cout, int tn);
int nlt_pack_conv_tile3_weights(int mode, const float* w_keras, int cin, int cout, int tn, unsigned short* packed, void* stream);
int nlt_conv_tile3_forward(int mode, int nprod, const float* src, int ld, int cin, int frames, int kobs, int h, int w,
                           const unsigned short* packed, const float* bias, int cout, int tn,
                           float* out, int ldo, float* mean_out, int ldm, int act, float alpha, void* stream);

/* ======================= bf16 channel mix (csrc/chmix_bf16.hip) =======================
 * 1x1 conv with bf16 activations / weights and fp32 accumulation on v_mfma_f32_16x16x32_bf16: the literal dense GEMM
 * of the path, used as the HBM-roofline stress point of BASELINE config 5 (2048^2 UV, bf16; SURVEY.md 8d "x64-ch").
 *   out[t][o] = act(sum_c x[t][c] * W[c][o] + b[o]);  x [texels,cin], out [texels,cout] bf16 (uint16 bit patterns),
 *   bias fp32; cin, cout in {32, 64, 128}.  W comes from a Keras (1,1,cin,cout) fp32 kernel through
 *   nlt_chmix_bf16_pack (rounded to bf16, round-to-nearest-even).
 *   replaces: tf.keras Conv2D(kernel_size=1) as built by nlt/networks/elements.py:26-31 (convnet.py:44,85), at a
 *   64-channel width and in bf16 -- a shape the released network does not contain (its 1x1 layers are 5->16 and 36->3).
 * Parity: against the oracle on bf16-rounded operands, within 1 bf16 ulp of the output. */
long nlt_chmix_bf16_packed_elems(int cin, int cout);
int nlt_chmix_bf16_pack(const float* w_keras, int cin, int cout, unsigned short* packed, void* stream);
int nlt_chmix_bf16_forward(const unsigned short* x, long texels, int cin, const unsigned short* packed,
                           const float* bias, int cout, int act, float alpha, unsigned short* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NLT_HIP_H_ */
