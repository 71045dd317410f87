/*
 * libgan_hip -- C ABI of the MI355X (gfx950) GAN training hot path.
 *
 * The reference (Youngjung/GAN) has no FFI: its layer primitives are Python
 * functions in models/ops.py that hand the arithmetic to TensorFlow-1.x ops.
 * Each entry point below replaces the TensorFlow op(s) named in its comment,
 * cited as the reference call site (file:line under the reference root).
 *
 * Conventions
 *   - All tensors are dense fp32, NHWC, in device memory owned by the caller.
 *   - The library never allocates or frees device memory and never
 *     synchronises; every launch is asynchronous on the caller's stream.
 *   - `ws` is a caller-provided scratch buffer of at least the number of bytes
 *     the matching *_workspace_bytes() query returns (it may be NULL when that
 *     query returns 0).
 *   - Return value 0 = success; non-zero = error, text in gan_last_error()
 *     (thread-local).  Nothing is launched when an argument check fails.
 *   - Results are deterministic: no floating-point atomics are used.
 */
#ifndef GAN_HIP_H
#define GAN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* gan_stream_t; /* a hipStream_t */

/* Geometry shared by conv2d and conv2d_transpose.  "big" is the strided side
 * (conv2d input / conv2d_transpose output), "small" the other side
 * (conv2d output / conv2d_transpose input).  The filter is always
 * [kh][kw][cb][cs]: that is HWIO for conv2d (ops.py:66) and
 * [height, width, output_channels, in_channels] for conv2d_transpose
 * (ops.py:80-81).  pt/pl are TensorFlow's pad_before for rows/columns
 * (SAME: ((ceil(hb/sh)-1)*sh + kh - hb)/2 rounded down; VALID: 0). */
typedef struct gan_conv_geom {
    int n;          /* batch */
    int hb, wb, cb; /* big side: height, width, channels */
    int hs, ws, cs; /* small side */
    int kh, kw;     /* filter taps */
    int sh, sw;     /* strides (1 or 2) */
    int pt, pl;     /* pad before (top, left) */
} gan_conv_geom;

/* Activation selectors for the fused epilogues. */
enum {
    GAN_ACT_NONE = 0,
    GAN_ACT_RELU = 1,    /* tf.nn.relu   (dcgan.py:368-380)            */
    GAN_ACT_LRELU = 2,   /* ops.lrelu    (ops.py:101-102), param = leak */
    GAN_ACT_TANH = 3,    /* tf.nn.tanh   (dcgan.py:385)                */
    GAN_ACT_SIGMOID = 4  /* tf.nn.sigmoid(dcgan.py:334,409)            */
};

const char* gan_last_error(void);
int gan_abi_version(void);

/* ---- tf.nn.conv2d + tf.nn.bias_add (ops.py:68-71) and their gradients ------------- */
size_t gan_conv_workspace_bytes(const gan_conv_geom* g);
/* y[small] = conv(x[big], w) + bias (bias may be NULL) */
int gan_conv2d_fwd(const gan_conv_geom* g, const float* x, const float* w, const float* bias,
                   float* y, void* ws, size_t ws_bytes, gan_stream_t stream);
/* gan_conv2d_fwd for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The split-K
 * slabs live in a 64 MiB device array built into the library (one copy per GPU the library runs on, held
 * until the process ends) instead of a caller workspace; the call itself allocates and frees nothing.
 * Results are bit-identical to gan_conv2d_fwd with the same plan.  A geometry whose slabs need more than
 * 64 MiB is refused with an error (use gan_conv2d_fwd).  All calls on one GPU share that GPU's array:
 * calls that may overlap on one GPU (different streams, different host threads) race and must be
 * serialised by the caller.  The first call on each device looks the array's address up and should be
 * made outside graph capture. */
int gan_conv2d_fwd_static_ws(const gan_conv_geom* g, const float* x, const float* w, const float* bias,
                             float* y, gan_stream_t stream);
/* dx[big] = Conv2DBackpropInput(dy[small], w) */
int gan_conv2d_dgrad(const gan_conv_geom* g, const float* dy, const float* w, float* dx,
                     void* ws, size_t ws_bytes, gan_stream_t stream);
/* dw (+)= Conv2DBackpropFilter(x[big], dy[small]); db (+)= sum over pixels of dy (db may be NULL).
 * accumulate != 0 adds into dw/db instead of overwriting (shared weights used twice in a step). */
int gan_conv2d_wgrad(const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                     int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream);

/* ---- tf.nn.conv2d_transpose + bias_add (ops.py:85-94) and their gradients --------- */
/* y[big] = conv2d_transpose(x[small], w) + bias */
int gan_deconv2d_fwd(const gan_conv_geom* g, const float* x, const float* w, const float* bias,
                     float* y, void* ws, size_t ws_bytes, gan_stream_t stream);
/* dx[small] = conv2d(dy[big], w) */
int gan_deconv2d_dgrad(const gan_conv_geom* g, const float* dy, const float* w, float* dx,
                       void* ws, size_t ws_bytes, gan_stream_t stream);
/* dw (+)= filter gradient with dy[big] as the conv input and x[small] as its output gradient;
 * db (+)= sum over pixels of dy[big] */
int gan_deconv2d_wgrad(const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                       int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream);

/* ---- tf.pad(x, REFLECT) in front of a VALID conv2d (cyclegan_module.py:106-109,124-125,143-144), folded into
 * the convolution's gather: x is the UNPADDED image, the geometry's pad_before (pt, pl) says by how many pixels
 * it is extended by reflection on each side (e.g. hb = hs, kh = 3, sh = 1, pt = 1 for the residual blocks), and
 * a tap outside the image reads the mirrored pixel instead of zero.  The input gradient (gan_conv2d_dgrad_reflect, same
 * geometry, dx the gradient of the UNPADDED image) is the transposed contraction on the padded geometry, left in the
 * workspace, and ONE kernel that sums its split-K slabs and folds the mirrored border back onto the image (the adjoint of
 * the padding); its workspace is sized by gan_conv2d_dgrad_reflect_workspace_bytes.  The two-step form, gan_conv2d_dgrad
 * on the padded geometry followed by gan_reflect_pad_bwd, gives the same values. */
int gan_conv2d_fwd_reflect(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                           void* ws, size_t ws_bytes, gan_stream_t stream);
int gan_conv2d_wgrad_reflect(const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                             int accumulate, void* ws, size_t ws_bytes, gan_stream_t stream);
size_t gan_conv2d_dgrad_reflect_workspace_bytes(const gan_conv_geom* g);
int gan_conv2d_dgrad_reflect(const gan_conv_geom* g, const float* dy, const float* w, float* dx,
                             void* ws, size_t ws_bytes, gan_stream_t stream);

/* ---- conv / deconv with the activation that follows it fused into the store ------------------
 * The reference applies an activation directly to a conv output in its layers without batch norm:
 * lrelu(conv2d(image)) (dcgan.py:328, cyclegan_module.py:20), tanh(deconv2d(h3)) (dcgan.py:385),
 * sigmoid(deconv2d(h2)) (dcgan.py:409), tanh(h4)/2. + 0.5 (text2image.py:198), tanh(conv2d) (cyclegan_module.py:145).
 *   *_fwd_act:      y = act(conv + bias) * act_scale + act_shift, in the contraction's epilogue.
 *   *_dgrad_gated:  the input gradient of the NEXT layer, multiplied on store by the activation derivative
 *                   of this layer: dx = dgrad(dy, w) * d(act(.)*scale+shift)/d(.) evaluated from the saved
 *                   activated tensor `gate` (shaped like dx).  What comes out is the gradient with respect to
 *                   the conv output in front of the activation, so no activation-backward pass runs. */
int gan_conv2d_fwd_act(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                       int act, float act_param, float act_scale, float act_shift, void* ws, size_t ws_bytes,
                       gan_stream_t stream);
int gan_deconv2d_fwd_act(const gan_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                         int act, float act_param, float act_scale, float act_shift, void* ws, size_t ws_bytes,
                         gan_stream_t stream);
int gan_conv2d_dgrad_gated(const gan_conv_geom* g, const float* dy, const float* w, float* dx,
                           const float* gate, int act, float act_param, float act_scale, float act_shift,
                           void* ws, size_t ws_bytes, gan_stream_t stream);
int gan_deconv2d_dgrad_gated(const gan_conv_geom* g, const float* dy, const float* w, float* dx,
                             const float* gate, int act, float act_param, float act_scale, float act_shift,
                             void* ws, size_t ws_bytes, gan_stream_t stream);

/* ---- tf.matmul(input_, matrix) + bias (ops.py:113-115) and its gradients ----------- */
size_t gan_linear_workspace_bytes(int rows, int in_dim, int out_dim);
int gan_linear_fwd(int rows, int in_dim, int out_dim, const float* x, const float* matrix,
                   const float* bias, float* y, void* ws, size_t ws_bytes, gan_stream_t stream);
int gan_linear_dgrad(int rows, int in_dim, int out_dim, const float* dy, const float* matrix,
                     float* dx, void* ws, size_t ws_bytes, gan_stream_t stream);
int gan_linear_wgrad(int rows, int in_dim, int out_dim, const float* x, const float* dy,
                     float* dmatrix, float* dbias, int accumulate, void* ws, size_t ws_bytes,
                     gan_stream_t stream);

/* ---- tf.contrib.layers.batch_norm (ops.py:38-44), fused with the activation that follows it
 * x is viewed as [groups][rows][c]; statistics are per (group, channel) over `rows`.
 * groups == 1 is batch norm; groups == batch with rows == H*W is instance norm (an extension:
 * the reference's CycleGAN gets it implicitly from batch_norm at batch 1).
 * Training forward: y = act(gamma*(x-mean)*rsqrt(var+eps)+beta) + residual (residual may be NULL),
 * biased variance; saves mean and rstd ([groups][c] each) for the backward; when moving_mean /
 * moving_var are non-NULL updates them in place, once per group in group order (groups > 1 with
 * moving statistics = several passes through the same layer stacked along the batch axis):
 * moving = momentum*moving + (1-momentum)*batch, with the batch variance multiplied by
 * rows/(rows-1) when unbiased_moving_var != 0. */
size_t gan_bn_workspace_bytes(int groups, int rows, int c);
int gan_bn_fwd_train(const float* x, const float* gamma, const float* beta, float* moving_mean,
                     float* moving_var, const float* residual, float* y, float* save_mean,
                     float* save_rstd, int groups, int rows, int c, float eps, float momentum,
                     int unbiased_moving_var, int act, float act_param, void* ws, size_t ws_bytes,
                     gan_stream_t stream);
/* The moving-average side effect of one more training forward on the same batch, from the
 * statistics gan_bn_fwd_train saved (groups == 1): what a second sess.run of an unchanged
 * sub-graph does to moving_mean / moving_variance (dcgan.py:228-247 runs the generator once per
 * run with identical weights and z), without recomputing the activations. */
int gan_bn_moving_update(float* moving_mean, float* moving_var, const float* save_mean,
                         const float* save_rstd, int rows, int c, float eps, float momentum,
                         int unbiased_moving_var, gan_stream_t stream);
/* The same update for `count` layers with one launch per 64 of them (all with the same eps / momentum / flavour of
 * variance): the replay of a whole generator is one call.  The jobs of a launch run concurrently: no two jobs among
 * jobs[64k .. 64k+63] may name the same moving_mean / moving_var (refused); a layer whose averages move twice takes
 * two calls. */
typedef struct {
    float* moving_mean;
    float* moving_var;
    const float* save_mean;
    const float* save_rstd;
    int rows, c;
} gan_bn_moving_job;
int gan_bn_moving_update_multi(const gan_bn_moving_job* jobs, int count, float eps, float momentum,
                               int unbiased_moving_var, gan_stream_t stream);
/* Inference forward (train=False, dcgan.py:427-436): uses the moving statistics. */
int gan_bn_fwd_eval(const float* x, const float* gamma, const float* beta, const float* moving_mean,
                    const float* moving_var, float* y, int rows, int c, float eps, int act,
                    float act_param, gan_stream_t stream);
/* Backward of the training forward (without the residual term, whose gradient is dy itself):
 * dx, and dgamma/dbeta (+)= when accumulate != 0.  dx may be NULL (input needs no gradient). */
int gan_bn_bwd(const float* x, const float* gamma, const float* beta, const float* save_mean,
               const float* save_rstd, const float* dy, float* dx, float* dgamma, float* dbeta,
               int accumulate, int groups, int rows, int c, int act, float act_param, void* ws,
               size_t ws_bytes, gan_stream_t stream);

/* ---- batch norm around a conv / deconv, with fewer launches -------------------------------------------------
 * No counterpart in the reference (TensorFlow schedules conv2d, bias_add and batch_norm as separate ops; these
 * entry points compute the same values): every layer of the reference's networks except the first and last is
 * conv / deconv -> batch_norm -> activation (dcgan.py:330-333,368-380; cyclegan_module.py:21-30,106-110,126-141),
 * and at training sizes both the contraction and the batch norm reduce in two launches -- split-K slabs then their
 * sum; block statistics then normalise.  Here the middle two become one launch:
 *   forward   gan_conv_fwd_bn_stats: y = op(x, w) + bias as gan_conv2d_fwd / gan_deconv2d_fwd / gan_conv2d_fwd_reflect
 *             compute it, and the batch norm's block statistics of y in bn_partial; then
 *             gan_bn_fwd_train_from_stats = gan_bn_fwd_train without its statistics launch.
 *   backward  gan_conv_dgrad_bn_sums: dx = input gradient of op (as gan_conv2d_dgrad / gan_deconv2d_dgrad /
 *             gan_conv2d_dgrad_reflect), where the op's INPUT was the output of the batch norm (+ activation) `bn`:
 *             the block sums of that batch norm's backward over dx are left in bn_partial (the reflect variant keeps its
 *             launch that folds the mirrored border back, so there the sums are a launch of their own); then
 *             gan_bn_bwd_tiled(.., sums_partial = bn_partial, sums_ready = 1, ..) = gan_bn_bwd without its sums launch.
 *   bias      gan_bn_bwd_tiled with colsum_partial != NULL also leaves the column sums of the rows of dx each of its
 *             workgroups stored; gan_conv_wgrad_colsum = the weight gradient of the layer in front of the batch norm,
 *             taking its bias gradient db from those instead of reading dx again in a launch of its own.
 *   fan-out   gan_add_bn_sums: out = a + b (as gan_add_fwd: the gradient of a tensor with two consumers, e.g. the input
 *             of a residual block, cyclegan_module.py:106-110) where that tensor was the output of the batch norm `bn`
 *             (rows x c per group): the same launch leaves the block sums of that batch norm's backward over `out` in
 *             bn_partial, for gan_bn_bwd_tiled(.., sums_ready = 1, ..).  out may be a or b.
 * All results are bit-identical to the separate entry points, except db through column-sum partials (same terms,
 * different grouping of the fp32 sum).  `op`: GAN_OP_*.  The shapes served are those for which
 * gan_bn_tile_blocks(groups, rows, c, backward) returns nb > 0 (channel count a multiple of 64, more than 1024 rows
 * per group -- 256 for the backward of stacked groups; the other shapes have one-launch batch norms already):
 * bn_partial / sums_partial hold 2 * groups * nb * c floats, colsum_partial groups * nb * c floats
 * (colsum_blocks = groups * nb); all tensors 16-byte aligned. */
enum { GAN_OP_CONV2D = 0, GAN_OP_DECONV2D = 1, GAN_OP_CONV2D_REFLECT = 2 };
typedef struct gan_bn_saved {
    const float* x;          /* the batch norm's input */
    const float* gamma;
    const float* beta;
    const float* save_mean;  /* as gan_bn_fwd_train saved them */
    const float* save_rstd;
    int groups;
    int act;                 /* the activation fused behind the batch norm */
    float act_param;
} gan_bn_saved;
int gan_bn_tile_blocks(int groups, int rows, int c, int backward);
int gan_conv_fwd_bn_stats(int op, const gan_conv_geom* g, const float* x, const float* w, const float* bias,
                          float* y, int groups, float* bn_partial, void* ws, size_t ws_bytes,
                          gan_stream_t stream);
int gan_bn_fwd_train_from_stats(const float* x, const float* gamma, const float* beta, float* moving_mean,
                                float* moving_var, const float* residual, float* y, float* save_mean,
                                float* save_rstd, int groups, int rows, int c, float eps, float momentum,
                                int unbiased_moving_var, int act, float act_param, const float* bn_partial,
                                gan_stream_t stream);
int gan_conv_dgrad_bn_sums(int op, const gan_conv_geom* g, const float* dy, const float* w, float* dx,
                           const gan_bn_saved* bn, float* bn_partial, void* ws, size_t ws_bytes,
                           gan_stream_t stream);
int gan_add_bn_sums(const float* a, const float* b, float* out, const gan_bn_saved* bn, int rows, int c,
                    float* bn_partial, gan_stream_t stream);
int gan_bn_bwd_tiled(const float* x, const float* gamma, const float* beta, const float* save_mean,
                     const float* save_rstd, const float* dy, float* dx, float* dgamma, float* dbeta,
                     int accumulate, int groups, int rows, int c, int act, float act_param, float* sums_partial,
                     int sums_ready, float* colsum_partial, gan_stream_t stream);
int gan_conv_wgrad_colsum(int op, const gan_conv_geom* g, const float* x, const float* dy, float* dw, float* db,
                          int accumulate, const float* colsum_partial, int colsum_blocks, void* ws,
                          size_t ws_bytes, gan_stream_t stream);

/* ---- stride-1 convolutions with few channels on one side, as a dense product plus a sum over taps ----------------
 * No counterpart in the reference (it calls tf.nn.conv2d for these layers like for any other): the 7 x 7, 64 -> 3
 * image head of the CycleGAN generators (cyclegan_module.py:143-144) fills 3 of the 32 columns of a matrix-pipe tile and
 * gathers every input value 49 times.  With T = kh * kw taps and rows of ld = T * cs rounded up to a multiple of 4:
 *   forward   Z[p, (t,o)] = sum_c x[p,c] w[t,c,o]  (gan_linear_fwd with the filter repacked by gan_taps_pack_filter to
 *             [cb][ld]), then gan_taps_sum_fwd: y[p,o] = bias[o] + sum_t Z[R(p+t), (t,o)], R = mirror (reflect != 0) or
 *             drop (zero padding); the padding is (k-1)/2 on every side, the stride 1.
 *   backward  gan_taps_spread_bwd: E[q, (t,o)] = sum over the pixels p whose tap t lands on q of dy[p,o]; then
 *             gan_linear_wgrad(x, E) gives the repacked filter gradient (and column sums whose centre-tap entries are the
 *             bias gradient), gan_taps_unpack_filter_grad puts both where they belong (accumulate as elsewhere), and
 *             gan_linear_dgrad(E, repacked filter) is the input gradient.
 *   thin INPUT side (the 3 -> 64 stem, cyclegan_module.py:124-125): its input gradient is
 *             Z = gan_linear_fwd(dy, filter repacked with transposed != 0: the filter is [t][o][c] there), then
 *             gan_taps_sum_bwd: dx[q,o] = sum_t sum over the pixels p whose tap t lands on q of Z[p, (t,o)].
 * cs <= 4, odd kh / kw.  Same values as the gan_conv2d_*_reflect / gan_conv2d_* entry points up to fp32 summation order. */
int gan_taps_pack_filter(const float* w, float* wp, int taps, int cb, int cs, int ld, int transposed,
                         gan_stream_t stream);
int gan_taps_unpack_filter_grad(const float* dwp, const float* dbp, float* dw, float* db, int taps, int cb, int cs,
                                int ld, int transposed, int accumulate, gan_stream_t stream);
int gan_taps_sum_fwd(const float* z, const float* bias, float* y, int n, int h, int w, int cs, int kh, int kw,
                     int ld, int reflect, gan_stream_t stream);
int gan_taps_sum_bwd(const float* z, float* dx, int n, int h, int w, int cs, int kh, int kw, int ld, int reflect,
                     gan_stream_t stream);
int gan_taps_spread_bwd(const float* dy, float* e, int n, int h, int w, int cs, int kh, int kw, int ld, int reflect,
                        gan_stream_t stream);

/* ---- stand-alone activations (ops.py:101-102; dcgan.py:334,368-385,409; text2image.py:198)
 * forward: y = act(x)*scale + shift; backward from the saved y: dx = dy*scale*act'(.) */
int gan_act_fwd(const float* x, float* y, long n, int act, float act_param, float scale, float shift,
                gan_stream_t stream);
int gan_act_bwd(const float* y, const float* dy, float* dx, long n, int act, float act_param,
                float scale, float shift, gan_stream_t stream);
/* y = a + b (residual add, cyclegan_module.py:110) */
int gan_add_fwd(const float* a, const float* b, float* y, long n, gan_stream_t stream);

/* ---- concat / conv_cond_concat (ops.py:23-28, 57-62; text2image.py:213-217) -----------
 * out[r][0:ca] = a[r][:], out[r][ca:ca+cb] = b[r / rows_per_b][:]   (rows_per_b = H*W broadcasts a
 * per-sample vector over the feature map; rows_per_b = 1 is a plain last-axis concat).
 * backward: da = dout[:, :ca]; db[g] = sum over the group's rows of dout[:, ca:].  da / db may be NULL. */
int gan_concat_fwd(const float* a, const float* b, float* out, long rows, int ca, int cb,
                   int rows_per_b, gan_stream_t stream);
int gan_concat_bwd(const float* dout, float* da, float* db, long rows, int ca, int cb,
                   int rows_per_b, gan_stream_t stream);

/* ---- tf.pad(..., "REFLECT") (cyclegan_module.py:106,108,124,143) ----------------------- */
int gan_reflect_pad_fwd(const float* x, float* y, int n, int h, int w, int c, int p,
                        gan_stream_t stream);
int gan_reflect_pad_bwd(const float* dy, float* dx, int n, int h, int w, int c, int p,
                        gan_stream_t stream);

/* ---- losses, fused with their gradient -------------------------------------------------
 * Each writes loss_out[0] = (accumulate ? loss_out[0] : 0) + weight*loss and, when the gradient
 * pointer is non-NULL, grad = weight * dloss/dinput.
 * sigmoid BCE: mean(max(x,0) - x*label + log1p(exp(-|x|)))  (dcgan.py:123-134, text2image.py:86-90,
 *              cyclegan_module.py:155-156)
 * mse const:   mean((x-target)^2)                             (cyclegan_module.py:152-153)
 * l1:          mean(|a-b|), gradient w.r.t. b                 (cyclegan_module.py:149-150) */
int gan_sigmoid_bce_mean(const float* logits, long n, float label, float weight, int accumulate,
                         float* loss_out, float* dlogits, gan_stream_t stream);
int gan_mse_const_mean(const float* x, long n, float target, float weight, int accumulate,
                       float* loss_out, float* dx, gan_stream_t stream);
int gan_l1_mean(const float* a, const float* b, long n, float weight, int accumulate,
                float* loss_out, float* db, gan_stream_t stream);
/* gan_l1_mean for image-sized terms (the cycle loss over whole 256 x 256 x 3 images, cyclegan.py:103-106 through
 * cyclegan_module.py:149-150): with a workspace of gan_l1_mean_workspace_bytes(n) bytes the elementwise work runs on up
 * to 256 workgroups instead of one, each leaves the sum of its contiguous range in the workspace and a second launch
 * adds those in block order.  Same results as gan_l1_mean up to the order of the fp32 sum; deterministic. */
size_t gan_l1_mean_workspace_bytes(long n);
int gan_l1_mean_ws(const float* a, const float* b, long n, float weight, int accumulate,
                   float* loss_out, float* db, void* ws, size_t ws_bytes, gan_stream_t stream);

/* ---- tf.train.AdamOptimizer(lr, beta1).minimize (dcgan.py:154-157) ----------------------
 * One fused step over a flat parameter group.  step_dev[0] (device int32) holds the number of
 * steps already taken; the call increments it, computes
 * lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) on the device (so a captured graph replays correctly) and
 * applies  m = b1*m+(1-b1)*g; v = b2*v+(1-b2)*g*g; theta -= lr_t*m/(sqrt(v)+eps)
 * with g = grad*grad_scale.  lr_scratch is one device float. */
int gan_adam_tf_step(float* theta, const float* grad, float* m, float* v, long n, float lr,
                     float beta1, float beta2, float eps, float grad_scale, int* step_dev,
                     float* lr_scratch, gan_stream_t stream);

/* ---- plan pinning (no counterpart in the reference: TensorFlow picks its conv algorithms
 * internally; this is the hook a host-side autotuner uses) ------------------------------------
 * Every conv / deconv / linear pass is one of three contractions over the geometry:
 *   mode 0: conv2d_fwd, deconv2d_dgrad, linear_fwd
 *   mode 1: conv2d_dgrad, deconv2d_fwd, linear_dgrad
 *   mode 2: conv2d_wgrad, deconv2d_wgrad, linear_wgrad
 * gan_conv_plan_set pins the tile configuration (fp32 MFMA 0: 128x128, 1: 128x64, 2: 64x64,
 * 3: 128x32; bf16 MFMA on fp32 operands split into three bf16 parts 4: 128x128, 5: 128x64,
 * 6: 64x64, 7: 128x128 and 8: 128x64 with the split-K halved inside the workgroup -- two groups of four waves
 * share one output tile --, 9: 128x32 for outputs of a few channels (modes 0 and 2; mode 1 maps it to 3); 4..9 need a
 * gathered-image channel count that is a multiple of 4; in mode 1 the small-side count must be even) and
 * the split-K factor used for one (mode, geometry) in this process; cfg < 0 removes the pin.
 * cfg may carry a tile order as well, cfg = tile configuration + 16 * order: which output tile a workgroup
 * computes as a function of its launch position (workgroups go to the 8 XCDs round robin, each XCD has its own
 * L2): 0 chosen by the library, 1 launch order with the k range fastest, 2 all k splits / parity classes / column
 * tiles of a row tile side by side on one XCD, 3 all tiles of one k range side by side, 4 launch order with the row
 * tile fastest.  Results do not depend on it.
 * gan_conv_plan_get reports the plan in effect (cfg in the same encoding) (`thin` = 1 when mode 1 takes the vector-ALU path
 * for outputs of <= 4 channels, which has no plan).  Changing a plan changes the workspace size
 * gan_conv_workspace_bytes reports.  gan_linear_as_conv_geom yields the geometry under which
 * the linear entry points look their plan up. */
int gan_conv_plan_set(int mode, const gan_conv_geom* g, int cfg, int splits);
int gan_conv_plan_get(int mode, const gan_conv_geom* g, int* cfg, int* splits, int* thin);
void gan_conv_plan_clear(void);
int gan_linear_as_conv_geom(int rows, int in_dim, int out_dim, gan_conv_geom* g);

/* ---- arithmetic of the contractions (no counterpart in the reference: tf.nn.conv2d computes in
 * fp32 on whatever device it runs) --------------------------------------------------------
 * By default a contraction whose channel counts are multiples of 4 runs on the bf16 matrix pipe:
 * every fp32 operand value is split exactly into three bf16 parts (8 + 8 + 8 significand bits) and
 * six of the nine part products are accumulated in fp32; the three dropped ones are together below
 * 2^-23 of the product, i.e. below the rounding of one fp32 multiplication.  gan_set_exact_f32(1)
 * keeps every contraction on the fp32 MFMA instead (bitwise an fmaf chain per output element);
 * the environment variable GAN_HIP_EXACT_F32=1 makes that the default of the process. */
void gan_set_exact_f32(int on);
int gan_get_exact_f32(void);

/* ---- test and tuning knobs (no counterpart in the reference; none is needed in production) ------
 * Each knob takes its initial value from an environment variable, read ONCE at the first call into
 * the library (no getenv on the launch path; a plan cannot change between a workspace query and the
 * call it sizes), and can be changed afterwards with gan_knob_set(name, value):
 *   "force_cfg"       GAN_HIP_FORCE_CFG=<0..9>   tile configuration of every contraction (-1: unset)
 *   "force_splits"    GAN_HIP_FORCE_SPLITS=<n>   split-K factor of every contraction (0: unset)
 *   "no_thin"         GAN_HIP_NO_THIN=1          thin outputs (<= 4 channels / linear outputs) stay on the MFMA path
 *   "slow_gather"     GAN_HIP_SLOW_GATHER=1      general gather arithmetic in the split kernel (bit-identical)
 *   "scalar_epilogue" GAN_HIP_SCALAR_EPILOGUE=1  dword stores instead of the 16-byte epilogue
 *   "three_launch_bn" GAN_HIP_THREE_LAUNCH_BN=1  every batch norm on the general three-launch path
 *   "two_launch_bn"   GAN_HIP_TWO_LAUNCH_BN=0    batch norms above 1024 rows per group (channel count a multiple of 64)
 *                                                on the three-launch path instead of the two-launch kernels whose
 *                                                apply workgroups merge the block statistics themselves
 *   "separate_colsum" GAN_HIP_SEPARATE_COLSUM=1  bias gradients as two launches after the weight gradient
 *                                                (bit-identical to the default, fused placement)
 *   "tile_order"      GAN_HIP_TILE_ORDER=<0..4>  which tile a contraction workgroup computes: 0 chosen per launch,
 *                                                1 launch order with the k range fastest, 2 grouped by row tile so
 *                                                that tiles sharing operand bytes run on one XCD, 3 grouped by k
 *                                                range, 4 launch order with the row tile fastest; 1000 + three
 *                                                digits sets it per mode (bit-identical results; for measurements)
 *   "rgb_gather"      GAN_HIP_RGB_GATHER=<0..2>  filter-row gather of the bf16 kernels for 3-channel images: 1 (default)
 *                                                forward and weight gradient, 2 weight gradient only, 0 off (those
 *                                                layers then run on the fp32 MFMA kernel)
 *   "row_gather_any"  GAN_HIP_ROW_GATHER_ANY=0   the same gather for every other channel count that is no multiple of 4
 *                                                (the 1 / 11 / 21-channel layers of the y-conditioned MNIST model): 0
 *                                                keeps those layers on the fp32 MFMA kernel
 *   "no_bn_ride"      GAN_HIP_NO_BN_RIDE=<1|2>
 *                                                1: gan_bn_tile_blocks reports 0, so callers keep the separate batch-norm
 *                                                launches; 2: the reductions of gan_conv_fwd_bn_stats / gan_conv_dgrad_bn_sums
 *                                                run as launches of their own behind the contraction (bit-identical)
 *   "rgb_direct"      GAN_HIP_RGB_DIRECT=0       forward passes over 3-channel images (5-row filters of up to 5 columns, output
 *                                                channels a multiple of 64) stay on the general kernels instead of the
 *                                                one-pass kernel
 *   "no_carry"        GAN_HIP_NO_CARRY=1         the last stage of a bias gradient's column sum never rides on a contraction
 *                                                launch that has no slab reduction (it runs as a launch of its own)
 * Changing force_cfg / force_splits changes the workspace sizes the queries report.  Not thread safe
 * against concurrent calls into the library. */
int gan_knob_set(const char* name, int value);

/* ---- utilities ------------------------------------------------------------------------- */
int gan_fill(float* x, long n, float value, gan_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GAN_HIP_H */
