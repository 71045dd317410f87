"""CycleGAN networks and criteria on the HIP hot path, mirroring the reference's
models/cyclegan_module.py: discriminator (:7-30), generator_resnet (:93-147) and the three criteria
(:149-156).

The reference's D is five 5x5 convolutions (ops.conv2d defaults) with strides 2,2,2,1,1, not the
4x4 "70x70 PatchGAN" of the paper; its normalisation is ops.batch_norm, which at the script's
batch size 1 is instance normalisation.  Both are reproduced as written.  generator_unet (:32-90)
cannot run in the reference itself (it calls the batch_norm class as a function and passes an int
as deconv2d's output_shape) and is not provided.
"""
from __future__ import division

from . import ops
from .ops import batch_norm, conv2d, deconv2d, lrelu


def discriminator(image, options, reuse=False, name="discriminator"):
    with ops.variable_scope(name) as scope:
        # input: [n, 256, 256, input_c_dim] at the reference's sizes
        if reuse:
            scope.reuse_variables()

        d_bn1 = batch_norm(name='d_bn1')
        d_bn2 = batch_norm(name='d_bn2')
        d_bn3 = batch_norm(name='d_bn3')

        h0 = conv2d(image, options.df_dim, name='d_h0_conv', act='lrelu')
        # -> [n, 128, 128, df_dim]
        h1 = d_bn1(conv2d(h0, options.df_dim * 2, name='d_h1_conv', norm_groups=1), act='lrelu')
        # -> [n, 64, 64, 2 df_dim]
        h2 = d_bn2(conv2d(h1, options.df_dim * 4, name='d_h2_conv', norm_groups=1), act='lrelu')
        # -> [n, 32, 32, 4 df_dim]
        h3 = d_bn3(conv2d(h2, options.df_dim * 8, d_h=1, d_w=1, name='d_h3_conv', norm_groups=1), act='lrelu')
        # -> [n, 32, 32, 8 df_dim] (stride 1)
        h4 = conv2d(h3, 1, d_h=1, d_w=1, name='d_h3_pred')
        # -> [n, 32, 32, 1]: one logit per patch
        return h4


def generator_unet(image, options, reuse=False, name="generator"):
    raise NotImplementedError(
        "generator_unet is dead code in the reference (cyclegan_module.py:44,59 call the batch_norm class "
        "as a function and give deconv2d an int output_shape); use generator_resnet (--use_resnet True, the default)")


def generator_resnet(image, options, reuse=False, name="generator"):
    with ops.variable_scope(name) as scope:
        # input: [n, 256, 256, input_c_dim] at the reference's sizes
        if reuse:
            scope.reuse_variables()

        def residule_block(x, dim, ks=3, s=1, name='res'):
            resBN1 = batch_norm(name=name + '_bn1')
            resBN2 = batch_norm(name=name + '_bn2')
            p = int((ks - 1) / 2)
            x, skip = ops.fork(x)       # the block's input feeds the convolutions and the skip connection
            # tf.pad(x, REFLECT by p) + VALID conv (cyclegan_module.py:106-109): the padding is folded into the
            # convolution's gather (padding='REFLECT'), no padded copy is written
            assert s == 1 and p * 2 + 1 == ks
            y = resBN1(conv2d(x, dim, ks, ks, s, s, padding='REFLECT', name=name + '_c1', norm_groups=1), act='relu')
            # resBN2(...) + x, the residual add fused into the normalisation kernel
            return resBN2(conv2d(y, dim, ks, ks, s, s, padding='REFLECT', name=name + '_c2', norm_groups=1), residual=skip)

        g_e1_bn = batch_norm(name='g_e1_bn')
        g_e2_bn = batch_norm(name='g_e2_bn')
        g_e3_bn = batch_norm(name='g_e3_bn')
        g_d1_bn = batch_norm(name='g_d1_bn')
        g_d2_bn = batch_norm(name='g_d2_bn')
        g_pred_bn = batch_norm(name='g_pred_bn')

        s = options.image_size
        s2 = int(s / 2)
        # the fast-neural-style layout the reference cites (cyclegan_module.py:121-122): 7x7 stem, two stride-2
        # convs down, nine residual blocks, two stride-2 transposed convs up, 7x7 head
        c1 = g_e1_bn(conv2d(image, options.gf_dim, 7, 7, 1, 1, padding='REFLECT', name='g_e1_c', norm_groups=1), act='relu')
        c2 = g_e2_bn(conv2d(c1, options.gf_dim * 2, 3, 3, 2, 2, name='g_e2_c', norm_groups=1), act='relu')
        c3 = g_e3_bn(conv2d(c2, options.gf_dim * 4, 3, 3, 2, 2, name='g_e3_c', norm_groups=1), act='relu')
        # nine residual blocks
        r = c3
        for i in range(1, 10):
            r = residule_block(r, options.gf_dim * 4, name='g_r%d' % i)

        d1 = deconv2d(r, [options.batch_size, s2, s2, options.gf_dim * 2], 3, 3, 2, 2, name='g_d1_dc', norm_groups=1)
        d1 = g_d1_bn(d1, act='relu')
        d2 = deconv2d(d1, [options.batch_size, s, s, options.gf_dim], 3, 3, 2, 2, name='g_d2_dc', norm_groups=1)
        d2 = g_d2_bn(d2, act='relu')
        pred = conv2d(d2, options.output_c_dim, 7, 7, 1, 1, padding='REFLECT', name='g_pred_c', norm_groups=1)
        pred = g_pred_bn(pred, act='tanh')

        return pred


def abs_criterion(in_, target, weight=1.0):
    """tf.reduce_mean(tf.abs(in_ - target)) (cyclegan_module.py:149-150), times `weight`."""
    return ops.l1_mean(in_, target, weight)


def mae_criterion(in_, target, weight=1.0):
    """tf.reduce_mean((in_-target)**2) (cyclegan_module.py:152-153; mean SQUARED error despite the
    name).  `target` is the constant the reference builds with ones_like / zeros_like."""
    return ops.mse_const_mean(in_, float(target), weight)


def sce_criterion(logits, labels, weight=1.0):
    """tf.reduce_mean(sigmoid_cross_entropy_with_logits) (cyclegan_module.py:155-156), constant labels."""
    return ops.sigmoid_cross_entropy_mean(logits, float(labels), weight)
