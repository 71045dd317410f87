"""Numpy restatement of the generator / discriminator graphs and train-step
schedules of the reference's three GAN models.  Test infrastructure only.

Each function cites the reference lines whose wiring it follows.
"""
import math

import numpy as np

from . import autodiff as A


def conv_out_size_same(size, stride):
    """dcgan.py:15-16"""
    return int(math.ceil(float(size) / float(stride)))


# ============================================================================
# DCGAN  (models/dcgan.py)
# ============================================================================
class DCGAN(object):
    def __init__(self, store, batch_size=64, output_height=64, output_width=64, c_dim=3,
                 y_dim=None, z_dim=100, gf_dim=64, df_dim=64, gfc_dim=1024, dfc_dim=1024,
                 lr=0.0002, beta1=0.5):
        self.st = store
        self.batch_size, self.output_height, self.output_width = batch_size, output_height, output_width
        self.c_dim, self.y_dim, self.z_dim = c_dim, y_dim, z_dim
        self.gf_dim, self.df_dim, self.gfc_dim, self.dfc_dim = gf_dim, df_dim, gfc_dim, dfc_dim
        self.lr, self.beta1 = lr, beta1
        # dcgan.py:57-68
        self.d_bn1 = A.batch_norm(name='d_bn1')
        self.d_bn2 = A.batch_norm(name='d_bn2')
        if not y_dim:
            self.d_bn3 = A.batch_norm(name='d_bn3')
        self.g_bn0 = A.batch_norm(name='g_bn0')
        self.g_bn1 = A.batch_norm(name='g_bn1')
        self.g_bn2 = A.batch_norm(name='g_bn2')
        if not y_dim:
            self.g_bn3 = A.batch_norm(name='g_bn3')
        self.d_adam = self.g_adam = None

    # dcgan.py:322-351
    def discriminator(self, tp, image, y=None):
        with self.st.scope("discriminator"):
            if not self.y_dim:
                h0 = A.lrelu(tp, A.conv2d(tp, image, self.df_dim, name='d_h0_conv'))
                h1 = A.lrelu(tp, self.d_bn1(tp, A.conv2d(tp, h0, self.df_dim * 2, name='d_h1_conv')))
                h2 = A.lrelu(tp, self.d_bn2(tp, A.conv2d(tp, h1, self.df_dim * 4, name='d_h2_conv')))
                h3 = A.lrelu(tp, self.d_bn3(tp, A.conv2d(tp, h2, self.df_dim * 8, name='d_h3_conv')))
                h4 = A.linear(tp, A.reshape(tp, h3, (self.batch_size, -1)), 1, 'd_h3_lin')
                return h4
            yb = A.reshape(tp, y, (self.batch_size, 1, 1, self.y_dim))
            x = A.conv_cond_concat(tp, image, yb)
            h0 = A.lrelu(tp, A.conv2d(tp, x, self.c_dim + self.y_dim, name='d_h0_conv'))
            h0 = A.conv_cond_concat(tp, h0, yb)
            h1 = A.lrelu(tp, self.d_bn1(tp, A.conv2d(tp, h0, self.df_dim + self.y_dim, name='d_h1_conv')))
            h1 = A.reshape(tp, h1, (self.batch_size, -1))
            h1 = A.concat(tp, [h1, y], 1)
            h2 = A.lrelu(tp, self.d_bn2(tp, A.linear(tp, h1, self.dfc_dim, 'd_h2_lin')))
            h2 = A.concat(tp, [h2, y], 1)
            return A.linear(tp, h2, 1, 'd_h3_lin')

    # dcgan.py:353-410 (train=True) and :412-462 (sampler, train=False)
    def generator(self, tp, z, y=None, train=True):
        bs = self.batch_size
        with self.st.scope("generator"):
            if not self.y_dim:
                s_h, s_w = self.output_height, self.output_width
                s_h2, s_w2 = conv_out_size_same(s_h, 2), conv_out_size_same(s_w, 2)
                s_h4, s_w4 = conv_out_size_same(s_h2, 2), conv_out_size_same(s_w2, 2)
                s_h8, s_w8 = conv_out_size_same(s_h4, 2), conv_out_size_same(s_w4, 2)
                s_h16, s_w16 = conv_out_size_same(s_h8, 2), conv_out_size_same(s_w8, 2)
                z_ = A.linear(tp, z, self.gf_dim * 8 * s_h16 * s_w16, 'g_h0_lin')
                h0 = A.reshape(tp, z_, (-1, s_h16, s_w16, self.gf_dim * 8))
                h0 = A.relu(tp, self.g_bn0(tp, h0, train))
                h1 = A.deconv2d(tp, h0, [bs, s_h8, s_w8, self.gf_dim * 4], name='g_h1')
                h1 = A.relu(tp, self.g_bn1(tp, h1, train))
                h2 = A.deconv2d(tp, h1, [bs, s_h4, s_w4, self.gf_dim * 2], name='g_h2')
                h2 = A.relu(tp, self.g_bn2(tp, h2, train))
                h3 = A.deconv2d(tp, h2, [bs, s_h2, s_w2, self.gf_dim * 1], name='g_h3')
                h3 = A.relu(tp, self.g_bn3(tp, h3, train))
                h4 = A.deconv2d(tp, h3, [bs, s_h, s_w, self.c_dim], name='g_h4')
                return A.tanh(tp, h4)
            s_h, s_w = self.output_height, self.output_width
            s_h2, s_h4 = int(s_h / 2), int(s_h / 4)
            s_w2, s_w4 = int(s_w / 2), int(s_w / 4)
            yb = A.reshape(tp, y, (bs, 1, 1, self.y_dim))
            z = A.concat(tp, [z, y], 1)
            h0 = A.relu(tp, self.g_bn0(tp, A.linear(tp, z, self.gfc_dim, 'g_h0_lin'), train))
            h0 = A.concat(tp, [h0, y], 1)
            h1 = A.relu(tp, self.g_bn1(tp, A.linear(tp, h0, self.gf_dim * 2 * s_h4 * s_w4, 'g_h1_lin'), train))
            h1 = A.reshape(tp, h1, (bs, s_h4, s_w4, self.gf_dim * 2))
            h1 = A.conv_cond_concat(tp, h1, yb)
            h2 = A.relu(tp, self.g_bn2(tp, A.deconv2d(tp, h1, [bs, s_h2, s_w2, self.gf_dim * 2], name='g_h2'), train))
            h2 = A.conv_cond_concat(tp, h2, yb)
            return A.sigmoid(tp, A.deconv2d(tp, h2, [bs, s_h, s_w, self.c_dim], name='g_h3'))

    def build(self, images, z, y=None):
        """Create every variable in the reference's creation order
        (generator, then discriminator: dcgan.py:107-108 / 113-114)."""
        tp = A.Tape(self.st)
        yn = None if y is None else tp.const(y)
        g = self.generator(tp, tp.const(z), yn)
        self.discriminator(tp, tp.const(images), yn)
        self.d_vars = self.st.select('d_')     # dcgan.py:146-147
        self.g_vars = self.st.select('g_')
        self.d_adam = A.Adam(self.st, self.d_vars, self.lr, self.beta1)   # dcgan.py:154-157
        self.g_adam = A.Adam(self.st, self.g_vars, self.lr, self.beta1)
        # constructing the graph does not run it: undo the moving-average side effect
        for n in list(self.st.values):
            if n.endswith('moving_mean'):
                self.st.values[n][...] = 0
            elif n.endswith('moving_variance'):
                self.st.values[n][...] = 1
        return g

    def d_losses(self, tp, images, z, y=None, g_needs_grad=False):
        yn = None if y is None else tp.const(y)
        if g_needs_grad:
            G = self.generator(tp, tp.const(z), yn)
        else:
            G = tp.const(self.generator(A.Tape(self.st), tp.const(z), yn).v)
        real_logits = self.discriminator(tp, tp.const(images), yn)
        fake_logits = self.discriminator(tp, G, yn)
        d_loss_real = A.sigmoid_ce_mean(tp, real_logits, 1.0)     # dcgan.py:129-132
        d_loss_fake = A.sigmoid_ce_mean(tp, fake_logits, 0.0)
        return d_loss_real, d_loss_fake

    def d_update(self, images, z, y=None):
        """dcgan.py:228-233 / 263-264"""
        tp = A.Tape(self.st)
        lr_, lf_ = self.d_losses(tp, images, z, y)
        d_loss = A.add(tp, lr_, lf_)                              # dcgan.py:139
        tp.backward(d_loss)
        self.d_adam.step(tp.grads(self.d_vars))
        return float(d_loss.v)

    def g_update(self, z, y=None):
        """dcgan.py:237-247 / 268-275"""
        tp = A.Tape(self.st)
        yn = None if y is None else tp.const(y)
        G = self.generator(tp, tp.const(z), yn)
        fake_logits = self.discriminator(tp, G, yn)
        g_loss = A.sigmoid_ce_mean(tp, fake_logits, 1.0)          # dcgan.py:133-134
        tp.backward(g_loss)
        self.g_adam.step(tp.grads(self.g_vars))
        return float(g_loss.v)

    def eval_losses(self, images, z, y=None):
        """The three logging re-evaluations (dcgan.py:249-260 / 277-279): forward only,
        post-update weights, each a separate run (each mutates BN moving averages)."""
        tp = A.Tape(self.st)
        yn = None if y is None else tp.const(y)
        G = self.generator(tp, tp.const(z), yn)
        err_d_fake = float(A.sigmoid_ce_mean(tp, self.discriminator(tp, G, yn), 0.0).v)
        tp = A.Tape(self.st)
        yn = None if y is None else tp.const(y)
        err_d_real = float(A.sigmoid_ce_mean(tp, self.discriminator(tp, tp.const(images), yn), 1.0).v)
        tp = A.Tape(self.st)
        yn = None if y is None else tp.const(y)
        G = self.generator(tp, tp.const(z), yn)
        err_g = float(A.sigmoid_ce_mean(tp, self.discriminator(tp, G, yn), 1.0).v)
        return err_d_fake, err_d_real, err_g

    def train_iteration(self, images, z, y=None, log_evals=False):
        """One iteration of dcgan.py:226-279: 1 D update, 2 G updates with the same z."""
        d = self.d_update(images, z, y)
        g1 = self.g_update(z, y)
        g2 = self.g_update(z, y)
        out = {'d_loss': d, 'g_loss_1': g1, 'g_loss_2': g2}
        if log_evals:
            f, r, g = self.eval_losses(images, z, y)
            out.update(errD_fake=f, errD_real=r, errG=g)
        return out

    def sample(self, z, y=None):
        tp = A.Tape(self.st)
        yn = None if y is None else tp.const(y)
        return self.generator(tp, tp.const(z), yn, train=False).v


# ============================================================================
# Text2Image / GAN-CLS  (models/text2image.py)
# ============================================================================
class Text2Image(object):
    def __init__(self, store, batch_size=64, image_size=64, z_dim=100, t_dim=256, gf_dim=64,
                 df_dim=64, caption_vector_length=1024, lr=0.0002, beta1=0.5):
        self.st = store
        self.batch_size, self.image_size, self.z_dim, self.t_dim = batch_size, image_size, z_dim, t_dim
        self.gf_dim, self.df_dim, self.caption_vector_length = gf_dim, df_dim, caption_vector_length
        self.lr, self.beta1 = lr, beta1
        # text2image.py:61-69
        self.g_bn0, self.g_bn1 = A.batch_norm(name='g_bn0'), A.batch_norm(name='g_bn1')
        self.g_bn2, self.g_bn3 = A.batch_norm(name='g_bn2'), A.batch_norm(name='g_bn3')
        self.d_bn1, self.d_bn2 = A.batch_norm(name='d_bn1'), A.batch_norm(name='d_bn2')
        self.d_bn3, self.d_bn4 = A.batch_norm(name='d_bn3'), A.batch_norm(name='d_bn4')

    # text2image.py:175-198 (train) / 148-172 (sampler)
    def generator(self, tp, t_z, t_text_embedding, train=True):
        bs = self.batch_size
        with self.st.scope('generator'):
            s = self.image_size
            s2, s4, s8, s16 = int(s / 2), int(s / 4), int(s / 8), int(s / 16)
            reduced = A.lrelu(tp, A.linear(tp, t_text_embedding, self.t_dim, 'g_embedding'))
            z_concat = A.concat(tp, [t_z, reduced], 1)
            z_ = A.linear(tp, z_concat, self.gf_dim * 8 * s16 * s16, 'g_h0_lin')
            h0 = A.reshape(tp, z_, (-1, s16, s16, self.gf_dim * 8))
            h0 = A.relu(tp, self.g_bn0(tp, h0, train))
            h1 = A.relu(tp, self.g_bn1(tp, A.deconv2d(tp, h0, [bs, s8, s8, self.gf_dim * 4], name='g_h1'), train))
            h2 = A.relu(tp, self.g_bn2(tp, A.deconv2d(tp, h1, [bs, s4, s4, self.gf_dim * 2], name='g_h2'), train))
            h3 = A.relu(tp, self.g_bn3(tp, A.deconv2d(tp, h2, [bs, s2, s2, self.gf_dim * 1], name='g_h3'), train))
            h4 = A.deconv2d(tp, h3, [bs, s, s, 3], name='g_h4')
            return A.affine(tp, A.tanh(tp, h4), 0.5, 0.5)

    # text2image.py:201-222
    def discriminator(self, tp, image, t_text_embedding):
        with self.st.scope("discriminator"):
            h0 = A.lrelu(tp, A.conv2d(tp, image, self.df_dim, name='d_h0_conv'))
            h1 = A.lrelu(tp, self.d_bn1(tp, A.conv2d(tp, h0, self.df_dim * 2, name='d_h1_conv')))
            h2 = A.lrelu(tp, self.d_bn2(tp, A.conv2d(tp, h1, self.df_dim * 4, name='d_h2_conv')))
            h3 = A.lrelu(tp, self.d_bn3(tp, A.conv2d(tp, h2, self.df_dim * 8, name='d_h3_conv')))
            reduced = A.lrelu(tp, A.linear(tp, t_text_embedding, self.t_dim, 'd_embedding'))
            tiled = A.tile_spatial(tp, reduced, 4, 4)
            h3_concat = A.concat(tp, [h3, tiled], 3)
            h3_new = A.lrelu(tp, self.d_bn4(tp, A.conv2d(tp, h3_concat, self.df_dim * 8, 1, 1, 1, 1, name='d_h3_conv_new')))
            return A.linear(tp, A.reshape(tp, h3_new, (self.batch_size, -1)), 1, 'd_h3_lin')

    def build(self, real, wrong, caption, z):
        tp = A.Tape(self.st)
        self.generator(tp, tp.const(z), tp.const(caption))
        self.discriminator(tp, tp.const(real), tp.const(caption))
        self.d_vars = self.st.select('d_')      # text2image.py:95-96
        self.g_vars = self.st.select('g_')
        self.d_adam = A.Adam(self.st, self.d_vars, self.lr, self.beta1)   # text2image.py:231-232
        self.g_adam = A.Adam(self.st, self.g_vars, self.lr, self.beta1)
        for n in list(self.st.values):
            if n.endswith('moving_mean'):
                self.st.values[n][...] = 0
            elif n.endswith('moving_variance'):
                self.st.values[n][...] = 1

    def d_update(self, real, wrong, caption, z):
        """text2image.py:82-92, 251-257"""
        tp = A.Tape(self.st)
        cap = tp.const(caption)
        fake = tp.const(self.generator(A.Tape(self.st), tp.const(z), cap).v)
        l_real = self.discriminator(tp, tp.const(real), cap)
        l_wrong = self.discriminator(tp, tp.const(wrong), cap)
        l_fake = self.discriminator(tp, fake, cap)
        d1 = A.sigmoid_ce_mean(tp, l_real, 1.0)
        d2 = A.sigmoid_ce_mean(tp, l_wrong, 0.0)
        d3 = A.sigmoid_ce_mean(tp, l_fake, 0.0)
        d_loss = A.add(tp, A.add(tp, d1, d2), d3)
        tp.backward(d_loss)
        self.d_adam.step(tp.grads(self.d_vars))
        return float(d_loss.v), float(d1.v), float(d2.v), float(d3.v)

    def g_update(self, caption, z):
        """text2image.py:86, 261-276"""
        tp = A.Tape(self.st)
        cap = tp.const(caption)
        fake = self.generator(tp, tp.const(z), cap)
        g_loss = A.sigmoid_ce_mean(tp, self.discriminator(tp, fake, cap), 1.0)
        tp.backward(g_loss)
        self.g_adam.step(tp.grads(self.g_vars))
        return float(g_loss.v)

    def train_iteration(self, real, wrong, caption, z):
        d = self.d_update(real, wrong, caption, z)
        g1 = self.g_update(caption, z)
        g2 = self.g_update(caption, z)
        return {'d_loss': d[0], 'd_loss1': d[1], 'd_loss2': d[2], 'd_loss3': d[3],
                'g_loss_1': g1, 'g_loss_2': g2}


# ============================================================================
# CycleGAN  (models/cyclegan_module.py, models/cyclegan.py)
# ============================================================================
def cyclegan_discriminator(tp, image, df_dim, name):
    """cyclegan_module.py:7-30"""
    st = tp.store
    with st.scope(name):
        d_bn1, d_bn2, d_bn3 = A.batch_norm(name='d_bn1'), A.batch_norm(name='d_bn2'), A.batch_norm(name='d_bn3')
        h0 = A.lrelu(tp, A.conv2d(tp, image, df_dim, name='d_h0_conv'))
        h1 = A.lrelu(tp, d_bn1(tp, A.conv2d(tp, h0, df_dim * 2, name='d_h1_conv')))
        h2 = A.lrelu(tp, d_bn2(tp, A.conv2d(tp, h1, df_dim * 4, name='d_h2_conv')))
        h3 = A.lrelu(tp, d_bn3(tp, A.conv2d(tp, h2, df_dim * 8, d_h=1, d_w=1, name='d_h3_conv')))
        return A.conv2d(tp, h3, 1, d_h=1, d_w=1, name='d_h3_pred')


def cyclegan_generator_resnet(tp, image, gf_dim, output_c_dim, image_size, batch_size, name):
    """cyclegan_module.py:93-147"""
    st = tp.store
    with st.scope(name):
        def residule_block(x, dim, ks=3, s=1, name='res'):
            bn1, bn2 = A.batch_norm(name=name + '_bn1'), A.batch_norm(name=name + '_bn2')
            p = int((ks - 1) / 2)
            y = A.reflect_pad(tp, x, p)
            y = bn1(tp, A.conv2d(tp, y, dim, ks, ks, s, s, padding='VALID', name=name + '_c1'))
            y = A.reflect_pad(tp, A.relu(tp, y), p)
            y = bn2(tp, A.conv2d(tp, y, dim, ks, ks, s, s, padding='VALID', name=name + '_c2'))
            return A.add(tp, y, x)

        g_e1_bn, g_e2_bn, g_e3_bn = A.batch_norm(name='g_e1_bn'), A.batch_norm(name='g_e2_bn'), A.batch_norm(name='g_e3_bn')
        g_d1_bn, g_d2_bn, g_pred_bn = A.batch_norm(name='g_d1_bn'), A.batch_norm(name='g_d2_bn'), A.batch_norm(name='g_pred_bn')
        s = image_size
        s2 = int(s / 2)
        c0 = A.reflect_pad(tp, image, 3)
        c1 = A.relu(tp, g_e1_bn(tp, A.conv2d(tp, c0, gf_dim, 7, 7, 1, 1, padding='VALID', name='g_e1_c')))
        c2 = A.relu(tp, g_e2_bn(tp, A.conv2d(tp, c1, gf_dim * 2, 3, 3, 2, 2, name='g_e2_c')))
        c3 = A.relu(tp, g_e3_bn(tp, A.conv2d(tp, c2, gf_dim * 4, 3, 3, 2, 2, name='g_e3_c')))
        r = c3
        for i in range(1, 10):
            r = residule_block(r, gf_dim * 4, name='g_r%d' % i)
        d1 = A.deconv2d(tp, r, [batch_size, s2, s2, gf_dim * 2], 3, 3, 2, 2, name='g_d1_dc')
        d1 = A.relu(tp, g_d1_bn(tp, d1))
        d2 = A.deconv2d(tp, d1, [batch_size, s, s, gf_dim], 3, 3, 2, 2, name='g_d2_dc')
        d2 = A.relu(tp, g_d2_bn(tp, d2))
        d2 = A.reflect_pad(tp, d2, 3)
        pred = A.conv2d(tp, d2, output_c_dim, 7, 7, 1, 1, padding='VALID', name='g_pred_c')
        return A.tanh(tp, g_pred_bn(tp, pred))


class ImagePool(object):
    """misc.py:249-267 restated with an injectable random source."""

    def __init__(self, maxsize=50, rand=None, randint=None):
        self.maxsize = maxsize
        self.num_img = 0
        self.images = []
        self.rand = rand or np.random.rand
        self.randint = randint or (lambda n: int(np.random.rand() * n))

    def __call__(self, image):
        if self.maxsize == 0:
            return image
        if self.num_img < self.maxsize:
            self.images.append(image)
            self.num_img += 1
            return image
        if self.rand() > 0.5:
            idx = self.randint(self.maxsize)
            tmp = [a.copy() for a in self.images[idx]]
            self.images[idx] = image
            return tmp
        return image


class CycleGAN(object):
    def __init__(self, store, batch_size=1, image_size=256, input_c_dim=3, output_c_dim=3,
                 gf_dim=64, df_dim=64, L1_lambda=10.0, use_lsgan=True, lr=0.0002, beta1=0.5,
                 max_size=50):
        self.st = store
        self.batch_size, self.image_size = batch_size, image_size
        self.input_c_dim, self.output_c_dim = input_c_dim, output_c_dim
        self.gf_dim, self.df_dim, self.L1_lambda, self.use_lsgan = gf_dim, df_dim, L1_lambda, use_lsgan
        self.lr, self.beta1 = lr, beta1
        self.pool = ImagePool(max_size)

    def G(self, tp, x, name):
        out_c = self.output_c_dim
        return cyclegan_generator_resnet(tp, x, self.gf_dim, out_c, self.image_size, self.batch_size, name)

    def D(self, tp, x, name):
        return cyclegan_discriminator(tp, x, self.df_dim, name)

    def criterionGAN(self, tp, logits, target):
        """mae_criterion (really MSE) if use_lsgan else sce_criterion: cyclegan.py:31-34"""
        if self.use_lsgan:
            return A.mse_mean(tp, logits, target)
        return A.sigmoid_ce_mean(tp, logits, target)

    def split(self, real_data):
        return (real_data[..., :self.input_c_dim],
                real_data[..., self.input_c_dim:self.input_c_dim + self.output_c_dim])

    def build(self, real_data):
        """Variable creation order of cyclegan.py:56-62."""
        a, b = self.split(real_data)
        tp = A.Tape(self.st)
        fb = self.G(tp, tp.const(a), "generatorA2B")
        fa = self.G(tp, tp.const(b), "generatorB2A")
        self.D(tp, fb, "discriminatorB")
        self.D(tp, fa, "discriminatorA")
        sel = self.st.select
        self.db_vars, self.da_vars = sel('discriminatorB'), sel('discriminatorA')   # cyclegan.py:112-115
        self.g_vars_a2b, self.g_vars_b2a = sel('generatorA2B'), sel('generatorB2A')
        mk = lambda names: A.Adam(self.st, names, self.lr, self.beta1)              # cyclegan.py:123-130
        self.da_adam, self.db_adam = mk(self.da_vars), mk(self.db_vars)
        self.ga2b_adam, self.gb2a_adam = mk(self.g_vars_a2b), mk(self.g_vars_b2a)
        for n in list(self.st.values):
            if n.endswith('moving_mean'):
                self.st.values[n][...] = 0
            elif n.endswith('moving_variance'):
                self.st.values[n][...] = 1

    def forward_fakes(self, real_data):
        """cyclegan.py:160-161"""
        a, b = self.split(real_data)
        tp = A.Tape(self.st)
        fake_B = self.G(tp, tp.const(a), "generatorA2B").v
        tp = A.Tape(self.st)
        fake_A = self.G(tp, tp.const(b), "generatorB2A").v
        return fake_A, fake_B

    def g_loss(self, tp, real_data, which, terms=None):
        """cyclegan.py:56-68.  `terms`, a dict, receives the GAN term and the two weighted cycle terms."""
        a, b = self.split(real_data)
        ra, rb = tp.const(a), tp.const(b)
        fake_B = self.G(tp, ra, "generatorA2B")
        fake_A_ = self.G(tp, fake_B, "generatorB2A")
        fake_A = self.G(tp, rb, "generatorB2A")
        fake_B_ = self.G(tp, fake_A, "generatorA2B")
        if which == 'a2b':
            gan = self.criterionGAN(tp, self.D(tp, fake_B, "discriminatorB"), 1.0)
        else:
            gan = self.criterionGAN(tp, self.D(tp, fake_A, "discriminatorA"), 1.0)
        cyc_a = A.scale(tp, A.l1_mean(tp, ra, fake_A_), self.L1_lambda)
        cyc_b = A.scale(tp, A.l1_mean(tp, rb, fake_B_), self.L1_lambda)
        if terms is not None:
            terms['gan'], terms['cyc_a'], terms['cyc_b'] = float(gan.v), float(cyc_a.v), float(cyc_b.v)
        return A.add(tp, A.add(tp, gan, cyc_a), cyc_b)

    def d_loss(self, tp, real, fake_sample, name, terms=None):
        """cyclegan.py:76-85.  `terms`, a dict, receives the two halves (each already divided by 2)."""
        l_real = self.criterionGAN(tp, self.D(tp, tp.const(real), name), 1.0)
        l_fake = self.criterionGAN(tp, self.D(tp, tp.const(fake_sample), name), 0.0)
        if terms is not None:
            terms['real'], terms['fake'] = 0.5 * float(l_real.v), 0.5 * float(l_fake.v)
        return A.scale(tp, A.add(tp, l_real, l_fake), 0.5)

    def train_iteration(self, real_data, detail=None):
        """cyclegan.py:159-180: five sequential runs.  `detail`, a dict, receives the ten loss terms
        (db_loss_real ... g_b2a_cyc_b) and, under 'grads', each optimiser group's gradients by name."""
        a, b = self.split(real_data)
        fake_A, fake_B = self.forward_fakes(real_data)
        fake_A, fake_B = self.pool([fake_A, fake_B])
        out = {}
        grads = {}

        def run(key, loss_fn, names, adam, prefix):
            tp = A.Tape(self.st)
            terms = {}
            l = loss_fn(tp, terms)
            tp.backward(l)
            grads[key] = tp.grads(names)
            adam.step(grads[key])
            if detail is not None:
                detail.update({prefix + k: v for k, v in terms.items()})
            return float(l.v)
        out['db_loss'] = run('db', lambda tp, t: self.d_loss(tp, b, fake_B, "discriminatorB", t),
                             self.db_vars, self.db_adam, 'db_loss_')
        out['g_loss_a2b'] = run('g_a2b', lambda tp, t: self.g_loss(tp, real_data, 'a2b', t),
                                self.g_vars_a2b, self.ga2b_adam, 'g_a2b_')
        out['da_loss'] = run('da', lambda tp, t: self.d_loss(tp, a, fake_A, "discriminatorA", t),
                             self.da_vars, self.da_adam, 'da_loss_')
        out['g_loss_b2a'] = run('g_b2a', lambda tp, t: self.g_loss(tp, real_data, 'b2a', t),
                                self.g_vars_b2a, self.gb2a_adam, 'g_b2a_')
        if detail is not None:
            detail['grads'] = grads
        return out
