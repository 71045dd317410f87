"""DCGAN on the HIP hot path, mirroring the reference's models/dcgan.py.

Same constructor shape (`DCGAN(sess, options)`), same network wiring (discriminator
dcgan.py:322-351, generator :353-410, sampler :412-462), same losses and variable split
(:123-147), same optimisers (:154-157) and the same per-iteration schedule (:226-279): one D
update, two G updates with the same z, then (optionally) the three logging re-evaluations.

Differences that are deliberate:
  * define-by-run instead of a TF graph: "building the model" runs each network once on zeros to
    create the variables in the reference's creation order, then packs the d_/g_ groups into flat
    arenas;
  * the reference's defects are not reproduced (pdb.set_trace at :153; the un-imported image
    helpers at :79,175,211,299,313);
  * `sess` is accepted and ignored (there is no session); `world_size > 1` adds data-parallel
    gradient all-reduce, which the reference does not have.
"""
from __future__ import division

import math
import os
import re
import time

import numpy as np
import torch

from . import ops
from . import variables as V
from .ops import batch_norm, concat, conv2d, conv_cond_concat, deconv2d, linear, lrelu
from .optim import AdamOptimizer
from .dp import rank_of, rank_seed, shard_start


def conv_out_size_same(size, stride):
    """dcgan.py:15-16"""
    return int(math.ceil(float(size) / float(stride)))


class DCGAN(object):
    def __init__(self, sess, options, store=None, world_size=1, process_group=None, load_data=True):
        self.sess = sess
        self.opts = options
        self.model = getattr(options, 'model', 'DCGAN')
        self.crop = getattr(options, 'crop', True)

        # dcgan.py:31-54: attribute names as in the reference, read from the same options
        self.batch_size = self.sample_num = options.batch_size
        for name in ('input_height', 'input_width', 'output_height', 'output_width', 'y_dim', 'z_dim', 'gf_dim', 'df_dim',
                     'gfc_dim', 'dfc_dim'):
            setattr(self, name, getattr(options, name))
        self.print_every = getattr(options, 'print_every', 1)

        self.world_size = world_size
        self.process_group = process_group
        self.rank = rank_of(world_size, process_group)
        self.stack_d_passes = True      # D update: D(real) and D(fake) as one stacked batch
        self.store = store if store is not None else V.VariableStore(seed=getattr(options, 'seed', 0))

        # the batch-norm layers by name (dcgan.py:56-68); the conditional networks are one layer shallower
        names = ['d_bn1', 'd_bn2', 'g_bn0', 'g_bn1', 'g_bn2'] + ([] if self.y_dim else ['d_bn3', 'g_bn3'])
        for name in names:
            setattr(self, name, batch_norm(name=name))

        self.dataset = options.dataset
        self.input_fname_pattern = getattr(options, 'input_fname_pattern', '*.jpg')
        self.checkpoint_dir = getattr(options, 'checkpoint_dir', './checkpoint')

        self.data_X = self.data_y = None
        c_dim = getattr(options, 'c_dim', None)
        if load_data and self.dataset == 'mnist' and os.path.exists(os.path.join("./data", "mnist", "train-images-idx3-ubyte")):
            self.data_X, self.data_y = self.load_mnist()
            c_dim = self.data_X[0].shape[-1]
        if c_dim is None:
            c_dim = 1 if self.dataset == 'mnist' else 3
        self.c_dim = c_dim
        self.grayscale = (self.c_dim == 1)

        self.build_model()

    # ------------------------------------------------------------------------------------ build
    def build_model(self):
        if self.crop:
            image_dims = [self.output_height, self.output_width, self.c_dim]
        else:
            image_dims = [self.input_height, self.input_width, self.c_dim]
        self.image_dims = image_dims
        dev = self.store.device
        with V.use_store(self.store), torch.no_grad():
            z = torch.zeros(self.batch_size, self.z_dim, device=dev)
            y = torch.zeros(self.batch_size, self.y_dim, device=dev) if self.y_dim else None
            inputs = torch.zeros([self.batch_size] + image_dims, device=dev)
            # variable creation order of dcgan.py:107-108 / 113-114: generator, then discriminator
            self.generator(z, y)
            self.discriminator(inputs, y, reuse=False)
            # building must not run the graph: undo the moving-average side effect of the dry run
            for name, var in self.store.vars.items():
                if name.endswith('moving_mean'):
                    var.data.zero_()
                elif name.endswith('moving_variance'):
                    var.data.fill_(1.0)

        t_vars = self.store.trainable_variables()
        self.d_vars = [var for var in t_vars if 'd_' in var.name]      # dcgan.py:146-147
        self.g_vars = [var for var in t_vars if 'g_' in var.name]
        opts = self.opts
        self.d_group = self.store.pack('d', self.d_vars)
        self.g_group = self.store.pack('g', self.g_vars)
        self.overlap_exchange = self.world_size > 1 or self.process_group is not None
        if self.overlap_exchange:
            # the layers nearest the loss finish their gradients first: their part of each arena is
            # exchanged while the backward pass of the remaining layers runs (see ops.backward)
            self.d_group.split_early('discriminator/' + ('d_h2_lin/Matrix' if self.y_dim else 'd_h3_conv/w'))
            self.g_group.split_early('generator/g_h2/w')
        if self.world_size > 1:
            self.d_group.broadcast(self.process_group)
            self.g_group.broadcast(self.process_group)
        self.d_optim = AdamOptimizer(opts.lr, beta1=opts.beta1).attach(self.d_group)   # dcgan.py:154-157
        self.g_optim = AdamOptimizer(opts.lr, beta1=opts.beta1).attach(self.g_group)
        self.counter = 1

    _d_bound = ()           # set per update; the networks record their boundary activations here
    _g_bound = None

    def _note_d_boundary(self, t):
        if isinstance(self._d_bound, list):
            self._d_bound.append(t)

    def _staged(self, optim, boundary):
        """Arguments of ops.backward for an update whose gradient exchange overlaps its backward pass."""
        if not self.overlap_exchange:
            return {}
        ws, pg = self.world_size, self.process_group
        return dict(boundary=boundary, at_boundary=lambda: optim.all_reduce_early(ws, pg))

    # ------------------------------------------------------------------------------------ networks
    def discriminator(self, image, y=None, reuse=False, passes=1, with_prob=True):
        """dcgan.py:322-351.  Returns (sigmoid(h), h) like the reference.

        passes=k: `image` (and `y`) hold k batches stacked along axis 0 and the call stands for k
        applications of the network (shared weights, batch-norm statistics per batch).
        with_prob=False skips the sigmoid of the first return value (None then), which the training
        losses never read (they are built on the logits, dcgan.py:129-134)."""
        bs = self.batch_size * passes
        with ops.variable_scope("discriminator") as scope:
            if reuse:
                scope.reuse_variables()
            if not self.y_dim:
                h0 = conv2d(image, self.df_dim, name='d_h0_conv', act='lrelu')
                h1 = self.d_bn1(conv2d(h0, self.df_dim * 2, name='d_h1_conv', norm_groups=passes), act='lrelu', passes=passes)
                h2 = self.d_bn2(conv2d(h1, self.df_dim * 4, name='d_h2_conv', norm_groups=passes), act='lrelu', passes=passes)
                self._note_d_boundary(h2)           # the layers after it own the early part of the d group
                h3 = self.d_bn3(conv2d(h2, self.df_dim * 8, name='d_h3_conv', norm_groups=passes), act='lrelu', passes=passes)
                h4 = linear(h3.reshape(bs, -1), 1, 'd_h3_lin')
                return (_sigmoid_value(h4) if with_prob else None), h4
            yb = y.reshape(bs, 1, 1, self.y_dim)
            x = conv_cond_concat(image, yb)
            h0 = conv2d(x, self.c_dim + self.y_dim, name='d_h0_conv', act='lrelu')
            h0 = conv_cond_concat(h0, yb)
            h1 = self.d_bn1(conv2d(h0, self.df_dim + self.y_dim, name='d_h1_conv', norm_groups=passes), act='lrelu', passes=passes)
            h1 = h1.reshape(bs, -1)
            h1 = concat([h1, y], 1)
            self._note_d_boundary(h1)
            h2 = self.d_bn2(linear(h1, self.dfc_dim, 'd_h2_lin'), act='lrelu', passes=passes)
            h2 = concat([h2, y], 1)
            h3 = linear(h2, 1, 'd_h3_lin')
            return (_sigmoid_value(h3) if with_prob else None), h3

    def generator(self, z, y=None, train=True):
        """dcgan.py:353-410 (and, with train=False, the sampler of :412-462)."""
        with ops.variable_scope("generator") as scope:
            if not train:
                scope.reuse_variables()
            bs = self.batch_size
            ng = 1 if train else None           # the deconvs feed training-mode batch norms of one group
            if not self.y_dim:
                # spatial sizes of the five stages, halving with ceil (dcgan.py:357-361)
                s_h, s_w = self.output_height, self.output_width
                (s_h2, s_w2), (s_h4, s_w4), (s_h8, s_w8), (s_h16, s_w16) = [
                    (conv_out_size_same(s_h, d), conv_out_size_same(s_w, d)) for d in (2, 4, 8, 16)]
                # project `z` and reshape
                self.t_z_, self.h0_w, self.h0_b = linear(z, self.gf_dim * 8 * s_h16 * s_w16, 'g_h0_lin', with_w=True)
                self.h0 = self.t_z_.reshape(-1, s_h16, s_w16, self.gf_dim * 8)
                h0 = self.g_bn0(self.h0, train=train, act='relu')
                self.h1, self.h1_w, self.h1_b = deconv2d(h0, [bs, s_h8, s_w8, self.gf_dim * 4], name='g_h1', with_w=True,
                                                           norm_groups=ng)
                h1 = self.g_bn1(self.h1, train=train, act='relu')
                self._g_bound = h1                  # the layers after it own the early part of the g group
                h2, self.h2_w, self.h2_b = deconv2d(h1, [bs, s_h4, s_w4, self.gf_dim * 2], name='g_h2', with_w=True,
                                                      norm_groups=ng)
                h2 = self.g_bn2(h2, train=train, act='relu')
                h3, self.h3_w, self.h3_b = deconv2d(h2, [bs, s_h2, s_w2, self.gf_dim * 1], name='g_h3', with_w=True,
                                                      norm_groups=ng)
                h3 = self.g_bn3(h3, train=train, act='relu')
                h4, self.h4_w, self.h4_b = deconv2d(h3, [bs, s_h, s_w, self.c_dim], name='g_h4', with_w=True, act='tanh')
                return h4                       # tanh(h4) of dcgan.py:385, applied in the deconv's epilogue
            s_h, s_w = self.output_height, self.output_width
            s_h2, s_h4 = int(s_h / 2), int(s_h / 4)
            s_w2, s_w4 = int(s_w / 2), int(s_w / 4)
            yb = y.reshape(bs, 1, 1, self.y_dim)
            z = concat([z, y], 1)
            h0 = self.g_bn0(linear(z, self.gfc_dim, 'g_h0_lin'), train=train, act='relu')
            h0 = concat([h0, y], 1)
            h1 = self.g_bn1(linear(h0, self.gf_dim * 2 * s_h4 * s_w4, 'g_h1_lin'), train=train, act='relu')
            h1 = h1.reshape(bs, s_h4, s_w4, self.gf_dim * 2)
            h1 = conv_cond_concat(h1, yb)
            self._g_bound = h1
            h2 = self.g_bn2(deconv2d(h1, [bs, s_h2, s_w2, self.gf_dim * 2], name='g_h2', norm_groups=ng), train=train, act='relu')
            h2 = conv_cond_concat(h2, yb)
            return deconv2d(h2, [bs, s_h, s_w, self.c_dim], name='g_h3', act='sigmoid')

    def sampler(self, z, y=None):
        """dcgan.py:412-462: the generator with batch-norm in inference mode."""
        with V.use_store(self.store), torch.no_grad():
            return self.generator(z, y, train=False)

    # ------------------------------------------------------------------------------------ updates
    def d_grads(self, images, z, y=None, keep_g=False):
        """Forward + backward of d_loss w.r.t. d_vars (the gradient half of minimize, dcgan.py:154-155):
        leaves the local gradients in the d group's arena.  Returns (d_loss_real, d_loss_fake) as
        1-element device tensors (d_loss is their sum, dcgan.py:139).

        keep_g=True records the generator pass (activations and batch-norm statistics) so that the
        next g_grads(reuse_g=True) on the same z / y, with the generator's weights untouched in
        between, does not run it again."""
        with V.use_store(self.store):
            self.d_optim.begin()
            self._g_pass = None
            self._d_bound = []
            if keep_g:
                with ops.moving_update_log() as log, ops.taping():
                    G_taped = self.generator(z, y)
                self._g_pass = (G_taped, log, z, y, self._g_bound)
                G = G_taped.detach()                           # gradients stop at d_vars
            else:
                with torch.no_grad():
                    G = self.generator(z, y)                   # gradients stop at d_vars: no G tape
            if self.stack_d_passes:
                # D(real) and D(fake) share weights and differ only in their batch-norm statistics:
                # both passes run as one batch of 2B with per-half statistics (dcgan.py:113-114)
                both = ops.stack_batches(images, G)
                y2 = None if y is None else ops.stack_batches(y, y)
                _, logits = self.discriminator(both, y2, reuse=True, passes=2, with_prob=False)
                d_loss_real, d_loss_fake = ops.stacked_sigmoid_cross_entropy_means(logits, (1.0, 0.0))
            else:
                _, D_logits = self.discriminator(images, y, reuse=False)
                _, D_logits_ = self.discriminator(G, y, reuse=True)
                d_loss_real = ops.sigmoid_cross_entropy_mean(D_logits, 1.0)      # dcgan.py:129-132
                d_loss_fake = ops.sigmoid_cross_entropy_mean(D_logits_, 0.0)
            ops.backward([d_loss_real, d_loss_fake], **self._staged(self.d_optim, self._d_bound))
            self._d_bound = ()
            self.d_optim.finish_grads()
        return d_loss_real.detach(), d_loss_fake.detach()

    def g_grads(self, z, y=None, reuse_g=False):
        """Forward + backward of g_loss w.r.t. g_vars (through D, whose weights get no gradient).

        The reference evaluates G(z) once in the d_optim run and again in the first g_optim run
        (dcgan.py:228-247) with the same z and the same generator weights: the second evaluation
        repeats the first bit for bit.  reuse_g=True takes the pass d_grads(keep_g=True) recorded
        instead of recomputing it, and applies the batch-norm moving-average updates that second
        evaluation would have made (gan_bn_moving_update rebuilds the batch variance from the saved 1/sqrt(var+eps),
        so the moving variance matches a second evaluation to float32 rounding, not bit for bit; the trained
        weights never read it)."""
        with V.use_store(self.store):
            self.g_optim.begin()
            kept, self._g_pass = getattr(self, '_g_pass', None), None
            self._d_bound = []
            if reuse_g and kept is not None and kept[2] is z and kept[3] is y:
                G, log, bound = kept[0], kept[1], kept[4]
                log.replay()
            else:
                G = self.generator(z, y)
                bound = self._g_bound
            _, D_logits_ = self.discriminator(G, y, reuse=True, with_prob=False)
            g_loss = ops.sigmoid_cross_entropy_mean(D_logits_, 1.0)          # dcgan.py:133-134
            ops.backward(g_loss, **self._staged(self.g_optim, [bound]))
            self._d_bound, self._g_bound = (), None
            self.g_optim.finish_grads()
        return g_loss.detach()

    def d_update(self, images, z, y=None, keep_g=False):
        """sess.run([d_optim, ...]) of dcgan.py:228-233 / 263-264."""
        losses = self.d_grads(images, z, y, keep_g=keep_g)
        self.d_optim.all_reduce(self.world_size, self.process_group)
        self.d_optim.adam(self.world_size)
        return losses

    def g_update(self, z, y=None, reuse_g=False):
        """sess.run([g_optim, ...]) of dcgan.py:237-247 / 268-275."""
        g_loss = self.g_grads(z, y, reuse_g=reuse_g)
        self.g_optim.all_reduce(self.world_size, self.process_group)
        self.g_optim.adam(self.world_size)
        return g_loss

    def eval_losses(self, images, z, y=None):
        """The three logging re-evaluations of dcgan.py:249-260 / 277-279: forward only, on the
        post-update weights, each its own run (so each also moves the BN moving averages)."""
        with V.use_store(self.store), torch.no_grad():
            errD_fake = ops.sigmoid_cross_entropy_mean(self.discriminator(self.generator(z, y), y, reuse=True)[1], 0.0)
            errD_real = ops.sigmoid_cross_entropy_mean(self.discriminator(images, y, reuse=True)[1], 1.0)
            errG = ops.sigmoid_cross_entropy_mean(self.discriminator(self.generator(z, y), y, reuse=True)[1], 1.0)
        return errD_fake, errD_real, errG

    def train_iteration(self, images, z, y=None, log_evals=False, share_g_pass=None):
        """One iteration of the loop body dcgan.py:226-279.

        share_g_pass (default on) lets the first generator update reuse the generator pass of the
        discriminator update instead of repeating it; the results are those of the run-by-run
        schedule (share_g_pass=False)."""
        if share_g_pass is None:
            share_g_pass = True
        d_real, d_fake = self.d_update(images, z, y, keep_g=share_g_pass)
        g1 = self.g_update(z, y, reuse_g=share_g_pass)
        # Run g_optim twice to make sure that d_loss does not go to zero (different from paper)
        g2 = self.g_update(z, y)
        out = {'d_loss_real': d_real, 'd_loss_fake': d_fake, 'g_loss_1': g1, 'g_loss_2': g2}
        if log_evals:
            f, r, g = self.eval_losses(images, z, y)
            out.update(errD_fake=f, errD_real=r, errG=g)
        return out

    def iteration_runner(self, images, z, y=None, segmented=None):
        """The loop body dcgan.py:226-275 as replayable HIP graphs over static input buffers.

        One GPU: the whole iteration is one graph.  Data parallel: the iteration is cut at its three
        gradient exchanges into four graphs sharing one memory pool (D pass + backward | D Adam +
        G update 1 | G Adam + G update 2 | G Adam) and the all-reduces of the flat gradient arenas
        are issued between them on the same stream.  Returns run(images, z, y=None) -> the dict
        train_iteration returns."""
        from .graph import iteration_runner
        run = iteration_runner(self.store, [self.d_optim, self.g_optim], lambda a, b, c: self.train_iteration(a, b, c),
                               [images, z, y], self.world_size, self.process_group, segmented)
        return lambda im, zz, yy=None: run(im, zz, yy)

    # ------------------------------------------------------------------------------------ train loop
    def _load_files_u8(self, files, pool_map=None):
        """dcgan.py:210-221 up to the scaling: centre-crop / resize a list of image files to one uint8 NHWC
        batch.  `pool_map(fn, items)` (order-keeping) spreads the files over worker threads."""
        from .misc import get_image_u8
        one = lambda f: get_image_u8(f, input_height=self.input_height, input_width=self.input_width,
                                     resize_height=self.output_height, resize_width=self.output_width, crop=self.crop,
                                     grayscale=self.grayscale)
        arr = np.array(pool_map(one, files) if pool_map is not None else [one(f) for f in files], dtype=np.uint8)
        return arr[:, :, :, None] if self.grayscale else arr

    @staticmethod
    def _scale_u8(images):
        """uint8 batch -> the network's input range, with the arithmetic of misc.py:90 followed by the
        float32 cast of dcgan.py:221 (`x / 127.5 - 1.` in float64): bit-identical on host and GPU."""
        if torch.is_tensor(images):
            return (images.double() / 127.5 - 1.0).float()
        return (np.asarray(images) / 127.5 - 1.).astype(np.float32)

    def _load_files(self, files, pool_map=None):
        """dcgan.py:210-221: a list of image files as one float32 NHWC batch in [-1, 1]."""
        return self._scale_u8(self._load_files_u8(files, pool_map))

    def train(self, data_X=None, data_y=None, max_iters=None, data_files=None, sample_every=100):
        """dcgan.py:151-320.  Data comes from in-memory arrays ([N,H,W,C] in the network's input
        range, plus one-hot labels for the conditional model: MNIST via load_mnist), or from image
        files: `data_files`, else ./data/<dataset>/<input_fname_pattern> as in the reference
        (:200-201).  Every `sample_every` iterations a grid of sampler outputs is written to
        opts.sample_dir (:287-317; 0 disables it).

        Batches are prepared ahead of the GPU on a background thread (prefetch.BatchPrefetcher), and the
        loop body is replayed as HIP graphs (iteration_runner) unless opts.use_graph is false; the
        logging evaluations, the sampler and the checkpoints run eagerly between replays.  Image files
        travel as uint8 and are scaled on the GPU; the first epoch also leaves them in GPU memory
        (opts.dataset_cache_gb, default 32: CelebA at 64x64 is 2.4 GB), so that later epochs decode nothing.

        Data parallel (world_size N): the ranks take consecutive batches of that sequence (N x batch_size
        items per step, each rank its own z), rank 0 prints, samples and saves.

        Deliberate deviation: the reference re-evaluates errD_fake / errD_real / errG after EVERY iteration
        (dcgan.py:249-260, 277-279), and each of those training-mode passes also moves the batch-norm moving
        averages.  Here they run on logged iterations only: with opts.print_every == 1 (the default) that is
        every iteration, as in the reference; a larger print_every (an extension) skips the evaluations and
        their moving-average updates in between, which changes sampler output and checkpointed statistics,
        never the trained weights.  Likewise the sample step runs the sampler only, not the reference's extra
        d_loss / g_loss evaluation (:289-296)."""
        from glob import glob
        from .misc import save_images
        from .prefetch import BatchPrefetcher, to_device
        from .summary import ScalarLog
        opts = self.opts
        if data_X is None and data_files is None:
            data_X, data_y = self.data_X, self.data_y
        if data_X is None and data_files is None and opts.dataset != 'mnist':
            data_files = sorted(glob(os.path.join("./data", opts.dataset, self.input_fname_pattern)))
        if data_X is None and not data_files:
            raise RuntimeError("no training data: pass data_X (and data_y) or data_files, or place MNIST under "
                               "./data/mnist / images under ./data/%s" % opts.dataset)
        dev = self.store.device
        W, R = self.world_size, self.rank
        rng = np.random.RandomState(getattr(opts, 'seed', 0))
        sample_num = min(self.batch_size, 64)
        sample_z = torch.from_numpy(rng.uniform(-1, 1, size=(self.batch_size, self.z_dim)).astype(np.float32)).to(dev)
        if R > 0:
            rng = np.random.RandomState(rank_seed(getattr(opts, 'seed', 0), R))
        sample_y = None
        if self.y_dim and data_y is not None:
            sample_y = torch.from_numpy(np.asarray(data_y[:self.batch_size], dtype=np.float32)).to(dev)
        sample_dir = getattr(opts, 'sample_dir', './samples')
        start_time = time.time()
        could_load, checkpoint_counter = self.load(self.checkpoint_dir)
        if could_load:
            self.counter = checkpoint_counter
        if R == 0:
            print(" [*] Load SUCCESS" if could_load else " [!] Load failed...")
        iters = 0
        use_graph = bool(getattr(opts, 'use_graph', True))
        scalars = ScalarLog(getattr(opts, 'log_dir', None), enabled=(R == 0 and bool(getattr(opts, 'write_scalars', False))))
        run = None
        cache, cached = None, [0]          # uint8 images of the file data set in GPU memory; batches stored so far
        for epoch in range(opts.nEpochs):
            n_items = len(data_X) if data_X is not None else len(data_files)
            batch_idxs = int(min(n_items, opts.train_size)) // (opts.batch_size * W)
            items = lambda idx: slice(shard_start(idx, R, W, opts.batch_size), shard_start(idx, R, W, opts.batch_size) + opts.batch_size)

            from_files = data_X is None
            if from_files and cache is None and epoch == 0:
                nbytes = batch_idxs * opts.batch_size * self.output_height * self.output_width * self.c_dim
                if 0 < nbytes <= float(getattr(opts, 'dataset_cache_gb', 32)) * 2 ** 30:
                    cache = torch.empty((batch_idxs * opts.batch_size, self.output_height, self.output_width, self.c_dim),
                                        dtype=torch.uint8, device=dev)

            def device_batches():
                """(idx, images, labels) on the GPU: from the resident uint8 copy once an epoch has filled
                it, else from the input pipeline."""
                if from_files and cache is not None and cached[0] >= batch_idxs:
                    for idx in range(0, batch_idxs):
                        labels = None
                        if self.y_dim and data_y is not None:
                            labels = torch.from_numpy(np.asarray(data_y[items(idx)], dtype=np.float32)).to(dev)
                        yield idx, self._scale_u8(cache[idx * opts.batch_size:(idx + 1) * opts.batch_size]), labels
                    return

                def make(idx):
                    sl = items(idx)
                    images = (np.asarray(data_X[sl], dtype=np.float32) if not from_files
                              else self._load_files_u8(data_files[sl], feed.pool_map))
                    labels = np.asarray(data_y[sl], dtype=np.float32) if (self.y_dim and data_y is not None) else None
                    return images, labels

                feed = BatchPrefetcher(make, range(0, batch_idxs), depth=int(getattr(opts, 'prefetch_batches', 4)),
                                       workers=int(getattr(opts, 'num_preprocess_threads', 4)))
                try:
                    for idx, host_batch in feed:
                        images, labels = to_device(host_batch, dev)
                        if from_files:
                            if cache is not None and cached[0] == idx:
                                cache[idx * opts.batch_size:(idx + 1) * opts.batch_size].copy_(images)
                                cached[0] = idx + 1
                            images = self._scale_u8(images)
                        yield idx, images, labels
                finally:
                    feed.close()

            for idx, batch_images, batch_labels in device_batches():
                batch_z = torch.from_numpy(
                    rng.uniform(-1, 1, [opts.batch_size, self.z_dim]).astype(np.float32)).to(dev)
                log = (idx % self.print_every == 0)
                if use_graph:
                    if run is None:
                        run = self.iteration_runner(batch_images, batch_z, batch_labels)
                    out = dict(run(batch_images, batch_z, batch_labels))
                    if log:
                        f, r, g = self.eval_losses(batch_images, batch_z, batch_labels)
                        out.update(errD_fake=f, errD_real=r, errG=g)
                else:
                    out = self.train_iteration(batch_images, batch_z, batch_labels, log_evals=log)
                self.counter += 1
                if log and R == 0:
                    print("Epoch: [%2d] [%4d/%4d] time: %4.4f, d_loss: %.8f, g_loss: %.8f"
                          % (epoch, idx, batch_idxs, time.time() - start_time,
                             float(out['errD_fake']) + float(out['errD_real']), float(out['errG'])))
                    scalars.add(self.counter, d_loss_real=out['d_loss_real'], d_loss_fake=out['d_loss_fake'],
                                d_loss=float(out['d_loss_real']) + float(out['d_loss_fake']), g_loss=out['g_loss_2'])
                if R == 0 and sample_every and np.mod(self.counter, sample_every) == 1 and (not self.y_dim or sample_y is not None):
                    samples = self.sampler(sample_z, sample_y).cpu().numpy()[:sample_num]
                    manifold_h = int(np.ceil(np.sqrt(samples.shape[0])))
                    manifold_w = int(np.floor(np.sqrt(samples.shape[0])))
                    if not os.path.exists(sample_dir):
                        os.makedirs(sample_dir)
                    save_images(samples[:manifold_h * manifold_w], [manifold_h, manifold_w],
                                os.path.join(sample_dir, 'train_{:02d}_{:04d}.png'.format(epoch, idx)))
                if R == 0 and np.mod(self.counter, 500) == 2:
                    self.save(opts.checkpoint_dir, self.counter)
                iters += 1
                if max_iters is not None and iters >= max_iters:
                    scalars.close()
                    return                  # (closing the generator stops the input pipeline)
        scalars.close()

    # ------------------------------------------------------------------------------------ data
    def load_mnist(self):
        """dcgan.py:464-499: idx files under ./data/mnist, train+test concatenated, shuffled with
        seed 547, scaled to [0,1], labels one-hot."""
        data_dir = os.path.join("./data", self.dataset)

        def read(name, offset, shape):
            with open(os.path.join(data_dir, name), 'rb') as fd:
                loaded = np.frombuffer(fd.read(), dtype=np.uint8)
            return loaded[offset:].reshape(shape).astype(np.float64)
        trX = read('train-images-idx3-ubyte', 16, (60000, 28, 28, 1))
        trY = read('train-labels-idx1-ubyte', 8, (60000,))
        teX = read('t10k-images-idx3-ubyte', 16, (10000, 28, 28, 1))
        teY = read('t10k-labels-idx1-ubyte', 8, (10000,))
        X = np.concatenate((trX, teX), axis=0)
        y = np.concatenate((trY, teY), axis=0).astype(np.int64)
        seed = 547
        np.random.seed(seed)
        np.random.shuffle(X)
        np.random.seed(seed)
        np.random.shuffle(y)
        y_vec = np.zeros((len(y), self.y_dim), dtype=np.float64)
        for i, label in enumerate(y):
            y_vec[i, y[i]] = 1.0
        return X / 255., y_vec

    # ------------------------------------------------------------------------------------ checkpoints
    @property
    def model_dir(self):
        """dcgan.py:501-503"""
        return "{}_{}_{}_{}_{}".format(self.model, self.dataset, self.batch_size, self.output_height, self.output_width)

    def save(self, checkpoint_dir, step):
        """dcgan.py:505-514: <checkpoint_dir>/<model_dir>/DCGAN.model-<step> (an .npz of every named
        variable, the Adam slots and step counts)."""
        model_name = "DCGAN.model"
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        if not os.path.exists(checkpoint_dir):
            os.makedirs(checkpoint_dir)
        return save_checkpoint(self.store, {'d': self.d_optim, 'g': self.g_optim},
                               os.path.join(checkpoint_dir, model_name), step)

    def load(self, checkpoint_dir):
        """dcgan.py:516-530: newest checkpoint of this model_dir; the step comes from the file name."""
        print(" [*] Reading checkpoints...")
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        path = latest_checkpoint(checkpoint_dir)
        if path:
            ckpt_name = os.path.basename(path)
            load_checkpoint(self.store, {'d': self.d_optim, 'g': self.g_optim}, path)
            counter = int(next(re.finditer(r"(\d+)(?!.*\d)", ckpt_name[:-len('.npz')])).group(0))
            print(" [*] Success to read {}".format(ckpt_name))
            return True, counter
        print(" [*] Failed to find a checkpoint")
        return False, 0


def _sigmoid_value(logits):
    """tf.nn.sigmoid(h) returned next to the logits (dcgan.py:334,351); value only, no gradient."""
    with torch.no_grad():
        return ops.sigmoid(logits.detach())


# ---------------------------------------------------------------------------------------------
# checkpoint files shared by the three models
# ---------------------------------------------------------------------------------------------
def save_checkpoint(store, optims, prefix, step):
    arrays = {'var/' + n: a for n, a in store.state_dict().items()}
    for key, opt in optims.items():
        g = opt.group
        arrays['adam/%s/m' % key] = g.m.cpu().numpy()
        arrays['adam/%s/v' % key] = g.v.cpu().numpy()
        arrays['adam/%s/step' % key] = opt.step_dev.cpu().numpy()
    path = "%s-%d.npz" % (prefix, step)
    np.savez(path, **arrays)
    with open(os.path.join(os.path.dirname(path), 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(path))
    return path


def latest_checkpoint(checkpoint_dir):
    index = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.exists(index):
        return None
    m = re.search(r'model_checkpoint_path: "(.*)"', open(index).read())
    if not m:
        return None
    path = os.path.join(checkpoint_dir, m.group(1))
    return path if os.path.exists(path) else None


def load_checkpoint(store, optims, path):
    data = np.load(path, allow_pickle=False)
    store.load_state_dict({k[len('var/'):]: data[k] for k in data.files if k.startswith('var/')})
    for key, opt in optims.items():
        g = opt.group
        g.m.copy_(torch.from_numpy(data['adam/%s/m' % key]).to(g.m.device))
        g.v.copy_(torch.from_numpy(data['adam/%s/v' % key]).to(g.v.device))
        opt.step_dev.copy_(torch.from_numpy(data['adam/%s/step' % key]).to(opt.step_dev.device))
