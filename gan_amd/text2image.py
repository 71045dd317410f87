"""Text-to-image GAN-CLS on the HIP hot path, mirroring the reference's models/text2image.py:
generator (text2image.py:175-198), sampler (:148-172), discriminator (:201-222), the matching-aware
losses of build_model (:73-128) and the per-batch schedule of train (:243-279): one D update over
(real, right text) / (wrong image, right text) / (fake, right text), then two G updates.

`train()` takes in-memory arrays; `train_from_files()` follows the reference's flowers pipeline
(load_training_data :287-322, get_training_batch :389-417) for the DS-SJE caption files, which are
plain .npy arrays.  The skip-thought captions live in an HDF5 file and need h5py.
"""
import glob
import os
import random

import numpy as np
import torch

from . import ops
from . import variables as V
from .dcgan import latest_checkpoint, load_checkpoint, save_checkpoint
from .dp import rank_of, rank_seed, shard_start
from .optim import AdamOptimizer


class Text2Image(object):
    '''Text-to-image GAN-CLS (text2image.py).  Options read (text2image.py:16-25 lists them): z_dim (noise, 100),
    t_dim (reduced text feature, 256), input_height (image size, 64), gf_dim / df_dim (first-layer filters, 64),
    caption_vector_length (2400 skip-thought / 1024 DS-SJE unless given), batch_size (64), textEncoder.'''

    def __init__(self, sess, options, store=None, world_size=1, process_group=None):
        self.sess = sess
        self.model = getattr(options, 'model', 'Text2Image')
        # text2image.py:36-59: the same attribute names, read from the same options
        encoders = {'skipthought': (2400, 5), 'DSSJE': (1024, 10)}      # caption vector length, captions per image
        if options.textEncoder not in encoders:
            raise ValueError('unknown text encoder')
        self.textEncoder = options.textEncoder
        default_length, self.nCapsPerImage = encoders[self.textEncoder]
        self.caption_vector_length = (options.caption_vector_length if options.caption_vector_length is not None
                                      else default_length)
        self.image_size = options.input_height
        for name in ('batch_size', 'z_dim', 't_dim', 'gf_dim', 'df_dim', 'nEpochs', 'dataset', 'lr', 'beta1'):
            setattr(self, name, getattr(options, name))
        for name, default in (('sample_dir', './samples'), ('checkpoint_dir', './checkpoint'), ('resume_model', None),
                              ('data_dir', 'data'), ('save_every_batch', 50)):
            setattr(self, name, getattr(options, name, default))
        self.use_graph = bool(getattr(options, 'use_graph', True))
        self.prefetch_batches = int(getattr(options, 'prefetch_batches', 4))
        self._runner = None
        self.world_size, self.process_group = world_size, process_group
        self.rank = rank_of(world_size, process_group)
        self.stack_d_passes = True      # D update: the three D applications as one stacked batch
        self.store = store if store is not None else V.VariableStore(seed=getattr(options, 'seed', 0))

        for name in ('g_bn0', 'g_bn1', 'g_bn2', 'g_bn3', 'd_bn1', 'd_bn2', 'd_bn3', 'd_bn4'):     # text2image.py:61-69
            setattr(self, name, ops.batch_norm(name=name))

        self.build_model()

    def build_model(self):
        """Create the variables in the reference's order (generator, then discriminator:
        text2image.py:80-82), split them by 'd_' / 'g_' (:94-96) and attach the two Adam optimisers
        (:231-232)."""
        dev = self.store.device
        bs, s = self.batch_size, self.image_size
        with V.use_store(self.store), torch.no_grad():
            z = torch.zeros(bs, self.z_dim, device=dev)
            cap = torch.zeros(bs, self.caption_vector_length, device=dev)
            img = torch.zeros(bs, s, s, 3, device=dev)
            self.generator(z, cap)
            self.discriminator(img, cap)
            for name, var in self.store.vars.items():
                if name.endswith('moving_mean'):
                    var.data.zero_()
                elif name.endswith('moving_variance'):
                    var.data.fill_(1.0)
        t_vars = self.store.trainable_variables()
        self.d_vars = [var for var in t_vars if 'd_' in var.name]
        self.g_vars = [var for var in t_vars if 'g_' in var.name]
        self.d_group = self.store.pack('d', self.d_vars)
        self.g_group = self.store.pack('g', self.g_vars)
        self.overlap_exchange = self.world_size > 1 or self.process_group is not None
        if self.overlap_exchange:
            # gradients of the layers nearest the loss are exchanged while the backward pass of the
            # remaining layers runs (ops.backward, DCGAN.build_model)
            self.d_group.split_early('discriminator/d_h3_conv/w')
            self.g_group.split_early('generator/g_h2/w')
        if self.world_size > 1:
            self.d_group.broadcast(self.process_group)
            self.g_group.broadcast(self.process_group)
        self.d_optim = AdamOptimizer(self.lr, beta1=self.beta1).attach(self.d_group)
        self.g_optim = AdamOptimizer(self.lr, beta1=self.beta1).attach(self.g_group)

    # Sample Images for a text embedding (text2image.py:148-172)
    def sampler(self, t_z, t_text_embedding):
        with V.use_store(self.store), torch.no_grad():
            return self.generator(t_z, t_text_embedding, train=False)

    # GENERATOR (text2image.py:175-198)
    def generator(self, t_z, t_text_embedding, train=True):
        with ops.variable_scope('generator') as scope:
            if not train:
                scope.reuse_variables()
            s = self.image_size
            s2, s4, s8, s16 = int(s / 2), int(s / 4), int(s / 8), int(s / 16)
            bs = self.batch_size

            reduced_text_embedding = ops.lrelu(ops.linear(t_text_embedding, self.t_dim, 'g_embedding'))
            z_concat = ops.concat([t_z, reduced_text_embedding], 1)
            z_ = ops.linear(z_concat, self.gf_dim * 8 * s16 * s16, 'g_h0_lin')
            h0 = z_.reshape(-1, s16, s16, self.gf_dim * 8)
            h0 = self.g_bn0(h0, train=train, act='relu')

            ng = 1 if train else None           # the deconvs feed training-mode batch norms of one group
            h1 = ops.deconv2d(h0, [bs, s8, s8, self.gf_dim * 4], name='g_h1', norm_groups=ng)
            h1 = self.g_bn1(h1, train=train, act='relu')
            self._g_bound = h1                  # the layers after it own the early part of the g group

            h2 = ops.deconv2d(h1, [bs, s4, s4, self.gf_dim * 2], name='g_h2', norm_groups=ng)
            h2 = self.g_bn2(h2, train=train, act='relu')

            h3 = ops.deconv2d(h2, [bs, s2, s2, self.gf_dim * 1], name='g_h3', norm_groups=ng)
            h3 = self.g_bn3(h3, train=train, act='relu')

            # tf.tanh(h4)/2. + 0.5, applied in the deconv's epilogue
            return ops.deconv2d(h3, [bs, s, s, 3], name='g_h4', act='tanh', act_scale=0.5, act_shift=0.5)

    # DISCRIMINATOR (text2image.py:201-222)
    def discriminator(self, image, t_text_embedding, reuse=False, passes=1, with_prob=True):
        """passes=k: `image` and `t_text_embedding` hold k batches stacked along axis 0 and the call
        stands for k applications of the network (shared weights, batch-norm statistics per batch).
        with_prob=False skips the sigmoid of the first return value (None then)."""
        bs = self.batch_size * passes
        with ops.variable_scope("discriminator") as scope:
            if reuse:
                scope.reuse_variables()

            h0 = ops.conv2d(image, self.df_dim, name='d_h0_conv', act='lrelu')                                       # 32
            h1 = self.d_bn1(ops.conv2d(h0, self.df_dim * 2, name='d_h1_conv', norm_groups=passes), act='lrelu', passes=passes)         # 16
            h2 = self.d_bn2(ops.conv2d(h1, self.df_dim * 4, name='d_h2_conv', norm_groups=passes), act='lrelu', passes=passes)         # 8
            if isinstance(self._d_bound, list):
                self._d_bound.append(h2)        # the layers after it own the early part of the d group
            h3 = self.d_bn3(ops.conv2d(h2, self.df_dim * 8, name='d_h3_conv', norm_groups=passes), act='lrelu', passes=passes)         # 4

            # ADD TEXT EMBEDDING TO THE NETWORK: expand_dims x2 + tile [1,4,4,1] + concat on channels
            reduced_text_embeddings = ops.lrelu(ops.linear(t_text_embedding, self.t_dim, 'd_embedding'))
            h3_concat = ops.conv_cond_concat(h3, reduced_text_embeddings.reshape(bs, 1, 1, self.t_dim))
            h3_new = self.d_bn4(ops.conv2d(h3_concat, self.df_dim * 8, 1, 1, 1, 1, name='d_h3_conv_new', norm_groups=passes), act='lrelu',
                                passes=passes)                                                                     # 4

            h4 = ops.linear(h3_new.reshape(bs, -1), 1, 'd_h3_lin')
            prob = None
            if with_prob:
                with torch.no_grad():
                    prob = ops.sigmoid(h4.detach())
            return prob, h4

    # ---------------------------------------------------------------------------------- updates
    _d_bound = ()           # set per update; the networks record their boundary activations here
    _g_bound = None

    def _staged(self, optim, boundary):
        """Arguments of ops.backward for an update whose gradient exchange overlaps its backward pass."""
        if not self.overlap_exchange:
            return {}
        ws, pg = self.world_size, self.process_group
        return dict(boundary=boundary, at_boundary=lambda: optim.all_reduce_early(ws, pg))

    def d_grads(self, real_image, wrong_image, real_caption, z, keep_g=False):
        """d_loss = d_loss1 + d_loss2 + d_loss3 (text2image.py:88-92) and its gradient w.r.t. d_vars.
        The three discriminator applications (real, wrong, fake image with the same captions) run
        as one stacked batch with per-application batch-norm statistics.  keep_g=True records the
        generator pass for the next g_grads(reuse_g=True), as in DCGAN.d_grads."""
        with V.use_store(self.store):
            self.d_optim.begin()
            self._g_pass = None
            self._d_bound = []
            if keep_g:
                with ops.moving_update_log() as log, ops.taping():
                    taped = self.generator(z, real_caption)
                self._g_pass = (taped, log, z, real_caption, self._g_bound)
                fake_image = taped.detach()
            else:
                with torch.no_grad():
                    fake_image = self.generator(z, real_caption)
            if self.stack_d_passes:
                images = ops.stack_batches(real_image, wrong_image, fake_image)
                captions = ops.stack_batches(real_caption, real_caption, real_caption)
                _, logits = self.discriminator(images, captions, reuse=True, passes=3, with_prob=False)
                d_loss1, d_loss2, d_loss3 = ops.stacked_sigmoid_cross_entropy_means(logits, (1.0, 0.0, 0.0))
            else:
                _, real_logits = self.discriminator(real_image, real_caption)
                _, wrong_logits = self.discriminator(wrong_image, real_caption, reuse=True)
                _, fake_logits = self.discriminator(fake_image, real_caption, reuse=True)
                d_loss1 = ops.sigmoid_cross_entropy_mean(real_logits, 1.0)
                d_loss2 = ops.sigmoid_cross_entropy_mean(wrong_logits, 0.0)
                d_loss3 = ops.sigmoid_cross_entropy_mean(fake_logits, 0.0)
            ops.backward([d_loss1, d_loss2, d_loss3], **self._staged(self.d_optim, self._d_bound))
            self._d_bound = ()
            self.d_optim.finish_grads()
        return d_loss1.detach(), d_loss2.detach(), d_loss3.detach(), fake_image

    def g_grads(self, real_caption, z, reuse_g=False):
        """g_loss (text2image.py:86) and its gradient w.r.t. g_vars.  reuse_g=True takes the
        generator pass the preceding d_grads(keep_g=True) recorded (same z, captions and generator
        weights: text2image.py:251-265 evaluates it twice) and replays its moving-average updates."""
        with V.use_store(self.store):
            self.g_optim.begin()
            kept, self._g_pass = getattr(self, '_g_pass', None), None
            if reuse_g and kept is not None and kept[2] is z and kept[3] is real_caption:
                fake_image, bound = kept[0], kept[4]
                kept[1].replay()
            else:
                fake_image = self.generator(z, real_caption)
                bound = self._g_bound
            _, fake_logits = self.discriminator(fake_image, real_caption, reuse=True, with_prob=False)
            g_loss = ops.sigmoid_cross_entropy_mean(fake_logits, 1.0)
            ops.backward(g_loss, **self._staged(self.g_optim, [bound]))
            self._g_bound = None
            self.g_optim.finish_grads()
        return g_loss.detach(), fake_image.detach()

    def d_update(self, real_image, wrong_image, real_caption, z, keep_g=False):
        out = self.d_grads(real_image, wrong_image, real_caption, z, keep_g=keep_g)
        self.d_optim.all_reduce(self.world_size, self.process_group)
        self.d_optim.adam(self.world_size)
        return out

    def g_update(self, real_caption, z, reuse_g=False):
        out = self.g_grads(real_caption, z, reuse_g=reuse_g)
        self.g_optim.all_reduce(self.world_size, self.process_group)
        self.g_optim.adam(self.world_size)
        return out

    def train_iteration(self, real_image, wrong_image, real_caption, z):
        """One batch of text2image.py:249-276: DISCR UPDATE, GEN UPDATE, GEN UPDATE TWICE."""
        d1, d2, d3, _ = self.d_update(real_image, wrong_image, real_caption, z, keep_g=True)
        g1, _ = self.g_update(real_caption, z, reuse_g=True)
        g2, gen = self.g_update(real_caption, z)
        return {'d_loss1': d1, 'd_loss2': d2, 'd_loss3': d3, 'g_loss_1': g1, 'g_loss_2': g2, 'generator': gen}

    def iteration_runner(self, real_image, wrong_image, real_caption, z, segmented=None):
        """train_iteration as replayable HIP graphs over static input buffers (one graph on one GPU, a
        chain cut at the gradient exchanges when data parallel).  Returns run(real, wrong, caption, z)."""
        from .graph import iteration_runner
        return iteration_runner(self.store, [self.d_optim, self.g_optim],
                                lambda a, b, c, d: self.train_iteration(a, b, c, d),
                                [real_image, wrong_image, real_caption, z], self.world_size, self.process_group, segmented)

    def _resume(self):
        """text2image.py:237-238: `--resume_model <path>` restores that checkpoint before training starts
        (the file save() wrote; its .npz suffix may be left out)."""
        path = self.resume_model
        if not path:
            return False
        if not os.path.exists(path) and os.path.exists(path + '.npz'):
            path = path + '.npz'
        load_checkpoint(self.store, {'d': self.d_optim, 'g': self.g_optim}, path)
        return True

    def _step(self, batch):
        """One iteration on a device batch (real, wrong, captions, z): graph replay unless use_graph is off."""
        if not self.use_graph:
            return self.train_iteration(*batch)
        if self._runner is None:
            self._runner = self.iteration_runner(*batch)
        return self._runner(*batch)

    def train(self, images, captions, max_iters=None, seed=0):
        """text2image.py:224-285 for in-memory data: images [N,s,s,3] floats in [0,1], captions
        [N, caption_vector_length].  The wrong image of a sample is a random other image
        (get_training_batch, :389-397).  Batches are prepared ahead of the GPU on a background thread."""
        from .prefetch import BatchPrefetcher, to_device
        dev = self.store.device
        W, R = self.world_size, self.rank          # data parallel: the ranks take consecutive batches
        rng = np.random.RandomState(rank_seed(seed, R))
        self._resume()
        n = len(images)
        iters = 0

        def make(batch_no):
            start = shard_start(batch_no, R, W, self.batch_size)
            sl = slice(start, start + self.batch_size)
            return (np.asarray(images[sl], dtype=np.float32),
                    np.asarray(images[rng.randint(0, n, self.batch_size)], dtype=np.float32),
                    np.asarray(captions[sl], dtype=np.float32),
                    rng.uniform(-1, 1, [self.batch_size, self.z_dim]).astype(np.float32))

        for i in range(self.nEpochs):
            batch_no = 0
            feed = BatchPrefetcher(make, range(n // (self.batch_size * W)), depth=self.prefetch_batches)
            for _, host_batch in feed:
                out = self._step(to_device(host_batch, dev))
                if R == 0:
                    d1, d2, d3 = float(out['d_loss1']), float(out['d_loss2']), float(out['d_loss3'])
                    print("epoch {} batch {}/{} (d1,d2,d3)=({:.4f}, {:.4f}, {:.4f}) (D,G)=({:.4f}, {:.4f})".format(
                        i, batch_no, n // (self.batch_size * W), d1, d2, d3, d1 + d2 + d3, float(out['g_loss_2'])))
                batch_no += 1
                if R == 0 and (batch_no % self.save_every_batch) == 0:
                    self.save(self.checkpoint_dir, batch_no)
                iters += 1
                if max_iters is not None and iters >= max_iters:
                    feed.close()
                    return
            if R == 0 and i % 5 == 0:
                self.save(self.checkpoint_dir, batch_no, note="epoch{}".format(i))

    # ---------------------------------------------------------------------------------- flowers files
    def load_training_data(self, data_dir=None, dataset=None, seed=None):
        """text2image.py:287-322 for dataset 'flowers': {'image_list', 'captions', 'data_length'}.

        DSSJE: captions are <data_dir>/flowers_caption_icml/<class>/<image>.npy, each [captions, 1024]
        (:303-316).  The reference takes its image list from the keys of flower_tv.hdf5 even then; the
        same names are the .npy stems, so the list is built from those when h5py is not there.  The
        first 75 % of the sorted list is the training split, shuffled (:296-298)."""
        data_dir = data_dir or self.data_dir
        dataset = dataset or self.dataset
        if dataset != 'flowers':
            raise NotImplementedError("only the flowers files are read here; MS-COCO batches come as arrays (train())")
        captions, image_list = {}, None
        h5_path = os.path.join(data_dir, 'flower_tv.hdf5')
        if self.textEncoder == 'skipthought' or os.path.exists(h5_path):
            try:
                import h5py
            except ImportError:
                if self.textEncoder == 'skipthought':
                    raise RuntimeError("skip-thought captions are stored in %s, which needs h5py" % h5_path)
                h5py = None
            if h5py is not None:
                with h5py.File(h5_path, 'r') as data:
                    captions = {k: np.array(v) for k, v in data.items()}
                image_list = sorted(captions)
        if self.textEncoder == 'DSSJE':
            icml = os.path.join(data_dir, 'flowers_caption_icml')
            captions = {}
            for classname in sorted(os.listdir(icml)):
                if os.path.isdir(os.path.join(icml, classname)):
                    for filename in glob.glob(os.path.join(icml, classname, '*.npy')):
                        captions[os.path.basename(filename[0:-4]) + '.jpg'] = np.load(filename, allow_pickle=False)
            if image_list is None:
                image_list = sorted(captions)
        img_75 = int(len(image_list) * 0.75)
        training_image_list = image_list[0:img_75]
        random.Random(seed).shuffle(training_image_list)
        return {'image_list': training_image_list, 'captions': captions, 'data_length': len(training_image_list)}

    def get_training_batch(self, batch_no, loaded_data, data_dir=None, rng=random):
        """text2image.py:389-417 (flowers): real images in list order, a uniformly drawn wrong image and
        one of the image's nCapsPerImage captions per sample, z ~ U(-1, 1).  Images are floats in [0, 1]."""
        from .misc import load_image_array
        data_dir = data_dir or self.data_dir
        bs, s = self.batch_size, self.image_size
        real_images = np.zeros((bs, s, s, 3), np.float32)
        wrong_images = np.zeros((bs, s, s, 3), np.float32)
        captions = np.zeros((bs, self.caption_vector_length), np.float32)
        names = loaded_data['image_list']
        image_files = []
        for cnt, i in enumerate(range(batch_no * bs, batch_no * bs + bs)):
            idx = i % len(names)
            image_file = os.path.join(data_dir, 'flowers/jpg/' + names[idx])
            real_images[cnt] = load_image_array(image_file, s, rng)
            wrong_image_id = rng.randint(0, len(names) - 1)
            wrong_images[cnt] = load_image_array(os.path.join(data_dir, 'flowers/jpg/' + names[wrong_image_id]), s, rng)
            random_caption = rng.randint(0, self.nCapsPerImage - 1)     # randint arguments are inclusive
            captions[cnt] = loaded_data['captions'][names[idx]][random_caption][0:self.caption_vector_length]
            image_files.append(image_file)
        z_noise = np.asarray([[rng.uniform(-1, 1) for _ in range(self.z_dim)] for _ in range(bs)], np.float32)
        return real_images, wrong_images, captions, z_noise, image_files

    def save_for_vis(self, real_images, generated_images, image_files):
        """text2image.py:343-358: the batch's real images (named after their files) and generated images as
        jpg files under <sample_dir>/text2image_<textEncoder> (scipy.misc.imsave's stretch to 0..255)."""
        from PIL import Image
        from .misc import bytescale
        sample_dir = os.path.join(self.sample_dir, 'text2image_' + self.textEncoder)
        if not os.path.exists(sample_dir):
            os.makedirs(sample_dir)
        real_images, generated_images = np.asarray(real_images), np.asarray(generated_images)
        for i in range(0, real_images.shape[0]):
            Image.fromarray(bytescale(real_images[i])).save(
                os.path.join(sample_dir, '{}_{}.jpg'.format(i, image_files[i].split('/')[-1])))
            Image.fromarray(bytescale(generated_images[i])).save(os.path.join(sample_dir, 'fake_image_{}.jpg'.format(i)))
        return sample_dir

    def train_from_files(self, data_dir=None, max_iters=None, seed=None):
        """The training loop of text2image.py:224-285 over the flowers files (batches read ahead of the GPU)."""
        from .prefetch import BatchPrefetcher, to_device
        dev = self.store.device
        W, R = self.world_size, self.rank          # data parallel: the ranks take consecutive batches
        if W > 1 and seed is None:
            seed = 0                               # every rank must shuffle the image list the same way
        rng = random.Random(rank_seed(seed, R))
        self._resume()
        loaded = self.load_training_data(data_dir, self.dataset, seed)
        if loaded['data_length'] == 0:
            raise RuntimeError("no flowers training images found under %s" % (data_dir or self.data_dir))
        n_batch = loaded['data_length'] // (self.batch_size * W)
        iters, out = 0, None
        n_feeds = (loaded['data_length'] + self.batch_size * W - 1) // (self.batch_size * W)   # the last one wraps around
        for i in range(self.nEpochs):
            batch_no = 0
            # one producer thread draws the random numbers of the batches in the order of the inline loop
            feed = BatchPrefetcher(lambda b: self.get_training_batch(b * W + R, loaded, data_dir, rng), range(n_feeds),
                                   depth=self.prefetch_batches)
            for _, host_batch in feed:
                real, wrong, cap, z, image_files = host_batch
                out = self._step(to_device((real, wrong, cap, z), dev))
                if R == 0:
                    d1, d2, d3 = float(out['d_loss1']), float(out['d_loss2']), float(out['d_loss3'])
                    print("epoch {} batch {}/{} (d1,d2,d3)=({:.4f}, {:.4f}, {:.4f}) (D,G)=({:.4f}, {:.4f})".format(
                        i, batch_no, n_batch, d1, d2, d3, d1 + d2 + d3, float(out['g_loss_2'])))
                batch_no += 1
                if R == 0 and (batch_no % self.save_every_batch) == 0:
                    print("Saving Images, Model")
                    self.save_for_vis(real.numpy(), out['generator'].cpu().numpy(), image_files)
                    self.save(self.checkpoint_dir, batch_no)
                iters += 1
                if max_iters is not None and iters >= max_iters:
                    feed.close()
                    return out
            if R == 0 and i % 5 == 0:
                self.save(self.checkpoint_dir, batch_no, note="epoch{}".format(i))
        return out

    @property
    def model_dir(self):
        """text2image.py:419-421"""
        return "{}_{}_{}_{}".format(self.model, self.dataset, self.batch_size, self.image_size)

    def save(self, checkpoint_dir, step, note=''):
        """text2image.py:330-341: <checkpoint_dir>/<model_dir>/<model>.model[_note]-<step>"""
        model_name = self.model + ".model" + (('_' + note) if note else '')
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        if not os.path.exists(checkpoint_dir):
            os.makedirs(checkpoint_dir)
        return save_checkpoint(self.store, {'d': self.d_optim, 'g': self.g_optim},
                               os.path.join(checkpoint_dir, model_name), step)

    def load(self, checkpoint_dir):
        path = latest_checkpoint(os.path.join(checkpoint_dir, self.model_dir))
        if not path:
            return False
        load_checkpoint(self.store, {'d': self.d_optim, 'g': self.g_optim}, path)
        return True
