"""CycleGAN trainer on the HIP hot path, mirroring the reference's models/cyclegan.py: graph wiring
of _build_model (:47-116), the four Adam optimisers (:123-130) and the five sequential runs of one
training iteration (:159-180).  Image loading, sampling to disk and summaries (host side) are not
part of the hot path; `train_iteration` takes the [N, s, s, 6] real_A|real_B tensor directly.
"""
from __future__ import division

import os
import time

import numpy as np
import torch

from . import ops
from . import variables as V
from .cyclegan_module import (abs_criterion, discriminator, generator_resnet, generator_unet, mae_criterion,
                              sce_criterion)
from .dcgan import latest_checkpoint, load_checkpoint, save_checkpoint
from .dp import rank_of, rank_seed, shard_start
from .optim import AdamOptimizer


class ImagePool(object):
    """misc.py:249-267: history of generated images fed to the discriminators.  Entries are the
    [fake_A, fake_B] pair of one iteration; `rand` / `randint` can be injected for reproducible tests."""

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
            self.num_img = self.num_img + 1
            return image
        if self.rand() > 0.5:
            idx = self.randint(self.maxsize)
            tmp = list(self.images[idx])
            self.images[idx] = image
            return tmp
        else:
            return image


class CycleGAN(object):
    def __init__(self, sess, opts, store=None, world_size=1, process_group=None):
        self.sess = sess
        # cyclegan.py:22-40: attribute names as in the reference
        self.batch_size, self.image_size, self.L1_lambda = opts.batch_size, opts.fine_size, opts.L1_lambda
        self.input_c_dim, self.output_c_dim = opts.input_nc, opts.output_nc
        self.dataset = opts.dataset
        self.dataset_dir = os.path.join(getattr(opts, 'data_dir', 'data'), opts.dataset)
        self.discriminator = discriminator
        self.generator = generator_resnet if opts.use_resnet else generator_unet
        self.criterionGAN = mae_criterion if opts.use_lsgan else sce_criterion
        self.options = opts                     # handed on to the network functions, which read these two from it
        opts.image_size, opts.output_c_dim = opts.fine_size, opts.output_nc

        self.model_dir = "%s_%s" % (self.dataset, self.image_size)
        self.world_size, self.process_group = world_size, process_group
        self.rank = rank_of(world_size, process_group)
        self.store = store if store is not None else V.VariableStore(seed=getattr(opts, 'seed', 0))

        self._build_model()
        self.pool = ImagePool(opts.max_size)

    def split(self, real_data):
        """real_A = real_data[..., :input_c_dim], real_B = the next output_c_dim channels (:53-54)."""
        a = real_data[:, :, :, :self.input_c_dim].contiguous()
        b = real_data[:, :, :, self.input_c_dim:self.input_c_dim + self.output_c_dim].contiguous()
        return a, b

    def _build_model(self):
        """Variable creation order of cyclegan.py:56-62; groups by name substring (:112-115)."""
        dev = self.store.device
        s = self.image_size
        with V.use_store(self.store), torch.no_grad():
            real_A = torch.zeros(self.batch_size, s, s, self.input_c_dim, device=dev)
            real_B = torch.zeros(self.batch_size, s, s, self.output_c_dim, device=dev)
            fake_B = self.generator(real_A, self.options, False, name="generatorA2B")
            fake_A = self.generator(real_B, self.options, False, name="generatorB2A")
            self.discriminator(fake_B, self.options, reuse=False, name="discriminatorB")
            self.discriminator(fake_A, self.options, reuse=False, name="discriminatorA")
            for name, var in self.store.vars.items():
                if name.endswith('moving_mean'):
                    var.data.zero_()
                elif name.endswith('moving_variance'):
                    var.data.fill_(1.0)
        t_vars = self.store.trainable_variables()
        self.db_vars = [var for var in t_vars if 'discriminatorB' in var.name]
        self.da_vars = [var for var in t_vars if 'discriminatorA' in var.name]
        self.g_vars_a2b = [var for var in t_vars if 'generatorA2B' in var.name]
        self.g_vars_b2a = [var for var in t_vars if 'generatorB2A' in var.name]
        opts = self.options
        groups = {'da': self.da_vars, 'db': self.db_vars, 'g_a2b': self.g_vars_a2b, 'g_b2a': self.g_vars_b2a}
        self.optims = {}
        for key, vs in groups.items():
            grp = self.store.pack(key, vs)
            if self.world_size > 1:
                grp.broadcast(self.process_group)
            self.optims[key] = AdamOptimizer(opts.lr, beta1=opts.beta1).attach(grp)     # cyclegan.py:123-130
        self.da_optim, self.db_optim = self.optims['da'], self.optims['db']
        self.g_a2b_optim, self.g_b2a_optim = self.optims['g_a2b'], self.optims['g_b2a']

    # ------------------------------------------------------------------------------------ runs
    def forward_fakes(self, real_data, keep=False):
        """sess.run([self.fake_A, self.fake_B]) (cyclegan.py:160-161).

        keep=True records both generator passes (activations + batch-norm statistics) so that the
        generator updates of the same iteration can take them instead of evaluating
        G_A2B(real_A) / G_B2A(real_B) again while that generator's weights are still unchanged."""
        real_A, real_B = self.split(real_data)
        self._kept = {}
        with V.use_store(self.store):
            if keep:
                with ops.taping():
                    with ops.moving_update_log() as log_b:
                        fake_B = self.generator(real_A, self.options, True, name="generatorA2B")
                    with ops.moving_update_log() as log_a:
                        fake_A = self.generator(real_B, self.options, True, name="generatorB2A")
                self._kept = {'a2b': (fake_B, log_b, real_data), 'b2a': (fake_A, log_a, real_data)}
            else:
                with torch.no_grad():
                    fake_B = self.generator(real_A, self.options, True, name="generatorA2B")
                    fake_A = self.generator(real_B, self.options, True, name="generatorB2A")
        return fake_A.detach(), fake_B.detach()

    def _first_pass(self, direction, real_data, source, want_tape):
        """G_<direction>(source image) for a generator update: the pass forward_fakes kept, while that
        generator has not been updated since (its moving averages still move as if it were
        evaluated again), or a fresh evaluation."""
        kept = self._kept.get(direction) if getattr(self, '_kept', None) else None
        name = "generatorA2B" if direction == 'a2b' else "generatorB2A"
        if kept is not None and kept[2] is real_data:
            kept[1].replay()
            return kept[0] if want_tape else kept[0].detach()
        return self.generator(source, self.options, True, name=name)

    def _finish(self, optim):
        optim.finish_grads()
        optim.all_reduce(self.world_size, self.process_group)
        optim.adam(self.world_size)

    def d_update(self, which, real, fake_sample):
        """db_optim / da_optim (cyclegan.py:76-85, 164-166, 173-175):
        loss = (criterionGAN(D(real), 1) + criterionGAN(D(fake_sample), 0)) / 2."""
        optim = self.db_optim if which == 'B' else self.da_optim
        name = "discriminator" + which
        with V.use_store(self.store):
            optim.begin()
            l_real = self.criterionGAN(self.discriminator(real, self.options, reuse=True, name=name), 1.0, 0.5)
            l_fake = self.criterionGAN(self.discriminator(fake_sample, self.options, reuse=True, name=name), 0.0, 0.5)
            ops.backward([l_real, l_fake])
            self._finish(optim)
        return l_real.detach(), l_fake.detach()

    def g_update(self, which, real_data):
        """g_a2b_optim / g_b2a_optim (cyclegan.py:56-68, 169-170, 178-179): the GAN term of one
        direction plus both L1 cycle terms, gradient w.r.t. that direction's generator only."""
        real_A, real_B = self.split(real_data)
        optim = self.g_a2b_optim if which == 'a2b' else self.g_b2a_optim
        with V.use_store(self.store):
            optim.begin()
            # the generator being updated needs its pass on tape; the other one only supplies a
            # constant input (tf.gradients prunes it the same way)
            # the translated image of the direction being trained feeds its discriminator AND the other
            # generator: ops.fork sums the two gradients
            fake_B = self._first_pass('a2b', real_data, real_A, want_tape=(which == 'a2b'))
            if which == 'a2b':
                fake_B, fake_B_d = ops.fork(fake_B)
            fake_A_ = self.generator(fake_B, self.options, True, name="generatorB2A")
            fake_A = self._first_pass('b2a', real_data, real_B, want_tape=(which == 'b2a'))
            if which == 'b2a':
                fake_A, fake_A_d = ops.fork(fake_A)
            fake_B_ = self.generator(fake_A, self.options, True, name="generatorA2B")
            if which == 'a2b':
                gan = self.criterionGAN(self.discriminator(fake_B_d, self.options, reuse=True, name="discriminatorB"), 1.0)
            else:
                gan = self.criterionGAN(self.discriminator(fake_A_d, self.options, reuse=True, name="discriminatorA"), 1.0)
            cyc_a = abs_criterion(real_A, fake_A_, self.L1_lambda)
            cyc_b = abs_criterion(real_B, fake_B_, self.L1_lambda)
            ops.backward([gan, cyc_a, cyc_b])
            self._finish(optim)
            if getattr(self, '_kept', None):
                self._kept.pop(which, None)         # this generator's weights have moved
        return gan.detach(), cyc_a.detach(), cyc_b.detach()

    def train_iteration(self, real_data):
        """The loop body cyclegan.py:159-180.  Losses come back as 1-element device tensors."""
        real_A, real_B = self.split(real_data)
        # Forward G network
        fake_A, fake_B = self.forward_fakes(real_data, keep=True)
        fake_A, fake_B = self._pooled(fake_A, fake_B)
        out = {}
        # Update D network
        r, f = self.d_update('B', real_B, fake_B)
        out['db_loss_real'], out['db_loss_fake'] = r, f
        # Update G network
        out['g_a2b_gan'], out['g_a2b_cyc_a'], out['g_a2b_cyc_b'] = self.g_update('a2b', real_data)
        # Update D network
        r, f = self.d_update('A', real_A, fake_A)
        out['da_loss_real'], out['da_loss_fake'] = r, f
        # Update G network
        out['g_b2a_gan'], out['g_b2a_cyc_a'], out['g_b2a_cyc_b'] = self.g_update('b2a', real_data)
        self._kept = {}
        return out

    def _pooled(self, fake_A, fake_B):
        """[fake_A, fake_B] = self.pool([fake_A, fake_B]) (cyclegan.py:162).  The pool's coin toss is host
        state, so a captured iteration is cut here: the pool is asked between two graphs and its answer is
        copied into the buffers the discriminator updates were captured on."""
        from .graph import SegmentedIteration
        if SegmentedIteration.active is None:
            if getattr(self, '_pool_bypass', False):        # warm-up runs of a capture leave the history alone
                return fake_A, fake_B
            return self.pool([fake_A, fake_B])
        out_A, out_B = torch.empty_like(fake_A), torch.empty_like(fake_B)

        def ask_pool():
            # the pool keeps copies: fake_A / fake_B are static buffers that the next replay overwrites
            a, b = self.pool([fake_A.clone(), fake_B.clone()])
            out_A.copy_(a)
            out_B.copy_(b)
        SegmentedIteration.cut(ask_pool)
        return out_A, out_B

    def iteration_runner(self, real_data):
        """train_iteration as a chain of HIP graphs over a static input buffer, cut at the image pool (and
        at the gradient exchanges when data parallel).  Returns run(real_data) -> the dict of losses."""
        from .graph import iteration_runner
        self._pool_bypass = True
        try:
            return iteration_runner(self.store, list(self.optims.values()), lambda a: self.train_iteration(a),
                                    [real_data], self.world_size, self.process_group, segmented=True)
        finally:
            self._pool_bypass = False

    @staticmethod
    def totals(out):
        """db_loss, g_loss_a2b, da_loss, g_loss_b2a as floats (the four scalars the reference logs)."""
        f = lambda k: float(out[k])
        return {'db_loss': f('db_loss_real') + f('db_loss_fake'),
                'g_loss_a2b': f('g_a2b_gan') + f('g_a2b_cyc_a') + f('g_a2b_cyc_b'),
                'da_loss': f('da_loss_real') + f('da_loss_fake'),
                'g_loss_b2a': f('g_b2a_gan') + f('g_b2a_cyc_a') + f('g_b2a_cyc_b')}

    def train(self, dataA=None, dataB=None, max_iters=None, seed=0, sample_every=100):
        """cyclegan.py:137-190.  dataA / dataB are lists of image files (default: the reference's
        ./<data_dir>/<dataset>/trainA|trainB/*.jpg, loaded with misc.load_data: resize to load_size,
        random crop to fine_size, random flip) or in-memory [N, s, s, c] float arrays in [-1, 1].
        Both are shuffled every epoch and paired by position; a batch is real_A | real_B on the
        channel axis.  Every `sample_every` iterations the translations of the current batch go to
        opts.sample_dir (sample_model, :216-236; 0 disables it).  Batches are prepared ahead of the GPU on a
        background thread, in loop order (the crop / flip draws keep their sequence); the iteration is
        replayed as HIP graphs unless opts.use_graph is false."""
        from glob import glob
        from .misc import load_data, save_images
        from .prefetch import BatchPrefetcher, to_device
        from .summary import ScalarLog
        opts = self.options
        dev = self.store.device
        W, R = self.world_size, self.rank      # data parallel: same shuffles on every rank, consecutive batches each
        order = rng = np.random.RandomState(seed)
        if W > 1:
            rng = np.random.RandomState(rank_seed(seed, R) + 1)       # this rank's crops and flips
        if dataA is None:
            dataA = sorted(glob('./{}/*.jpg'.format(self.dataset_dir + '/trainA')))
            dataB = sorted(glob('./{}/*.jpg'.format(self.dataset_dir + '/trainB')))
        if len(dataA) == 0 or len(dataB) == 0:
            raise RuntimeError("no training data under %s/trainA and %s/trainB" % (self.dataset_dir, self.dataset_dir))
        from_files = isinstance(dataA[0], str)
        counter = 1
        start_time = time.time()
        loaded = self.load(opts.checkpoint_dir)
        if R == 0:
            print(" [*] Load SUCCESS" if loaded else " [!] Load failed...")
        iters = 0
        out = None
        run = None
        scalars = ScalarLog(getattr(opts, 'log_dir', None), enabled=(R == 0 and bool(getattr(opts, 'write_scalars', False))))
        for epoch in range(opts.nEpochs):
            ia, ib = order.permutation(len(dataA)), order.permutation(len(dataB))
            batch_idxs = int(min(min(len(dataA), len(dataB)), getattr(opts, 'train_size', 1e8))) // (self.batch_size * W)

            def make(idx, ia=ia, ib=ib):
                start = shard_start(idx, R, W, self.batch_size)
                sl = slice(start, start + self.batch_size)
                if from_files:
                    return np.array([load_data((dataA[a], dataB[b]), flip=getattr(opts, 'flip', True),
                                               load_size=getattr(opts, 'load_size', 286), fine_size=self.image_size, rng=rng)
                                     for a, b in zip(ia[sl], ib[sl])]).astype(np.float32)
                return np.concatenate([np.asarray(dataA)[ia[sl]], np.asarray(dataB)[ib[sl]]], axis=3).astype(np.float32)

            feed = BatchPrefetcher(make, range(0, batch_idxs), depth=int(getattr(opts, 'prefetch_batches', 4)))
            for idx, host_batch in feed:
                real = to_device(host_batch, dev)
                if getattr(opts, 'use_graph', True):
                    if run is None:
                        run = self.iteration_runner(real)
                    out = run(real)
                else:
                    out = self.train_iteration(real)
                counter += 1
                if R == 0:
                    print("Epoch: [%2d] [%4d/%4d] time: %4.4f" % (epoch, idx, batch_idxs, time.time() - start_time))
                    if scalars.path:
                        scalars.add(counter, **self.totals(out))
                if R == 0 and sample_every and np.mod(counter, sample_every) == 1:
                    sample_dir = getattr(opts, 'sample_dir', './samples')
                    if not self.sample_model(sample_dir, epoch, idx):
                        # no testA / testB files (in-memory data): the translations of the current batch
                        fake_A, fake_B = self.forward_fakes(real)
                        if not os.path.exists(sample_dir):
                            os.makedirs(sample_dir)
                        save_images(fake_A.cpu().numpy(), [self.batch_size, 1], '{}/A_{:02d}_{:04d}.jpg'.format(sample_dir, epoch, idx))
                        save_images(fake_B.cpu().numpy(), [self.batch_size, 1], '{}/B_{:02d}_{:04d}.jpg'.format(sample_dir, epoch, idx))
                if R == 0 and np.mod(counter, 1000) == 2:
                    self.save(opts.checkpoint_dir, counter)
                iters += 1
                if max_iters is not None and iters >= max_iters:
                    feed.close()
                    scalars.close()
                    return out
        scalars.close()
        return out

    def sample_model(self, sample_dir, epoch, idx, rng=np.random):
        """cyclegan.py:216-236: translate batch_size shuffled testA / testB pairs and write the real and the
        translated images as <sample_dir>/{A,B}_<epoch>_<idx>[_real].jpg.  Returns False (nothing written)
        when the data set has no test images."""
        from glob import glob
        from .misc import load_data, save_images
        dataA = sorted(glob('./{}/*.jpg'.format(self.dataset_dir + '/testA')))
        dataB = sorted(glob('./{}/*.jpg'.format(self.dataset_dir + '/testB')))
        if len(dataA) < self.batch_size or len(dataB) < self.batch_size:
            return False
        rng.shuffle(dataA)
        rng.shuffle(dataB)
        batch_files = zip(dataA[:self.batch_size], dataB[:self.batch_size])
        sample_images = np.array([load_data(f, False, True, fine_size=self.image_size) for f in batch_files]).astype(np.float32)
        fake_A, fake_B = self.forward_fakes(torch.from_numpy(sample_images).to(self.store.device))
        if not os.path.exists(sample_dir):
            os.makedirs(sample_dir)
        save_images(sample_images[:, :, :, 0:3], [self.batch_size, 1], '{}/A_{:02d}_{:04d}_real.jpg'.format(sample_dir, epoch, idx))
        save_images(sample_images[:, :, :, 3:6], [self.batch_size, 1], '{}/B_{:02d}_{:04d}_real.jpg'.format(sample_dir, epoch, idx))
        save_images(fake_A.cpu().numpy(), [self.batch_size, 1], '{}/A_{:02d}_{:04d}.jpg'.format(sample_dir, epoch, idx))
        save_images(fake_B.cpu().numpy(), [self.batch_size, 1], '{}/B_{:02d}_{:04d}.jpg'.format(sample_dir, epoch, idx))
        return True

    def test_files(self, opts):
        """cyclegan.py:238-265: translate every testA (AtoB) or testB (BtoA) image with the checkpointed
        generator and write <test_dir>/A2B_<name> or B2A_<name>.  Returns the written paths."""
        from glob import glob
        from .misc import load_test_data, save_images
        if opts.which_direction == 'AtoB':
            sample_files = sorted(glob('./{}/*.jpg'.format(self.dataset_dir + '/testA')))
        elif opts.which_direction == 'BtoA':
            sample_files = sorted(glob('./{}/*.jpg'.format(self.dataset_dir + '/testB')))
        else:
            raise Exception('--which_direction must be AtoB or BtoA')
        print(" [*] Load SUCCESS" if self.load(opts.checkpoint_dir) else " [!] Load failed...")
        if not os.path.exists(opts.test_dir):
            os.makedirs(opts.test_dir)
        written = []
        for sample_file in sample_files:
            print('Processing image: ' + sample_file)
            sample_image = np.array([load_test_data(sample_file, self.image_size)]).astype(np.float32)
            fake = self.test(torch.from_numpy(sample_image).to(self.store.device), opts.which_direction)
            prefix = 'A2B_' if opts.which_direction == 'AtoB' else 'B2A_'
            written.append('{}/{}{}'.format(opts.test_dir, prefix, sample_file.split('/')[-1]))
            save_images(fake.cpu().numpy(), [1, 1], written[-1])
        return written

    def test(self, images, which_direction='AtoB'):
        """cyclegan.py:238-265 without the file I/O: translate a batch with one generator."""
        if which_direction not in ('AtoB', 'BtoA'):
            raise Exception('--which_direction must be AtoB or BtoA')
        name = "generatorA2B" if which_direction == 'AtoB' else "generatorB2A"
        with V.use_store(self.store), torch.no_grad():
            return self.generator(images, self.options, True, name=name)

    def save(self, checkpoint_dir, step):
        """cyclegan.py:192-201"""
        model_name = "CycleGAN.model"
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        if not os.path.exists(checkpoint_dir):
            os.makedirs(checkpoint_dir)
        return save_checkpoint(self.store, self.optims, os.path.join(checkpoint_dir, model_name), step)

    def load(self, checkpoint_dir):
        """cyclegan.py:203-214"""
        checkpoint_dir = os.path.join(checkpoint_dir, self.model_dir)
        print(" [*] Reading checkpoint from {}...".format(checkpoint_dir))
        path = latest_checkpoint(checkpoint_dir)
        if path:
            load_checkpoint(self.store, self.optims, path)
            return True
        return False
