"""Command line of the reference (main.py:104-191) over the MI355X hot path.

Every flag of the reference's parser is accepted with the same name, type and default, so the
scripts (scripts/DCGAN_mnist.sh, scripts/Text2Image_flowers.sh, scripts/CycleGAN.sh) run unchanged
with `python -m gan_amd.main` in place of `python main.py`.  Flags that only feed models outside the
hot path (TextDecoder, CrossModalRepr*) are parsed and ignored; asking for those models raises.

Extension: `--synthetic_iters N` trains N iterations on synthetic data of the model's input shape
(uniform images, normal captions), for machines without the datasets.
"""
from __future__ import print_function

import argparse
import os
from os.path import join

import numpy as np


def str2bool(v):
    """main.py:18-24: yes/true/t/y/1 and no/false/f/n/0, any case."""
    word = v.lower()
    if word in ('yes', 'true', 't', 'y', '1'):
        return True
    if word in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


# The reference's flag surface (main.py:104-188) as a table: name, type, default.  `bool` stands for str2bool.
# Flags in the second table only feed models outside the hot path; they are accepted so that existing command
# lines keep parsing.
REFERENCE_FLAGS = [
    ('model', str, 'DCGAN'), ('nEpochs', int, 25), ('save_every_batch', int, 50), ('print_every', int, 1),
    ('batch_size', int, 64), ('train_size', int, 1e8), ('load_size', int, 286), ('fine_size', int, 256),
    ('ngf', int, 32), ('ndf', int, 64), ('input_nc', int, 3), ('output_nc', int, 3), ('niter', int, 200),
    ('lr', float, 0.0002), ('beta1', float, 0.5), ('flip', bool, True), ('which_direction', str, 'AtoB'),
    ('phase', str, 'train'), ('save_latest_freq', int, 5000), ('continue_train', bool, False),
    ('serial_batches', bool, False), ('serial_batch_iter', bool, True), ('checkpoint_dir', str, './checkpoint'),
    ('sample_dir', str, './samples'), ('test_dir', str, './test'), ('L1_lambda', float, 10.0),
    ('use_resnet', bool, True), ('use_lsgan', bool, True), ('max_size', int, 50),
    ('z_dim', int, 100), ('t_dim', int, 256), ('gf_dim', int, 64), ('df_dim', int, 64), ('gfc_dim', int, 1024),
    ('dfc_dim', int, 1024), ('caption_vector_length', int, None), ('resume_model', str, None),
    ('input_height', int, 108), ('input_width', int, None), ('output_height', int, 64), ('output_width', int, None),
    ('dataset', str, 'mnist'), ('data_dir', str, 'data'), ('log_dir', str, 'logs'),
    ('input_fname_pattern', str, '*.jpg'), ('crop', bool, True), ('visualize', bool, False), ('note', str, ''),
    ('textEncoder', str, 'skipthought'),
]
OUT_OF_PATH_FLAGS = [
    ('gru_num_units', int, 512), ('gru_num_layers', int, 2), ('sequence_length', int, 30),
    ('dropout_keep_prob', float, 0.5), ('decay_every_nEpochs', int, 8), ('decay_factor', float, 0.5),
    ('values_per_input_shard', int, 2300), ('input_queue_capacity_factor', int, 2),
    ('num_input_reader_threads', int, 1), ('num_preprocess_threads', int, 4),
    ('inception_checkpoint_file', str, 'checkpoint/inception_v3.ckpt'), ('vocab_file', str, 'vocab.txt'),
    ('num_lstm_units', int, 1024), ('embedding_size', int, 1024), ('image_size', int, 64), ('vocab_size', int, 2500),
    ('use_TextNLL', bool, True), ('use_CaptionNLL', bool, True), ('use_GAN', bool, True), ('use_wrongImage', bool, True),
]


def build_parser():
    parser = argparse.ArgumentParser(description='GAN training on MI355X with the flags of the reference\'s main.py')
    for name, typ, default in REFERENCE_FLAGS + OUT_OF_PATH_FLAGS:
        parser.add_argument('--' + name, type=str2bool if typ is bool else typ, default=default,
                            help='as in the reference (default %r)' % (default,))

    # extensions
    parser.add_argument('--synthetic_iters', type=int, default=0,
                        help='train this many iterations on synthetic data of the model\'s input shape')
    parser.add_argument('--seed', type=int, default=0, help='seed of the weight initialisers and synthetic data')
    parser.add_argument('--use_graph', type=str2bool, default=True,
                        help='replay the training iteration as HIP graphs; false issues every launch eagerly')
    parser.add_argument('--write_scalars', type=str2bool, default=True,
                        help='append the losses of every logged step to <log_dir>/<model_dir>/scalars.jsonl '
                             '(stands in for the tf.summary scalars)')
    parser.add_argument('--dist_backend', default='nccl',
                        help='collective backend when started by torch.distributed.run with several ranks: nccl (RCCL, '
                             'one GPU per rank) or gloo (rehearsal: every rank on the one visible GPU)')
    parser.add_argument('--dataset_cache_gb', type=float, default=32.0,
                        help='DCGAN image files: keep the cropped / resized uint8 images in GPU memory after the first '
                             'epoch when they fit in this many GiB (0: decode the files every epoch)')
    parser.add_argument('--contraction_plans', default=None,
                        help='JSON file of pinned contraction plans (gan_amd.autotune.save / tools/insitu_tune.py): every '
                             'run then launches the same kernels with the same split-K summation order; layer shapes the '
                             'file does not hold keep the library\'s heuristic plan (or are tuned on first use when '
                             'GAN_HIP_AUTOTUNE=1)')
    parser.add_argument('--prefetch_batches', type=int, default=4,
                        help='batches the host input pipeline prepares ahead of the GPU (num_preprocess_threads decode the files)')
    return parser


def synthetic_train(model, opts):
    """N iterations on seeded synthetic inputs of the shapes and ranges the real loaders deliver
    (data parallel: every rank draws its own)."""
    import torch
    rank = getattr(model, 'rank', 0)
    rng = np.random.default_rng(opts.seed + 7919 * rank)
    dev = model.store.device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    b = opts.batch_size
    out = None
    for it in range(opts.synthetic_iters):
        if opts.model == 'DCGAN':
            lo = 0.0 if opts.dataset == 'mnist' else -1.0
            shape = [b] + list(model.image_dims)
            y = None
            if model.y_dim:
                y = t(np.eye(model.y_dim)[rng.integers(0, model.y_dim, b)])
            out = model.train_iteration(t(rng.uniform(lo, 1, shape)), t(rng.uniform(-1, 1, (b, model.z_dim))), y)
        elif opts.model == 'Text2Image':
            s = model.image_size
            out = model.train_iteration(t(rng.uniform(0, 1, (b, s, s, 3))), t(rng.uniform(0, 1, (b, s, s, 3))),
                                        t(rng.standard_normal((b, model.caption_vector_length))),
                                        t(rng.uniform(-1, 1, (b, model.z_dim))))
        else:
            s = opts.fine_size
            out = model.train_iteration(t(rng.uniform(-1, 1, (b, s, s, opts.input_nc + opts.output_nc))))
        if rank == 0 and it % max(1, opts.print_every) == 0:
            print("iter %4d  " % it + "  ".join("%s %.5f" % (k, float(v)) for k, v in sorted(out.items())
                                                 if getattr(v, 'numel', lambda: 0)() == 1))
    return out


def main(opts):
    if opts.input_width is None:
        opts.input_width = opts.input_height
    if opts.output_width is None:
        opts.output_width = opts.output_height
    if len(opts.note) > 0:
        opts.note = '_' + opts.note

    # path for checkpoint
    os.makedirs(opts.checkpoint_dir, exist_ok=True)         # (several ranks may get here together)

    # path for sample images
    opts.sample_dir = join(opts.sample_dir, opts.model, opts.dataset)
    print('sample_dir = {}'.format(opts.sample_dir))
    os.makedirs(opts.sample_dir, exist_ok=True)

    if getattr(opts, 'contraction_plans', None):
        from . import autotune
        autotune.load(opts.contraction_plans)
    sess = None     # the reference's tf.Session has no counterpart: kernels launch on the current HIP stream
    # one process per GPU under torch.distributed.run: every rank a replica, gradients summed per update
    from .dp import init_from_env
    world, rank = init_from_env(getattr(opts, 'dist_backend', 'nccl'))
    dp = dict(world_size=world) if world > 1 else {}
    if opts.model == 'DCGAN':
        from .dcgan import DCGAN
        opts.y_dim = None
        if opts.dataset == 'mnist':
            opts.y_dim = 10
            if opts.synthetic_iters:        # what load_mnist would have set (dcgan.py:72-75)
                opts.c_dim = 1
        model = DCGAN(sess, opts, load_data=not opts.synthetic_iters, **dp)
    elif opts.model == 'Text2Image':
        from .text2image import Text2Image
        model = Text2Image(sess, opts, **dp)
    elif opts.model == 'CycleGAN':
        from .cyclegan import CycleGAN
        model = CycleGAN(sess, opts, **dp)
    elif opts.model in ('TextDecoder', 'CrossModalRepr', 'CrossModalRepr_wWrong'):
        raise Exception("model %s is outside the accelerated GAN hot path" % opts.model)
    else:
        raise Exception("undefined model")

    if rank == 0:
        show_all_variables(model.store)

    # follow checkpoint_dir's convention
    model_dir = model.model_dir
    opts.log_dir = os.path.join(opts.log_dir, model_dir + opts.note)

    # run train or validation
    if opts.phase == 'train':
        if opts.synthetic_iters:
            return synthetic_train(model, opts)
        if opts.model in ('DCGAN', 'CycleGAN'):
            model.train()           # MNIST idx files / image files under ./data, as in the reference
        else:
            return model.train_from_files()     # <data_dir>/flowers/jpg + caption files, as in the reference
    elif opts.phase == 'valid':
        raise Exception("phase 'valid' is not used by the reference's GAN models")
    else:
        if opts.model == 'CycleGAN':
            return model.test_files(opts)       # cyclegan.py:238-265 (the reference's main only loads)
        loaded = model.load(opts.checkpoint_dir)
        if not (loaded[0] if isinstance(loaded, tuple) else loaded):
            raise Exception("[!] Train a model first, then run test mode")

    # Below is codes for visualization (main.py:98-101)
    if opts.model == 'DCGAN' and rank == 0:
        from .misc import visualize
        OPTION = 1
        visualize(model, opts, OPTION)
    return None


def show_all_variables(store):
    """misc.show_all_variables (misc.py:18-20): one line per trainable variable and the total."""
    total = 0
    for v in store.trainable_variables():
        n = int(np.prod(v.shape)) if v.shape else 1
        total += n
        print("%-60s %-20s %d" % (v.name, v.shape, n))
    print("Total size of variables: %d" % total)
    print("Total bytes of variables: %d" % (4 * total))
    return total


if __name__ == '__main__':
    try:
        main(build_parser().parse_args())
    finally:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
