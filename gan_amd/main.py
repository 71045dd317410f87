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
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    elif v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    else:
        raise argparse.ArgumentTypeError('Boolean value expected.')


def build_parser():
    parser = argparse.ArgumentParser(description='')
    parser.add_argument('--model', default='DCGAN', help='model to use [DCGAN, Text2Image, CycleGAN]')
    parser.add_argument('--nEpochs', type=int, default=25, help='# of epoch')
    parser.add_argument('--save_every_batch', type=int, default=50,
                        help='Save Model/Samples every x iterations over batches(does not overwrite previously saved models)')
    parser.add_argument('--print_every', type=int, default=1,
                        help='print the debug information every print_freq iterations')
    parser.add_argument('--batch_size', type=int, default=64, help='# images in batch')
    parser.add_argument('--train_size', type=int, default=1e8, help='# images used to train')
    parser.add_argument('--load_size', type=int, default=286, help='scale images to this size')
    parser.add_argument('--fine_size', type=int, default=256, help='then crop to this size')
    parser.add_argument('--ngf', type=int, default=32, help='# of gen filters in first conv layer')
    parser.add_argument('--ndf', type=int, default=64, help='# of discri filters in first conv layer')
    parser.add_argument('--input_nc', type=int, default=3, help='# of input image channels')
    parser.add_argument('--output_nc', type=int, default=3, help='# of output image channels')
    parser.add_argument('--niter', type=int, default=200, help='# of iter at starting learning rate')
    parser.add_argument('--lr', type=float, default=0.0002, help='initial learning rate for adam')
    parser.add_argument('--beta1', type=float, default=0.5, help='momentum term of adam')
    parser.add_argument('--flip', type=str2bool, default=True, help='if flip the images for data argumentation')
    parser.add_argument('--which_direction', default='AtoB', help='AtoB or BtoA')
    parser.add_argument('--phase', default='train', help='train, valid, test')
    parser.add_argument('--save_latest_freq', type=int, default=5000,
                        help='save the latest model every latest_freq sgd iterations (overwrites the previous latest model)')
    parser.add_argument('--continue_train', type=str2bool, default=False,
                        help='if continue training, load the latest model: 1: true, 0: false')
    parser.add_argument('--serial_batches', type=str2bool, default=False,
                        help='f 1, takes images in order to make batches, otherwise takes them randomly')
    parser.add_argument('--serial_batch_iter', type=str2bool, default=True, help='iter into serial image list')
    parser.add_argument('--checkpoint_dir', default='./checkpoint', help='models are saved here')
    parser.add_argument('--sample_dir', default='./samples', help='sample are saved here')
    parser.add_argument('--test_dir', default='./test', help='test sample are saved here')
    parser.add_argument('--L1_lambda', type=float, default=10.0, help='weight on L1 term in objective')
    parser.add_argument('--use_resnet', type=str2bool, default=True, help='generation network using reidule block')
    parser.add_argument('--use_lsgan', type=str2bool, default=True, help='gan loss defined in lsgan')
    parser.add_argument('--max_size', type=int, default=50,
                        help='max size of image pool, 0 means do not use image pool')

    parser.add_argument('--z_dim', type=int, default=100, help="Noise Dimension")
    parser.add_argument('--t_dim', type=int, default=256, help='Text feature dimension')
    parser.add_argument('--gf_dim', type=int, default=64, help='Number of conv in the first layer gen.')
    parser.add_argument('--df_dim', type=int, default=64, help='Number of conv in the first layer discr.')
    parser.add_argument('--gfc_dim', type=int, default=1024, help='Dimension of gen untis for fully connected layer [1024]')
    parser.add_argument('--dfc_dim', type=int, default=1024, help='Dimension of discrim untis for for fully connected layer [1024]')
    parser.add_argument('--caption_vector_length', type=int, default=None, help='Caption Vector Length')
    parser.add_argument('--resume_model', type=str, default=None, help='Pre-Trained Model Path, to resume from')
    parser.add_argument('--input_height', type=int, default=108, help="The size of image to use (will be center cropped)")
    parser.add_argument('--input_width', type=int, default=None,
                        help="The size of image to use (will be center cropped), None=input_height")
    parser.add_argument('--output_height', type=int, default=64, help="The size of output image to produce")
    parser.add_argument('--output_width', type=int, default=None, help="The size of output image to produce , None=output_height")
    parser.add_argument('--dataset', default='mnist', help="The name of dataset [celebA, mnist, lsun, horse2zebra, mscoco, flowers]")
    parser.add_argument('--data_dir', type=str, default="data", help='Data Directory')
    parser.add_argument('--log_dir', type=str, default="logs", help='tf.Summary Directory')
    parser.add_argument('--input_fname_pattern', default='*.jpg', help="Glob pattern of filename of input images [*]")
    parser.add_argument('--crop', type=str2bool, default=True, help="True for training, False for testing [False]")
    parser.add_argument('--visualize', type=str2bool, default=False, help="True for visualizing, False for nothing [False]")
    parser.add_argument('--note', type=str, default='', help="")

    parser.add_argument('--textEncoder', type=str, default='skipthought', help="skipthought / DSSJE")

    # flags of models outside the hot path: accepted so that existing command lines keep parsing
    for name, typ, default in [('gru_num_units', int, 512), ('gru_num_layers', int, 2), ('sequence_length', int, 30),
                               ('dropout_keep_prob', float, 0.5), ('decay_every_nEpochs', int, 8),
                               ('decay_factor', float, 0.5), ('values_per_input_shard', int, 2300),
                               ('input_queue_capacity_factor', int, 2), ('num_input_reader_threads', int, 1),
                               ('num_preprocess_threads', int, 4),
                               ('inception_checkpoint_file', str, 'checkpoint/inception_v3.ckpt'),
                               ('vocab_file', str, 'vocab.txt'), ('num_lstm_units', int, 1024),
                               ('embedding_size', int, 1024), ('image_size', int, 64), ('vocab_size', int, 2500)]:
        parser.add_argument('--' + name, type=typ, default=default, help="")
    for name in ('use_TextNLL', 'use_CaptionNLL', 'use_GAN', 'use_wrongImage'):
        parser.add_argument('--' + name, type=str2bool, default=True, help="")

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
