"""Host-side image helpers around the hot path, after the reference's models/misc.py: reading and
normalising training images (CelebA-style centre crop, CycleGAN resize / random crop / flip), and
writing sample grids.  Nothing here touches the GPU; arrays are numpy, NHWC, float.

scipy.misc.imread / imresize / imsave, which the reference calls, no longer exist; they are restated
on Pillow.  `imresize` keeps the old function's behaviour of stretching a non-uint8 input to the
full 0..255 range before resizing (scipy's `bytescale`), because the reference feeds it the float
arrays its `imread` returns and so trains on contrast-stretched images (misc.py:38-42, 72-90).
"""
from __future__ import division

import numpy as np
from PIL import Image


# ------------------------------------------------------------------------------------ file I/O
def imread(path, grayscale=False):
    """misc.py:38-42: float array, HxWx3 RGB or HxW luminance (ITU-R 601-2, as `flatten=True`)."""
    with Image.open(path) as im:
        if grayscale:
            return np.asarray(im.convert('F'), dtype=np.float64)
        return np.asarray(im.convert('RGB'), dtype=np.float64)


def bytescale(data):
    """scipy.misc.bytescale with default arguments: uint8 passes through, anything else is mapped
    affinely from [min, max] onto 0..255."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    cmin, cmax = float(data.min()), float(data.max())
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1.0
    scaled = (data - cmin) * (255.0 / cscale)
    return (scaled.clip(0, 255) + 0.5).astype(np.uint8)


def imresize(arr, size):
    """scipy.misc.imresize(arr, [height, width]) with its default bilinear filter: uint8 result."""
    im = Image.fromarray(bytescale(arr))
    im = im.resize((int(size[1]), int(size[0])), resample=Image.BILINEAR)
    return np.asarray(im)


def imsave(images, size, path):
    """misc.py:68-70: tile `images` (values in [0, 1]) into a size[0] x size[1] grid and write it."""
    image = np.squeeze(merge(images, size))
    Image.fromarray(bytescale(image)).save(path)
    return path


def save_images(images, size, image_path):
    """misc.py:35-36: generator output in [-1, 1] -> image grid file."""
    return imsave(inverse_transform(images), size, image_path)


def load_image_array(image_file, image_size, rng=np.random):
    """utils/image_processing.py:35-52 (the Text2Image loader): RGB (grey images repeated over three
    channels), resized to image_size x image_size as floats in [0, 1] -- skimage.transform.resize's
    output range -- and mirrored with probability 0.5.  Bilinear, like skimage's default order 1."""
    with Image.open(image_file) as im:
        img = np.asarray(im.convert('RGB'), dtype=np.float32) / 255.0
    planes = [np.asarray(Image.fromarray(img[:, :, c], mode='F').resize((image_size, image_size), resample=Image.BILINEAR))
              for c in range(3)]
    img_resized = np.stack(planes, axis=2)
    if rng.random() > 0.5:
        img_resized = np.fliplr(img_resized)
    return np.clip(img_resized, 0.0, 1.0).astype('float32')


# ------------------------------------------------------------------------------------ DCGAN inputs
def center_crop(x, crop_h, crop_w=None, resize_h=64, resize_w=64):
    """misc.py:72-80"""
    if crop_w is None:
        crop_w = crop_h
    h, w = x.shape[:2]
    j = int(round((h - crop_h) / 2.))
    i = int(round((w - crop_w) / 2.))
    return imresize(x[j:j + crop_h, i:i + crop_w], [resize_h, resize_w])


def transform_u8(image, input_height, input_width, resize_height=64, resize_width=64, crop=True):
    """The uint8 image that misc.py:82-90 then scales: centre crop (or none) and resize."""
    if crop:
        return center_crop(image, input_height, input_width, resize_height, resize_width)
    return imresize(image, [resize_height, resize_width])


def transform(image, input_height, input_width, resize_height=64, resize_width=64, crop=True):
    """misc.py:82-90: to [-1, 1]"""
    return np.array(transform_u8(image, input_height, input_width, resize_height, resize_width, crop)) / 127.5 - 1.


def get_image(image_path, input_height, input_width, resize_height=64, resize_width=64, crop=True, grayscale=False):
    """misc.py:28-33"""
    image = imread(image_path, grayscale)
    return transform(image, input_height, input_width, resize_height, resize_width, crop)


def get_image_u8(image_path, input_height, input_width, resize_height=64, resize_width=64, crop=True, grayscale=False):
    """get_image before its last step: get_image(...) == get_image_u8(...) / 127.5 - 1. (in float64).  A quarter
    of the bytes to keep or move; the scaling can then run where the batch is consumed."""
    return transform_u8(imread(image_path, grayscale), input_height, input_width, resize_height, resize_width, crop)


def inverse_transform(images):
    """misc.py:92-93"""
    return (images + 1.) / 2.


def merge(images, size):
    """misc.py:47-66: images [N,h,w,c] laid out row-major on a grid of size[0] rows x size[1]
    columns (missing cells stay black); one channel gives a 2-D result."""
    images = np.asarray(images)
    if images.ndim != 4 or images.shape[3] not in (1, 3, 4):
        raise ValueError('in merge(images,size) images parameter must have dimensions: HxW or HxWx3 or HxWx4')
    n, h, w, c = images.shape
    rows, cols = int(size[0]), int(size[1])
    if n > rows * cols:
        raise ValueError('merge: %d images do not fit a %d x %d grid' % (n, rows, cols))
    cells = np.zeros((rows * cols, h, w, c))
    cells[:n] = images
    grid = cells.reshape(rows, cols, h, w, c).transpose(0, 2, 1, 3, 4).reshape(rows * h, cols * w, c)
    return grid[:, :, 0] if c == 1 else grid


# ------------------------------------------------------------------------------------ CycleGAN inputs
def load_test_data(image_path, fine_size=256):
    """misc.py:269-273"""
    img = imresize(imread(image_path), [fine_size, fine_size])
    return img / 127.5 - 1


def gray2rgb_ifneeded(img):
    """misc.py:275-280"""
    if img.ndim == 2:
        return np.repeat(img[:, :, None], 3, 2)
    return img


def preprocess_A_and_B(img_A, img_B, load_size=286, fine_size=256, flip=True, is_test=False, rng=np.random):
    """misc.py:304-321: resize to load_size, one shared random crop of fine_size, one shared flip."""
    if is_test:
        img_A = imresize(img_A, [fine_size, fine_size])
        img_B = imresize(img_B, [fine_size, fine_size])
    else:
        img_A = imresize(img_A, [load_size, load_size])
        img_B = imresize(img_B, [load_size, load_size])
        h1 = int(np.ceil(rng.uniform(1e-2, load_size - fine_size)))
        w1 = int(np.ceil(rng.uniform(1e-2, load_size - fine_size)))
        img_A = img_A[h1:h1 + fine_size, w1:w1 + fine_size]
        img_B = img_B[h1:h1 + fine_size, w1:w1 + fine_size]
        if flip and rng.random() > 0.5:
            img_A = np.fliplr(img_A)
            img_B = np.fliplr(img_B)
    return img_A, img_B


def load_data(image_path, flip=True, is_test=False, load_size=286, fine_size=256, rng=np.random):
    """misc.py:282-297: (path_A, path_B) -> [fine_size, fine_size, 6] in [-1, 1], A | B on the channel axis."""
    img_A = gray2rgb_ifneeded(imread(image_path[0]))
    img_B = gray2rgb_ifneeded(imread(image_path[1]))
    img_A, img_B = preprocess_A_and_B(img_A, img_B, load_size, fine_size, flip=flip, is_test=is_test, rng=rng)
    img_A = img_A / 127.5 - 1.
    img_B = img_B / 127.5 - 1.
    return np.concatenate((img_A, img_B), axis=2)


# ------------------------------------------------------------------------------------ DCGAN visualisation
def visualize(model, config, option, out_dir='./samples', rng=np.random):
    """misc.py:175-198, options 0 and 1: sampler grids written under ./samples.  Option 0: one grid for
    z ~ U(-0.5, 0.5).  Option 1: for each of the first 100 z coordinates one grid in which that coordinate
    runs over arange(0, 1, 1/batch_size) and the others are zero (MNIST: with random class labels).
    Options 2-4 build GIF animations with moviepy, which is outside this path."""
    import math
    from time import gmtime, strftime
    import torch
    if option not in (0, 1):
        raise NotImplementedError("visualize: options 2-4 write GIFs through moviepy")
    import os
    if not os.path.exists(out_dir):
        os.makedirs(out_dir)
    dev = model.store.device
    image_frame_dim = int(math.ceil(config.batch_size ** .5))
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    written = []
    if option == 0:
        z_sample = rng.uniform(-0.5, 0.5, size=(config.batch_size, model.z_dim))
        samples = model.sampler(to_dev(z_sample)).cpu().numpy()
        written.append(save_images(samples, [image_frame_dim, image_frame_dim],
                                   os.path.join(out_dir, 'test_%s.png' % strftime("%Y%m%d%H%M%S", gmtime()))))
        return written
    values = np.arange(0, 1, 1. / config.batch_size)
    for idx in range(min(100, model.z_dim)):
        print(" [*] %d" % idx)
        z_sample = np.zeros([config.batch_size, model.z_dim])
        for kdx, z in enumerate(z_sample):
            z[idx] = values[kdx]
        if config.dataset == "mnist":
            y = rng.choice(10, config.batch_size)
            y_one_hot = np.zeros((config.batch_size, 10))
            y_one_hot[np.arange(config.batch_size), y] = 1
            samples = model.sampler(to_dev(z_sample), to_dev(y_one_hot))
        else:
            samples = model.sampler(to_dev(z_sample))
        written.append(save_images(samples.cpu().numpy(), [image_frame_dim, image_frame_dim],
                                   os.path.join(out_dir, 'test_arange_%s.png' % (idx))))
    return written
