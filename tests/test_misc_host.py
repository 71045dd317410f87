"""Host-side image helpers (gan_amd/misc.py) against the behaviour of the reference's models/misc.py."""
import numpy as np
import pytest
from PIL import Image


def test_merge_lays_images_out_row_major():
    from gan_amd.misc import merge
    imgs = np.stack([np.full((2, 3, 3), i, dtype=np.float64) for i in range(5)])
    grid = merge(imgs, [2, 3])
    assert grid.shape == (4, 9, 3)
    assert grid[0, 0, 0] == 0 and grid[0, 3, 0] == 1 and grid[0, 6, 0] == 2     # first row: images 0, 1, 2
    assert grid[2, 0, 0] == 3 and grid[2, 3, 0] == 4 and grid[2, 6, 0] == 0     # second row: 3, 4, empty cell
    gray = merge(imgs[:, :, :, :1], [3, 2])
    assert gray.shape == (6, 6) and gray[2, 3] == 3
    with pytest.raises(ValueError):
        merge(np.zeros((2, 4, 4, 2)), [1, 2])
    with pytest.raises(ValueError):
        merge(imgs, [2, 2])


def test_bytescale_and_imresize_follow_scipy_misc():
    from gan_amd.misc import bytescale, imresize
    u8 = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert bytescale(u8) is u8 or np.array_equal(bytescale(u8), u8)
    f = np.array([[10.0, 20.0], [30.0, 50.0]])
    np.testing.assert_array_equal(bytescale(f), [[0, 64], [128, 255]])          # stretched to the full range
    out = imresize(np.full((8, 8, 3), 77, dtype=np.uint8), [4, 6])
    assert out.shape == (4, 6, 3) and out.dtype == np.uint8 and (out == 77).all()


def test_transform_center_crop_and_inverse(tmp_path):
    from gan_amd.misc import center_crop, get_image, inverse_transform, transform
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (40, 30, 3)).astype(np.uint8)
    c = center_crop(img, 20, 10, 20, 10)                  # crop only: resize to the same size is the identity
    np.testing.assert_array_equal(c, img[10:30, 10:20])
    t = transform(img, 20, 10, 20, 10, crop=True)
    assert t.min() >= -1.0 and t.max() <= 1.0
    np.testing.assert_allclose(inverse_transform(t) * 255.0, img[10:30, 10:20], atol=1e-9)
    path = str(tmp_path / "x.png")
    Image.fromarray(img).save(path)
    g = get_image(path, 20, 10, resize_height=8, resize_width=8, crop=True)
    assert g.shape == (8, 8, 3) and -1.0 <= g.min() and g.max() <= 1.0
    gg = get_image(path, 20, 10, resize_height=8, resize_width=8, crop=False, grayscale=True)
    assert gg.shape == (8, 8)
    # the uint8 form that travels to the GPU: scaling it there (float64, then the float32 cast) is get_image
    import torch
    from gan_amd.dcgan import DCGAN
    from gan_amd.misc import get_image_u8
    for kw in (dict(crop=True), dict(crop=False, grayscale=True)):
        u8 = get_image_u8(path, 20, 10, resize_height=8, resize_width=8, **kw)
        want = get_image(path, 20, 10, resize_height=8, resize_width=8, **kw).astype(np.float32)
        assert u8.dtype == np.uint8
        np.testing.assert_array_equal(DCGAN._scale_u8(u8), want)
        np.testing.assert_array_equal(DCGAN._scale_u8(torch.from_numpy(u8.copy())).numpy(), want)


def test_save_images_writes_the_grid(tmp_path):
    from gan_amd.misc import save_images
    imgs = np.stack([np.full((4, 4, 3), v) for v in (-1.0, 0.0, 1.0, 0.5)])
    path = save_images(imgs, [2, 2], str(tmp_path / "grid.png"))
    back = np.asarray(Image.open(path))
    assert back.shape == (8, 8, 3)
    assert back[0, 0, 0] == 0 and back[0, 4, 0] == 128 and back[4, 0, 0] == 255 and back[4, 4, 0] == 191


def test_cyclegan_pair_loading(tmp_path):
    from gan_amd.misc import load_data, load_test_data, preprocess_A_and_B
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (50, 60, 3)).astype(np.uint8)
    b = rng.integers(0, 256, (50, 60)).astype(np.uint8)          # a grey image becomes 3 equal channels
    pa, pb = str(tmp_path / "a.png"), str(tmp_path / "b.png")
    Image.fromarray(a).save(pa)
    Image.fromarray(b).save(pb)
    ab = load_data((pa, pb), load_size=36, fine_size=32, rng=np.random.RandomState(3))
    assert ab.shape == (32, 32, 6) and ab.min() >= -1.0 and ab.max() <= 1.0
    np.testing.assert_array_equal(ab[..., 3], ab[..., 4])
    again = load_data((pa, pb), load_size=36, fine_size=32, rng=np.random.RandomState(3))
    np.testing.assert_array_equal(ab, again)                      # same seed: same crop and flip
    ta, tb = preprocess_A_and_B(a, a, load_size=36, fine_size=32, is_test=True)
    assert ta.shape == (32, 32, 3) and np.array_equal(ta, tb)
    assert load_test_data(pa, fine_size=16).shape == (16, 16, 3)


def test_text2image_flowers_file_pipeline(tmp_path, monkeypatch):
    """load_training_data / get_training_batch of the Text2Image model over a tiny flowers tree
    (DS-SJE captions as .npy, jpg images): list split and shuffle, caption choice, image range.
    Host logic only: the model object is created without building its networks."""
    import random
    from PIL import Image
    from gan_amd.text2image import Text2Image
    rng = np.random.default_rng(0)
    (tmp_path / "flowers" / "jpg").mkdir(parents=True)
    for cls in ("class_00001", "class_00002"):
        (tmp_path / "flowers_caption_icml" / cls).mkdir(parents=True)
    names = ["image_%05d" % i for i in range(8)]
    caps = {}
    for i, n in enumerate(names):
        Image.fromarray(rng.integers(0, 256, (40 + i, 50, 3)).astype(np.uint8)).save(str(tmp_path / "flowers" / "jpg" / (n + ".jpg")))
        caps[n] = rng.standard_normal((10, 1024)).astype(np.float32)
        np.save(str(tmp_path / "flowers_caption_icml" / ("class_0000%d" % (1 + i % 2)) / (n + ".npy")), caps[n])
    m = Text2Image.__new__(Text2Image)
    m.textEncoder, m.data_dir, m.dataset, m.batch_size, m.image_size = 'DSSJE', str(tmp_path), 'flowers', 4, 16
    m.caption_vector_length, m.nCapsPerImage, m.z_dim = 12, 10, 5
    data = m.load_training_data(seed=3)
    assert data['data_length'] == 6 and sorted(data['image_list']) == [n + ".jpg" for n in names[:6]]   # first 75 %
    assert set(data['captions']) == {n + ".jpg" for n in names}
    real, wrong, cap, z, files = m.get_training_batch(1, data, rng=random.Random(5))
    assert real.shape == wrong.shape == (4, 16, 16, 3) and cap.shape == (4, 12) and z.shape == (4, 5)
    assert 0.0 <= real.min() and real.max() <= 1.0 and np.abs(z).max() <= 1.0
    # batch 1 of size 4 over 6 images wraps around the list (text2image.py:399-400)
    want = [data['image_list'][i % 6] for i in range(4, 8)]
    assert [f.split('/')[-1] for f in files] == want
    for row, f in zip(cap, files):
        stem = f.split('/')[-1][:-4]
        assert any(np.array_equal(row, caps[stem][k][:12]) for k in range(10))
    with pytest.raises(NotImplementedError):
        m.load_training_data(dataset='mscoco')
