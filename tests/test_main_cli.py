"""The command line keeps the reference's flag surface (main.py:104-188) and dispatches the three
GAN models (main.py:66-82)."""
import numpy as np
import pytest


def parse(argv):
    from gan_amd.main import build_parser
    return build_parser().parse_args(argv)


def test_defaults_match_the_reference_parser():
    o = parse([])
    want = dict(model='DCGAN', nEpochs=25, save_every_batch=50, print_every=1, batch_size=64, train_size=1e8,
                load_size=286, fine_size=256, ngf=32, ndf=64, input_nc=3, output_nc=3, niter=200, lr=0.0002,
                beta1=0.5, flip=True, which_direction='AtoB', phase='train', save_latest_freq=5000,
                continue_train=False, serial_batches=False, serial_batch_iter=True, checkpoint_dir='./checkpoint',
                sample_dir='./samples', test_dir='./test', L1_lambda=10.0, use_resnet=True, use_lsgan=True,
                max_size=50, z_dim=100, t_dim=256, gf_dim=64, df_dim=64, gfc_dim=1024, dfc_dim=1024,
                caption_vector_length=None, resume_model=None, input_height=108, input_width=None,
                output_height=64, output_width=None, dataset='mnist', data_dir='data', log_dir='logs',
                input_fname_pattern='*.jpg', crop=True, visualize=False, note='', textEncoder='skipthought')
    for k, v in want.items():
        assert getattr(o, k) == v, k


def test_reference_script_command_lines_parse():
    # scripts/DCGAN_mnist.sh, scripts/Text2Image_flowers.sh, scripts/CycleGAN.sh
    o = parse("--dataset mnist --input_height 28 --output_height 28 --phase train".split())
    assert (o.dataset, o.input_height, o.output_height) == ('mnist', 28, 28)
    o = parse("--dataset flowers --input_height 64 --model Text2Image --nEpochs 200 --textEncoder DSSJE".split())
    assert (o.model, o.textEncoder, o.nEpochs) == ('Text2Image', 'DSSJE', 200)
    o = parse("--model CycleGAN --batch_size 1 --nEpochs 600 --dataset horse2zebra".split())
    assert (o.model, o.batch_size, o.dataset) == ('CycleGAN', 1, 'horse2zebra')
    assert parse(["--use_lsgan", "false"]).use_lsgan is False and parse(["--flip", "0"]).flip is False
    with pytest.raises(SystemExit):
        parse(["--use_lsgan", "maybe"])


@pytest.mark.gpu
@pytest.mark.parametrize("argv", [
    "--model DCGAN --dataset mnist --input_height 28 --output_height 28 --batch_size 8 --gf_dim 8 --df_dim 8 "
    "--gfc_dim 32 --dfc_dim 32 --z_dim 16",
    "--model DCGAN --dataset celebA --output_height 16 --batch_size 8 --gf_dim 8 --df_dim 8 --z_dim 16",
    "--model Text2Image --dataset flowers --input_height 64 --textEncoder DSSJE --batch_size 4 --gf_dim 4 --df_dim 4 "
    "--t_dim 8 --caption_vector_length 12 --z_dim 10",
    "--model CycleGAN --dataset horse2zebra --batch_size 1 --fine_size 32 --gf_dim 4 --df_dim 4 --max_size 2",
])
def test_models_train_from_the_command_line_on_synthetic_data(argv, tmp_path):
    from gan_amd.main import main
    o = parse(argv.split() + ["--synthetic_iters", "3", "--checkpoint_dir", str(tmp_path / "ckpt"),
                              "--sample_dir", str(tmp_path / "samples")])
    out = main(o)
    vals = [float(v) for v in out.values() if getattr(v, "numel", lambda: 0)() == 1]
    assert len(vals) >= 4 and np.isfinite(vals).all()


@pytest.mark.gpu
def test_pinned_contraction_plans_reproduce_a_run_bit_for_bit(tmp_path):
    """--contraction_plans: a run that pins the plans another run ended up with (gan_amd.autotune.save) launches the same
    kernels with the same split-K summation order, so its losses are the first run's bit for bit; the plans come back
    from the library as the file states them."""
    import ctypes
    import json
    from gan_amd import _lib as L, autotune
    from gan_amd.main import main
    argv = ("--model DCGAN --dataset celebA --output_height 32 --batch_size 16 --gf_dim 16 --df_dim 16 --z_dim 16 "
            "--synthetic_iters 2 --use_graph false").split()
    dirs = lambda tag: ["--checkpoint_dir", str(tmp_path / ("c" + tag)), "--sample_dir", str(tmp_path / ("s" + tag))]
    plans = str(tmp_path / "plans.json")
    autotune.reset()
    autotune.enable(True)
    try:
        first = main(parse(argv + dirs("a")))
        autotune.save(plans)
    finally:
        autotune.enable(False)
        autotune.reset()
    rows = json.load(open(plans))
    assert rows, "the tuner saw no layer shape"
    second = main(parse(argv + dirs("b") + ["--contraction_plans", plans]))
    try:
        for k in first:
            if getattr(first[k], "numel", lambda: 0)() == 1:
                assert float(first[k]) == float(second[k]), k
        for row in rows:
            g = L.ConvGeom(*row["geom"])
            for mode, (cfg, splits, _, _) in row["plans"].items():
                c, s, thin = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                L.check(L.lib.gan_conv_plan_get(int(mode), ctypes.byref(g), ctypes.byref(c), ctypes.byref(s), ctypes.byref(thin)))
                # (the planner may round a split count down to what the k range divides into)
                assert thin.value or (c.value == cfg and 0 < s.value <= splits), (row["geom"], mode)
    finally:
        autotune.reset()


@pytest.mark.gpu
def test_undefined_and_out_of_scope_models_raise(tmp_path):
    from gan_amd.main import main
    common = ["--checkpoint_dir", str(tmp_path / "c"), "--sample_dir", str(tmp_path / "s")]
    with pytest.raises(Exception, match="undefined model"):
        main(parse(["--model", "nope"] + common))
    with pytest.raises(Exception, match="outside the accelerated"):
        main(parse(["--model", "TextDecoder"] + common))


@pytest.mark.gpu
def test_training_from_image_files_and_sample_grids(tmp_path):
    """The reference's file pipelines end to end on tiny images: DCGAN (centre crop + resize,
    dcgan.py:200-221, sample grid :287-317) and CycleGAN (resize / random crop / flip of an A|B pair,
    misc.py:282-321, translations written like sample_model)."""
    from PIL import Image
    from gan_amd.cyclegan import CycleGAN
    from gan_amd.dcgan import DCGAN
    rng = np.random.default_rng(0)
    files = []
    for i in range(20):
        p = str(tmp_path / ("img%02d.jpg" % i))
        Image.fromarray(rng.integers(0, 256, (30, 26, 3)).astype(np.uint8)).save(p)
        files.append(p)
    o = parse(("--model DCGAN --dataset celebA --input_height 24 --output_height 16 --batch_size 8 --gf_dim 8 "
               "--df_dim 8 --z_dim 16 --nEpochs 1").split() + ["--checkpoint_dir", str(tmp_path / "ck"),
                                                              "--sample_dir", str(tmp_path / "smp")])
    o.input_width, o.output_width, o.y_dim = 24, 16, None
    m = DCGAN(None, o, load_data=False)
    m.train(data_files=files, sample_every=2)
    grids = sorted((tmp_path / "smp").glob("train_*.png"))
    assert m.counter == 3 and len(grids) == 1               # 20 files / batch 8 = 2 iterations, one grid
    assert np.asarray(Image.open(grids[0])).shape == (16 * 3, 16 * 2, 3)     # 8 samples -> 3 x 2 grid
    o = parse("--model CycleGAN --dataset h2z --batch_size 1 --fine_size 32 --load_size 36 --gf_dim 4 --df_dim 4 "
              "--max_size 2 --nEpochs 1".split() + ["--checkpoint_dir", str(tmp_path / "ck2"),
                                                    "--sample_dir", str(tmp_path / "smp2")])
    c = CycleGAN(None, o)
    out = c.train(files[:3], files[3:6], sample_every=2)
    assert all(np.isfinite(float(v)) for v in out.values())
    assert len(list((tmp_path / "smp2").glob("A_*.jpg"))) == 1 and all(opt.steps_taken == 3 for opt in c.optims.values())


@pytest.mark.gpu
def test_text2image_trains_from_the_flowers_files(tmp_path):
    """`--model Text2Image --dataset flowers --textEncoder DSSJE` reads <data_dir>/flowers/jpg and the
    DS-SJE caption .npy files like the reference (text2image.py:287-322, 389-417) and runs its schedule."""
    from PIL import Image
    from gan_amd.main import main
    rng = np.random.default_rng(0)
    (tmp_path / "flowers" / "jpg").mkdir(parents=True)
    (tmp_path / "flowers_caption_icml" / "class_00001").mkdir(parents=True)
    for i in range(12):
        name = "image_%05d" % i
        Image.fromarray(rng.integers(0, 256, (70, 80, 3)).astype(np.uint8)).save(str(tmp_path / "flowers" / "jpg" / (name + ".jpg")))
        np.save(str(tmp_path / "flowers_caption_icml" / "class_00001" / (name + ".npy")),
                rng.standard_normal((10, 1024)).astype(np.float32))
    o = parse(("--model Text2Image --dataset flowers --input_height 64 --textEncoder DSSJE --batch_size 4 --gf_dim 4 "
               "--df_dim 4 --t_dim 8 --caption_vector_length 12 --z_dim 10 --nEpochs 1 --save_every_batch 2").split()
              + ["--data_dir", str(tmp_path), "--checkpoint_dir", str(tmp_path / "ck"), "--sample_dir", str(tmp_path / "smp")])
    out = main(o)           # 9 training images / batch 4 -> 3 batches
    assert all(np.isfinite(float(out[k])) for k in ('d_loss1', 'd_loss2', 'd_loss3', 'g_loss_1', 'g_loss_2'))
    assert out['generator'].shape == (4, 64, 64, 3)
    assert (tmp_path / "ck" / "Text2Image_flowers_4_64").exists()       # the end-of-epoch checkpoint
    vis = tmp_path / "smp" / "Text2Image" / "flowers" / "text2image_DSSJE"   # save_for_vis after batch 2
    assert len(list(vis.glob("fake_image_*.jpg"))) == 4 and len(list(vis.glob("*_image_*.jpg"))) >= 8


@pytest.mark.gpu
def test_reference_directory_layout_end_to_end(tmp_path, monkeypatch):
    """Run from a directory laid out as the reference expects (./data/<dataset>/...), through main():
    DCGAN trains on ./data/faces/*.jpg and then writes the test_arange grids of main.py:98-101 /
    misc.py:181-198 under ./samples; CycleGAN trains on trainA / trainB, samples testA / testB pairs
    (cyclegan.py:216-236) and, in the test phase, translates every testA image (cyclegan.py:238-265)."""
    from PIL import Image
    from gan_amd.main import main
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(1)

    def images(folder, n, size=(30, 26)):
        (tmp_path / folder).mkdir(parents=True)
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, size + (3,)).astype(np.uint8)).save(str(tmp_path / folder / ("%03d.jpg" % i)))

    images("data/faces", 17)
    main(parse("--model DCGAN --dataset faces --input_height 24 --output_height 16 --batch_size 8 --gf_dim 8 --df_dim 8 "
               "--z_dim 12 --nEpochs 2 --print_every 100".split()))
    grids = sorted((tmp_path / "samples").glob("test_arange_*.png"))
    assert len(grids) == 12                                  # one per z coordinate (100 when z_dim >= 100)
    assert np.asarray(Image.open(str(grids[0]))).shape == (16 * 3, 16 * 3, 3)      # ceil(sqrt(8)) = 3
    assert (tmp_path / "checkpoint" / "DCGAN_faces_8_16_16").exists()              # counter 2
    import json
    recs = [json.loads(l) for l in open(str(tmp_path / "logs" / "DCGAN_faces_8_16_16" / "scalars.jsonl"))]
    assert [r["step"] for r in recs] == [2, 4]               # print_every 100: the first batch of each epoch
    assert all(abs(r["d_loss"] - r["d_loss_real"] - r["d_loss_fake"]) < 1e-6 and np.isfinite(r["g_loss"]) for r in recs)

    for part in ("trainA", "trainB", "testA", "testB"):
        images("data/h2z/" + part, 3, (40, 40))
    common = ("--model CycleGAN --dataset h2z --batch_size 1 --fine_size 32 --load_size 36 --gf_dim 4 --df_dim 4 "
              "--max_size 2 --nEpochs 34").split()
    main(parse(common))                                      # 102 iterations: sample at counter 101, checkpoint at 2
    smp = tmp_path / "samples" / "CycleGAN" / "h2z"
    assert sorted(p.name for p in smp.glob("*.jpg")) == ["A_33_0000.jpg", "A_33_0000_real.jpg", "B_33_0000.jpg",
                                                         "B_33_0000_real.jpg"]
    recs = [json.loads(l) for l in open(str(tmp_path / "logs" / "h2z_32" / "scalars.jsonl"))]
    assert len(recs) == 102 and sorted(recs[0]) == ["da_loss", "db_loss", "g_loss_a2b", "g_loss_b2a", "step", "time"]
    written = main(parse(common + ["--phase", "test", "--which_direction", "AtoB"]))
    assert [p.split("/")[-1] for p in written] == ["A2B_000.jpg", "A2B_001.jpg", "A2B_002.jpg"]
    assert np.asarray(Image.open(written[0])).shape == (32, 32, 3)
    with pytest.raises(Exception, match="which_direction"):
        main(parse(common + ["--phase", "test", "--which_direction", "sideways"]))
