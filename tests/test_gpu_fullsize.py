"""The BASELINE configurations at their full sizes, where the numpy oracle is too slow to run whole:
size-independent properties of the contraction kernels on every CycleGAN-256 / Text2Image layer shape
(the trilinear identity <conv(x,w),dy> = <x,dgrad(dy,w)> = <w,wgrad(x,dy)> plus oracle spot checks of
sampled output pixels), the full-size y-conditioned MNIST DCGAN against the oracle, and full-size
Text2Image / CycleGAN iterations (graph replay == eager, finite losses and gradients)."""
import argparse
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RNG = np.random.default_rng(4242)

# n, h, w, cb, cs, k, s, padding  (conv2d view: big [n,h,w,cb] -> small [n,.,.,cs])
FULL_SHAPES = [
    (1, 262, 262, 3, 64, 7, 1, 'VALID'),      # cyclegan g_e1_c on the reflect-padded image
    (1, 256, 256, 64, 128, 3, 2, 'SAME'),     # g_e2_c; transposed: g_d2_dc
    (1, 128, 128, 128, 256, 3, 2, 'SAME'),    # g_e3_c; transposed: g_d1_dc
    (1, 66, 66, 256, 256, 3, 1, 'VALID'),     # the 18 residual-block convolutions
    (1, 262, 262, 64, 3, 7, 1, 'VALID'),      # g_pred_c
    (1, 256, 256, 3, 64, 5, 2, 'SAME'),       # cyclegan d_h0_conv
    (1, 128, 128, 64, 128, 5, 2, 'SAME'),
    (1, 64, 64, 128, 256, 5, 2, 'SAME'),
    (1, 32, 32, 256, 512, 5, 1, 'SAME'),      # d_h3_conv, stride 1
    (1, 32, 32, 512, 1, 5, 1, 'SAME'),        # d_h3_pred
    (64, 4, 4, 768, 512, 1, 1, 'SAME'),       # text2image d_h3_conv_new (1x1 on the text-concatenated map)
    (64, 28, 28, 11, 11, 5, 2, 'SAME'),       # MNIST d_h0_conv
    (64, 14, 14, 138, 128, 5, 2, 'SAME'),     # MNIST g_h2 (transposed)
]


@pytest.mark.parametrize("case", FULL_SHAPES)
def test_trilinear_identity_and_spot_checks_at_full_size(case):
    from gan_amd import _lib as L
    from gpu_util import dev, geom_for, host
    n, h, w, cb, cs, k, s, padding = case
    g = geom_for(*case)
    x = RNG.standard_normal((n, h, w, cb)).astype(np.float32)
    wt = (RNG.standard_normal((k, k, cb, cs)) * 0.05).astype(np.float32)
    dy = RNG.standard_normal((n, g.hs, g.ws, cs)).astype(np.float32)
    xd, wd, dyd = dev(x), dev(wt), dev(dy)
    y = torch.empty((n, g.hs, g.ws, cs), device='cuda')
    dx = torch.empty_like(xd)
    dw = torch.empty_like(wd)
    db = torch.empty(cs, device='cuda')
    gp = ctypes.byref(g)
    wp, wn = L.workspace(int(L.lib.gan_conv_workspace_bytes(gp)))
    st = L.stream()
    L.check(L.lib.gan_conv2d_fwd(gp, L.ptr(xd), L.ptr(wd), None, L.ptr(y), wp, wn, st))
    L.check(L.lib.gan_conv2d_dgrad(gp, L.ptr(dyd), L.ptr(wd), L.ptr(dx), wp, wn, st))
    L.check(L.lib.gan_conv2d_wgrad(gp, L.ptr(xd), L.ptr(dyd), L.ptr(dw), L.ptr(db), 0, wp, wn, st))
    torch.cuda.synchronize()
    yh = host(y).astype(np.float64)
    s1 = float((yh * dy).sum())
    s2 = float((host(dx).astype(np.float64) * x).sum())
    s3 = float((host(dw).astype(np.float64) * wt).sum())
    scale = float(np.abs(yh * dy).sum())            # the size of what cancels in the three sums
    assert abs(s1 - s2) <= 1e-6 * scale and abs(s1 - s3) <= 1e-6 * scale, (s1, s2, s3, scale)
    np.testing.assert_allclose(host(db), dy.reshape(-1, cs).sum(0, dtype=np.float64), rtol=0,
                               atol=1e-4 * float(np.abs(dy).sum(axis=(0, 1, 2)).max()))
    # oracle spot checks: sampled output pixels recomputed in float64 from the definition
    # y[n,i,j,:] = sum_{a,b,c} x[n, i*s+a-pt, j*s+b-pl, c] * w[a,b,c,:]   (ops.py:68, TF SAME/VALID)
    ymax = float(np.abs(yh).max())
    for _ in range(24):
        b_, i, j = RNG.integers(0, n), RNG.integers(0, g.hs), RNG.integers(0, g.ws)
        if _ < 8:                                    # force border pixels in
            i, j = [0, g.hs - 1][_ % 2], [0, g.ws - 1][(_ // 2) % 2]
        acc = np.zeros(cs)
        for a in range(k):
            for b in range(k):
                r, c = i * s + a - g.pt, j * s + b - g.pl
                if 0 <= r < h and 0 <= c < w:
                    acc += x[b_, r, c].astype(np.float64) @ wt[a, b].astype(np.float64)
        assert np.abs(yh[b_, i, j] - acc).max() <= 1e-4 * ymax, (case, b_, i, j)


def test_mnist_conditional_dcgan_full_size_against_oracle():
    """BASELINE config 0 (scripts/DCGAN_mnist.sh): batch 64, 28x28x1, y_dim 10, z 100, default widths."""
    from oracle import autodiff as A
    from oracle import nets
    from gan_amd.dcgan import DCGAN
    from gpu_util import dev
    opts = argparse.Namespace(model='DCGAN', dataset='mnist', batch_size=64, input_height=28, input_width=28,
                              output_height=28, output_width=28, y_dim=10, z_dim=100, gf_dim=64, df_dim=64,
                              gfc_dim=1024, dfc_dim=1024, c_dim=1, lr=0.0002, beta1=0.5, crop=True, nEpochs=1,
                              train_size=1e8, print_every=1, checkpoint_dir='./checkpoint', seed=0)
    st = A.Store(seed=3)
    ref = nets.DCGAN(st, batch_size=64, output_height=28, output_width=28, c_dim=1, y_dim=10, z_dim=100,
                     gf_dim=64, df_dim=64, gfc_dim=1024, dfc_dim=1024, lr=0.0002, beta1=0.5)
    ref.build(np.zeros((64, 28, 28, 1), np.float32), np.zeros((64, 100), np.float32), np.zeros((64, 10), np.float32))
    gpu = DCGAN(None, opts, load_data=False)
    assert list(gpu.store.vars) == list(st.values)
    gpu.store.load_state_dict(st.values)
    rng = np.random.default_rng(3)
    for it in range(2):
        im = rng.uniform(0, 1, (64, 28, 28, 1)).astype(np.float32)
        z = rng.uniform(-1, 1, (64, 100)).astype(np.float32)
        y = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 64)]
        want = ref.train_iteration(im, z, y)
        out = gpu.train_iteration(dev(im), dev(z), dev(y))
        got = {'d_loss': float(out['d_loss_real']) + float(out['d_loss_fake']),
               'g_loss_1': float(out['g_loss_1']), 'g_loss_2': float(out['g_loss_2'])}
        for k_ in got:      # stated tolerance: 2e-3 relative on the losses of the first two iterations
            assert abs(got[k_] - want[k_]) <= 2e-3 * max(1.0, abs(want[k_])), (it, k_, got[k_], want[k_])


class _act_recorder(object):
    """Records the output of every activation of the given kinds that the model applies on the GPU side (the
    leaky-relu fused into d_h0_conv's epilogue, the relus / leaky-relus fused into the batch norms), in call order,
    as (kind, tensor)."""

    def __init__(self, kinds=('lrelu',)):
        self.kinds = kinds

    def __enter__(self):
        import gan_amd.dcgan as D
        from gan_amd import ops
        self.records = []
        self.saved = (D.conv2d, ops.batch_norm.__call__)
        rec, kinds = self.records, self.kinds

        def conv2d(x, *a, **k):
            y = self.saved[0](x, *a, **k)
            if k.get('act') in kinds:
                rec.append((k['act'], y))
            return y

        def bn_call(bn, x, *a, **k):
            y = self.saved[1](bn, x, *a, **k)
            if k.get('act') in kinds:
                rec.append((k['act'], y))
            return y
        D.conv2d, ops.batch_norm.__call__ = conv2d, bn_call
        return self

    @property
    def outputs(self):
        return [t for _, t in self.records]

    def __exit__(self, *exc):
        import gan_amd.dcgan as D
        from gan_amd import ops
        D.conv2d, ops.batch_norm.__call__ = self.saved


_lrelu_recorder = _act_recorder


def test_dcgan64_full_size_against_oracle():
    """BASELINE config 1, the benchmarked one: DCGAN 64x64x3, batch 64, gf_dim = df_dim = 64, default
    contraction arithmetic.  The D update's gradients of the first iteration and the losses of two whole
    iterations (1 D + 2 G updates each) against the float64 oracle.

    A leaky-relu's derivative jumps at 0, and among the 1.6e7 units of the two discriminator passes a few
    hundred sit within float32 rounding of the kink: there any float32 evaluation may take the other branch
    than float64 does, and each such unit changes the gradients below it by the whole gradient it switches.
    The units whose float64 pre-activation lies within 2e-5 of the kink are therefore masked on both sides:
    the oracle's backward pass takes the branch the GPU took there (read back from the GPU's activations; the
    forward values are untouched).  With them out of the way every gradient of the discriminator is held to
    3e-4 (max-norm) and the layers above the last leaky-relu to 1e-5; biases that feed a batch norm are noise
    on both sides."""
    from oracle import autodiff as A
    from oracle import nets
    from oracle import tf_ops as T
    from gan_amd.dcgan import DCGAN
    from gpu_util import check_group_grads, dev, host
    opts = argparse.Namespace(model='DCGAN', dataset='celebA', batch_size=64, input_height=108, input_width=108,
                              output_height=64, output_width=64, y_dim=None, z_dim=100, gf_dim=64, df_dim=64,
                              gfc_dim=1024, dfc_dim=1024, c_dim=3, lr=0.0002, beta1=0.5, crop=True, nEpochs=1,
                              train_size=1e8, print_every=1, checkpoint_dir='./checkpoint', seed=0)
    st = A.Store(seed=5, dtype=np.float64)
    ref = nets.DCGAN(st, batch_size=64)
    ref.build(np.zeros((64, 64, 64, 3)), np.zeros((64, 100)))
    for n in st.values:      # both sides start from the same float32 numbers
        st.values[n] = st.values[n].astype(np.float32).astype(np.float64)
    gpu = DCGAN(None, opts, load_data=False)
    assert list(gpu.store.vars) == list(st.values)
    gpu.store.load_state_dict(st.values)
    rng = np.random.default_rng(5)
    im = rng.uniform(-1, 1, (64, 64, 64, 3)).astype(np.float32)
    z = rng.uniform(-1, 1, (64, 100)).astype(np.float32)
    # gradients of d_loss w.r.t. d_vars at the initial weights: GPU first, recording its leaky-relu outputs
    with _lrelu_recorder() as rec:
        gpu.d_grads(dev(im), dev(z))
    torch.cuda.synchronize()
    assert gpu.stack_d_passes and len(rec.outputs) == 4           # h0..h3 of the stacked (real; fake) batch
    gpu_act = [host(t) for t in rec.outputs]
    margin, calls, masked = 2e-5, [0], [0]
    plain_lrelu = A.lrelu

    def lrelu_following_the_gpu_near_the_kink(tp, x, leak=0.2):
        i = calls[0]
        calls[0] += 1
        half = gpu_act[i % 4][(i // 4) * 64:(i // 4 + 1) * 64]    # the oracle runs D(real) then D(fake)
        assert half.shape == x.v.shape
        near = np.abs(x.v) < margin
        masked[0] += int(near.sum())
        upper = np.where(near, half >= 0, x.v >= 0)
        return tp.op(T.lrelu(x.v, leak), [x], lambda g: x.accumulate(np.where(upper, g, x.v.dtype.type(leak) * g)))
    tp = A.Tape(ref.st)
    A.lrelu = lrelu_following_the_gpu_near_the_kink
    try:
        lr_, lf_ = ref.d_losses(tp, im.astype(np.float64), z.astype(np.float64))
    finally:
        A.lrelu = plain_lrelu
    assert calls[0] == 8
    tp.backward(A.add(tp, lr_, lf_))
    want_g = tp.grads(ref.d_vars)
    total_units = sum(a.size for a in gpu_act)
    print("units masked at the leaky-relu kink: %d of %d" % (masked[0], total_units))
    # the mask must stay a handful of units: a wrong pre-activation could otherwise hide behind it
    assert masked[0] < 1e-3 * total_units
    check_group_grads(gpu.d_group, want_g, tol=3e-4)
    top = ('d_h3_lin/Matrix', 'd_h3_lin/bias', 'd_bn3/beta', 'd_bn3/gamma')
    for v in gpu.d_group.variables:
        if v.name.split('/', 1)[1] in top:
            from gpu_util import assert_close
            assert_close(host(v.grad), want_g[v.name], "grad " + v.name, tol=1e-5, elementwise=False)
    _check_g_update_gradients(ref, gpu, z)
    for it in range(2):
        want = ref.train_iteration(im.astype(np.float64), z.astype(np.float64))
        out = gpu.train_iteration(dev(im), dev(z))
        got = {'d_loss': float(out['d_loss_real']) + float(out['d_loss_fake']),
               'g_loss_1': float(out['g_loss_1']), 'g_loss_2': float(out['g_loss_2'])}
        # stated tolerance: 2e-3 on the first iteration's losses, 1e-2 on the second's (Adam's first steps move
        # every weight by about lr whatever the size of its gradient, which amplifies the float32 differences
        # of the first iteration; the 200-step trajectory tests bound the growth)
        tol = 2e-3 if it == 0 else 1e-2
        for k_ in got:
            assert abs(got[k_] - want[k_]) <= tol * max(1.0, abs(want[k_])), (it, k_, got[k_], want[k_])
        im = rng.uniform(-1, 1, (64, 64, 64, 3)).astype(np.float32)
        z = rng.uniform(-1, 1, (64, 100)).astype(np.float32)


def _check_g_update_gradients(ref, gpu, z):
    """The G update's gradients at the benchmarked size (dcgan.py:237-247: backward through D into G -- the
    deconvolutions' input and weight gradients, the gated tanh) against the float64 oracle, at the initial weights.
    The relu units of the generator and the leaky-relu units of the discriminator pass that sit within 2e-5 of
    their kink in float64 take, in the oracle's backward pass, the branch the GPU took (read from its activations),
    as in the D update's comparison; their number is printed and bounded.  Every generator gradient to 3e-4
    (max-norm); biases in front of a batch norm are noise on both sides."""
    from oracle import autodiff as A
    from oracle import tf_ops as T
    from gpu_util import assert_close, check_group_grads, dev, host
    with _act_recorder(('relu', 'lrelu')) as rec:
        g_loss = gpu.g_grads(dev(z))
    torch.cuda.synchronize()
    kinds = [k for k, _ in rec.records]
    assert kinds == ['relu'] * 4 + ['lrelu'] * 4, kinds          # g_bn0..3, then h0..h3 of D(G(z))
    gpu_act = {'relu': [host(t) for k, t in rec.records if k == 'relu'],
               'lrelu': [host(t) for k, t in rec.records if k == 'lrelu']}
    margin, calls, masked = 2e-5, {'relu': 0, 'lrelu': 0}, [0]
    plain = (A.relu, A.lrelu)

    def follow(kind, x):
        act = gpu_act[kind][calls[kind]]
        calls[kind] += 1
        assert act.shape == x.v.shape
        near = np.abs(x.v) < margin
        masked[0] += int(near.sum())
        return near, act

    def relu(tp, x):
        near, act = follow('relu', x)
        on = np.where(near, act > 0, x.v > 0)
        return tp.op(T.relu(x.v), [x], lambda g: x.accumulate(np.where(on, g, 0.0)))

    def lrelu(tp, x, leak=0.2):
        near, act = follow('lrelu', x)
        upper = np.where(near, act >= 0, x.v >= 0)
        return tp.op(T.lrelu(x.v, leak), [x], lambda g: x.accumulate(np.where(upper, g, x.v.dtype.type(leak) * g)))
    tp = A.Tape(ref.st)
    A.relu, A.lrelu = relu, lrelu
    try:
        G = ref.generator(tp, tp.const(z.astype(np.float64)))
        want_loss = A.sigmoid_ce_mean(tp, ref.discriminator(tp, G), 1.0)
    finally:
        A.relu, A.lrelu = plain
    assert calls == {'relu': 4, 'lrelu': 4}
    tp.backward(want_loss)
    want_g = tp.grads(ref.g_vars)
    total_units = sum(a.size for v in gpu_act.values() for a in v)
    print("G update: units masked at a relu / leaky-relu kink: %d of %d" % (masked[0], total_units))
    assert masked[0] < 1e-3 * total_units
    assert_close([float(g_loss)], [float(want_loss.v)], "g_loss at the initial weights", tol=1e-5)
    check_group_grads(gpu.g_group, want_g, tol=3e-4)


def test_dcgan64_full_size_loss_trajectory_200_steps():
    """The benchmarked configuration over 200 iterations (1 D + 2 G updates each, models/dcgan.py:226-279)
    against the committed float64-oracle trajectory tests/golden/dcgan64_traj200.npz (generator script
    alongside: same weight seed, same data generator, so only losses and seeds are stored).

    At this size, on noise images, training is chaotic in the numerical sense: Adam turns the rounding noise of
    10^7 near-zero gradients into +-lr steps and relu / leaky-relu units flip at their kink, so ANY float32
    evaluation leaves the float64 trajectory quickly -- the oracle itself, run in float32 numpy, is 0.06 off
    after 5 iterations, 0.6 after 12 and 1.7 after 28 (`losses_f32` in the fixture, the yardstick).  Stated
    tolerance: (a) over the 60 iterations the yardstick covers, the running maximum of |loss - float64 oracle|
    stays within 4x the yardstick's running maximum (+ 2e-3); (b) the first iteration agrees to 1e-3 of the loss; (c) over
    all 200 iterations every loss is finite and the mean of each loss over the last 100 iterations agrees with
    the float64 oracle's to 25 %: the two runs have decorrelated by then but sample the same dynamics."""
    import os
    import sys
    from gan_amd.dcgan import DCGAN
    from gpu_util import dev
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    try:
        import make_dcgan64_traj as M
    finally:
        sys.path.remove(here)
    gold = np.load(os.path.join(here, "dcgan64_traj200.npz"))
    assert int(gold['weight_seed']) == M.WEIGHT_SEED and int(gold['data_seed']) == M.DATA_SEED and int(gold['batch']) == 64
    want, yard = gold['losses'], gold['losses_f32']
    st, _ = M.initial_oracle()
    opts = argparse.Namespace(model='DCGAN', dataset='celebA', batch_size=64, input_height=108, input_width=108,
                              output_height=64, output_width=64, y_dim=None, z_dim=100, gf_dim=64, df_dim=64,
                              gfc_dim=1024, dfc_dim=1024, c_dim=3, lr=0.0002, beta1=0.5, crop=True, nEpochs=1,
                              train_size=1e8, print_every=1, checkpoint_dir='./checkpoint', seed=0)
    gpu = DCGAN(None, opts, load_data=False)
    assert list(gpu.store.vars) == list(st.values)
    gpu.store.load_state_dict(st.values)
    rng = np.random.default_rng(M.DATA_SEED)
    outs = []
    for it in range(want.shape[0]):
        im, z = M.draw(rng)
        outs.append(gpu.train_iteration(dev(im), dev(z)))           # losses stay on the device until the end
    torch.cuda.synchronize()
    got = np.array([[float(o['d_loss_real']) + float(o['d_loss_fake']), float(o['g_loss_1']), float(o['g_loss_2'])]
                    for o in outs])
    assert np.isfinite(got).all()
    assert gpu.d_optim.steps_taken == 200 and gpu.g_optim.steps_taken == 400
    n = yard.shape[0]
    dev_gpu = np.maximum.accumulate(np.abs(got[:n] - want[:n]).max(axis=1))
    dev_f32 = np.maximum.accumulate(np.abs(yard - want[:n]).max(axis=1))
    tail_gpu, tail_ref = got[100:].mean(axis=0), want[100:].mean(axis=0)
    print("running max |loss - float64 oracle| at iterations 1 / 5 / 10 / 20 / 60: gpu %s, float32 numpy oracle %s; "
          "mean losses over iterations 100-199: gpu %s, float64 oracle %s"
          % (np.round(dev_gpu[[0, 4, 9, 19, n - 1]], 4), np.round(dev_f32[[0, 4, 9, 19, n - 1]], 4),
             np.round(tail_gpu, 3), np.round(tail_ref, 3)))
    assert dev_gpu[0] <= 1e-3 * max(1.0, float(np.abs(want[0]).max()))      # losses of magnitude ~10 here
    assert (dev_gpu <= 4 * dev_f32 + 2e-3).all(), np.nonzero(dev_gpu > 4 * dev_f32 + 2e-3)[0][:5]
    assert (np.abs(tail_gpu - tail_ref) <= 0.25 * np.abs(tail_ref)).all()


def _finite_grads(optims):
    for o in optims:
        assert bool(torch.isfinite(o.group.grads).all())
        assert float(o.group.grads.abs().max()) > 0


def test_text2image_full_size_graph_replay_equals_eager():
    from gan_amd.graph import CapturedIteration, model_state_restore, model_state_snapshot
    from gan_amd.main import build_parser
    from gan_amd.text2image import Text2Image
    from gpu_util import dev
    o = build_parser().parse_args("--dataset flowers --input_height 64 --model Text2Image --textEncoder DSSJE".split())
    eager, graphed = Text2Image(None, o), Text2Image(None, o)
    graphed.store.load_state_dict(eager.store.state_dict())
    rng = np.random.default_rng(1)
    batches = [[dev(rng.uniform(0, 1, (64, 64, 64, 3))), dev(rng.uniform(0, 1, (64, 64, 64, 3))),
                dev(rng.standard_normal((64, 1024))), dev(rng.uniform(-1, 1, (64, 100)))] for _ in range(2)]
    optims = [graphed.d_optim, graphed.g_optim]
    fn = lambda a, b, c, d: {k: v for k, v in graphed.train_iteration(a, b, c, d).items() if k != 'generator'}
    step = CapturedIteration(fn, batches[0], snapshot=lambda: model_state_snapshot(graphed.store, optims),
                             restore=lambda s: model_state_restore(graphed.store, optims, s))
    for b in batches:
        e = eager.train_iteration(*b)
        g = step(*b)
        torch.cuda.synchronize()
        for k in g:
            assert np.isfinite(float(e[k])) and float(e[k]) == float(g[k]), k
    _finite_grads([eager.d_optim, eager.g_optim])
    for name, v in eager.store.vars.items():
        assert torch.equal(v.data, graphed.store.vars[name].data), name
    gen = e['generator']
    assert gen.shape == (64, 64, 64, 3) and 0.0 <= float(gen.min()) and float(gen.max()) <= 1.0


def test_cyclegan_full_size_iteration():
    """256x256, gf_dim = df_dim = 64, batch 1: two iterations run, losses and gradients are finite,
    the untrained LSGAN discriminator losses sit near their known starting values, the image pool
    hands back what it stored, and repeating the run from the same state gives the same numbers."""
    from gan_amd.cyclegan import CycleGAN
    from gan_amd.main import build_parser
    from gpu_util import dev
    o = build_parser().parse_args("--model CycleGAN --batch_size 1 --dataset horse2zebra --max_size 2".split())
    rng = np.random.default_rng(2)
    data = [dev(rng.uniform(-1, 1, (1, 256, 256, 6))) for _ in range(2)]
    runs = []
    for _ in range(2):
        m = CycleGAN(None, o)
        outs = [CycleGAN.totals(m.train_iteration(d)) for d in data]
        torch.cuda.synchronize()
        runs.append(outs)
        _finite_grads(list(m.optims.values()))
        assert m.pool.num_img == 2 and all(opt.steps_taken == 2 for opt in m.optims.values())
        out = m.test(data[0][..., :3].contiguous(), 'AtoB')
        assert out.shape == (1, 256, 256, 3) and float(out.abs().max()) <= 1.0
    for a, b in zip(*runs):
        for k in a:
            assert np.isfinite(a[k]) and a[k] == b[k], k
    # cycle terms: L1_lambda * mean|x - G(G(x))| of U(-1,1) images against a tanh output is O(lambda * 0.5)
    assert 1.0 < runs[0][0]['g_loss_a2b'] < 30.0 and 0.0 < runs[0][0]['db_loss'] < 5.0


def _golden(module):
    import importlib
    import os
    import sys
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    try:
        return importlib.import_module(module), here
    finally:
        sys.path.remove(here)


def test_text2image_full_size_losses_against_the_float64_oracle_fixture():
    """BASELINE config 3 of the hot path, Text2Image GAN-CLS at 64x64, batch 64 (models/text2image.py:73-128,
    243-279): 12 iterations (1 D update on real / wrong / fake + 2 G updates each) against the committed
    float64-oracle fixture tests/golden/t2i64_traj.npz (generator script alongside: seeds only, the weights and
    the inputs are drawn again here).  Stated tolerance, as for DCGAN-64: every loss of the first iteration within
    1e-3 (relative, losses below 1 absolute); afterwards the running maximum of |loss - float64 oracle| stays within
    4x that of the oracle run in float32 (+ 2e-3), the yardstick for what float32 rounding does to this
    trajectory (0.004 after 1 iteration, 0.07 after 6, 0.4 after 7 for the float32 ORACLE)."""
    import os
    from gan_amd.main import build_parser
    from gan_amd.text2image import Text2Image
    from gpu_util import dev
    M, here = _golden("make_fullsize_pins")
    gold = np.load(os.path.join(here, "t2i64_traj.npz"))
    assert int(gold['weight_seed']) == M.T2I_WEIGHT_SEED and int(gold['data_seed']) == M.T2I_DATA_SEED
    assert [str(c) for c in gold['columns']] == list(M.T2I_COLUMNS)
    want, yard = gold['losses'], gold['losses_f32']
    st, _ = M.t2i_oracle()
    o = build_parser().parse_args("--dataset flowers --input_height 64 --model Text2Image --textEncoder DSSJE".split())
    gpu = Text2Image(None, o)
    assert list(gpu.store.vars) == list(st.values)
    gpu.store.load_state_dict(st.values)
    rng = np.random.default_rng(M.T2I_DATA_SEED)
    outs = []
    for it in range(want.shape[0]):
        outs.append(gpu.train_iteration(*[dev(a) for a in M.t2i_draw(rng)]))
    torch.cuda.synchronize()
    got = np.array([[float(o_[k]) for k in M.T2I_COLUMNS] for o_ in outs])
    assert np.isfinite(got).all()
    first = np.abs(got[0] - want[0]) / np.maximum(1.0, np.abs(want[0]))
    dev_gpu = np.maximum.accumulate(np.abs(got - want).max(axis=1))
    dev_f32 = np.maximum.accumulate(np.abs(yard - want).max(axis=1))
    print("Text2Image 64x64: first iteration's relative loss deviations %s; running max |loss - float64 oracle| at "
          "iterations 1 / 3 / 6 / 12: gpu %s, float32 numpy oracle %s"
          % (np.array2string(first, precision=2), np.round(dev_gpu[[0, 2, 5, 11]], 4), np.round(dev_f32[[0, 2, 5, 11]], 4)))
    assert (first <= 1e-3).all(), first
    assert (dev_gpu <= 4 * dev_f32 + 2e-3).all(), np.nonzero(dev_gpu > 4 * dev_f32 + 2e-3)[0][:5]


def test_cyclegan_full_size_against_the_float64_oracle_fixture():
    """CycleGAN at 256x256, batch 1, gf_dim = df_dim = 64 (models/cyclegan.py:47-116,159-180) against the committed
    float64-oracle fixture tests/golden/cyclegan256_pin.npz.
    (A) Each of the four updates evaluated at the initial weights: its loss terms to 1e-4 and, for every variable, a
    checksum of its gradient -- L2 norm and projection onto a seeded +-1 vector -- within max(5e-4, 4x the float32
    oracle's deviation for that variable, the float32 oracle's worst deviation over the group) of the variable's
    gradient norm: the generators are ~50 normalised layers deep with relu
    kinks among 4M units per layer, and the float32 oracle's own deviation from float64 is the yardstick for
    what float32 arithmetic can deliver there.  Biases in front of a batch norm (zero true gradient) must stay tiny.
    (B) Two iterations of the schedule: all ten loss terms, the first iteration's within 1e-3 (relative, terms
    below 1 absolute) and 4x the yardstick, the second's within 4x the yardstick + 2e-3."""
    import os
    from gan_amd.cyclegan import CycleGAN
    from gan_amd.main import build_parser
    from gpu_util import dev, host
    M, here = _golden("make_fullsize_pins")
    gold = np.load(os.path.join(here, "cyclegan256_pin.npz"))
    assert int(gold['weight_seed']) == M.CYC_WEIGHT_SEED and int(gold['data_seed']) == M.CYC_DATA_SEED
    assert int(gold['proj_seed']) == M.CYC_PROJ_SEED and [str(c) for c in gold['columns']] == list(M.CYC_COLUMNS)
    st, _ = M.cyc_oracle()
    initial = {n: v.copy() for n, v in st.values.items()}
    o = build_parser().parse_args("--model CycleGAN --batch_size 1 --dataset horse2zebra".split())
    assert o.max_size == 50 and o.fine_size == 256
    m = CycleGAN(None, o)
    assert list(m.store.vars) == list(st.values)
    rng = np.random.default_rng(M.CYC_DATA_SEED)
    real = dev(M.cyc_draw(rng))
    real_A, real_B = m.split(real)
    # ---- (A) the four updates, each from the initial weights
    m.store.load_state_dict(initial)
    fake_A, fake_B = m.forward_fakes(real)
    runs = (('db', lambda: m.d_update('B', real_B, fake_B), ('db_loss_real', 'db_loss_fake')),
            ('g_a2b', lambda: m.g_update('a2b', real), ('g_a2b_gan', 'g_a2b_cyc_a', 'g_a2b_cyc_b')),
            ('da', lambda: m.d_update('A', real_A, fake_A), ('da_loss_real', 'da_loss_fake')),
            ('g_b2a', lambda: m.g_update('b2a', real), ('g_b2a_gan', 'g_b2a_cyc_a', 'g_b2a_cyc_b')))
    col = {c: i for i, c in enumerate(M.CYC_COLUMNS)}
    worst = {}
    for key, run, terms in runs:
        m.store.load_state_dict(initial)
        vals = run()
        torch.cuda.synchronize()
        for t, v in zip(terms, vals):
            w = float(gold['init_losses'][col[t]])
            assert abs(float(v) - w) <= 1e-4 * max(1.0, abs(w)), (t, float(v), w)
        names = [str(n) for n in gold[key + '_names']]
        grads = {v.name: host(v.grad).astype(np.float64) for v in m.optims[key].group.variables}
        assert list(grads) == names
        n64, p64 = gold[key + '_norm'], gold[key + '_proj']
        n32, p32 = gold[key + '_norm_f32'], gold[key + '_proj_f32']
        biggest = float(n64.max())
        bad = []
        w_gpu = w_f32 = 0.0
        live = n64 >= 1e-5 * biggest
        group_f32 = float((np.maximum(np.abs(n32 - n64), np.abs(p32 - p64)) / n64)[live].max())
        for i, name in enumerate(names):
            g = grads[name]
            nrm = float(np.sqrt((g * g).sum()))
            if n64[i] < 1e-5 * biggest:                    # a bias in front of a batch norm: rounding noise on both sides
                assert nrm < 1e-3 * biggest, (name, nrm, biggest)
                continue
            prj = float((g * M.cyc_projection(name, g.shape)).sum())
            e_gpu = max(abs(nrm - n64[i]), abs(prj - p64[i])) / n64[i]
            e_f32 = max(abs(n32[i] - n64[i]), abs(p32[i] - p64[i])) / n64[i]
            w_gpu, w_f32 = max(w_gpu, e_gpu), max(w_f32, e_f32)
            # (which units flip at a kink is a matter of chance per variable: the float32 oracle's worst variable of
            # the group bounds what chance does, besides 4x the variable's own deviation)
            if e_gpu > max(5e-4, 4 * e_f32, group_f32):
                bad.append((name, e_gpu, e_f32, group_f32))
        worst[key] = (w_gpu, w_f32)
        assert not bad, bad
    print("CycleGAN 256x256 gradient checksums at the initial weights, worst deviation / gradient norm per group "
          "(gpu, float32 numpy oracle): %s" % {k: ("%.1e" % a, "%.1e" % b) for k, (a, b) in worst.items()})
    # ---- (B) the schedule
    m = CycleGAN(None, o)
    m.store.load_state_dict(initial)
    want, yard = gold['losses'], gold['losses_f32']
    rng = np.random.default_rng(M.CYC_DATA_SEED)
    for it in range(want.shape[0]):
        out = m.train_iteration(dev(M.cyc_draw(rng)))
        torch.cuda.synchronize()
        got = np.array([float(out[c]) for c in M.CYC_COLUMNS])
        d_gpu, d_f32 = np.abs(got - want[it]), np.abs(yard[it] - want[it])
        print("CycleGAN 256x256 iteration %d: max |term - float64 oracle| gpu %.2e, float32 numpy oracle %.2e"
              % (it, d_gpu.max(), d_f32.max()))
        if it == 0:
            assert (d_gpu <= 1e-3 * np.maximum(1.0, np.abs(want[it]))).all(), (got, want[it])
            assert (d_gpu <= 4 * d_f32 + 1e-4).all(), (d_gpu, d_f32)
        else:
            assert (d_gpu <= 4 * d_f32.max() + 2e-3).all(), (d_gpu, d_f32)
    assert m.pool.num_img == 2 and all(opt.steps_taken == 2 for opt in m.optims.values())
