"""Generates the float64-oracle fixtures that pin Text2Image and CycleGAN at their BASELINE sizes:

  t2i64_traj.npz      Text2Image GAN-CLS, 64x64x3, batch 64, t_dim 256, 1024-d captions (models/text2image.py:73-128,
                      243-279 of the reference): d_loss1..3, g_loss_1, g_loss_2 of 12 iterations from the float64
                      oracle (`losses`) and from the oracle run in float32 (`losses_f32`, the yardstick for what
                      rounding alone does to the trajectory).
  cyclegan256_pin.npz CycleGAN 256x256, batch 1, gf_dim = df_dim = 64, L1_lambda 10, LSGAN, pool of 50
                      (models/cyclegan.py:47-116,159-180): (A) each of the four updates evaluated at the initial
                      weights: its loss terms (`init_losses`) and a checksum of every variable's gradient, its L2
                      norm and its projection onto a seeded +-1 vector (`<group>_norm`, `<group>_proj`, float64;
                      `..._f32` the float32 oracle's); (B) the ten loss terms of 2 iterations of the schedule
                      (float64 `losses`, float32 `losses_f32`).

Only losses, checksums and seeds are stored: weights come from the oracle's seeded initialisers rounded to float32
and inputs from one seeded generator, exactly as the tests that replay the fixtures draw them.  The reference ships
no fixture and TensorFlow 1.x cannot run here: like the other files of this directory this pins the ORACLE.
    python tests/golden/make_fullsize_pins.py t2i | cyclegan
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import autodiff as A          # noqa: E402
from oracle import nets                   # noqa: E402

T2I_WEIGHT_SEED, T2I_DATA_SEED, T2I_BATCH, T2I_ITERS = 11, 64111, 64, 12
T2I_COLUMNS = ('d_loss1', 'd_loss2', 'd_loss3', 'g_loss_1', 'g_loss_2')
CYC_WEIGHT_SEED, CYC_DATA_SEED, CYC_PROJ_SEED, CYC_ITERS = 13, 256013, 7, 2
CYC_COLUMNS = ('db_loss_real', 'db_loss_fake', 'g_a2b_gan', 'g_a2b_cyc_a', 'g_a2b_cyc_b',
               'da_loss_real', 'da_loss_fake', 'g_b2a_gan', 'g_b2a_cyc_a', 'g_b2a_cyc_b')
CYC_GROUPS = ('db', 'g_a2b', 'da', 'g_b2a')


def _round_to_f32(st, dtype):
    for n in st.values:
        st.values[n] = st.values[n].astype(np.float32).astype(dtype)


# ------------------------------------------------------------------------------------------ Text2Image
def t2i_oracle(dtype=np.float64):
    st = A.Store(seed=T2I_WEIGHT_SEED, dtype=dtype)
    net = nets.Text2Image(st, batch_size=T2I_BATCH)
    z = np.zeros((T2I_BATCH, 64, 64, 3))
    net.build(z, z, np.zeros((T2I_BATCH, 1024)), np.zeros((T2I_BATCH, 100)))
    _round_to_f32(st, dtype)
    return st, net


def t2i_draw(rng):
    """Synthetic inputs of the configuration: real / wrong images U[0,1), captions N(0,1), z U(-1,1)."""
    real = rng.uniform(0, 1, (T2I_BATCH, 64, 64, 3)).astype(np.float32)
    wrong = rng.uniform(0, 1, (T2I_BATCH, 64, 64, 3)).astype(np.float32)
    cap = rng.standard_normal((T2I_BATCH, 1024)).astype(np.float32)
    z = rng.uniform(-1, 1, (T2I_BATCH, 100)).astype(np.float32)
    return real, wrong, cap, z


def make_t2i():
    out = {}
    for key, dtype in (('losses', np.float64), ('losses_f32', np.float32)):
        st, net = t2i_oracle(dtype)
        rng = np.random.default_rng(T2I_DATA_SEED)
        losses = np.zeros((T2I_ITERS, len(T2I_COLUMNS)))
        t0 = time.time()
        for it in range(T2I_ITERS):
            o = net.train_iteration(*[a.astype(dtype) for a in t2i_draw(rng)])
            losses[it] = [o[k] for k in T2I_COLUMNS]
            print("%s iteration %d  %s  (%.0f s)" % (key, it, np.round(losses[it], 6), time.time() - t0), flush=True)
        out[key] = losses
    np.savez(os.path.join(HERE, "t2i64_traj.npz"), columns=np.array(T2I_COLUMNS), weight_seed=T2I_WEIGHT_SEED,
             data_seed=T2I_DATA_SEED, batch=T2I_BATCH, **out)


# ------------------------------------------------------------------------------------------ CycleGAN
def cyc_oracle(dtype=np.float64):
    st = A.Store(seed=CYC_WEIGHT_SEED, dtype=dtype)
    net = nets.CycleGAN(st, batch_size=1, image_size=256, gf_dim=64, df_dim=64, L1_lambda=10.0, use_lsgan=True,
                        max_size=50)
    net.build(np.zeros((1, 256, 256, 6)))
    _round_to_f32(st, dtype)
    return st, net


def cyc_draw(rng):
    return rng.uniform(-1, 1, (1, 256, 256, 6)).astype(np.float32)


def cyc_projection(name, shape):
    """The +-1 vector a variable's gradient is projected on: seeded by the variable's name."""
    seed = [CYC_PROJ_SEED] + [ord(c) for c in name]
    return np.random.default_rng(seed).integers(0, 2, size=shape).astype(np.float64) * 2 - 1


def cyc_checksums(names, grads):
    g = [np.asarray(grads[n], dtype=np.float64) for n in names]
    return (np.array([np.sqrt((a * a).sum()) for a in g]),
            np.array([(a * cyc_projection(n, a.shape)).sum() for n, a in zip(names, g)]))


def cyc_updates_from_initial_weights(net, real, restore):
    """The four updates of cyclegan.py:159-180, each evaluated at the INITIAL weights (`restore()` puts them back in
    front of every one): loss terms and gradients with no Adam step in between, so that what is pinned is the
    gradient arithmetic of each update, not the amplification of rounding noise by the previous update's step.
    Returns ({term: value}, {group: {name: gradient}})."""
    a, b = net.split(real)
    restore()
    fake_A, fake_B = net.forward_fakes(real)
    jobs = (('db', lambda tp, t: net.d_loss(tp, b, fake_B, "discriminatorB", t), net.db_vars, 'db_loss_'),
            ('g_a2b', lambda tp, t: net.g_loss(tp, real, 'a2b', t), net.g_vars_a2b, 'g_a2b_'),
            ('da', lambda tp, t: net.d_loss(tp, a, fake_A, "discriminatorA", t), net.da_vars, 'da_loss_'),
            ('g_b2a', lambda tp, t: net.g_loss(tp, real, 'b2a', t), net.g_vars_b2a, 'g_b2a_'))
    terms, grads = {}, {}
    for key, loss_fn, names, prefix in jobs:
        restore()
        tp = A.Tape(net.st)
        t = {}
        tp.backward(loss_fn(tp, t))
        grads[key] = tp.grads(names)
        terms.update({prefix + k: v for k, v in t.items()})
    restore()
    return terms, grads


def make_cyclegan():
    out = {}
    # part A: every update's gradient at the initial weights
    for suffix, dtype in (('', np.float64), ('_f32', np.float32)):
        st, net = cyc_oracle(dtype)
        initial = {n: v.copy() for n, v in st.values.items()}

        def restore():
            for n, v in initial.items():
                st.values[n] = v.copy()
        t0 = time.time()
        real = cyc_draw(np.random.default_rng(CYC_DATA_SEED)).astype(dtype)
        terms, grads = cyc_updates_from_initial_weights(net, real, restore)
        names = {'db': net.db_vars, 'g_a2b': net.g_vars_a2b, 'da': net.da_vars, 'g_b2a': net.g_vars_b2a}
        out['init_losses' + suffix] = np.array([terms[k] for k in CYC_COLUMNS])
        for g in CYC_GROUPS:
            nrm, prj = cyc_checksums(names[g], grads[g])
            out['%s_norm%s' % (g, suffix)] = nrm
            out['%s_proj%s' % (g, suffix)] = prj
            out['%s_names' % g] = np.array(names[g])
        print("float%d gradients at the initial weights  %s  (%.0f s)" % (np.dtype(dtype).itemsize * 8,
              np.round(out['init_losses' + suffix], 6), time.time() - t0), flush=True)
    # part B: the schedule, two iterations
    for suffix, dtype in (('', np.float64), ('_f32', np.float32)):
        st, net = cyc_oracle(dtype)
        rng = np.random.default_rng(CYC_DATA_SEED)
        losses = np.zeros((CYC_ITERS, len(CYC_COLUMNS)))
        t0 = time.time()
        for it in range(CYC_ITERS):
            d = {}
            net.train_iteration(cyc_draw(rng).astype(dtype), detail=d)
            losses[it] = [d[k] for k in CYC_COLUMNS]
            print("float%d iteration %d  %s  (%.0f s)" % (np.dtype(dtype).itemsize * 8, it, np.round(losses[it], 6),
                                                          time.time() - t0), flush=True)
        out['losses' + suffix] = losses
    np.savez(os.path.join(HERE, "cyclegan256_pin.npz"), columns=np.array(CYC_COLUMNS), weight_seed=CYC_WEIGHT_SEED,
             data_seed=CYC_DATA_SEED, proj_seed=CYC_PROJ_SEED, **out)


if __name__ == "__main__":
    {'t2i': make_t2i, 'cyclegan': make_cyclegan}[sys.argv[1]]()
