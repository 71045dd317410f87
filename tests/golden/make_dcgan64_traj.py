"""Generates tests/golden/dcgan64_traj200.npz: the float64 numpy oracle's loss trajectory over 200
iterations of the benchmarked configuration (DCGAN 64x64x3, batch 64, gf_dim = df_dim = 64, lr 2e-4,
beta1 0.5; one D update + two G updates per iteration, models/dcgan.py:226-279 of the reference).

Only the 200 x 3 losses and the seeds are stored: the initial weights come from the oracle's seeded
initialisers (rounded to float32, so that the GPU side starts from the very same numbers) and every
iteration's images and z are drawn from one seeded generator, exactly as the test that replays the
trajectory draws them.  `losses_f32` (second mode, below) holds the first 60 iterations from the oracle run in
float32: the yardstick for what rounding alone does to this trajectory.  The reference ships no fixture and TensorFlow 1.x cannot run here: like the
other files of this directory this pins the ORACLE.  About half an hour on 8 cores:
    python tests/golden/make_dcgan64_traj.py [iterations]
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import autodiff as A          # noqa: E402
from oracle import nets                   # noqa: E402

WEIGHT_SEED, DATA_SEED, BATCH = 5, 64064, 64


def initial_oracle(dtype=np.float64):
    st = A.Store(seed=WEIGHT_SEED, dtype=dtype)
    net = nets.DCGAN(st, batch_size=BATCH)
    net.build(np.zeros((BATCH, 64, 64, 3)), np.zeros((BATCH, 100)))
    for n in st.values:
        st.values[n] = st.values[n].astype(np.float32).astype(dtype)
    return st, net


def draw(rng):
    im = rng.uniform(-1, 1, (BATCH, 64, 64, 3)).astype(np.float32)
    z = rng.uniform(-1, 1, (BATCH, 100)).astype(np.float32)
    return im, z


def _state(st, net, rng, losses, done):
    return {'values': st.values, 'adam': [(a.t, a.m, a.v) for a in (net.d_adam, net.g_adam)],
            'rng': rng.bit_generator.state, 'losses': losses, 'done': done}


def main():
    """A run that is interrupted resumes from the state it last left under /tmp (every 10 iterations)."""
    import pickle
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ckpt = "/tmp/dcgan64_traj_state_%d.pkl" % iters
    st, net = initial_oracle()
    rng = np.random.default_rng(DATA_SEED)
    losses = np.zeros((iters, 3))
    first = 0
    if os.path.exists(ckpt):
        with open(ckpt, "rb") as f:
            s = pickle.load(f)
        st.values.update(s['values'])
        for a, (t, m, v) in zip((net.d_adam, net.g_adam), s['adam']):
            a.t, a.m, a.v = t, m, v
        rng.bit_generator.state = s['rng']
        losses, first = s['losses'], s['done']
        print("resuming after iteration %d" % first, flush=True)
    t0 = time.time()
    for it in range(first, iters):
        im, z = draw(rng)
        out = net.train_iteration(im.astype(np.float64), z.astype(np.float64))
        losses[it] = (out['d_loss'], out['g_loss_1'], out['g_loss_2'])
        if it % 10 == 0 or it == iters - 1:
            print("iteration %d  d_loss %.6f  g_loss %.6f %.6f  (%.0f s)" % ((it,) + tuple(losses[it]) + (time.time() - t0,)),
                  flush=True)
        if it % 10 == 9 and it + 1 < iters:
            with open(ckpt + ".tmp", "wb") as f:
                pickle.dump(_state(st, net, rng, losses, it + 1), f, protocol=4)
            os.replace(ckpt + ".tmp", ckpt)
    np.savez(os.path.join(HERE, "dcgan64_traj%d.npz" % iters), losses=losses, columns=np.array(['d_loss', 'g_loss_1', 'g_loss_2']),
             weight_seed=WEIGHT_SEED, data_seed=DATA_SEED, batch=BATCH)


def add_float32_yardstick(n):
    """Adds `losses_f32`: the first n iterations of the same trajectory from the oracle run in float32 (numpy),
    i.e. how far float32 rounding alone moves this trajectory away from the float64 one.
        python tests/golden/make_dcgan64_traj.py f32 40"""
    path = os.path.join(HERE, "dcgan64_traj200.npz")
    gold = dict(np.load(path))
    st, net = initial_oracle(np.float32)
    rng = np.random.default_rng(DATA_SEED)
    out = np.zeros((n, 3))
    for it in range(n):
        im, z = draw(rng)
        o = net.train_iteration(im, z)
        out[it] = (o['d_loss'], o['g_loss_1'], o['g_loss_2'])
        print("float32 iteration %d  deviation from float64 %.3e" % (it, np.abs(out[it] - gold['losses'][it]).max()), flush=True)
    gold['losses_f32'] = out
    np.savez(path, **gold)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "f32":
        add_float32_yardstick(int(sys.argv[2]))
    else:
        main()
