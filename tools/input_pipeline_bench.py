#!/usr/bin/env python
"""DCGAN-64 training from image FILES on one MI355X: what the host input pipeline costs.

Writes `--images` CelebA-sized (178x218) JPEG files of noise into a temporary directory and runs
DCGAN.train over them: one epoch (every batch decoded by the input pipeline's threads while the GPU
trains) and three epochs (epochs 2 and 3 read the uint8 copy the first one left in GPU memory).
Prints one JSON line."""
import argparse
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gan_amd.dcgan import DCGAN  # noqa: E402
from gan_amd.main import build_parser  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=2048)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    from PIL import Image
    rng = np.random.default_rng(0)
    tmp = tempfile.mkdtemp(prefix="gan_files_")
    files = []
    for i in range(args.images):
        small = rng.integers(0, 256, (28, 23, 3), dtype=np.uint8)
        f = os.path.join(tmp, "%06d.jpg" % i)
        Image.fromarray(small).resize((178, 218), Image.BICUBIC).save(f, quality=90)
        files.append(f)

    def run(epochs, **kw):
        o = build_parser().parse_args(("--dataset celebA --input_height 108 --output_height 64 --crop true --nEpochs %d "
                                       "--print_every 1000000 --num_preprocess_threads %d --checkpoint_dir %s"
                                       % (epochs, args.threads, os.path.join(tmp, "ck"))).split())
        o.y_dim, o.c_dim, o.input_width, o.output_width = None, 3, 108, 64
        for k, v in kw.items():
            setattr(o, k, v)
        m = DCGAN(None, o, load_data=False)
        m.train(data_files=files[:64], sample_every=0)          # capture, workspace, allocator pools
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            m.train(data_files=files, sample_every=0)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    t0 = time.perf_counter()
    o = build_parser().parse_args("--dataset celebA --input_height 108 --output_height 64".split())
    o.y_dim, o.c_dim, o.input_width, o.output_width = None, 3, 108, 64
    m = DCGAN(None, o, load_data=False)
    for i in range(0, 256, 64):
        m._load_files(files[i:i + 64])
    inline_ms = (time.perf_counter() - t0) / 4 * 1e3
    t1 = run(1)
    t3 = run(3)
    batches = args.images // 64
    print(json.dumps({
        "images": args.images, "decode_threads": args.threads,
        "inline_decode_ms_per_batch_one_thread": round(inline_ms, 2),
        "epoch_decoding_ms_per_batch": round(t1 / batches * 1e3, 3),
        "epoch_resident_ms_per_batch": round((t3 - t1) / (2 * batches) * 1e3, 3),
        "images_per_s_decoding": round(args.images / t1, 1),
        "images_per_s_resident": round(2 * args.images / (t3 - t1), 1)}))


if __name__ == "__main__":
    main()
