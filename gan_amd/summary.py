"""Scalar training log: what the reference sends to tf.summary scalars (dcgan.py:136-142 d_loss / g_loss /
d_loss_real / d_loss_fake, cyclegan.py:87-100 the four CycleGAN losses, written through a FileWriter on
every run, dcgan.py:165,234-247) goes to <log_dir>/scalars.jsonl, one JSON object per logged step:
{"step": counter, "time": seconds since start, <name>: value, ...}.  Histogram and image summaries have
no counterpart; the sampler grids cover the images."""
import json
import os
import time


class ScalarLog(object):
    def __init__(self, log_dir, enabled=True):
        self.path = None
        self._t0 = time.time()
        self._fd = None
        if enabled and log_dir:
            os.makedirs(log_dir, exist_ok=True)
            self.path = os.path.join(log_dir, "scalars.jsonl")
            self._fd = open(self.path, "a")

    def add(self, step, **scalars):
        if self._fd is None:
            return
        rec = {"step": int(step), "time": round(time.time() - self._t0, 4)}
        rec.update({k: float(v) for k, v in scalars.items()})
        self._fd.write(json.dumps(rec) + "\n")
        self._fd.flush()

    def close(self):
        if self._fd is not None:
            self._fd.close()
            self._fd = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
