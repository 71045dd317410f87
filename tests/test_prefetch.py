"""Host input pipeline (gan_amd/prefetch.py): order, random-sequence equivalence with the inline loop,
error propagation and shutdown.  CPU only."""
import threading
import time

import numpy as np
import pytest
import torch

from gan_amd.prefetch import BatchPrefetcher, to_device


def test_batches_arrive_in_order_and_equal_the_inline_loop():
    def inline(n):
        rng = np.random.RandomState(3)
        return [(rng.uniform(-1, 1, (4, 5)).astype(np.float32), rng.randint(0, 10, 4)) for _ in range(n)]

    rng = np.random.RandomState(3)
    make = lambda i: (rng.uniform(-1, 1, (4, 5)).astype(np.float32), rng.randint(0, 10, 4), "name%d" % i, None)
    got = list(BatchPrefetcher(make, range(7), depth=2, workers=2, pin=False))
    assert [i for i, _ in got] == list(range(7))
    for (i, (a, b, name, nothing)), (ea, eb) in zip(got, inline(7)):
        assert torch.is_tensor(a) and a.dtype == torch.float32 and torch.is_tensor(b)
        np.testing.assert_array_equal(a.numpy(), ea)
        np.testing.assert_array_equal(b.numpy(), eb)
        assert name == "name%d" % i and nothing is None


def test_pool_map_keeps_item_order_and_runs_on_workers():
    seen = set()

    def slow_square(v):
        seen.add(threading.current_thread().name)
        time.sleep(0.01 * (5 - v))          # later items finish first
        return v * v

    feed = BatchPrefetcher(lambda i: np.asarray(feed.pool_map(slow_square, range(5)), np.int64) + i, range(3),
                           depth=1, workers=4, pin=False)
    out = [b.numpy().tolist() for _, b in feed]
    assert out == [[0 + i, 1 + i, 4 + i, 9 + i, 16 + i] for i in range(3)]
    assert len(seen) > 1 and threading.current_thread().name not in seen


def test_producer_runs_ahead_but_not_beyond_depth():
    made = []
    feed = BatchPrefetcher(lambda i: made.append(i) or np.zeros(1, np.float32), range(20), depth=3, pin=False)
    it = iter(feed)
    next(it)
    time.sleep(0.3)
    # one handed over + `depth` queued + one finished and waiting for a slot
    assert 3 <= len(made) <= 5
    feed.close()
    n = len(made)
    time.sleep(0.2)
    assert len(made) == n and not feed._thread.is_alive()


def test_exception_in_make_surfaces_at_its_batch():
    def make(i):
        if i == 2:
            raise KeyError("bad file %d" % i)
        return np.full(2, i, np.float32)

    feed = BatchPrefetcher(make, range(5), depth=4, pin=False)
    got = []
    with pytest.raises(KeyError, match="bad file 2"):
        for i, b in feed:
            got.append(i)
    assert got == [0, 1]
    assert not feed._thread.is_alive()


def test_leaving_the_loop_early_stops_the_producer_and_second_pass_is_refused():
    feed = BatchPrefetcher(lambda i: np.zeros(3, np.float32), range(1000), depth=2, pin=False)
    for i, b in feed:
        if i == 3:
            break
    feed.close()
    assert not feed._thread.is_alive()
    with pytest.raises(RuntimeError):
        list(feed)
    with pytest.raises(ValueError):
        BatchPrefetcher(lambda i: None, range(2), depth=0)


def test_to_device_maps_over_nested_batches():
    batch = (torch.ones(2), [torch.zeros(1), "x"], None)
    out = to_device(batch, torch.device("cpu"))
    assert out[0].equal(batch[0]) and out[1][1] == "x" and out[2] is None
