"""Segmented HIP-graph runner with RCCL all-reduces on a one-rank group (rehearsal of the multi-GPU path)."""
import os, sys, faulthandler
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
from test_gpu_dcgan import make_opts
from gan_amd.dcgan import DCGAN
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("joined", flush=True)
opts = make_opts()
rng = np.random.default_rng(0)
batches = [(torch.from_numpy(rng.uniform(-1, 1, (8, 16, 16, 3)).astype(np.float32)).cuda(),
            torch.from_numpy(rng.uniform(-1, 1, (8, 10)).astype(np.float32)).cuda()) for _ in range(4)]
eager = DCGAN(None, opts, load_data=False)
seg = DCGAN(None, opts, load_data=False, process_group=dist.group.WORLD)
seg.store.load_state_dict(eager.store.state_dict())
run = seg.iteration_runner(*batches[0], segmented=True)
print("captured", flush=True)
for i, (im, z) in enumerate(batches * 3):
    e = eager.train_iteration(im, z)
    g = run(im, z)
    torch.cuda.synchronize()
    assert all(float(e[k]) == float(g[k]) for k in e), (i, e, g)
    print("iteration", i, "ok", flush=True)
dist.destroy_process_group()
print("done", flush=True)
