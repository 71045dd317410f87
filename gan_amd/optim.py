"""tf.train.AdamOptimizer(lr, beta1).minimize(loss, var_list) over one flat arena
(dcgan.py:154-157, text2image.py:231-232, cyclegan.py:123-130).

The step count lives in device memory and the bias-corrected learning rate is computed on the
device, so a captured HIP graph of the update replays with the right lr_t at every step.
Under data parallelism the group's flat gradient arena is all-reduced (sum) in one collective
and the 1/world_size average is folded into the Adam kernel's gradient scale.
"""
import torch

from . import _lib
from ._lib import check, lib, ptr, stream


class AdamOptimizer(object):
    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = float(learning_rate), float(beta1), float(beta2), float(epsilon)
        self.group = None
        self.step_dev = None
        self.lr_scratch = None
        self._early = None      # the early part's exchange in flight (a work handle, or True once done)

    def attach(self, group):
        """Bind to a packed variable group (the var_list)."""
        self.group = group
        dev = group.values.device
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.lr_scratch = torch.zeros(1, dtype=torch.float32, device=dev)
        return self

    @property
    def steps_taken(self):
        return int(self.step_dev.item())

    def begin(self):
        self.group.begin_update()

    def finish_grads(self):
        """End of the backward pass: gradient slots nothing wrote to are zero-filled."""
        self.group.end_update()

    def all_reduce(self, world_size=1, process_group=None):
        """Sum the group's flat gradient arena over the ranks (one collective per update).  Inside a
        segmented graph capture this is a cut: the collective runs eagerly between two graphs."""
        if world_size > 1 or process_group is not None:
            from .graph import SegmentedIteration
            SegmentedIteration.cut(lambda: self._all_reduce_now(process_group))

    def all_reduce_early(self, world_size=1, process_group=None):
        """At the boundary of a staged backward pass (ops.backward): start summing the finished part
        of the gradient arena over the ranks while the rest of the pass still runs.  The exchange
        goes on the collective library's own stream; all_reduce() later sums the remaining part and
        joins both.  A cut of a segmented graph capture, like all_reduce."""
        if (world_size > 1 or process_group is not None) and self.group.early_offset is not None:
            self.group.finish_early()
            from .graph import SegmentedIteration
            SegmentedIteration.cut(lambda: self._all_reduce_early_now(process_group))

    def _all_reduce_early_now(self, process_group):
        import torch.distributed as dist
        part = self.group.early_grads()
        if part.is_cuda and dist.get_backend(process_group) == "gloo":
            host = part.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=process_group)
            part.copy_(host)
            self._early = True
        else:
            self._early = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=process_group, async_op=True)

    def _all_reduce_now(self, process_group):
        import torch.distributed as dist
        early, self._early = self._early, None
        grads = self.group.grads if early is None else self.group.late_grads()
        try:
            self._sum(grads, process_group)
        finally:
            if early is not None and early is not True:
                early.wait()            # the current stream waits for the early part's exchange

    def _sum(self, grads, process_group):
        import torch.distributed as dist
        if grads.is_cuda and dist.get_backend(process_group) == "gloo":
            # gloo (CPU rigs and rehearsals with several ranks on one GPU) moves host memory:
            # stage the arena through the host explicitly instead of relying on gloo's own
            # device-tensor path, which was seen to stall behind a replayed HIP graph
            host = grads.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=process_group)
            grads.copy_(host)
        else:
            dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=process_group)

    def adam(self, world_size=1):
        """The fused TF-Adam step; the 1/world_size gradient average is applied inside the kernel."""
        g = self.group
        check(lib.gan_adam_tf_step(ptr(g.values), ptr(g.grads), ptr(g.m), ptr(g.v), g.numel, self.lr, self.beta1,
                                   self.beta2, self.eps, 1.0 / float(world_size), ptr(self.step_dev),
                                   ptr(self.lr_scratch), stream()))

    def apply(self, world_size=1, process_group=None):
        """Finish the update: zero-fill untouched gradient slots, all-reduce, fused Adam step."""
        self.finish_grads()
        self.all_reduce(world_size, process_group)
        self.adam(world_size)
