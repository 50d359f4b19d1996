"""
Batch data parallelism, one process per GPU (replaces the single-process ``nn.DataParallel`` shim of
reference timbre_trap/utils/experiments.py:67-78 and its use at experiments/train.py:166-168).

Clips are independent; the only exchange per step is the gradient: ONE all-reduce (RCCL over xGMI
when the backend is 'nccl', gloo on CPU for tests) of the flat fp32 gradient buffer (2.46 MB at
model_complexity 2), averaged over ranks.  Per-rank batch sizes must be equal for the mean over the
global batch (reference computes the loss on the gathered batch) to equal the mean of rank means.
"""

import os

import torch
import torch.distributed as dist


def force_dist():
    """TTRAP_FORCE_DIST=1 keeps the N > 1 code path alive at world size 1 (process group, broadcast of the parameters, the
    asynchronous gradient all-reduce and its stream wait): how RCCL -- its coexistence with libttrap_hip.so and torch's bundled HIP
    runtime in one process, communicator setup under HSA_ENABLE_IPC_MODE_LEGACY=0, ``work.wait()`` as a stream wait -- is exercised
    on a 1-GPU box (tests/test_gpu_distributed.py)."""
    return os.environ.get('TTRAP_FORCE_DIST') == '1'


def dist_active():
    """True where the data-parallel exchange runs: an initialised process group of more than one rank, or of ONE rank under
    TTRAP_FORCE_DIST=1."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_dist())


def init_process_group_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for one process (unless TTRAP_FORCE_DIST=1
    and the launcher's rendezvous variables are present)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and not (force_dist() and 'RANK' in os.environ and 'MASTER_PORT' in os.environ):
        return 0, 1, 0
    rank = int(os.environ['RANK'])
    local_rank = int(os.environ.get('LOCAL_RANK', rank))
    if backend is None:
        # TTRAP_DIST_BACKEND=gloo lets several ranks share one GPU (functional test of the N > 1 path on a 1-GPU box)
        backend = os.environ.get('TTRAP_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def allreduce_gradients(flat_grad, world_size=None, async_op=False):
    """Average one flat gradient buffer over all ranks with a single collective."""
    if not dist_active():
        return None
    world_size = world_size or dist.get_world_size()
    flat_grad.div_(world_size)
    return dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, async_op=async_op)


class GradientSync:
    """
    The data-parallel exchange of one train step, split in two so that independent work can be issued in between:

        sync.start(flat_grad)      pre-divide by the world size, launch ONE asynchronous all-reduce(SUM)
        ...                        e.g. the next batch's CQT on the compute stream (no dependence on the weights)
        sync.finish()              the CURRENT stream waits for the collective (the host does not block with RCCL)

    With the 'nccl' backend (RCCL) the collective runs on the process group's own HIP stream, ordered after the kernels
    already enqueued on the current stream; ``finish`` makes the current stream wait on its completion event, so
    ``opt.step()`` issued afterwards sees the averaged gradient.  With gloo (CPU tests, 1-GPU functional tests) ``finish``
    blocks the host instead; the arithmetic is identical.
    """

    def __init__(self, world_size=None):
        self.world = world_size
        self.work = None

    def start(self, grads):
        """``grads``: a FusedAdamW (preferred: its detached gradient views are re-attached first, so that a gradient autograd
        deposited into a fresh tensor after ``model.zero_grad()`` is in the flat buffer BEFORE it is averaged) or the flat
        gradient buffer itself (the caller then vouches that every ``p.grad`` still is its view)."""
        if self.work is not None:
            raise RuntimeError('GradientSync.start called twice without finish')
        flat_grad = grads.sync_views() if hasattr(grads, 'sync_views') else grads
        self.work = allreduce_gradients(flat_grad, self.world, async_op=True)
        return self.work

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None


def broadcast_parameters(flat_param, src=0):
    if dist_active():
        dist.broadcast(flat_param, src=src)


class DataParallel(torch.nn.DataParallel):
    """
    Attribute-forwarding wrapper with the reference shim's surface (``model.sliCQ`` etc. resolve on the
    wrapped module, ``.module`` unwraps).  It IS a ``torch.nn.DataParallel`` -- reference experiments/train.py:506-508
    and :525-527 unwrap / re-wrap around ``torch.save`` with ``isinstance(model, torch.nn.DataParallel)`` -- but it never
    replicates: it holds ONE replica, this process's (``device_ids`` is kept empty, which is stock DataParallel's own
    "run the module directly" case), and the gradients are averaged across processes when ``sync_gradients`` is called
    (or by FusedAdamW users through ``allreduce_gradients`` on the flat buffer).
    """

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        torch.nn.Module.__init__(self)          # not DataParallel.__init__: no device bookkeeping, no replication
        if device_ids is not None and len(device_ids) > 1:
            import warnings
            warnings.warn('timbre_trap DataParallel holds ONE replica per process: device_ids=%s is ignored and this process uses '
                          'one GPU only.  For %d GPUs launch one process per GPU (torchrun --nproc-per-node %d) -- see INTEGRATION.md.'
                          % (list(device_ids), len(device_ids), len(device_ids)), RuntimeWarning, stacklevel=2)
        self.module = module
        self.device_ids = []
        self.output_device = None
        self.src_device_obj = None
        self.dim = dim

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.module, name)

    def sync_gradients(self):
        if not dist_active():
            return
        grads = [p.grad for p in self.module.parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        allreduce_gradients(flat)
        o = 0
        for g in grads:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
