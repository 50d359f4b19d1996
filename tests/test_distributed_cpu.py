"""
CPU-only, world_size 2 over gloo: the data-parallel exchange of the train step (one all-reduce of the flat gradient
buffer, averaged over ranks) and the DataParallel shim.  The GPU path uses the same calls with backend 'nccl' (= RCCL).
"""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from timbre_trap.utils import DataParallel, GradientSync, allreduce_gradients, init_process_group_from_env
    from timbre_trap.utils.distributed import broadcast_parameters
    r, w, _ = init_process_group_from_env(backend='gloo')
    assert (r, w) == (rank, world)
    # 1. one collective over a flat gradient buffer: mean over ranks
    g = torch.arange(614490, dtype=torch.float32) * (rank + 1)
    allreduce_gradients(g, world)
    ok1 = torch.allclose(g, torch.arange(614490, dtype=torch.float32) * 1.5)
    # 1b. the split form used by the train step: start (async) -> independent work -> finish; same values, bit for bit
    g2 = torch.arange(614490, dtype=torch.float32) * (rank + 1)
    sync = GradientSync(world)
    sync.start(g2)
    other = torch.ones(1000).cumsum(0)                                # independent work issued while the collective is in flight
    sync.finish()
    ok1 = ok1 and torch.equal(g2, g) and float(other[-1]) == 1000.0
    try:
        sync.start(g2); sync.start(g2)
        ok1 = False
    except RuntimeError:
        sync.finish()
    # 2. parameters start identical on every rank
    p = torch.full((10,), float(rank))
    broadcast_parameters(p)
    ok2 = bool((p == 0).all())
    # 3. the attribute-forwarding shim: data-parallel SGD on a toy module equals the single-process step on the full batch
    torch.manual_seed(0)
    lin = torch.nn.Linear(4, 3)
    model = DataParallel(lin)
    assert model.in_features == 4 and model.module is lin           # attribute passthrough (reference shim surface)
    # reference train.py:506-508 / :525-527 unwrap and re-wrap with exactly this check
    assert isinstance(model, torch.nn.DataParallel) and not isinstance(model.module, torch.nn.DataParallel)
    full_x = torch.arange(32, dtype=torch.float32).view(8, 4) / 10
    x = full_x[rank * 4:(rank + 1) * 4]                              # equal per-rank batches
    model(x).pow(2).mean().backward()
    model.sync_gradients()
    ref = torch.nn.Linear(4, 3)
    ref.load_state_dict(lin.state_dict())
    ref(full_x).pow(2).mean().backward()
    ok3 = torch.allclose(lin.weight.grad, ref.weight.grad, atol=1e-6) and torch.allclose(lin.bias.grad, ref.bias.grad, atol=1e-6)
    out.put((rank, ok1, ok2, ok3))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2():
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results == [(0, True, True, True), (1, True, True, True)]


def test_dataparallel_warns_about_ignored_device_ids():
    """Unmodified multi-GPU train.py (`DataParallel(model, device_ids=gpu_ids)`, reference experiments/train.py:166-168) must not
    silently run on one GPU."""
    import warnings
    from timbre_trap.utils import DataParallel
    lin = torch.nn.Linear(2, 2)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        DataParallel(lin, device_ids=[0, 1, 2, 3])
        assert any('one process per GPU' in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        DataParallel(lin, device_ids=[0])
        DataParallel(lin)
        assert not w


def test_gradient_sync_reattaches_through_the_optimizer():
    """GradientSync.start(opt) asks the optimizer for its flat buffer AFTER re-attaching detached gradient views."""
    from timbre_trap.utils import GradientSync

    class FakeOpt:
        calls = 0

        def sync_views(self):
            FakeOpt.calls += 1
            return torch.ones(3)
    os.environ.pop('WORLD_SIZE', None)
    s = GradientSync(1)
    s.start(FakeOpt())
    s.finish()
    assert FakeOpt.calls == 1


def test_single_process_is_a_noop():
    from timbre_trap.utils import allreduce_gradients, init_process_group_from_env
    os.environ.pop('WORLD_SIZE', None)
    assert init_process_group_from_env() == (0, 1, 0)
    g = torch.ones(5)
    assert allreduce_gradients(g) is None and bool((g == 1).all())


def test_bare_bench_gpus_n_without_devices_refuses():
    """`python bench.py --gpus 2` with no WORLD_SIZE and fewer than 2 visible GPUs must exit non-zero before any launch
    (it used to run -- and report -- a 1-GPU benchmark)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'TTRAP_DIST_BACKEND')}
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and 'refusing' in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith('{')]


def test_bench_world_size_mismatch_is_an_error():
    """WORLD_SIZE set by a launcher but different from --gpus: error, not a measurement of another configuration."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and 'WORLD_SIZE' in (out.stderr + out.stdout)
