"""CPU, world_size 2, gloo: the exchange step of the N > 1 path (flat head-gradient bucket all-reduce), sharding, and the
self-starting launcher of `bench.py --gpus N` (bf/training/helpers.py:129-142)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from single_shot_detection_amd import launch
from single_shot_detection_amd.distributed import GradBucket, grad_sink, shard_batch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    conv_a = torch.nn.Conv2d(8, 12, 3, padding=1)
    conv_b = torch.nn.Conv2d(8, 4, 3, padding=1)
    for m in (conv_a, conv_b):   # channels_last parameter memory like the head convs
        m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    params = list(conv_a.parameters()) + list(conv_b.parameters())
    g = torch.Generator().manual_seed(100 + rank)
    for p in params:
        grad = torch.randn(p.shape, generator=g)
        p.grad = grad.contiguous(memory_format=torch.channels_last) if grad.dim() == 4 else grad
    local = [p.grad.clone() for p in params]
    # two buckets, both rings started before either is finished (the bench starts the head bucket, runs the rest of the
    # backward, starts the second bucket, then finishes both)
    buckets = [GradBucket(list(conv_a.parameters())), GradBucket(list(conv_b.parameters()))]
    for b in buckets:
        b.start_()
    for b in buckets:
        b.finish_()
    gathered = [None] * world
    dist.all_gather_object(gathered, [x.numpy() for x in local])
    for i, p in enumerate(params):
        want = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        assert torch.allclose(p.grad, want, atol=1e-6), (rank, i)
        assert p.grad.stride() == local[i].stride()
    assert all(b.copied_last == len(b.params) for b in buckets)       # gradients made elsewhere: copied into their slots once ...
    assert all(p.grad.data_ptr() == p._ssdk_grad_view.data_ptr() for p in params)   # ... and .grad now IS the slot
    # zero-copy step: a producer that writes its result into the slot (what the heads / conv backward do) -> nothing is moved
    for p in params:
        p.grad = None
    local2 = []
    for p in params:
        sink = grad_sink(p)
        assert sink is not None and sink.stride() == p.stride()
        sink.copy_(torch.randn(p.shape, generator=g))
        local2.append(sink.clone())
        p.grad = sink
        assert grad_sink(p) is None        # a parameter that already holds a gradient must be accumulated into, not overwritten
    for b in buckets:
        b.start_()
    for b in buckets:
        b.finish_()
    assert all(b.copied_last == 0 for b in buckets)
    dist.all_gather_object(gathered, [x.numpy() for x in local2])
    for i, p in enumerate(params):
        want = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        assert torch.allclose(p.grad, want, atol=1e-6), (rank, i)
    np.save(os.path.join(out_dir, f'ok{rank}.npy'), np.array([1]))
    dist.destroy_process_group()


def test_grad_bucket_allreduce_gloo_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ok{r}.npy') for r in range(world))


def test_shard_batch_covers_everything_once():
    items = list(range(13))
    for world in (1, 2, 4, 8):
        parts = [shard_batch(items, r, world) for r in range(world)]
        assert sum(parts, []) == items


def test_bucket_is_noop_without_process_group():
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    GradBucket([p]).allreduce_()
    assert torch.equal(p.grad, torch.full((3,), 2.0))


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no torchrun around it: the parent spawns the two ranks, they rendezvous on 127.0.0.1 and rank 0's
    JSON line comes back on the parent's stdout (--rendezvous-only: the hot path itself needs a GPU)."""
    env = dict(os.environ, SSDK_BENCH_BACKEND='gloo')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--rendezvous-only'], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['ranks'] == 2 and out['rank_sum'] == 1.0


def test_launcher_reports_a_failed_rank_and_stops_the_others(tmp_path):
    script = tmp_path / 'child.py'
    script.write_text('import os, sys, time\n'
                      'rank = int(os.environ["RANK"])\n'
                      'assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1"\n'
                      'if rank == 1:\n    sys.exit(7)\n'
                      'time.sleep(120)\n')
    import time
    t0 = time.time()
    code = launch.launch(3, [sys.executable, str(script)])
    assert code == 7 and time.time() - t0 < 60


def test_launcher_kills_a_rank_that_ignores_sigterm(tmp_path):
    """A survivor that does not die on SIGTERM (stuck in a collective, or with a handler installed) is killed once the grace period has
    passed: the launcher returns the failed rank's code instead of waiting for ever."""
    script = tmp_path / 'child.py'
    script.write_text('import os, signal, sys, time\n'
                      'signal.signal(signal.SIGTERM, signal.SIG_IGN)\n'
                      'if int(os.environ["RANK"]) == 1:\n    time.sleep(1)\n    sys.exit(5)\n'
                      'time.sleep(300)\n')
    import time
    t0 = time.time()
    code = launch.launch(2, [sys.executable, str(script)], grace=2.0)
    assert code == 5 and time.time() - t0 < 60


def test_launcher_retries_when_the_rendezvous_port_was_taken(tmp_path):
    """The port is released before the ranks bind it; a rank that reports EADDRINUSE makes the launcher start over on another port."""
    marker = tmp_path / 'first_attempt_done'
    script = tmp_path / 'child.py'
    script.write_text('import os, sys\n'
                      f'marker = {str(marker)!r}\n'
                      'if int(os.environ["RANK"]) == 0 and not os.path.exists(marker):\n'
                      '    open(marker, "w").write(os.environ["MASTER_PORT"])\n'
                      '    sys.stderr.write("RuntimeError: The server socket has failed to listen: EADDRINUSE (Address already in use)\\n")\n'
                      '    sys.exit(1)\n'
                      'if int(os.environ["RANK"]) == 0:\n'
                      '    open(marker + ".second", "w").write(os.environ["MASTER_PORT"])\n'
                      'sys.exit(0)\n')
    assert launch.launch(2, [sys.executable, str(script)], grace=2.0) == 0
    assert marker.exists() and (tmp_path / 'first_attempt_done.second').exists()   # (two attempts ran; the second may reuse the freed port)


def test_launcher_does_not_retry_an_unrelated_address_in_use(tmp_path):
    """'Address already in use' from some OTHER socket of the job (not the rendezvous listen / bind) is an ordinary failure: one attempt."""
    count = tmp_path / 'attempts'
    script = tmp_path / 'child.py'
    script.write_text('import os, sys\n'
                      f'count = {str(count)!r}\n'
                      'if int(os.environ["RANK"]) == 0:\n'
                      '    open(count, "a").write("x")\n'
                      '    sys.stderr.write("OSError: [Errno 98] Address already in use (metrics exporter)\\n")\n'
                      '    sys.exit(3)\n'
                      'sys.exit(0)\n')
    assert launch.launch(2, [sys.executable, str(script)], grace=2.0) == 3
    assert count.read_text() == 'x'


def _sums_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from single_shot_detection_amd.ops import allreduce_sums_
    # the packed buffer of a tower layer: per level (2 C + 1) doubles = (sum, sum of squares, rows)
    C, levels = 8, 3
    buf = torch.zeros((levels * (2 * C + 1),), dtype=torch.float64)
    for lv in range(levels):
        off = lv * (2 * C + 1)
        buf[off:off + 2 * C] = torch.arange(2 * C, dtype=torch.float64) + 100 * lv + 1000 * rank
        buf[off + 2 * C] = 10 * (lv + 1) + rank          # rows of this rank at this level
    allreduce_sums_(buf)
    for lv in range(levels):
        off = lv * (2 * C + 1)
        want = 2 * (torch.arange(2 * C, dtype=torch.float64) + 100 * lv) + 1000
        assert torch.equal(buf[off:off + 2 * C], want)
        assert buf[off + 2 * C].item() == 20 * (lv + 1) + 1
    np.save(os.path.join(out_dir, f'sums{rank}.npy'), np.array([1]))
    dist.destroy_process_group()


def test_sync_batchnorm_packed_sums_allreduce_gloo_world2(tmp_path):
    """The exchange step of synchronised BatchNorm: one all-reduce(sum) of the packed per-level (sum, sum^2, rows) buffers."""
    mp.spawn(_sums_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f'sums{r}.npy') for r in range(2))


class _ToyPredictor(torch.nn.Module):
    """features -> heads, like detection.detector.Predictor: the heads' gradients are complete before the features' are."""

    def __init__(self):
        super().__init__()
        self.features = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 8, 3, padding=1))
        self.heads = torch.nn.ModuleList([torch.nn.Conv2d(8, 6, 3, padding=1), torch.nn.Conv2d(8, 4, 3, padding=1)])
        self.unused = torch.nn.Linear(4, 4)   # takes no part in forward(): its gradient is zero on every rank

    def forward(self, x):
        f = self.features(x)
        return torch.cat([h(f).flatten(1) for h in self.heads], 1)


def _bdp_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from single_shot_detection_amd.distributed import BucketedDataParallel
    torch.manual_seed(100 + rank)          # different initial weights per rank: the wrapper broadcasts rank 0's
    net = BucketedDataParallel(_ToyPredictor())
    w0 = [p.detach().clone() for p in net.parameters()]
    gathered = [None] * world
    dist.all_gather_object(gathered, [w.numpy() for w in w0])
    assert all(np.array_equal(a, b) for a, b in zip(gathered[0], gathered[rank]))
    assert list(net.state_dict())[0].startswith('module.')
    x = torch.randn((2, 3, 9, 9), generator=torch.Generator().manual_seed(7 + rank))   # this rank's own images
    params = [p for p in net.parameters()]
    # reference: every rank's local gradients, no exchange, averaged by hand
    with net.no_sync():
        net(x).square().sum().backward()
    local = [None if p.grad is None else p.grad.clone() for p in params]
    assert net.start_order == []
    dist.all_gather_object(gathered, [None if g is None else g.numpy() for g in local])
    for p in params:
        p.grad = None
    # the exchange: two rings, the heads' started from its last parameter's hook -- before the features' gradients existed
    net(x).square().sum().backward()
    assert net.start_order == [0, 1] and 0 in net.started_early, (net.start_order, net.started_early)
    for i, p in enumerate(params):
        if gathered[0][i] is None:
            assert float(p.grad.abs().max()) == 0.0      # the unused layer: a zero gradient, not a hang and not None
            continue
        want = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
        assert torch.allclose(p.grad, want, atol=1e-6, rtol=1e-5), (rank, i)
        assert p.grad.data_ptr() == p._ssdk_grad_view.data_ptr()          # .grad IS the bucket slot
    # a second step: counters were reset, gradients accumulate into the slots' values only after zero_grad
    for p in params:
        p.grad = None
    net(x).square().sum().backward()
    assert net.start_order == [0, 1]
    # a backward pass that RAISES behind the heads (their ring has started, the end-of-backward callback never runs) must not leave the
    # wrapper deaf: the next pass resets the counters, finishes the orphaned ring, and exchanges as before
    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')
    for p in params:
        p.grad = None
    hook = net.module.features.register_forward_hook(lambda m, i, o: _Boom.apply(o))
    try:
        net(x).square().sum().backward()
        raise AssertionError('the backward pass should have raised')
    except RuntimeError as e:
        assert 'boom' in str(e)
    hook.remove()
    for p in params:
        p.grad = None
    net(x).square().sum().backward()
    assert net.start_order == [0, 1] and 0 in net.started_early, (net.start_order, net.started_early)
    for i, p in enumerate(params):
        if gathered[0][i] is not None:
            want = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
            assert torch.allclose(p.grad, want, atol=1e-6, rtol=1e-5), (rank, i)
    np.save(os.path.join(out_dir, f'bdp{rank}.npy'), np.array([1]))
    dist.destroy_process_group()


def test_bucketed_data_parallel_gloo_world2(tmp_path):
    """distributed.BucketedDataParallel (what detection.init(distributed=True) wraps the predictor in, and what bench.py --gpus N runs):
    broadcast of rank 0's weights, hook-driven start of the heads' ring, averaged gradients equal to the hand-made average of the ranks'
    local ones, zero gradients for parameters that took no part, no_sync()."""
    world = 2
    mp.spawn(_bdp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'bdp{r}.npy') for r in range(world))


def _bdp_one_rank_raises_worker(rank, world, port, out_dir):
    """Only rank 1's backward pass raises (behind the heads: its first ring has started, its second has not); rank 0 finishes its step.
    Rank 1's next pass first issues the collective it owes, so the ranks stay paired: no hang, and the following step averages correctly."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=__import__('datetime').timedelta(seconds=60))
    from single_shot_detection_amd.distributed import BucketedDataParallel
    torch.manual_seed(5)
    net = BucketedDataParallel(_ToyPredictor())
    params = [p for p in net.parameters()]
    x = torch.randn((2, 3, 9, 9), generator=torch.Generator().manual_seed(17 + rank))
    with net.no_sync():
        net(x).square().sum().backward()
    gathered = [None] * world
    dist.all_gather_object(gathered, [None if p.grad is None else p.grad.numpy().copy() for p in params])
    for p in params:
        p.grad = None

    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.view_as(t)

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError('boom')
    if rank == 1:
        hook = net.module.features.register_forward_hook(lambda m, i, o: _Boom.apply(o))
        try:
            net(x).square().sum().backward()
            raise AssertionError('the backward pass should have raised')
        except RuntimeError as e:
            assert 'boom' in str(e)
        hook.remove()
    else:
        net(x).square().sum().backward()      # waits in bucket 1's all-reduce until rank 1 issues the one it owes
    for p in params:
        p.grad = None
    net(x).square().sum().backward()          # rank 1: abort_step_() first (the missing collective), then an ordinary step
    assert net.recovered_steps == (1 if rank == 1 else 0)
    assert net.start_order == [0, 1] and 0 in net.started_early
    for i, p in enumerate(params):
        if gathered[0][i] is not None:
            want = sum(torch.from_numpy(gathered[r][i]) for r in range(world)) / world
            assert torch.allclose(p.grad, want, atol=1e-6, rtol=1e-5), (rank, i)
    np.save(os.path.join(out_dir, f'one{rank}.npy'), np.array([1]))
    dist.destroy_process_group()


def test_bucketed_data_parallel_one_rank_raises(tmp_path):
    """Advisor (round 4): a backward pass that raises on ONE rank must not leave the ranks' collectives unpaired."""
    world = 2
    mp.spawn(_bdp_one_rank_raises_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'one{r}.npy') for r in range(world))


def _bdp_order_worker(rank, world, port, out_dir):
    """Buckets whose gradients complete in another order than their indices: the rings still start in index order on every rank (and none
    early past an unfinished predecessor); timing fields are filled when asked for."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from single_shot_detection_amd.distributed import BucketedDataParallel
    torch.manual_seed(9)
    toy = _ToyPredictor()
    groups = [list(toy.features.parameters()), list(toy.heads.parameters()), list(toy.unused.parameters())]   # the features complete LAST
    net = BucketedDataParallel(toy, groups=groups)
    net.collect_timing = True
    x = torch.randn((2, 3, 9, 9), generator=torch.Generator().manual_seed(27 + rank))
    net(x).square().sum().backward()
    assert net.start_order == [0, 1, 2], net.start_order
    assert 2 not in net.started_early                      # (the heads were complete first, but started only behind bucket 0, in the hook of ITS last gradient)
    t = net.exchange_timing()
    assert len(t) == 3 and all(e['exposed_ms'] is not None and e['exposed_ms'] >= 0.0 for e in t) and t[0]['bytes'] == net.buckets[0].nbytes
    np.save(os.path.join(out_dir, f'ord{rank}.npy'), np.array([1]))
    dist.destroy_process_group()


def test_bucketed_data_parallel_index_order_and_timing(tmp_path):
    world = 2
    mp.spawn(_bdp_order_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f'ord{r}.npy') for r in range(world))
