"""CPU, world_size 2, gloo: the multi-GPU path (frame sharding + the single all_gather of result records)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, n_frames, ret):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sncal_amd
    from sncal_amd.dist import shard_range, pack_records, gather_records
    start, stop = shard_range(n_frames, rank, world)
    counts = [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]
    # every rank "computes" records for its own frames only: frame id encoded in the payload
    ids = torch.arange(start, stop)
    kpts = ids.float()[:, None, None].expand(-1, 57, 3).contiguous() + 0.25
    rec = (ids % 251).to(torch.uint8)[:, None].expand(-1, 136).contiguous()
    allrec = gather_records(pack_records(kpts, rec), counts)
    ok = allrec.shape == (n_frames, 57 * 3 * 4 + 136)
    k_all = allrec[:, :684].contiguous().view(torch.float32).reshape(n_frames, 57, 3)
    ok = ok and torch.equal(k_all[:, 0, 0], torch.arange(n_frames).float() + 0.25)
    ok = ok and torch.equal(allrec[:, 684], (torch.arange(n_frames) % 251).to(torch.uint8))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _worker_log(rank, world, port, n_frames, batch, ret):
    """The deferred form the pipeline uses: every rank logs its frames batch after batch (the last batch short, the shards ragged),
    ONE collective after the last batch (dist.RecordLog.gather)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sncal_amd
    from sncal_amd.dist import shard_range, pack_records, RecordLog
    start, stop = shard_range(n_frames, rank, world)
    counts = [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]
    log = RecordLog()
    calls = {'n': 0}
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        calls['n'] += 1
        return real(*a, **k)
    dist.all_gather_into_tensor = counting
    for a in range(start, stop, batch):
        ids = torch.arange(a, min(a + batch, stop))
        kpts = ids.float()[:, None, None].expand(-1, 57, 3).contiguous() + 0.25
        rec = (ids % 251).to(torch.uint8)[:, None].expand(-1, 136).contiguous()
        extra = (ids % 13).to(torch.uint8)[:, None].expand(-1, 136).contiguous()          # a second record per frame (C4 / noisy runs)
        log.add(pack_records(kpts, rec, extra))
    ok = len(log) == stop - start and calls['n'] == 0                                    # nothing exchanged while batches run
    allrec = log.gather(counts)
    ok = ok and calls['n'] == 1 and len(log) == 0                                        # ONE collective for the whole run
    ok = ok and allrec.shape == (n_frames, 57 * 3 * 4 + 2 * 136)
    k_all = allrec[:, :684].contiguous().view(torch.float32).reshape(n_frames, 57, 3)
    ok = ok and torch.equal(k_all[:, 5, 1], torch.arange(n_frames).float() + 0.25)
    ok = ok and torch.equal(allrec[:, 684 + 7], (torch.arange(n_frames) % 251).to(torch.uint8))
    ok = ok and torch.equal(allrec[:, 684 + 136], (torch.arange(n_frames) % 13).to(torch.uint8))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _run_log(n_frames, batch, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_log, args=(world, port, n_frames, batch, ret), nprocs=world, join=True)
    return dict(ret)


def test_deferred_gather_world2_even_batches():
    assert _run_log(64, 8) == {0: True, 1: True}


def test_deferred_gather_world2_ragged_shards_and_short_last_batch():
    assert _run_log(37, 8) == {0: True, 1: True}          # rank 0: 19 frames = 8 + 8 + 3, rank 1: 18 = 8 + 8 + 2


def test_record_log_without_process_group_returns_local_records():
    import sncal_amd
    from sncal_amd.dist import RecordLog
    log = RecordLog()
    a, b = torch.arange(6, dtype=torch.uint8).reshape(2, 3), torch.arange(6, 15, dtype=torch.uint8).reshape(3, 3)
    log.add(a)
    log.add(b)
    assert torch.equal(log.gather(), torch.cat([a, b]))
    assert len(log) == 0


def _run(n_frames, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, n_frames, ret), nprocs=world, join=True)
    return dict(ret)


def test_shard_ranges_cover_all_frames():
    import sncal_amd
    from sncal_amd.dist import shard_range
    for n in (0, 1, 7, 64, 512, 1025):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_gather_world2_even():
    assert _run(128) == {0: True, 1: True}


def test_gather_world2_ragged():
    assert _run(37) == {0: True, 1: True}


def _bench_cmd(*flags):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py'), *flags], env=env, capture_output=True, text=True, timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun must start 2 ranks itself (the driver runs exactly that command line);
    --dry-dist takes the same launcher and the path's single collective on gloo, without GPU work."""
    import json
    p = _bench_cmd('--gpus', '2', '--dry-dist')
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['gathered_ranks'] == [0, 1] and out['records'] == 8


def test_bench_refuses_more_gpus_than_visible():
    """No GPU in the CPU container: asking for 2 must fail loudly instead of reporting a 1-GPU run as n_gpus 2."""
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    p = _bench_cmd('--gpus', '2', '--steps', '1', '--warmup', '1')
    assert p.returncode != 0 and 'GPU(s) visible' in (p.stderr + p.stdout)
