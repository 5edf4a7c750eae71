"""CPU, world_size 2 over gloo: the frame-sharded N>1 path (no data-path collective).  Each rank
parses its own slice of the stream's IDR pictures; the union is every picture exactly once, in
order, and the control-plane reductions used by bench.py behave."""
import hashlib
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, path, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from minivideo_amd import dist as mvd
    from tests.util import Stream
    mvd.init("gloo", rank, world)
    data = np.fromfile(path, np.uint8)
    with Stream(data) as s:
        n = s.idr_count
        lo, hi = mvd.shard(n, rank, world)
        sums = {}
        for k in range(lo, hi):
            rc, packed = s.packed(k)
            assert rc == 1
            sums[k] = hashlib.md5(packed.tobytes()).hexdigest()
    allsums = mvd.gather_objects(sums)
    t = mvd.max_over_ranks(1.0 + rank)
    import torch.distributed as dist
    dist.barrier()
    if rank == 0:
        q.put((allsums, t, n))
    dist.destroy_process_group()


def test_two_rank_frame_sharding(tmp_path):
    from minivideo_amd import gen
    stream, packed = gen.make_stream(9, 6, 7, seed=77, profile="high")
    path = str(tmp_path / "s.264")
    stream.tofile(path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 300)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    allsums, t, n = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert n == 7 and t == 2.0
    merged = {}
    for d in allsums:
        assert not (set(d) & set(merged))          # no picture decoded twice
        merged.update(d)
    assert sorted(merged) == list(range(7))        # every picture exactly once
    for k in range(7):
        assert merged[k] == hashlib.md5(packed[k].tobytes()).hexdigest()


def test_shard_is_balanced_partition():
    from minivideo_amd.dist import shard
    for n in (0, 1, 7, 512, 513):
        for w in (1, 2, 3, 8):
            cuts = [shard(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_bench_refuses_a_rank_count_that_does_not_match():
    """`bench.py --gpus 8` must never run on one GPU and print n_gpus: 1.  Started bare it launches the ranks
    itself (and fails when the devices are not there); under a launcher, --gpus and WORLD_SIZE must agree."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "HIP device" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, env=dict(env, WORLD_SIZE="4", RANK="0"))
    assert r.returncode != 0 and "does not match WORLD_SIZE=4" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_strong_scaling_shares_cover_the_batch():
    """--strong P: the ranks' shares are contiguous, disjoint and add up to P (config 5: 512 pictures over 8 GPUs)."""
    from minivideo_amd.dist import shard
    for P, world in ((512, 8), (512, 3), (7, 8), (2048, 5)):
        cuts = [shard(P, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == P
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
