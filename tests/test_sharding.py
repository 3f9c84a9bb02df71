"""CPU tests of the N>1 path: shard partition properties and the diagnostic all-reduce over gloo
(world_size 2), the same code path RCCL serves on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cmx import sharding


@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 1000, 10**8, 10**8 + 17])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_shards_tile_the_range(n, world):
    prev = 0
    sizes = []
    for r in range(world):
        lo, hi = sharding.shard_bounds(n, r, world)
        assert lo == prev and lo <= hi <= n
        if r > 0 and lo < n:
            assert lo % sharding.ALIGN == 0       # 16-byte aligned columns on every rank
        sizes.append(hi - lo)
        prev = hi
    assert prev == n
    assert max(sizes) - min(sizes) <= 2 * sharding.ALIGN      # balanced to within one aligned block (+ ragged tail)


def test_bad_shard_request():
    with pytest.raises(ValueError):
        sharding.shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(99)
        full = torch.rand(3, n, dtype=torch.float64, generator=g)      # same columns on every rank
        lo, hi = sharding.shard_bounds(n, rank, world)
        local = full[:, lo:hi].sum(dim=1)                               # what cmx_column_sums_* yields per rank
        total = sharding.allreduce_sums(local.clone())
        q.put((rank, total.numpy(), full.sum(dim=1).numpy(), hi - lo))
    finally:
        dist.destroy_process_group()


def test_diagnostic_allreduce_gloo_world2():
    world, n = 2, 100_003
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sum(r[3] for r in res) == n
    for _, total, expect, _ in res:
        np.testing.assert_allclose(total, expect, rtol=1e-12)


def test_allreduce_is_noop_without_process_group():
    t = torch.tensor([1.0, 2.0], dtype=torch.float64)
    assert sharding.allreduce_sums(t) is t and t.tolist() == [1.0, 2.0]
