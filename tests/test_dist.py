"""Multi-GPU host logic on CPU: LPT chromosome sharding and the collective callbacks over
torch.distributed with the gloo backend (world_size 2)."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

from genrich_amd import synth
from genrich_amd.dist import lpt_partition, merge_peaks
from genrich_amd.lib import PEAK_DTYPE


def test_lpt_partition_hg38():
    lens = synth.HG38_LENS
    for world, bound in ((2, 2.0), (4, 3.98), (8, 7.72)):
        owner = lpt_partition(lens, world)
        assert sorted(set(owner)) == list(range(world))
        load = [sum(l for l, o in zip(lens, owner) if o == r) for r in range(world)]
        assert sum(load) == sum(lens)
        assert sum(lens) / max(load) >= bound - 0.02  # LPT speed-up bounds
    assert lpt_partition(lens, 1) == [0] * len(lens)


def test_merge_peaks_orders_by_chrom_then_start():
    a = np.array([(2, 10, 20, 1, 1, 1, 1), (0, 50, 60, 1, 1, 1, 1)], dtype=PEAK_DTYPE)
    b = np.array([(0, 5, 9, 1, 1, 1, 1), (1, 7, 9, 1, 1, 1, 1)], dtype=PEAK_DTYPE)
    m = merge_peaks([a, b])
    assert list(zip(m["chrom"], m["start"])) == [(0, 5), (0, 50), (1, 7), (2, 10)]


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from genrich_amd.dist import Collectives

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    coll = Collectives(device="cpu")
    # allreduce of the fixed-point fragLen parts
    buf = (C.c_int64 * 2)(10 + rank, 1 << (40 + rank))
    assert coll.allreduce_i64(buf, 2, None) == 0
    # a concatenation of tables of different lengths, the way the library makes one out of the same callback (gx_host_coll.h):
    # every rank writes its records into its own region of a zeroed buffer, the sum over ranks is the concatenation
    counts = [2, 3]
    tab = np.zeros((sum(counts), 2), dtype=np.int64)
    at = sum(counts[:rank])
    tab[at:at + counts[rank], 0] = np.arange(counts[rank]) + 100 * rank
    tab[at:at + counts[rank], 1] = 7 + rank
    cat = (C.c_int64 * tab.size)(*tab.reshape(-1).tolist())
    assert coll.allreduce_i64(cat, tab.size, None) == 0
    got = np.array(list(cat), dtype=np.int64).reshape(-1, 2)
    q.put((rank, [buf[0], buf[1]], got.tolist()))
    dist.destroy_process_group()


def test_collective_callbacks_gloo_world2():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, red, gathered in res:
        assert red == [21, (1 << 40) + (1 << 41)]
        assert gathered == [[0, 7], [1, 7], [100, 8], [101, 8], [102, 8]]
