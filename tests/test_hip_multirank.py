"""GPU: the chromosome-sharded path (2 ranks sharing the one GPU of the test box, collectives
over gloo) must give exactly the single-rank result: same lambda, same peaks, same q-values."""
import os
import socket

import numpy as np
import pytest

import backends as B
import synth

pytestmark = pytest.mark.gpu

LENS = [200_000, 150_000, 90_000, 40_000, 16_000]


def _case():
    tr = synth.make_fragments(LENS, 60_000, 5, peak_every=20_000, tower_every=70_000, frac_tower=0.1)
    ct = synth.make_fragments(LENS, 40_000, 6, uniform_only=True)
    return tr, ct


def _run(gx, tr, ct):
    gx.sample_begin(0, None)
    gx.push_events(tr)
    gx.sample_end()
    gx.sample_begin(1, None)
    gx.push_events(ct)
    _, lam, fac = gx.sample_end()
    gx.pvalues()
    gx.find_peaks()
    return lam, fac, gx.get_peaks()


def _worker(rank, world, port, q):
    import torch.distributed as dist

    import genrich_amd
    from genrich_amd.dist import Collectives, lpt_partition

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    tr, ct = _case()
    owner = lpt_partition(LENS, world)
    owned = np.array([o == rank for o in owner], dtype=np.uint8)
    gx = genrich_amd.Genrich(B.make_params(pq=0.2, qval=True, min_auc=20.0))
    gx.set_chroms(LENS)
    gx.set_owned(owned)
    coll = Collectives(device="cpu")
    gx.set_collectives(rank, world, coll.allreduce_i64, coll.allgather_tab)
    sel = lambda ev: ev[owned[ev["chrom"]].astype(bool)]  # noqa: E731
    lam, fac, peaks = _run(gx, sel(tr), sel(ct))
    q.put((rank, lam, fac, peaks.tobytes()))
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank():
    import torch.multiprocessing as mp

    import genrich_amd
    from genrich_amd.dist import merge_peaks
    from genrich_amd.lib import PEAK_DTYPE

    tr, ct = _case()
    gx = genrich_amd.Genrich(B.make_params(pq=0.2, qval=True, min_auc=20.0))
    gx.set_chroms(LENS)
    lam1, fac1, peaks1 = _run(gx, tr, ct)
    assert len(peaks1) > 0
    gx.close()

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, lam, fac, _ in res:
        assert np.float32(lam).tobytes() == np.float32(lam1).tobytes()
        assert np.float32(fac).tobytes() == np.float32(fac1).tobytes()
    merged = merge_peaks([np.frombuffer(r[3], dtype=PEAK_DTYPE) for r in res])
    assert merged.tobytes() == peaks1.tobytes(), "sharded peaks (coordinates, AUC, p, q) differ from the single-rank run"
