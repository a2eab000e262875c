"""GPU: the chromosome-sharded path (2 ranks sharing the one GPU of the test box, collectives through
host callbacks over gloo) must give exactly the single-rank result -- same lambda / factor, same peak
bytes -- with a control and -q, with fractional weights (ATAC geometry + -s multimapping: the case
where the fixed-point fragLen all-reduce matters) and with three replicates (Fisher + global q).
The library's own RCCL path is run with a single rank (two ranks cannot share one GPU under RCCL):
same code, one-member communicator."""
import os
import socket

import numpy as np
import pytest

import backends as B
import synth

pytestmark = pytest.mark.gpu

LENS = [200_000, 150_000, 90_000, 40_000, 16_000]


def _scenario(name):
    """(params, [(treatment, control or None)] per replicate)"""
    if name == "ctrl_q":
        tr = synth.make_fragments(LENS, 60_000, 5, peak_every=20_000, tower_every=70_000, frac_tower=0.1)
        ct = synth.make_fragments(LENS, 40_000, 6, uniform_only=True)
        return B.make_params(pq=0.2, qval=True, min_auc=20.0), [(tr, ct)]
    if name == "atac_multimap":
        tr = synth.make_fragments(LENS, 60_000, 7, peak_every=20_000, tower_every=70_000, frac_tower=0.1)
        tr = synth.atac_events(synth.add_multimap(tr, LENS, 0.25, seed=8), LENS, d=100)
        return B.make_params(pq=0.01, min_auc=20.0), [(tr, None)]
    if name == "reps3_q":
        reps = [(synth.make_fragments(LENS, 40_000, sd, peak_every=20_000, tower_every=70_000, frac_tower=0.1), None)
                for sd in (11, 13, 15)]
        return B.make_params(pq=0.3, qval=True, min_auc=20.0), reps
    if name == "plain_p":  # one sample, -p, unit weights: every rank must keep the single-rank fast path
        tr = synth.make_fragments(LENS, 80_000, 21, peak_every=20_000, tower_every=70_000, frac_tower=0.1)
        return B.make_params(pq=0.01, min_auc=20.0), [(tr, None)]
    if name == "noctrl_q":  # one sample without a control, -q: the dense BH exchange
        tr = synth.make_fragments(LENS, 80_000, 23, peak_every=20_000, tower_every=70_000, frac_tower=0.1)
        return B.make_params(pq=0.3, qval=True, min_auc=20.0), [(tr, None)]
    raise KeyError(name)


def _run(gx, reps, owned=None):
    sel = (lambda ev: ev) if owned is None else (lambda ev: ev[owned[ev["chrom"]].astype(bool)])
    scal = []
    for tr, ct in reps:
        gx.sample_begin(0, None)
        gx.push_events(sel(tr))
        frag, _, _ = gx.sample_end()
        if ct is not None:
            gx.sample_begin(1, None)
            gx.push_events(sel(ct))
            _, lam, fac = gx.sample_end()
        else:
            lam, fac = gx.sample_no_control(), 1.0
        gx.pvalues()
        scal.append((frag, lam, fac))
    gx.find_peaks()
    return scal, gx.get_peaks()


def _worker(rank, world, port, q, name):
    import torch.distributed as dist

    import genrich_amd
    from genrich_amd.dist import Collectives, lpt_partition

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    params, reps = _scenario(name)
    owner = lpt_partition(LENS, world)
    owned = np.array([o == rank for o in owner], dtype=np.uint8)
    gx = genrich_amd.Genrich(params)
    gx.set_chroms(LENS)
    gx.set_owned(owned)
    coll = Collectives(device="cpu")
    gx.set_collectives(rank, world, coll.allreduce_i64)
    scal, peaks = _run(gx, reps, owned)
    q.put((rank, scal, peaks.tobytes(), gx.path_info()))
    dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("ctrl_q", 2), ("atac_multimap", 2), ("reps3_q", 2), ("plain_p", 2), ("noctrl_q", 2),
                                        ("reps3_q", 3), ("ctrl_q", 4)])
def test_two_ranks_equal_one_rank(name, world):
    """(... or three, or four: more ranges of the p axis for the range-partitioned BH exchange, ranks that own a single
    small chromosome)"""
    import torch.multiprocessing as mp

    import genrich_amd
    from genrich_amd.dist import merge_peaks
    from genrich_amd.lib import PEAK_DTYPE

    params, reps = _scenario(name)
    gx = genrich_amd.Genrich(params)
    gx.set_chroms(LENS)
    scal1, peaks1 = _run(gx, reps)
    assert len(peaks1) > 0
    gx.close()
    # ... and the oracle agrees with the single-rank run
    o = B.Oracle(params)
    o.set_chroms(LENS)
    scalo, peakso = _run(o, reps)
    assert peakso.tobytes() == peaks1.tobytes()

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, name)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, scal, _, flags in res:
        if name in ("ctrl_q", "reps3_q"):
            assert flags & 64, "the range-partitioned BH exchange"
        if name == "noctrl_q":
            assert flags & 32, "the p-value histogram must have travelled as the dense all-reduce"
        if name == "plain_p":
            # lambda reached every rank ahead of the tile stage (the early all-reduce of the closed form of fragLen): the
            # fused kernel wrote the sweep's bits, and the sweep walked the loose slots, as on one rank
            assert flags & 1 and flags & 2, flags
        for (f, lam, fac), (f1, lam1, fac1) in zip(scal, scal1):
            assert f == f1, "fragLen must be the exact fixed-point sum whatever the number of ranks"
            assert np.float32(lam).tobytes() == np.float32(lam1).tobytes()
            assert np.float32(fac).tobytes() == np.float32(fac1).tobytes()
    merged = merge_peaks([np.frombuffer(r[2], dtype=PEAK_DTYPE) for r in res])
    assert merged.tobytes() == peaks1.tobytes(), "sharded peaks (coordinates, AUC, p, q) differ from the single-rank run"


def _withholding_worker(rank, world, port, q):
    """Rank 1 takes part in every exchange but contributes nothing to the big one (the dense p-value histogram)."""
    import torch.distributed as dist

    import genrich_amd
    from genrich_amd.dist import Collectives, lpt_partition

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    params, reps = _scenario("noctrl_q")
    owner = lpt_partition(LENS, world)
    owned = np.array([o == rank for o in owner], dtype=np.uint8)
    gx = genrich_amd.Genrich(params)
    gx.set_chroms(LENS)
    gx.set_owned(owned)
    coll = Collectives(device="cpu")

    def allreduce(buf, n, user):
        if rank == 1 and n > 4096:
            np.ctypeslib.as_array(buf, shape=(n,))[:] = 0   # "lost": this rank's histogram never arrives
        return coll.allreduce_i64(buf, n, user)

    gx.set_collectives(rank, world, allreduce)
    try:
        _run(gx, reps, owned)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))
    dist.destroy_process_group()


def test_a_rank_whose_histogram_is_lost_fails_every_rank():
    """computeQval 377-382: the lengths collected with the p-values must add up to the genome length.  On N ranks that is
    checked on the table AFTER the exchange: a rank whose contribution never arrived would otherwise leave plausible, wrong
    q-values behind.  Every rank must return GX_ERR_PVAL."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_withholding_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = sorted([q.get(timeout=120) for _ in procs])
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
    finally:
        for p in procs:   # (a rank that never came back must not outlive the test)
            if p.is_alive():
                p.kill()
    for rank, msg in res:
        assert "error -8" in msg and "does not match p-value length" in msg, (rank, msg)


@pytest.mark.parametrize("name", ["ctrl_q", "reps3_q", "plain_p", "noctrl_q"])
def test_rccl_path_with_one_rank(name, monkeypatch):
    """gx_set_rccl + GX_FORCE_COLL=1: the all-reduce of the fragLen words and the all-gather of the BH
    records run through RCCL on the library's stream (a communicator of one rank) and must change nothing."""
    import genrich_amd
    from genrich_amd.lib import rccl_unique_id

    params, reps = _scenario(name)
    gx = genrich_amd.Genrich(params)
    gx.set_chroms(LENS)
    scal1, peaks1 = _run(gx, reps)
    gx.close()
    monkeypatch.setenv("GX_FORCE_COLL", "1")
    g2 = genrich_amd.Genrich(params)
    g2.set_chroms(LENS)
    g2.set_rccl(0, 1, rccl_unique_id())
    scal2, peaks2 = _run(g2, reps)
    assert scal2 == scal1
    assert peaks2.tobytes() == peaks1.tobytes()
    if name == "plain_p":
        assert g2.path_info() & 2, "the loose-slot sweep must survive the collectives"
    if name in ("ctrl_q", "reps3_q"):
        assert g2.path_info() & 64, "the range-partitioned BH exchange (RCCL, one rank: send / recv to itself)"
    if name == "noctrl_q":
        assert g2.path_info() & 32, "the dense all-reduce of the p-value histogram (RCCL, one rank)"



def _rccl_worker(rank, world, port, q, name):
    """One process per GPU; the library's own RCCL communicator (gx_set_rccl) carries every exchange, the 128-byte unique id
    travels over a gloo group."""
    import torch
    import torch.distributed as dist

    import genrich_amd
    from genrich_amd.dist import lpt_partition
    from genrich_amd.lib import rccl_unique_id

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    torch.cuda.set_device(rank)
    params, reps = _scenario(name)
    params.device = rank
    owner = lpt_partition(LENS, world)
    owned = np.array([o == rank for o in owner], dtype=np.uint8)
    gx = genrich_amd.Genrich(params)
    gx.set_chroms(LENS)
    gx.set_owned(owned)
    box = [rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    gx.set_rccl(rank, world, box[0])
    scal, peaks = _run(gx, reps, owned)
    q.put((rank, scal, peaks.tobytes(), gx.path_info(), gx.rccl_nranks()))
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["plain_p", "noctrl_q", "ctrl_q", "reps3_q", "atac_multimap"])
def test_rccl_two_gpus_equal_one_rank(name):
    """The multi-rank RCCL execution itself -- the early all-reduce, the dense BH all-reduce, the range-partitioned
    exchange with its grouped send / recv -- on two real GPUs.  Skipped where fewer than two are visible (the round's
    one-GPU test boxes): the first node with two runs it."""
    import torch
    import torch.multiprocessing as mp

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import genrich_amd
    from genrich_amd.dist import merge_peaks
    from genrich_amd.lib import PEAK_DTYPE

    params, reps = _scenario(name)
    gx = genrich_amd.Genrich(params)
    gx.set_chroms(LENS)
    scal1, peaks1 = _run(gx, reps)
    gx.close()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q, name)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for _, scal, _, flags, nranks in res:
        assert nranks == 2
        if name == "plain_p":
            assert flags & 2
        if name == "noctrl_q":
            assert flags & 32
        if name in ("ctrl_q", "reps3_q"):
            assert flags & 64
        for (f, lam, fac), (f1, lam1, fac1) in zip(scal, scal1):
            assert f == f1 and np.float32(lam).tobytes() == np.float32(lam1).tobytes() and np.float32(fac).tobytes() == np.float32(fac1).tobytes()
    merged = merge_peaks([np.frombuffer(r[2], dtype=PEAK_DTYPE) for r in res])
    assert merged.tobytes() == peaks1.tobytes()
