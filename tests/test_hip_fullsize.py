"""GPU, BASELINE.json's full size (hg38, 50 M fragments, configs[1]): size-independent properties
that need no oracle, the whole genome against the oracle byte for byte (every contig, every interval,
every peak), and oracle parity on a mid-size slice of another stream."""
import numpy as np
import pytest

import backends as B
import synth

pytestmark = pytest.mark.gpu

LENS = synth.HG38_LENS


@pytest.fixture(scope="module")
def full_run():
    import genrich_amd
    ev = synth.make_fragments(LENS, 50_000_000, seed=1)
    gx = genrich_amd.Genrich(B.make_params(pq=0.01))
    gx.set_chroms(LENS)
    gx.sample_begin(0, None)
    gx.push_events(ev)
    frag, _, _ = gx.sample_end()
    lam = gx.sample_no_control()
    gx.pvalues()
    gx.find_peaks()
    return ev, gx, frag, lam


def test_fraglen_is_the_sum_of_fragment_lengths(full_run):
    """Linearity: sum over intervals of len * pileup == sum over fragments of their length (exact
    for unit weights), and lambda is its float quotient by the genome length (calcLambda 1831)."""
    ev, gx, frag, lam = full_run
    want = int((ev["end"].astype(np.int64) - ev["start"].astype(np.int64)).sum())
    assert frag == float(want)
    assert gx.genome_len == sum(LENS)
    assert np.float32(lam) == np.float32(want / sum(LENS))


def test_intervals_partition_every_chromosome(full_run):
    ev, gx, _, _ = full_run
    total = 0
    for c in (0, 7, 20, 23, 24):
        end, cols = gx.get_intervals(-1, c)
        assert end[-1] == LENS[c]
        assert np.all(np.diff(end.astype(np.int64)) > 0), "interval ends must be strictly increasing"
        # run-length property: adjacent intervals never carry the same pileup (2241: break only where the difference != 0)
        assert np.all(cols["expt"][1:] != cols["expt"][:-1])
        # number of intervals = distinct positions with a non-zero net difference (+1 closing)
        sel = ev[ev["chrom"] == c]
        pos = np.concatenate([sel["start"], sel["end"]]).astype(np.int64)
        w = np.concatenate([np.ones(len(sel), np.int64), -np.ones(len(sel), np.int64)])
        order = np.argsort(pos, kind="stable")
        pos, w = pos[order], w[order]
        first = np.concatenate([[True], pos[1:] != pos[:-1]])
        net = np.add.reduceat(w, np.flatnonzero(first))
        upos = pos[first]
        nbreak = int(((net != 0) & (upos > 0) & (upos < LENS[c])).sum())
        assert len(end) == nbreak + 1
        # pileup checksum: sum(len * cov) on this chromosome == bases covered by its fragments
        lens_iv = np.diff(np.concatenate([[0], end.astype(np.int64)]))
        assert int(round(float((lens_iv * cols["expt"].astype(np.float64)).sum()))) == int(
            (np.minimum(sel["end"], LENS[c]).astype(np.int64) - sel["start"]).sum())
        total += len(end)
    assert total > 0


def test_peaks_are_ordered_separated_and_above_threshold(full_run):
    _, gx, _, _ = full_run
    pk = gx.get_peaks()
    assert len(pk) > 50_000
    key = pk["chrom"].astype(np.int64) * (1 << 32) + pk["start"]
    assert np.all(np.diff(key) > 0), "peaks must be in chromosome-table order, then by position"
    same = pk["chrom"][1:] == pk["chrom"][:-1]
    gap = pk["start"][1:].astype(np.int64) - pk["end"][:-1].astype(np.int64)
    assert np.all(gap[same] > 100), "two peaks closer than maxGap would have been one candidate"
    assert np.all(pk["auc"] >= 200.0) and np.all(pk["p"] > 2.0)
    assert np.all(pk["end"] > pk["start"]) and np.all(pk["summit"] < pk["end"] - pk["start"])
    assert gx.peak_bp == int((pk["end"].astype(np.int64) - pk["start"]).sum())


def test_whole_genome_is_the_oracles_bytes(full_run):
    """All 25 contigs / 50 M fragments through the HIP path and through the oracle: the peak list must be
    the same bytes and every chromosome's interval ends, pileups and -log10 p the same bits."""
    ev, gx, frag, lam = full_run
    o = B.Oracle(B.make_params(pq=0.01))
    so = B.run_case(o, dict(lens=LENS, replicates=[dict(save=None, treat=ev, ctrl=None)]))
    assert so[0][0] == frag and np.float32(so[0][1]).tobytes() == np.float32(lam).tobytes()
    po, ph = o.get_peaks(), gx.get_peaks()
    assert len(po) == len(ph) > 50_000
    assert po.tobytes() == ph.tobytes(), "peak lists differ"
    assert o.peak_bp == gx.peak_bp
    total = 0
    for c in range(len(LENS)):
        eo, co = o.get_intervals(-1, c)
        eh, ch = gx.get_intervals(-1, c)
        assert np.array_equal(eo, eh), f"interval ends differ on contig {c}"
        assert np.array_equal(co["expt"].view(np.uint32), ch["expt"].view(np.uint32)), f"pileups differ on contig {c}"
        assert np.array_equal(co["p"].view(np.uint32), ch["p"].view(np.uint32)), f"p-values differ on contig {c}"
        total += len(eo)
    assert total == gx.interval_total()
    o.close()


def test_rerun_is_bit_identical(full_run):
    """Determinism: integer atomics and ordered float sums only -> the same bytes every run."""
    import genrich_amd
    ev, gx, frag, lam = full_run
    first = gx.get_peaks().tobytes()
    g2 = genrich_amd.Genrich(B.make_params(pq=0.01))
    g2.set_chroms(LENS)
    g2.sample_begin(0, None)
    for part in np.array_split(ev[::-1], 7):  # different order and batching of the same events
        g2.push_events(part)
    f2, _, _ = g2.sample_end()
    l2 = g2.sample_no_control()
    g2.pvalues()
    g2.find_peaks()
    assert (f2, l2) == (frag, lam)
    assert g2.get_peaks().tobytes() == first


def test_midsize_slice_against_oracle():
    """chr19, chr20, chr21, chr22, chrY, chrM of the same generator (~280 Mbp, 4.5 M fragments):
    the oracle finishes in seconds; intervals and peaks must agree exactly."""
    import genrich_amd
    sub = [LENS[i] for i in (18, 19, 20, 21, 23, 24)]
    ev = synth.make_fragments(sub, 4_500_000, seed=3)
    case = dict(lens=sub, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.01)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = genrich_amd.Genrich(params)
    sh = B.run_case(h, case)
    assert so[0][0] == sh[0][0] and np.float32(so[0][1]) == np.float32(sh[0][1])
    po, ph = o.get_peaks(), h.get_peaks()
    assert len(po) == len(ph) > 1000
    for f in ("chrom", "start", "end", "summit"):
        assert np.array_equal(po[f], ph[f]), f
    assert np.array_equal(po["auc"].view(np.uint32), ph["auc"].view(np.uint32)), "AUC must be bit-identical"
    assert po.tobytes() == ph.tobytes()
    for c in (0, 5):
        eo, co = o.get_intervals(-1, c)
        eh, ch = h.get_intervals(-1, c)
        assert np.array_equal(eo, eh)
        assert np.array_equal(co["expt"].view(np.uint32), ch["expt"].view(np.uint32))
        assert np.array_equal(co["p"].view(np.uint32), ch["p"].view(np.uint32))


# ---- BASELINE.json's other GPU configs at full size: all 25 contigs against the oracle, byte for byte -------------
# (configs[2] control + -q, configs[3] ATAC geometry + -s multimapping weights, configs[4] three replicates + Fisher + -q;
#  savePileupCtrl Genrich.c:2052-2161, saveFragAtac 2728-2749 + addFrac / subFrac 2311-2488, combinePval 612-667)

def _whole_genome_against_oracle(case, params, min_peaks, setups=(None,)):
    """One oracle run, and one run of the HIP library per entry of `setups` (a function that prepares the fresh context --
    a hint, a switch -- or None): everything of every run against the oracle.  Returns the runs' gx_path_info flags."""
    import genrich_amd
    o = B.Oracle(params)
    so = B.run_case(o, case)
    po = o.get_peaks()
    nrep = len(case["replicates"])
    whiches = [-1] + (list(range(nrep)) if nrep > 1 else [])
    runs = []
    for setup in setups:
        h = genrich_amd.Genrich(params)
        again = 0
        if isinstance(setup, tuple):   # (fn, n): the case n times on the same context (what a context learns from a run: the last one counts)
            setup, again = setup[0], setup[1] - 1
        if setup is not None:
            setup(h)
        sh = B.run_case(h, case)
        for _ in range(again):
            h.reset()
            sh = B.run_case(h, case)
        flags = h.path_info()
        for k, ((fo, lo, co), (fh, lh, ch)) in enumerate(zip(so, sh)):
            # fragLen: the device's sum is the exact sum of the reference's float products, rounded once, and is compared bit
            # for bit with the oracle's exact sum; the reference's own double accumulation rounds at some of its additions once
            # the sum has passed 2^26 (fractional weights only) -- counted by the oracle, half a unit in the last place each
            # (DESIGN.md section 2).  lambda, the float that everything downstream uses, must be the same bits.
            B.assert_fraglen(o, k, fo, fh)
            assert np.float32(lo).tobytes() == np.float32(lh).tobytes()
            if co is not None:
                assert np.float32(co).tobytes() == np.float32(ch).tobytes()
        ph = h.get_peaks()
        assert len(po) == len(ph) >= min_peaks, (len(po), len(ph))
        assert po.tobytes() == ph.tobytes(), "peak lists differ"
        assert o.peak_bp == h.peak_bp
        runs.append((h, flags))
    total = 0
    for which in whiches:
        for c in range(len(LENS)):
            eo, co = o.get_intervals(which, c)   # (one contig of the oracle's table at a time, against every run)
            for h, _ in runs:
                eh, ch = h.get_intervals(which, c, piles=False)
                assert np.array_equal(eo, eh), f"interval ends differ on contig {c} (array {which})"
                for k in ("p", "q"):
                    assert np.array_equal(co[k].view(np.uint32), ch[k].view(np.uint32)), f"{k} differs on contig {c} (array {which})"
            if which == -1:
                total += len(eo)
    out = []
    for h, flags in runs:
        assert total == h.interval_total()
        h.close()
        out.append(flags)
    o.close()
    return out[0] if len(out) == 1 else out


def test_fullsize_config3_control_and_q_is_the_oracles_bytes():
    # (peaks of ~1,000 fragments every 200 kb: strong enough to stay significant after the genome-wide correction --
    # SURVEY 8(d)'s generator caveat -- so the q-mode sweep sees > 10^4 peaks, not just the towers)
    t = synth.make_fragments(LENS, 50_000_000, seed=1, peak_every=200_000, tower_every=50_000_000)
    c = synth.make_fragments(LENS, 50_000_000, seed=2, uniform_only=True)
    case = dict(lens=LENS, replicates=[dict(save=None, treat=t, ctrl=c)])
    flags = _whole_genome_against_oracle(case, B.make_params(pq=0.05, qval=True), 1_000)
    assert flags & 1, "the tile stage of a unit-weight sample is k_sbtile"


def test_fullsize_config4_atac_multimap_is_the_oracles_bytes():
    """BASELINE.json configs[3] on BOTH of its device paths, against one oracle run: (1) as `genrich-amd -s` and bench.py run
    it -- the library is told that fractional weights may come (gx_expect_fractional), the first sample already leaves level 1
    as pair records with a weight class and goes through k_sbtile<.., FRAC>: THE PATH bench.py TIMES; (2) without the hint:
    the unit-weight level 1 meets a fractional weight, gives the sample up (ST_SB_FRAC) and the general chain builds it."""
    ev = synth.make_fragments(LENS, 50_000_000, seed=1)
    ev = synth.atac_events(synth.add_multimap(ev, LENS, 0.10, seed=11), LENS, d=100)
    case = dict(lens=LENS, replicates=[dict(save=None, treat=ev, ctrl=None)])
    # (3) the hinted context's SECOND run -- what bench.py's timed steps are: the context has seen the fractions, lambda comes with the
    # sample's end, and the sweep walks the loose slots all the same (k_loose_late, round 6)
    hinted, plain, second = _whole_genome_against_oracle(case, B.make_params(pq=0.01), 10_000,
                                                         setups=(lambda h: h.expect_fractional(True), None,
                                                                 (lambda h: h.expect_fractional(True), 2)))
    assert second & 1 and second & 128 and second & 2 and second & 16384, "the second run sweeps the loose slots, bits written late"
    assert not hinted & 2, "the first run's early lambda did not stand (fractions): it took the tight table"
    assert hinted & 1 and hinted & 16 and hinted & 128, "the hinted run is the fused tile stage on fractional pair records"
    assert not hinted & 4, "... from the first sample on: nothing was sent back"
    assert not plain & 1 and plain & 4, "without the hint the first fractional weight sends the sample to the general chain"


def test_fullsize_config5_three_replicates_fisher_q_is_the_oracles_bytes():
    reps = [dict(save=None, treat=synth.make_fragments(LENS, 50_000_000, seed=s), ctrl=None) for s in (1, 3, 5)]
    case = dict(lens=LENS, replicates=reps)
    _whole_genome_against_oracle(case, B.make_params(pq=0.05, qval=True), 1_000)


def test_fullsize_config2_with_excluded_regions_is_the_oracles_bytes():
    """The reference's own flagship command line is `-e chrM,chrY -E <N-gap BED files>` (README.md:463-465): BASELINE's config 2
    stream with ~800 excluded regions (one touching position 0, one reaching a chromosome's end: synth.excluded_regions) and
    two skipped contigs -- the whole genome against the oracle: peaks, interval ends, p (SKIP inside the regions), lambda with
    the regions' bases taken out of the genome length (calcLambda 1819-1827; the bedPos / save toggling of savePileupExpt
    2185-2263 and saveLambda 1847-1876)."""
    skip = [False] * len(LENS)
    skip[23] = skip[24] = True   # chrY, chrM
    beds = synth.excluded_regions(LENS, n=800, seed=7, skip=(23, 24))
    assert beds[0][0] == 0 and beds[1][-1] == LENS[1] and sum(len(b) for b in beds) // 2 > 700
    ev = synth.make_fragments(LENS, 50_000_000, seed=1)
    case = dict(lens=LENS, skip=skip, beds=beds, replicates=[dict(save=None, treat=ev, ctrl=None)])
    import genrich_amd
    params = B.make_params(pq=0.01)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = genrich_amd.Genrich(params)
    sh = B.run_case(h, case)
    B.assert_fraglen(o, 0, so[0][0], sh[0][0])
    assert np.float32(so[0][1]).tobytes() == np.float32(sh[0][1]).tobytes()
    excl = sum(b[i + 1] - b[i] for b in beds for i in range(0, len(b), 2))
    assert o.genome_len == h.genome_len == sum(L for L, s in zip(LENS, skip) if not s) - excl
    po, ph = o.get_peaks(), h.get_peaks()
    assert len(po) == len(ph) > 40_000
    assert po.tobytes() == ph.tobytes(), "peak lists differ"
    assert o.peak_bp == h.peak_bp
    total = 0
    for c in range(len(LENS)):
        if skip[c]:
            continue
        eo, co = o.get_intervals(-1, c)
        eh, ch = h.get_intervals(-1, c)
        assert np.array_equal(eo, eh), f"interval ends differ on contig {c}"
        for k in ("expt", "ctrl", "p"):
            assert np.array_equal(co[k].view(np.uint32), ch[k].view(np.uint32)), f"{k} differs on contig {c}"
        if beds[c]:
            assert (co["p"] == -1.0).any(), "an excluded region carries SKIP"
        total += len(eo)
    assert total == h.interval_total()
    o.close()
    h.close()
