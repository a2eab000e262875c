"""GPU parity of the round-3 fast paths against the CPU oracle, and proof that they are the paths that ran:
  * k_sbtile (gx_sbtile.h): level 2 of the bucket sort fused with the tile passes;
  * the peak sweep on the tile stage's loose slots (gx_kernels.h LooseCtl): no k_pack_pval round trip;
  * the way back to the general chain when a super-bucket does not fit k_sbtile's LDS or the sample holds
    fractional weights (Genrich.c:2311-2488 addFrac / subFrac).
Every combination must give the oracle's bits (the reference's: oracle pinned in tests/test_oracle.py)."""
import numpy as np
import pytest

import backends as B
import synth
from test_hip_parity import assert_same_run, hip_backend

pytestmark = pytest.mark.gpu

FUSED, LOOSE, FELL_BACK, PT_GREW, PAIRS, FRAC_PAIRS, PILES_MADE = 1, 2, 4, 8, 16, 128, 256
MERGE_P, PACK_HIST, LAZY_Q, LATE_LOOSE, Q_LOOSE = 1024, 2048, 8192, 16384, 32768


def _case(seed=11, n=90_000, lens=(400_000, 123_457, 16_384, 4_097, 5), **kw):
    lens = list(lens)
    ev = synth.make_fragments(lens[:4], n, seed, peak_every=20_000, tower_every=150_000, **kw)
    return dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])


def _run(case, params):
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    return o, h, flags


@pytest.mark.parametrize("no_fused,no_loose", [(False, False), (True, False), (False, True), (True, True)])
def test_every_combination_of_the_fast_paths_is_the_oracles_bits(monkeypatch, no_fused, no_loose):
    if no_fused:
        monkeypatch.setenv("GX_NO_FUSED", "1")
    if no_loose:
        monkeypatch.setenv("GX_NO_LOOSE", "1")
    o, h, flags = _run(_case(), B.make_params(pq=0.01, min_auc=50.0))
    assert h.n_peaks > 0
    assert bool(flags & FUSED) == (not no_fused)
    assert bool(flags & LOOSE) == (not no_loose)
    assert not flags & FELL_BACK


def test_default_run_takes_both_fast_paths_and_a_second_sweep_agrees():
    case = _case(seed=5, n=150_000)
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    B.run_case(o, case)
    h = hip_backend(params)
    B.run_case(h, case)
    assert h.path_info() & (FUSED | LOOSE) == FUSED | LOOSE
    first = h.get_peaks().copy()
    h.find_peaks()  # once more, on the same loose slots
    again = h.get_peaks()
    assert first.tobytes() == again.tobytes() == o.get_peaks().tobytes()


@pytest.mark.parametrize("q_loose", [True, False])
def test_q_mode_uses_the_fused_tile_stage_and_since_round_6_the_loose_slots(monkeypatch, q_loose):
    if not q_loose:
        monkeypatch.setenv("GX_NO_Q_LOOSE", "1")
    o, h, flags = _run(_case(seed=3), B.make_params(pq=0.05, qval=True, min_auc=20.0))
    assert flags & FUSED and bool(flags & LOOSE) == q_loose and bool(flags & Q_LOOSE) == q_loose, flags


def test_a_control_uses_the_fused_tile_stage_for_both_samples():
    lens = [300_000, 70_001]
    t = synth.make_fragments(lens, 70_000, 21, peak_every=20_000, tower_every=150_000)
    c = synth.make_fragments(lens, 50_000, 22, uniform_only=True)
    case = dict(lens=lens, replicates=[dict(save=None, treat=t, ctrl=c)])
    o, h, flags = _run(case, B.make_params(pq=0.01, min_auc=20.0))
    assert flags & FUSED and not flags & LOOSE


@pytest.mark.parametrize("n_pile", [30_000, 44_000, 58_000])
def test_a_super_bucket_beyond_the_key_array_is_worked_off_in_rounds(n_pile):
    # 49 tiles -> 4 tiles per super-bucket; 30,000 / 44,000 fragments inside one of them = more keys than SBT_KEYCAP (46,976)
    # but no more pair records than the slots take (64 K): rounds of tiles, no fall-back
    lens = [200_000]
    rng = np.random.default_rng(9)
    ev = synth.make_fragments(lens, 4_000, 2, peak_every=20_000, tower_every=150_000)
    pile = np.zeros(n_pile, dtype=B.EVENT_DTYPE)
    pile["start"] = 66_000 + rng.integers(0, 12_000, size=len(pile))
    pile["end"] = pile["start"] + 100 + rng.integers(0, 200, size=len(pile))
    pile["count"] = 1
    case = dict(lens=lens, replicates=[dict(save=None, treat=np.concatenate([ev, pile]), ctrl=None)])
    o, h, flags = _run(case, B.make_params(pq=0.01, min_auc=20.0))
    assert flags & FUSED and not flags & FELL_BACK
    assert flags & LOOSE


def test_a_super_bucket_beyond_the_lds_goes_back_to_the_general_chain():
    # 49 tiles -> 4 tiles per super-bucket; 70,000 fragments inside one of them: more pair records than k_sbtile's slots take
    lens = [200_000]
    rng = np.random.default_rng(9)
    ev = synth.make_fragments(lens, 20_000, 2, peak_every=20_000, tower_every=150_000)
    pile = np.zeros(70_000, dtype=B.EVENT_DTYPE)
    pile["start"] = 66_000 + rng.integers(0, 12_000, size=len(pile))
    pile["end"] = pile["start"] + 100 + rng.integers(0, 200, size=len(pile))
    pile["count"] = 1
    case = dict(lens=lens, replicates=[dict(save=None, treat=np.concatenate([ev, pile]), ctrl=None)])
    o, h, flags = _run(case, B.make_params(pq=0.01, min_auc=20.0))
    assert flags & FELL_BACK and not flags & FUSED
    assert flags & LOOSE  # (the general chain's tile kernel writes the sweep's bits too)


def test_fractional_weights_ride_the_pair_records_from_the_second_sample_on():
    """The first sample with a fractional weight is turned away by the fused kernel (unit-weight records) and built on the
    general chain; the context then writes pair records with a weight class and the next samples stay fused."""
    lens = [300_000, 70_001]
    ev = synth.add_multimap(synth.make_fragments(lens, 60_000, 31, peak_every=20_000, tower_every=150_000), lens, 0.3, 32)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None), dict(save=None, treat=ev[::2].copy(), ctrl=None),
                                       dict(save=None, treat=ev[1::2].copy(), ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    assert flags & FELL_BACK and flags & FUSED and flags & 16 and not flags & LOOSE


def test_fractional_hint_keeps_the_first_sample_fused():
    lens = [300_000, 70_001]
    ev = synth.add_multimap(synth.make_fragments(lens, 60_000, 31, peak_every=20_000, tower_every=150_000), lens, 0.3, 32)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    h.expect_fractional(True)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    assert flags & FUSED and flags & 16 and not flags & FELL_BACK


def test_fractional_hint_on_unit_weight_data_keeps_the_early_lambda_and_the_loose_sweep():
    """`genrich-amd -s` on a file without multimappers: the hint (gx_expect_fractional) only selects the kernels that can
    carry a weight class; the closed form of fragLen, lambda ahead of the tile stage and the sweep on the loose slots
    stay until a count > 1 really arrives (processPair, Genrich.c:3122-3176, gives every alignment of a unique read count 1)."""
    case = _case(seed=13, n=120_000)
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    h.expect_fractional(True)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    assert flags & FUSED and flags & PAIRS and flags & FRAC_PAIRS and flags & LOOSE and not flags & FELL_BACK
    # a second run of the same context, the hint withdrawn: unit-weight records again, same bits
    h.reset()
    h.expect_fractional(False)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    assert flags & FUSED and flags & LOOSE and not flags & FRAC_PAIRS


def test_what_a_context_has_learned_about_fractions_is_not_the_hints_to_clear():
    """A hinted context that then SEES a fractional weight (Scalars::fracSeen) does without the early lambda from the next
    sample on, and withdrawing the hint does not bring it back: the next run still writes records with a weight class."""
    lens = [300_000, 70_001]
    unit = synth.make_fragments(lens, 60_000, 31, peak_every=20_000, tower_every=150_000)
    ev = synth.add_multimap(unit, lens, 0.3, 32)
    params = B.make_params(pq=0.01, min_auc=20.0)
    h = hip_backend(params)
    h.expect_fractional(True)
    # (the third run: lambda comes with the sample's end, the loose slots are swept all the same -- k_loose_late, round 6)
    for i, (treat, want_loose) in enumerate(((unit, True), (ev, False), (unit, True))):
        case = dict(lens=lens, replicates=[dict(save=None, treat=treat, ctrl=None)])
        o = B.Oracle(params)
        so = B.run_case(o, case)
        h.reset()
        sh = B.run_case(h, case)
        flags = h.path_info()
        assert_same_run(o, h, so, sh, case)
        assert flags & FUSED and flags & FRAC_PAIRS and not flags & FELL_BACK
        assert bool(flags & LOOSE) == want_loose, (flags, want_loose)
        assert bool(flags & LATE_LOOSE) == (i == 2), flags
        if i == 1:
            h.expect_fractional(False)   # (the third run: no hint any more -- what was learned stays)
        o.close()


def test_fractional_weights_without_pair_records_stay_on_the_general_chain(monkeypatch):
    monkeypatch.setenv("GX_NO_FRAC_PAIRS", "1")
    lens = [300_000, 70_001]
    ev = synth.add_multimap(synth.make_fragments(lens, 60_000, 31, peak_every=20_000, tower_every=150_000), lens, 0.3, 32)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None), dict(save=None, treat=ev[::2].copy(), ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    assert flags & FELL_BACK and not flags & FUSED and not flags & LOOSE


def test_fractional_pairs_atac_geometry_towers_and_a_dense_sample():
    """ATAC cut sites with -s weights on the fractional pair records: heavy tiles (a tower: the whole workgroup), bins
    beyond the key array (rounds), and a sample dense enough for the all-bins launch -- bits of the oracle everywhere."""
    lens = [200_000, 90_000]
    params = B.make_params(pq=0.01, min_auc=20.0)
    warm = synth.add_multimap(synth.make_fragments(lens, 2_000, 5), lens, 0.5, 6)
    tr = synth.make_fragments(lens, 70_000, 41, peak_every=20_000, tower_every=70_000, frac_tower=0.25)
    tr = synth.atac_events(synth.add_multimap(tr, lens, 0.25, seed=42), lens, d=100)
    case = dict(lens=lens, replicates=[dict(save=None, treat=warm, ctrl=None), dict(save=None, treat=tr, ctrl=None)])
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    assert flags & FUSED and flags & 16


def test_tiles_without_intervals_and_unsaved_chromosomes_under_the_loose_sweep():
    # long empty stretches (tiles whose slots k_scan_iv has to fill), a chromosome without reads, one that is
    # not saved (its tiles hold records but no intervals), peaks that run across tile borders
    lens = [250_000, 40_000, 90_000, 30_000]
    rng = np.random.default_rng(4)
    parts = []
    for ci, centre in [(0, 4_096), (0, 8_192), (0, 200_000), (2, 45_056), (3, 12_288)]:
        e = np.zeros(900, dtype=B.EVENT_DTYPE)
        e["chrom"] = ci
        e["start"] = centre - 300 + rng.integers(0, 400, size=len(e))
        e["end"] = e["start"] + 150 + rng.integers(0, 100, size=len(e))
        e["count"] = 1
        parts.append(e)
    ev = np.concatenate(parts)
    rng.shuffle(ev)
    for save in (None, [1, 1, 0, 1]):
        case = dict(lens=lens, replicates=[dict(save=save, treat=ev, ctrl=None)])
        o, h, flags = _run(case, B.make_params(pq=0.01, min_auc=20.0, max_gap=300))
        assert flags & FUSED and flags & LOOSE
        assert h.n_peaks >= 3


def test_a_flat_significant_plateau_across_a_tile_without_records():
    # one significant interval that spans a whole tile with no breakpoint in it: the tile's slot must carry the
    # previous interval's end (k_scan_iv) for the candidate walk to see the right lengths
    lens = [600_000]
    e = np.zeros(40, dtype=B.EVENT_DTYPE)
    e["start"] = 3_000
    e["end"] = 14_000
    e["count"] = 1
    bg = synth.make_fragments(lens, 300, 8, uniform_only=True)
    bg = bg[(bg["end"] < 2_500) | (bg["start"] > 20_000)]
    case = dict(lens=lens, replicates=[dict(save=None, treat=np.concatenate([e, bg]), ctrl=None)])
    o, h, flags = _run(case, B.make_params(pq=0.01, min_auc=20.0))
    assert flags & FUSED and flags & LOOSE
    assert h.n_peaks >= 1


def test_page_tables_grow_when_a_list_outgrows_its_row(monkeypatch):
    """RETRY_PT: with one page per (XCD class, super-bucket) list (GX_PT_JMAX=1; 16 normally) and everything in one
    super-bucket (GX_SBSHIFT=8) a list of 90,000 fragments needs more pages than a row of the page table holds --
    k_sort1 raises ST_PT_FULL, the host grows the tables and builds the sample again: same bits as the oracle, and the
    flag says that this is what happened.  (The kernels behind k_sort1 run on the short lists before the host has
    seen the flag: they must stay inside the table rows -- list_len, gx_sort.h; round 3 found them reading past.)"""
    monkeypatch.setenv("GX_PT_JMAX", "1")
    monkeypatch.setenv("GX_SBSHIFT", "8")
    o, h, flags = _run(_case(seed=23, n=90_000), B.make_params(pq=0.01, min_auc=50.0))
    assert h.n_peaks > 0
    assert flags & PT_GREW


def test_page_tables_do_not_grow_in_an_ordinary_run():
    o, h, flags = _run(_case(seed=23, n=90_000), B.make_params(pq=0.01, min_auc=50.0))
    assert not flags & PT_GREW


@pytest.mark.parametrize("frac", [False, True])
def test_half_size_bins_and_the_128_key_level_1(monkeypatch, frac):
    """A dense sample takes bins of half the size; beyond 4096 of them level 1's second pass scatters to 128 fine bins per
    coarse one (two owner wavefronts).  Forced here on a 36 Mbp genome with 2-tile bins (4,395 half-size bins)."""
    monkeypatch.setenv("GX_SBSHIFT", "1")
    monkeypatch.setenv("GX_FORCE_HALF_BINS", "1")
    lens = [20_000_000, 16_000_000]
    ev = synth.make_fragments(lens, 400_000, 77, peak_every=200_000, tower_every=5_000_000)
    params = B.make_params(pq=0.01, min_auc=20.0)
    reps = [dict(save=None, treat=ev, ctrl=None)]
    if frac:
        warm = synth.add_multimap(synth.make_fragments(lens, 2_000, 5), lens, 0.5, 6)
        reps = [dict(save=None, treat=warm, ctrl=None), dict(save=None, treat=synth.add_multimap(ev, lens, 0.2, 78), ctrl=None)]
    case = dict(lens=lens, replicates=reps)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    flags = h.path_info()
    assert_same_run(o, h, so, sh, case)
    assert flags & FUSED and flags & 16


def test_replicates_keep_their_exact_pileups_until_somebody_asks():
    """Three replicates without control, Fisher, -q (combinePval Genrich.c:612-667): a replicate's pileup floats -- the
    reference's Pileup.cov, which only -f / -k print -- are not written when the next replicate reuses the loose slots;
    the replicate keeps its exact pileups instead and the floats are made when somebody asks, here after the peaks
    are out.  They must be the bits a single-sample run of that replicate gives (there the oracle has them)."""
    lens = [300_000, 120_000, 4_097]
    reps = [dict(save=None, treat=synth.make_fragments(lens, 60_000, 100 + r, peak_every=25_000, tower_every=110_000), ctrl=None)
            for r in range(3)]
    case = dict(lens=lens, replicates=reps)
    params = B.make_params(pq=0.05, qval=True, min_auc=20.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    assert not h.path_info() & PILES_MADE, "no pileup float was written on the way to the peaks"
    assert_same_run(o, h, so, sh, case)      # (ends / p / q of every array, the peaks: asks for no pileups of a replicate)
    for r, rep in enumerate(reps):
        o1 = B.Oracle(params)
        B.run_case(o1, dict(lens=lens, replicates=[rep]))
        for c in range(len(lens)):
            e1, c1 = o1.get_intervals(-1, c)
            eh, ch = h.get_intervals(r, c)   # the late request: from what the replicate kept
            assert np.array_equal(e1, eh)
            assert np.array_equal(c1["expt"].view(np.uint32), ch["expt"].view(np.uint32)), (r, c)
            assert np.array_equal(c1["ctrl"].view(np.uint32), ch["ctrl"].view(np.uint32)), (r, c)
        o1.close()
    assert h.path_info() & PILES_MADE
    h.reset()                                 # the kept buffers go back to the pool: a second run of the same context
    sh = B.run_case(h, case)
    assert not h.path_info() & PILES_MADE
    assert_same_run(o, h, so, sh, case)


def test_switches_are_read_when_the_context_is_made_and_can_be_set_on_it(monkeypatch):
    """The GX_* test switches are parsed once, in gx_create; gx_set_knob changes one on a live context (bench.py's
    `materialised` loop).  Both routes must give the oracle's bits and say which path ran."""
    case = _case(seed=21, n=60_000)
    params = B.make_params(pq=0.01, min_auc=50.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    assert h.path_info() & LOOSE
    monkeypatch.setenv("GX_NO_LOOSE", "1")     # (too late for this context: it was read when the context was made)
    h.reset()
    sh = B.run_case(h, case)
    assert h.path_info() & LOOSE
    h.set_knob("GX_NO_LOOSE", 1)
    h.reset()
    sh = B.run_case(h, case)
    assert not h.path_info() & LOOSE
    assert_same_run(o, h, so, sh, case)
    h.set_knob("GX_NO_LOOSE", 0)
    h.reset()
    sh = B.run_case(h, case)
    assert h.path_info() & LOOSE
    assert_same_run(o, h, so, sh, case)
    with pytest.raises(RuntimeError):
        h.set_knob("GX_NO_SUCH_SWITCH", 1)
    h2 = hip_backend(params)                   # a new context sees the environment
    sh = B.run_case(h2, case)
    assert not h2.path_info() & LOOSE
    assert_same_run(o, h2, so, sh, case)


def test_a_pileup_that_does_not_return_to_zero_is_an_error(monkeypatch):
    """savePileupExpt 2283-2289: behind a chromosome's last base the difference array must be back at zero ("finishes at
    %f (not 0.0)").  On the device: the closing interval of a chromosome's last tile may only hold the fragments that end
    at the chromosome's length (they have no end record) -- checked by the scan over the tiles, GX_ERR_ARR otherwise.
    A healthy sample with fragments that reach the end passes; with the weight of those ends damaged behind level 1 of the
    sort (GX_FAULT=1: as if a record had been lost) every tile-stage variant must refuse the sample."""
    lens = [150_000]
    ev = synth.make_fragments(lens, 30_000, 3, peak_every=20_000)
    tail = np.array([(0, lens[0] - 180 - 3 * k, lens[0], 1) for k in range(40)], dtype=B.EVENT_DTYPE)  # 40 fragments end at len
    case = dict(lens=lens, replicates=[dict(save=None, treat=np.concatenate([ev, tail]), ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=50.0)
    _run(case, params)                                    # healthy: the oracle's bits, no error
    monkeypatch.setenv("GX_FAULT", "1")
    for knobs in ({}, {"GX_NO_PAIRS": "1"}, {"GX_NO_FUSED": "1"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        h = hip_backend(params)
        with pytest.raises(RuntimeError, match="return to 0"):
            B.run_case(h, case)
        h.close()
        for k in knobs:
            monkeypatch.delenv(k)


# ---- 8-byte events (gx_event8, gx_push_events_packed): the same sample, half the bytes (saveInterval's arguments, Genrich.c:2516-2519) ----

def _push_mixed(h, ev, lens, where_packed=0):
    """The sample through gx_push_events_packed for every event that fits the 8-byte form and gx_push_events for the rest,
    in pieces (odd counts among them)."""
    from genrich_amd.lib import pack_events
    h.set_chroms(lens)
    h.sample_begin(0, None)
    p8, rest = pack_events(ev)
    cuts = [0, 1, 1 + 4097, len(p8) // 2 | 1, len(p8)]
    for a, b in zip(cuts[:-1], cuts[1:]):
        if b > a:
            h.push_events_packed(p8[a:b])
    if len(rest):
        h.push_events(rest)
    out = h.sample_end()
    h.sample_no_control()
    h.pvalues()
    h.find_peaks()
    return out, len(p8), len(rest)


def test_packed_events_give_the_sixteen_byte_paths_bits():
    """Unit-weight fragments plus what does NOT fit eight bytes (a fragment of 70,000 bases, an interval that ends before it
    starts): the packed pieces are read in place by k_sort_a<.., PACKED>, the others travel as gx_event -- oracle's bits."""
    lens = [400_000, 123_457, 16_384]
    ev = synth.make_fragments(lens, 90_000, 21, peak_every=20_000, tower_every=150_000)
    extra = np.zeros(3, dtype=B.EVENT_DTYPE)
    extra["chrom"], extra["start"], extra["end"], extra["count"] = [0, 0, 1], [1000, 250_000, 5000], [71_000, 249_990, 5100], [1, 1, 1]
    ev = np.concatenate([ev, extra])
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    (frag, _, _), n8, nrest = _push_mixed(h, ev, lens)
    assert n8 > 80_000 and nrest == 2
    flags = h.path_info()
    assert flags & FUSED and flags & PAIRS and flags & 512 and flags & LOOSE, flags
    assert frag == so[0][0]
    assert h.get_peaks().tobytes() == o.get_peaks().tobytes()
    for c in range(len(lens)):
        eo, co = o.get_intervals(-1, c)
        eh, ch = h.get_intervals(-1, c)
        assert np.array_equal(eo, eh)
        for k in ("expt", "p"):
            assert np.array_equal(co[k].view(np.uint32), ch[k].view(np.uint32)), (k, c)


def test_packed_events_with_weight_classes_and_on_the_general_chain(monkeypatch):
    """All eight counts in the 8-byte form (hinted: fractional pair records), and the same pieces through the general chain
    (GX_NO_FUSED: k_sort1 reads 16-byte events, so the library unpacks them first)."""
    lens = [300_000, 70_001]
    ev = synth.add_multimap(synth.make_fragments(lens, 60_000, 31, peak_every=20_000, tower_every=150_000), lens, 0.3, 32)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    B.run_case(o, case)
    for general in (False, True):
        if general:
            monkeypatch.setenv("GX_NO_FUSED", "1")
        h = hip_backend(params)
        h.expect_fractional(True)
        _, n8, nrest = _push_mixed(h, ev, lens)
        flags = h.path_info()
        assert bool(flags & 512) == (not general) and bool(flags & FUSED) == (not general), flags
        assert h.get_peaks().tobytes() == o.get_peaks().tobytes()
        for c in range(len(lens)):
            eo, co = o.get_intervals(-1, c)
            eh, ch = h.get_intervals(-1, c)
            assert np.array_equal(eo, eh)
            assert np.array_equal(co["p"].view(np.uint32), ch["p"].view(np.uint32))
        h.close()


def test_packed_events_in_device_memory_aligned_or_not():
    """A caller's device buffer of 8-byte events: read in place when it starts on a 16-byte boundary and has an even count,
    unpacked by the library first otherwise -- the same peaks either way."""
    import ctypes as C
    from genrich_amd.lib import pack_events
    # (device memory from the HIP runtime the library itself runs on: a second runtime in the process -- torch's -- sees no GPU)
    hip = C.CDLL("libamdhip64.so.7")   # (already mapped: the library links against it)
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    lens = [400_000, 123_457]
    ev = synth.make_fragments(lens, 80_001, 5, peak_every=20_000, tower_every=150_000)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    B.run_case(o, case)
    p8, rest = pack_events(ev)
    assert len(rest) == 0
    for shift, drop in ((0, len(p8) & 1), (1, 0), (0, 1 - (len(p8) & 1))):   # aligned + even, misaligned, odd count
        h = hip_backend(params)   # (first: the context brings the device up)
        n = len(p8) - drop
        buf = C.c_void_p()
        assert hip.hipMalloc(C.byref(buf), 8 * (len(p8) + 2)) == 0
        assert hip.hipMemcpy(buf.value + 8 * shift, p8.ctypes.data, 8 * n, 1) == 0   # hipMemcpyHostToDevice
        h.set_chroms(lens)
        h.sample_begin(0, None)
        h.push_events_packed(buf.value + 8 * shift, where=2, n=n)
        if drop:
            h.push_events(ev[n:])
        h.sample_end()
        h.sample_no_control()
        h.pvalues()
        h.find_peaks()
        in_place = shift == 0 and n % 2 == 0
        assert bool(h.path_info() & 512) == in_place, (shift, n, h.path_info())
        assert h.get_peaks().tobytes() == o.get_peaks().tobytes()
        h.close()
        assert hip.hipFree(buf) == 0


# ---- -E regions on the fused tile stage (round 6; savePileupExpt's bedPos / save walk, Genrich.c:2185-2263) ----

def _bed_case(seed=41):
    """Three chromosomes over several bins, regions of every kind: touching position 0, reaching a chromosome's end (so the
    closing interval lies inside one), spanning whole tiles and whole bins, two edges in one tile, a region of one base."""
    lens = [2_600_000, 1_100_000, 300_000]
    ev = synth.make_fragments(lens, 260_000, seed, peak_every=20_000, tower_every=300_000)
    beds = [[0, 9_000, 50_000, 50_001, 123_000, 123_900, 400_000, 1_500_000, 2_000_000, 2_000_700],
            [5_000, 6_000, 600_000, 640_123, 1_090_000, 1_100_000],
            []]
    return dict(lens=lens, beds=beds, replicates=[dict(save=None, treat=ev, ctrl=None)])


@pytest.mark.parametrize("fused", [True, False])
def test_excluded_regions_on_the_fused_tile_stage_and_on_the_general_chain(monkeypatch, fused):
    if not fused:
        monkeypatch.setenv("GX_NO_BED_FUSED", "1")
    o, h, flags = _run(_bed_case(), B.make_params(pq=0.01, min_auc=20.0))
    assert h.n_peaks > 0
    assert bool(flags & FUSED) == fused and bool(flags & PAIRS) == fused and not flags & FELL_BACK, flags
    assert not flags & LOOSE   # (lambda only comes with the sample's end: the bases inside the regions leave the closed form there)


def test_excluded_regions_with_a_control_and_q_on_the_fused_tile_stage():
    case = _bed_case(43)
    case["replicates"][0]["ctrl"] = synth.make_fragments(case["lens"], 200_000, 44, uniform_only=True)
    o, h, flags = _run(case, B.make_params(pq=0.05, qval=True, min_auc=20.0))
    assert flags & FUSED and flags & PAIRS and not flags & FELL_BACK, flags


def test_a_tower_inside_an_excluded_region_comes_off_fraglen():
    """Thousands of keys in one tile (the whole workgroup's tile outside a region) that lies INSIDE a -E region: nothing is
    emitted, and the pileup over its bases leaves the closed form of fragLen like any other excluded base (round 6: sbt_heavy saw
    an inactive tile and left it in; found by reading, not by a run)."""
    lens = [1_000_000, 300_000]
    ev = synth.make_fragments(lens, 60_000, 47, frac_tower=0.3, tower_every=500_000, peak_every=50_000)
    beds = [[200_000, 300_000, 700_000, 750_010], []]   # the first tower inside a region, the second one on a region's edge
    case = dict(lens=lens, beds=beds, replicates=[dict(save=None, treat=ev, ctrl=None)])
    o, h, flags = _run(case, B.make_params(pq=0.01, min_auc=20.0))
    assert flags & FUSED and flags & PAIRS and not flags & FELL_BACK, flags


# ---- the control merge that scores its own intervals (round 6: k_merge2<.., true> -> k_pairs_missed -> k_pack_ep2) ----

@pytest.mark.parametrize("merge_p", [True, False])
@pytest.mark.parametrize("kind", ["plain", "multimap", "bed", "deep"])
def test_control_merge_with_and_without_its_own_p_values(monkeypatch, merge_p, kind):
    """savePval (Genrich.c:1720-1794) both ways: the merge looks p up itself and leaves (end, p) -- whole pileups below 256 from the
    table, fractional pileups (multimapping on either side) and very deep ones (a tower of > 256 reads) by k_pairs_missed, -E
    regions as SKIP -- or leaves both pileups for k_pack_pairs (GX_NO_MERGE_P, the path until round 5).  assert_same_run asks for
    the pileup floats too: with the merge's own p-values they come from a second merge, run on request."""
    if not merge_p:
        monkeypatch.setenv("GX_NO_MERGE_P", "1")
    lens = [900_000, 250_000, 4_097]
    tr = synth.make_fragments(lens, 120_000, 61, peak_every=20_000, tower_every=150_000,
                              frac_tower=0.25 if kind == "deep" else 0.02)
    ct = synth.make_fragments(lens, 100_000, 62, uniform_only=True)
    if kind == "multimap":
        tr = synth.add_multimap(tr, lens, 0.2, seed=63)
        ct = synth.add_multimap(ct, lens, 0.1, seed=64)
    case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=ct)])
    if kind == "bed":
        case["beds"] = [[0, 5_000, 100_000, 180_500, 890_000, 900_000], [10, 11], []]
    for params in (B.make_params(pq=0.01, min_auc=20.0), B.make_params(pq=0.05, qval=True, min_auc=20.0)):
        o, h, flags = _run(case, params)
        assert bool(flags & MERGE_P) == merge_p, flags
        assert h.path_info() & PILES_MADE   # (assert_same_run asked for the intervals' pileup floats)


# ---- BH's histogram summed by the tight table's kernel (round 6: k_pack_pval<.., HIST> -> k_bh_from_dense) ----

@pytest.mark.parametrize("pack_hist", [True, False])
@pytest.mark.parametrize("kind", ["plain", "deep", "multimap", "bed", "skipped"])
def test_q_values_of_a_single_replicate_from_the_pileup_histogram(monkeypatch, pack_hist, kind):
    """hashPval / computeQval (Genrich.c:300-401) for one replicate without a control: p is a function of the pileup, so the table
    {p -> bp} is made of "bp at V" sums collected while the tight table is written -- whole pileups in LDS, the pileups beyond the
    table p(V) (a tower deeper than 2184 reads) straight into the hash table -- or, as until round 5, by a hash insertion per interval
    (GX_NO_PACK_HIST; also what fractional weights and -E regions take).  The lengths must add up to the genome (377-382: checked on
    the device), and q, the peaks and their AUC must be the oracle's bits."""
    if not pack_hist:
        monkeypatch.setenv("GX_NO_PACK_HIST", "1")
    lens = [700_000, 250_000, 4_097, 90_000]
    tr = synth.make_fragments(lens, 140_000, 71, peak_every=20_000, tower_every=300_000, frac_tower=0.2 if kind == "deep" else 0.02)
    if kind == "multimap":
        tr = synth.add_multimap(tr, lens, 0.2, seed=72)
    case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=None)])
    if kind == "bed":
        case["beds"] = [[0, 5_000, 100_000, 180_500], [10, 11], [], []]
    if kind == "skipped":
        case["skip"] = [False, True, False, False]
    o, h, flags = _run(case, B.make_params(pq=0.05, qval=True, min_auc=20.0))
    assert h.n_peaks > 0
    assert bool(flags & PACK_HIST) == (pack_hist and kind in ("plain", "deep", "skipped")), flags


# ---- q looked up where it is read (round 6: k_sig_from_p + k_q_fill_cands; the whole array on request) ----

@pytest.mark.parametrize("lazy", [True, False])
@pytest.mark.parametrize("kind", ["single", "ctrl", "reps3", "skipped_ctrl"])
def test_q_values_inside_the_candidates_only_and_the_whole_array_on_request(monkeypatch, lazy, kind):
    """computeQval's q never falls as p grows (Genrich.c:392-399: a running minimum over the sorted values), so callPeaks' test
    `q > threshold` (1015) is `p >= the smallest p whose q passes`: the sweep's bits come from one compare per interval, updatePeak's
    q-values (943-970) from the run's {p, q} table for the candidates' intervals only, and gx_get_intervals has the whole array looked
    up when somebody asks (the table of the last run stays until the next one wants it).  GX_NO_LAZY_Q: every interval's q by
    k_qlookup, as until round 5.  Either way: the oracle's q per interval, peaks and AUC bits."""
    if not lazy:
        monkeypatch.setenv("GX_NO_LAZY_Q", "1")
    lens = [600_000, 200_000, 4_097, 70_000]
    reps = []
    for sd in ([81] if kind in ("single", "ctrl", "skipped_ctrl") else [81, 83, 85]):
        tr = synth.make_fragments(lens, 120_000, sd, peak_every=20_000, tower_every=250_000)
        ct = synth.make_fragments(lens, 100_000, sd + 100, peak_every=0, tower_every=0) if kind in ("ctrl", "skipped_ctrl") else None
        reps.append(dict(save=None, treat=tr, ctrl=ct))
    case = dict(lens=lens, replicates=reps)
    if kind == "skipped_ctrl":
        case["skip"] = [False, True, False, False]
    o, h, flags = _run(case, B.make_params(pq=0.05, qval=True, min_auc=20.0))
    assert h.n_peaks > 0
    assert bool(flags & LAZY_Q) == (lazy or bool(flags & Q_LOOSE)), flags   # (-q on the loose slots never makes the array unasked)
    # a second sweep on the same context: the first one's table is released and built again, q asked for AFTER the peaks
    h.find_peaks()
    assert h.get_peaks().tobytes() == o.get_peaks().tobytes()
    for c in range(len(lens)):
        if case.get("skip", [False] * len(lens))[c]:
            continue
        eh, ch = h.get_intervals(-1, c, piles=False)
        eo, co = o.get_intervals(-1, c)
        assert np.array_equal(eh, eo)
        assert np.array_equal(ch["q"].view(np.uint32), co["q"].view(np.uint32))


# ---- the loose slots swept although lambda came with the sample's end (round 6: k_loose_late) ----

@pytest.mark.parametrize("kind", ["multimap", "deep", "empty_tiles", "knob_off"])
def test_fractional_weights_sweep_the_loose_slots_once_the_table_is_there(kind):
    """Fractional weights (-s: addFrac / subFrac, Genrich.c:2300-2376) take the closed form of fragLen away, so lambda -- and with it
    from which pileup on an interval is significant -- is only known behind the tile stage.  The sweep still walks the intervals where
    savePileupExpt (2197-2273) left them: one pass over the pileups writes the significance bits and the fillers of the tiles' unused
    slots, and p comes from the whole table p(V) (a pileup like 7 1/3 has no entry in the compact one).  A pileup beyond the table
    (`deep`) or the switch send the run to the tight table as before.  Either way the oracle's bits."""
    lens = [500_000, 150_000, 4_097, 60_000]
    if kind == "empty_tiles":
        lens = [2_000_000, 150_000]   # (long stretches without a fragment: tiles without intervals, filled by k_scan_iv)
    unit = synth.make_fragments(lens, 40_000 if kind == "empty_tiles" else 110_000, 91, peak_every=20_000, tower_every=250_000,
                                frac_tower=0.2 if kind == "deep" else 0.02)
    ev = synth.add_multimap(unit, lens, 0.3, 92)
    params = B.make_params(pq=0.01, min_auc=20.0)
    h = hip_backend(params)
    h.expect_fractional(True)
    if kind == "knob_off":
        h.set_knob("GX_NO_LATE_LOOSE", 1)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    o = B.Oracle(params)
    so = B.run_case(o, case)
    for run in range(2):   # (the first run learns that the data holds fractions -- its early lambda does not stand --, the second knows)
        h.reset()
        sh = B.run_case(h, case)
        flags = h.path_info()
        assert_same_run(o, h, so, sh, case)
        assert flags & FUSED and flags & FRAC_PAIRS and not flags & FELL_BACK, flags
        if run == 1:
            want = kind in ("multimap", "empty_tiles")
            assert bool(flags & LOOSE) == want and bool(flags & LATE_LOOSE) == want, flags
    assert h.n_peaks > 0
    o.close()


# ---- -q on the loose slots (round 6: k_pack_pval<.., HIST, false>, k_bh_small, k_loose_late, k_peak_both<.., PVQ>) ----

@pytest.mark.parametrize("kind", ["plain", "skipped", "fault", "deep", "twice"])
def test_q_values_on_the_loose_slots_without_a_tight_table(monkeypatch, kind):
    """One replicate without a control and -q: p is a function of the pileup, and so is q (computeQval 352-401 maps equal p to equal
    q) -- BH's histogram is summed from the loose slots, q tabulated by whole pileup, callPeaks' test (1015) becomes "from this
    pileup on", and updatePeak (943-970) takes p and q from two LDS tables.  The tight table is made when gx_get_intervals asks.
    `fault`: the device is told that q is no threshold on the pileup -- the run is repeated on the tight table, and the context
    stays there; `deep`: a pileup beyond the table p(V) takes the tight table from the start.  The oracle's bits every time."""
    if kind == "fault":
        monkeypatch.setenv("GX_FAULT", "2")
    lens = [700_000, 250_000, 4_097, 90_000]
    tr = synth.make_fragments(lens, 140_000, 171, peak_every=20_000, tower_every=300_000, frac_tower=0.2 if kind == "deep" else 0.02)
    case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=None)])
    if kind == "skipped":
        case["skip"] = [False, True, False, False]
    params = B.make_params(pq=0.05, qval=True, min_auc=20.0)
    o, h, flags = _run(case, params)
    assert h.n_peaks > 0
    assert bool(flags & Q_LOOSE) == (kind in ("plain", "skipped", "twice")), flags
    assert flags & PACK_HIST or kind == "fault" or kind == "deep", flags
    if kind == "twice":   # the same context once more: the intervals were asked for, so the replicate has its tight table now
        h.find_peaks()
        assert h.get_peaks().tobytes() == o.get_peaks().tobytes()
        assert not h.path_info() & Q_LOOSE
        h.reset()
        sh = B.run_case(h, case)
        assert h.path_info() & Q_LOOSE
        assert h.get_peaks().tobytes() == o.get_peaks().tobytes()
        h.find_peaks()    # ... and twice in a row on the loose slots
        assert h.path_info() & Q_LOOSE
        assert h.get_peaks().tobytes() == o.get_peaks().tobytes()
