"""GPU parity: the HIP library (through its C ABI) against the CPU oracle on the same inputs.
Integer structure (interval ends, pileups, peak coordinates) must be bit-exact; -log10 p/q within
1e-5 (north_star), and we additionally report/require bit-equality where the libm difference
cannot matter."""
import numpy as np
import pytest

import backends as B
import golden_cases as G
import synth

pytestmark = pytest.mark.gpu


def hip_backend(params):
    import genrich_amd
    return genrich_amd.Genrich(params)


def run_both(case, params):
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    sh = B.run_case(h, case)
    return o, h, so, sh


def assert_same_run(o, h, so, sh, case, tol=1e-5):
    for (fo, lo, co), (fh, lh, ch) in zip(so, sh):
        assert fo == fh, ("fragLen", fo, fh)
        assert np.float32(lo).tobytes() == np.float32(lh).tobytes(), ("lambda", lo, lh)
        if co is not None:
            assert np.float32(co).tobytes() == np.float32(ch).tobytes(), ("factor", co, ch)
    assert o.genome_len == h.genome_len
    nbit = 0
    nrep = len(case["replicates"])
    whiches = [-1] + (list(range(nrep)) if nrep > 1 else [])
    for which, c in [(w, c) for w in whiches for c in range(len(case["lens"]))]:
        eo, co = o.get_intervals(which, c)
        eh, chh = h.get_intervals(which, c)
        assert np.array_equal(eo, eh), f"interval ends differ on chrom {c} (array {which})"
        if len(case["replicates"]) == 1:
            assert np.array_equal(co["expt"].view(np.uint32), chh["expt"].view(np.uint32)), "expt pileup bits"
            assert np.array_equal(co["ctrl"].view(np.uint32), chh["ctrl"].view(np.uint32)), "ctrl pileup bits"
        for k in ("p", "q"):
            a, b = co[k].astype(np.float64), chh[k].astype(np.float64)
            fin = np.isfinite(a) & (np.abs(a) < 1e30)
            assert np.array_equal(fin, np.isfinite(b) & (np.abs(b) < 1e30))
            assert np.all(np.abs(a[fin] - b[fin]) <= tol * np.maximum(1.0, np.abs(a[fin]))), k
            assert np.array_equal(co[k][~fin], chh[k][~fin])
            nbit += int((co[k].view(np.uint32) != chh[k].view(np.uint32)).sum())
    po, ph = o.get_peaks(), h.get_peaks()
    assert len(po) == len(ph), (len(po), len(ph))
    for f in ("chrom", "start", "end", "summit"):
        assert np.array_equal(po[f], ph[f]), f
    for f in ("auc", "p", "q"):
        assert np.allclose(po[f], ph[f], rtol=1e-5, atol=1e-5), f
    assert o.peak_bp == h.peak_bp
    return nbit


SUPPORTED = ["basic", "atac", "ctrl_q", "multimap", "atac_odd", "reps3", "reps3_p_missing",
             "ctrl_only_chrom", "nopeaks_log"]


@pytest.mark.parametrize("name", SUPPORTED)
def test_golden_case(name):
    meta, case, params, names = G.load_case(name)
    o, h, so, sh = run_both(case, params)
    nbit = assert_same_run(o, h, so, sh, case)
    assert nbit == 0, f"{nbit} p/q values differ in their last bits from the host-libm oracle"
    if meta["ref_peaks"]:
        assert h.n_peaks == meta["ref_peaks"][0][0]


@pytest.mark.parametrize("qval", [False, True])
def test_random_treatment_only(qval):
    lens = [300_000, 70_001, 16_384, 16_385, 5]
    ev = synth.make_fragments(lens[:4], 60_000, 3, peak_every=20_000, tower_every=100_000)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.05 if qval else 0.01, qval=qval, min_auc=50.0)
    o, h, so, sh = run_both(case, params)
    assert_same_run(o, h, so, sh, case)
    assert h.n_peaks > 0


def test_empty_and_edge_inputs():
    lens = [50_000, 20_000]
    # one fragment only; fragments touching 0 and the chromosome end; duplicates at one base
    ev = np.array([(0, 0, 10, 1), (0, 49_990, 50_000, 1), (0, 49_990, 60_000, 1), (1, 5, 5, 1)]
                  + [(0, 1000, 1200, 1)] * 500, dtype=B.EVENT_DTYPE)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=1.0)
    o, h, so, sh = run_both(case, params)
    assert_same_run(o, h, so, sh, case)


def test_multimap_fractional_weights():
    lens = [120_000, 40_000]
    ev = synth.make_fragments(lens, 20_000, 9, peak_every=10_000, tower_every=60_000)
    ev = synth.add_multimap(ev, lens, 0.3, 10)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o, h, so, sh = run_both(case, params)
    assert_same_run(o, h, so, sh, case)


def test_random_with_control_q():
    lens = [250_000, 100_000, 33_000]
    tr = synth.make_fragments(lens, 50_000, 21, peak_every=20_000, tower_every=90_000, frac_tower=0.1)
    ct = synth.make_fragments(lens, 40_000, 22, uniform_only=True)
    case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=ct)])
    params = B.make_params(pq=0.2, qval=True, min_auc=20.0)
    o, h, so, sh = run_both(case, params)
    assert_same_run(o, h, so, sh, case)
    assert h.n_peaks > 0


def test_random_three_replicates():
    lens = [150_000, 60_000]
    reps = []
    for r in range(3):
        tr = synth.make_fragments(lens, 25_000, 31 + r, peak_every=15_000, tower_every=70_000)
        ct = synth.make_fragments(lens, 20_000, 41 + r, uniform_only=True) if r != 1 else None
        reps.append(dict(save=None, treat=tr, ctrl=ct))
    case = dict(lens=lens, replicates=reps)
    params = B.make_params(pq=0.05, qval=True, min_auc=20.0)
    o, h, so, sh = run_both(case, params)
    assert_same_run(o, h, so, sh, case)
    assert h.n_peaks > 0
