"""GPU parity: the HIP library (through its C ABI) against the CPU oracle on the same inputs.
Everything must be bit-exact: interval ends, pileups, peak coordinates, and -log10 p / q / AUC too
(north_star asks for 1e-5; the library re-evaluates with the host's libm the few p-values whose
double lies next to a float rounding boundary, gx_math.h, so the floats are the reference's)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import backends as B
from genrich_amd.lib import _libm as _LIBM
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
    for k, ((fo, lo, co), (fh, lh, ch)) in enumerate(zip(so, sh)):
        # The device's fragLen is the exact sum of the reference's float products, rounded once: the oracle's exact sum bit
        # for bit; the reference's own double accumulation may round additions past 2^26 (fractional weights only; counted
        # by the oracle).  lambda must agree in every case.
        B.assert_fraglen(o, k, fo, fh)
        assert np.float32(lo).tobytes() == np.float32(lh).tobytes(), ("lambda", lo, lh)
        if co is not None:
            assert np.float32(co).tobytes() == np.float32(ch).tobytes(), ("factor", co, ch)
    assert o.genome_len == h.genome_len
    nbit = {"p": 0, "q": 0}
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
            nbit[k] += int((co[k].view(np.uint32) != chh[k].view(np.uint32)).sum())
    po, ph = o.get_peaks(), h.get_peaks()
    assert len(po) == len(ph), (len(po), len(ph))
    for f in ("chrom", "start", "end", "summit"):
        assert np.array_equal(po[f], ph[f]), f
    for f in ("auc", "p", "q"):
        assert np.array_equal(po[f].view(np.uint32), ph[f].view(np.uint32)), f
    assert o.peak_bp == h.peak_bp
    assert nbit == {"p": 0, "q": 0}, f"{nbit} p/q values differ in their last bits from the host-libm oracle"
    return nbit


SUPPORTED = ["basic", "atac", "ctrl_q", "multimap", "atac_odd", "reps3", "reps3_p_missing",
             "ctrl_only_chrom", "nopeaks_log", "bedx", "bedx_noctrl"]


@pytest.mark.parametrize("name", SUPPORTED)
def test_golden_case(name):
    meta, case, params, names = G.load_case(name)
    o, h, so, sh = run_both(case, params)
    nbit = assert_same_run(o, h, so, sh, case)
    assert nbit == {"p": 0, "q": 0}, f"{nbit} p/q values differ in their last bits from the host-libm oracle"
    if meta["ref_peaks"]:
        assert h.n_peaks == meta["ref_peaks"][0][0]


@pytest.mark.parametrize("name", G.case_names())
def test_golden_text_outputs_byte_identical(name, tmp_path):
    """narrowPeak / -f / -k text produced from the HIP path (gx_emit.cpp over the C ABI) must be
    byte-identical to the files the reference itself wrote (tests/golden/<case>/out.*)."""
    meta, case, params, names = G.load_case(name)
    h = hip_backend(params)
    B.run_case(h, case)
    nrep = len(case["replicates"])
    pile = str(tmp_path / "pile")
    for r, rm in enumerate(meta["replicates"]):
        cname = None if rm["control"] is None else (
            "null" if rm["control"] == "null" else meta["tmp_prefix"] + rm["ctrl_name"])
        h.write_pile(r, names, meta["tmp_prefix"] + rm["expt_name"], cname, pile, append=r > 0)
    peaks_opt = "-X" not in meta["args"]
    h.write_log(nrep, names, params.qval_opt, peaks_opt, params.thr, str(tmp_path / "log"))
    assert open(pile, "rb").read() == G.read_gz(name, "out.pile")
    assert open(tmp_path / "log", "rb").read() == G.read_gz(name, "out.log")
    if peaks_opt:
        h.write_narrowpeak(names, str(tmp_path / "np"))
        assert open(tmp_path / "np", "rb").read() == G.read_gz(name, "out.narrowPeak")


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


# ---- scalar device functions against the host ---------------------------------------------

def test_device_log10f_is_the_hosts():
    """saveQval's log10f (Genrich.c:221, 226): the device restatement must equal the host libm
    bit for bit (k and the genome length are integers >= 1)."""
    h = hip_backend(B.make_params())
    rng = np.random.default_rng(1)
    x = np.concatenate([np.arange(1, 200_001), rng.integers(1, 2**32, 300_000),
                        rng.integers(1, 2**62, 100_000)]).astype(np.float32)
    got = h.selftest(0, x)
    import ctypes as C
    want = np.array([_LIBM.log10f(C.c_float(v)) for v in x[:60_000]], dtype=np.float32)
    assert np.array_equal(got[:60_000].view(np.uint32), want.view(np.uint32))
    # the rest against numpy's double log10 rounded once (differs from libm only on hard cases)
    ref = np.log10(x.astype(np.float64)).astype(np.float32)
    assert np.max(np.abs(got.astype(np.float64) - ref)) < 2e-6


def test_device_calc_pval_vs_oracle():
    """calcPval on a grid of (treatment, control) values: every bit of the host-libm oracle.  The device
    evaluates in double with its own libm; results next to a float rounding boundary ("risky",
    gx_math.h RISK_B = 2^-40) are re-evaluated on the host.  The test also measures what that margin
    rests on: the device's doubles differ from the host's by far less than RISK_B."""
    from genrich_amd.lib import selftest_host
    h = hip_backend(B.make_params())
    lib = B.Oracle.lib()
    rng = np.random.default_rng(2)
    n = 400_000
    expt = (rng.integers(0, 400_000, n) / 120.0).astype(np.float32)
    ctrl = np.where(rng.random(n) < 0.5, rng.random(n) * 7.5, rng.random(n) * 300).astype(np.float32)
    ctrl[:10] = [0, -1, 7, 7.0000005, 1e-30, 6.9999995, 3, 3, 3, 3]
    expt[:10] = [5, 5, 0, 1, 1, 1, 0, 1e6, 3e38, 1e-3]
    got, dd, nrisky = h.selftest2(1, expt, ctrl)
    hw, hd = selftest_host(1, expt, ctrl)   # (== the oracle: tests/test_abi.py)
    nbad = int((got.view(np.uint32) != hw.view(np.uint32)).sum())
    assert nbad == 0, f"{nbad} of {n} p-values differ in the last bit"
    want = np.array([lib.gxo_calc_pval(float(e), float(c)) for e, c in zip(expt[:100_000], ctrl[:100_000])], dtype=np.float32)
    assert np.array_equal(got[:100_000].view(np.uint32), want.view(np.uint32))
    ok = (hd > 1e-45) & (hd < 1e30)   # results that are not zero as a float
    rel = np.abs(dd[ok] - hd[ok]) / hd[ok]
    print(f"risky: {nrisky} of {n}; max relative device/host difference of the doubles: {rel.max():.3g}")
    assert rel.max() < 2.0 ** -41, "the device's doubles are too far from the host's for the RISK_B = 2^-38 margin"
    assert 0 < nrisky < n * 1e-3


def test_device_fisher_vs_host():
    from genrich_amd.lib import selftest_host
    h = hip_backend(B.make_params())
    rng = np.random.default_rng(21)
    n = 200_000
    sums = (rng.random(n) * 60).astype(np.float32)
    sums[:1000] = (rng.random(1000) * 0.9).astype(np.float32)
    dfs = (2 * rng.integers(1, 17, n)).astype(np.float32)
    got, dd, nrisky = h.selftest2(3, sums, dfs)
    hw, hd = selftest_host(3, sums, dfs)
    assert np.array_equal(got.view(np.uint32), hw.view(np.uint32))
    ok = (hd > 1e-45) & (hd < 1e30)
    rel = np.abs(dd[ok] - hd[ok]) / hd[ok]
    print(f"risky: {nrisky} of {n}; max relative device/host difference of the doubles: {rel.max():.3g}")
    assert rel.max() < 2.0 ** -41


def test_device_fisher_closed_form_rounds_to_the_reference_algorithms_floats():
    """The merge kernels combine replicates through the closed form of the even-df chi-squared tail (gx_math.h fisher_fast),
    not through the reference's pgamma series: every float must still be the one the reference's algorithm gives with the
    host's libm (the values next to a rounding boundary are the host's), and the two doubles must stay far inside the margin
    behind that rule (RISK_B = 2^-38; measured on the CPU: 0.012 x)."""
    from genrich_amd.lib import selftest_host
    h = hip_backend(B.make_params())
    rng = np.random.default_rng(22)
    n = 400_000
    sums = np.concatenate([(rng.random(n // 4) * 60), 10.0 ** (-6 + 8 * rng.random(n // 4)), rng.random(n // 4) * 3.0,
                           10.0 ** (rng.random(n // 4) * 38.0)]).astype(np.float32)
    dfs = (2 * rng.integers(2, 33, n)).astype(np.float32)
    got, dd, nrisky = h.selftest2(4, sums, dfs)
    want, hd = selftest_host(3, sums, dfs)          # the reference's algorithm, this machine's libm
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ok = (hd > 1e-40) & (hd < 1e30)
    rel = np.abs(dd[ok] - hd[ok]) / hd[ok]
    print(f"risky: {nrisky} of {n}; max relative difference closed form (device) / pgamma series (host): {rel.max():.3g}")
    assert 0 < nrisky < n // 1000
    assert rel.max() < 2.0 ** -41


def test_device_getval_all_residues():
    h = hip_backend(B.make_params())
    lib = B.Oracle.lib()
    import ctypes as C
    v = np.concatenate([np.arange(0, 120 * 40), np.arange(120 * 5000, 120 * 5000 + 360)]).astype(np.int32)
    got = h.selftest(2, v.view(np.float32))
    neg = C.c_int(0)
    want = np.array([lib.gxo_getval(int(x), C.byref(neg)) for x in v], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_pileups_beyond_the_lookup_tables():
    """Towers deeper than the p-value tables (2^18 / 2^16 in 1/120 units = 2184x / 546x): the
    direct evaluation must give the same bits as the tabulated one."""
    lens = [60_000, 30_000]
    base = synth.make_fragments(lens, 6_000, 17, peak_every=10_000, tower_every=40_000)
    tower = np.array([(0, 20_000 + (i % 7), 20_200 + (i % 11), 1) for i in range(3_000)]
                     + [(1, 5_000, 5_300, 1)] * 700, dtype=B.EVENT_DTYPE)
    tr = np.concatenate([base, tower])
    ct = np.concatenate([synth.make_fragments(lens, 5_000, 18, uniform_only=True),
                         np.array([(0, 20_050, 20_150, 1)] * 900, dtype=B.EVENT_DTYPE)])
    for ctrl in (None, ct):
        case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=ctrl)])
        params = B.make_params(pq=0.05, qval=ctrl is not None, min_auc=20.0)
        o, h, so, sh = run_both(case, params)
        assert_same_run(o, h, so, sh, case)
        e, cols = h.get_intervals(-1, 0)
        assert cols["expt"].max() > 2500


def test_wide_record_path(tmp_path):
    """Genomes with more tiles than a 4-byte key can address route every event through the
    8-byte record stream; GX_FORCE_REC64 forces that path on small inputs."""
    import subprocess
    import sys
    import os
    code = (
        "import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.');\n"
        "import numpy as np, backends as B, golden_cases as G, genrich_amd\n"
        "for name in ('ctrl_q', 'multimap', 'bedx'):\n"
        "    meta, case, params, names = G.load_case(name)\n"
        "    o = B.Oracle(params); B.run_case(o, case)\n"
        "    h = genrich_amd.Genrich(params); B.run_case(h, case)\n"
        "    assert o.get_peaks().tobytes() == h.get_peaks().tobytes(), name\n"
        "    for c in range(len(case['lens'])):\n"
        "        eo, co = o.get_intervals(-1, c); eh, ch = h.get_intervals(-1, c)\n"
        "        assert np.array_equal(eo, eh) and np.array_equal(co['p'].view(np.uint32), ch['p'].view(np.uint32)), name\n"
        "print('wide ok')\n")
    env = dict(os.environ, GX_FORCE_REC64="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert res.returncode == 0 and "wide ok" in res.stdout, res.stderr[-2000:]


def test_fraglen_closed_form_guards():
    """fragLen is normally the closed form (sum of fragment lengths, k_convert) plus a correction
    for the few intervals whose float product len * val rounds: covered stretches longer than two
    tiles (3 x 6,000,001 = 18,000,003 is not a float) and deep pileups (20,001 x 999)."""
    lens = [300_000, 9_000_000]
    long3 = np.array([(1, 1_000_000, 7_000_001, 1)] * 3, dtype=B.EVENT_DTYPE)  # nothing else on chrom 1
    bg = synth.make_fragments(lens[:1], 4000, seed=3)
    deep = np.array([(0, 100_000, 100_150, 1)] * 1500 + [(1, 8_000_000, 8_000_999, 1)] * 20_001, dtype=B.EVENT_DTYPE)
    for extra in (long3, deep, np.concatenate([long3, deep])):
        tr = np.concatenate([bg, extra])
        case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=bg)])
        o, h, so, sh = run_both(case, B.make_params(pq=0.01, min_auc=20.0))
        assert_same_run(o, h, so, sh, case)
        closed = float((tr["end"].astype(np.int64) - tr["start"]).sum())
        assert so[0][0] != closed  # the correction, not luck: the reference's rounded sum is not the closed form


def test_fraglen_general_path_on_unit_data():
    """GX_FORCE_SLOWFRAG routes ordinary unit-weight data through the per-interval walk: same
    fragLen / lambda / peaks as the oracle (and hence as the closed form)."""
    import subprocess
    import sys
    import os
    code = (
        "import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, '.');\n"
        "import numpy as np, backends as B, golden_cases as G, genrich_amd\n"
        "for name in ('basic', 'ctrl_q', 'atac'):\n"
        "    meta, case, params, names = G.load_case(name)\n"
        "    o = B.Oracle(params); so = B.run_case(o, case)\n"
        "    h = genrich_amd.Genrich(params); sh = B.run_case(h, case)\n"
        "    assert [x[0] for x in so] == [x[0] for x in sh], (name, so, sh)\n"
        "    assert o.get_peaks().tobytes() == h.get_peaks().tobytes(), name\n"
        "print('slow ok')\n")
    env = dict(os.environ, GX_FORCE_SLOWFRAG="1")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert res.returncode == 0 and "slow ok" in res.stdout, res.stderr[-2000:]


def test_giant_candidate_takes_the_wavefront_walk():
    """A peak of several thousand intervals is walked by a whole wavefront (k_peak_walk); ordinary
    ones by one thread each (k_peak_short).  Same in-order float AUC either way."""
    rng = np.random.default_rng(11)
    lens = [2_000_000]
    bg = synth.make_fragments(lens, 15_000, seed=5)
    st = rng.integers(500_000, 560_000, 40_000).astype(np.uint32)
    ln = rng.integers(150, 400, 40_000).astype(np.uint32)
    big = np.zeros(len(st), dtype=B.EVENT_DTYPE)
    big["chrom"], big["start"], big["end"], big["count"] = 0, st, st + ln, 1
    case = dict(lens=lens, replicates=[dict(save=None, treat=np.concatenate([bg, big]), ctrl=None)])
    o, h, so, sh = run_both(case, B.make_params(pq=0.01, min_auc=20.0))
    assert_same_run(o, h, so, sh, case)
    pk = h.get_peaks()
    e, _ = h.get_intervals(-1, 0)
    widest = max(np.searchsorted(e, q["end"]) - np.searchsorted(e, q["start"]) for q in pk)
    assert widest > 1024, widest
    assert np.array_equal(o.get_peaks()["auc"].view(np.uint32), pk["auc"].view(np.uint32))


def test_crowded_tile_takes_the_chunked_bucket_path():
    """More endpoint records in one super-bucket than the one-pass LDS sort holds (32,768): the
    level-2 bucket kernel falls back to its chunked two-pass path."""
    rng = np.random.default_rng(13)
    lens = [400_000]
    bg = synth.make_fragments(lens, 3000, seed=7)
    st = rng.integers(100_000, 104_000, 70_000).astype(np.uint32)
    ln = rng.integers(100, 200, 70_000).astype(np.uint32)
    crowd = np.zeros(len(st), dtype=B.EVENT_DTYPE)
    crowd["chrom"], crowd["start"], crowd["end"], crowd["count"] = 0, st, st + ln, 1
    case = dict(lens=lens, replicates=[dict(save=None, treat=np.concatenate([bg, crowd]), ctrl=bg)])
    o, h, so, sh = run_both(case, B.make_params(pq=0.01, min_auc=20.0))
    assert_same_run(o, h, so, sh, case)
    assert h.n_peaks >= 1


@pytest.mark.parametrize("name", ["basic", "ctrl_q", "bedx"])
def test_without_pileup_floats(name):
    """gx_set_keep_pileups(0): interval ends, p, q and peaks are what they are with the pileup
    floats kept; asking for the pileups is an error, not a silent zero."""
    meta, case, params, names = G.load_case(name)
    full = hip_backend(params)
    B.run_case(full, case)
    lean = hip_backend(params)
    lean.set_keep_pileups(False)
    B.run_case(lean, case)
    assert full.get_peaks().tobytes() == lean.get_peaks().tobytes()
    for c in range(len(case["lens"])):
        ef, cf = full.get_intervals(-1, c)
        el, cl = lean.get_intervals(-1, c, piles=False)
        assert np.array_equal(ef, el)
        assert np.array_equal(cf["p"].view(np.uint32), cl["p"].view(np.uint32))
        assert np.array_equal(cf["q"].view(np.uint32), cl["q"].view(np.uint32))
    with pytest.raises(RuntimeError):
        lean.get_intervals(-1, 0)


def test_thousands_of_small_contigs():
    """An assembly with alternate / unplaced contigs: 2,500 chromosomes, many shorter than one tile,
    some without any read, some skipped (-e)."""
    rng = np.random.default_rng(17)
    lens = [int(x) for x in rng.integers(300, 30_000, 2500)]
    lens[7] = 1  # a one-base contig
    ev = synth.make_fragments(lens, 400_000, seed=9)
    ev = ev[(ev["chrom"] % 11) != 3]  # contigs without reads
    ct = synth.make_fragments(lens, 300_000, seed=10, uniform_only=True)
    skip = [(i % 97) == 5 for i in range(len(lens))]
    case = dict(lens=lens, skip=skip, replicates=[dict(save=None, treat=ev, ctrl=ct)])
    for qval in (False, True):
        params = B.make_params(pq=0.05 if qval else 0.01, qval=qval, min_auc=10.0)
        o, h, so, sh = run_both(case, params)
        assert_same_run(o, h, so, sh, case)
        assert qval or h.n_peaks > 1000


@pytest.mark.parametrize("name", ["basic", "atac", "bedx_noctrl"])
def test_q_values_without_control(name):
    """-q on a control-less replicate (the BH table over a p array that came from the pack kernel)."""
    meta, case, _, names = G.load_case(name)
    params = B.make_params(pq=0.2, qval=True, min_auc=5.0)
    o, h, so, sh = run_both(case, params)
    nbit = assert_same_run(o, h, so, sh, case)
    assert nbit == {"p": 0, "q": 0}
    qs = np.concatenate([h.get_intervals(-1, c)[1]["q"] for c in range(len(case["lens"]))])
    assert len(np.unique(qs)) > 3


def test_q_values_without_control_deep_and_fractional():
    """... with fractional pileups (-s) and pileups beyond the p-value table."""
    lens = [300_000]
    bg = synth.make_fragments(lens, 6000, seed=21)
    mm = synth.add_multimap(bg, lens, 0.3, seed=22)
    deep = np.array([(0, 150_000, 150_200, 1)] * 2600, dtype=B.EVENT_DTYPE)
    case = dict(lens=lens, replicates=[dict(save=None, treat=np.concatenate([mm, deep]), ctrl=None)])
    o, h, so, sh = run_both(case, B.make_params(pq=0.2, qval=True, min_auc=5.0))
    assert_same_run(o, h, so, sh, case)
    e, cols = h.get_intervals(-1, 0)
    assert cols["expt"].max() > 2500 and (cols["expt"] % 1 != 0).any()


def test_events_from_several_host_and_device_segments():
    """One sample fed in pieces: host pushes and device-resident segments of odd sizes (the level-1
    chunk bookkeeping of k_convert runs over the concatenation)."""
    import ctypes as C
    # plain HIP runtime calls -- through the very runtime instance the library under test is linked
    # to (a process may hold a second copy, e.g. the one bundled with torch)
    hip_backend(B.make_params(pq=0.01)).close()
    paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
    rocm = [q for q in paths if "/torch/" not in q] or paths
    hip = C.CDLL(rocm[0])
    lens = [3_000_000, 1_500_000]
    ev = synth.make_fragments(lens, 60_000, seed=31)
    ct = synth.make_fragments(lens, 50_000, seed=32, uniform_only=True)
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=ct)])
    params = B.make_params(pq=0.01, min_auc=20.0)
    o = B.Oracle(params)
    so = B.run_case(o, case)
    h = hip_backend(params)
    h.set_chroms(lens)

    def feed(events):
        cuts = [0, 9_001, 9_001 + 8_192, 30_011, len(events)]
        keep = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = events[a:b]
            if (a // 7) % 2 == 0:
                h.push_events(part)
            else:
                raw = np.ascontiguousarray(part).tobytes()
                ptr = C.c_void_p()
                assert hip.hipMalloc(C.byref(ptr), C.c_size_t(len(raw))) == 0
                assert hip.hipMemcpy(ptr, raw, C.c_size_t(len(raw)), 1) == 0  # synchronous host -> device
                keep.append(ptr)  # must stay allocated until the sample is closed
                h.push_events_device(ptr.value, len(part))
        return keep

    h.sample_begin(0, None)
    k1 = feed(ev)
    frag, _, _ = h.sample_end()
    h.sample_begin(1, None)
    k2 = feed(ct)
    _, lam, fac = h.sample_end()
    h.pvalues()
    h.find_peaks()
    for ptr in k1 + k2:
        hip.hipFree(ptr)
    assert_same_run(o, h, so, [(frag, lam, fac)], case)
    assert h.n_peaks > 0


def _random_case(seed, scale=1):
    """scale > 1: the same mix on chromosomes / samples `scale` times larger (tools/fuzz_hip_vs_oracle.py)"""
    rng = np.random.default_rng(seed)
    nch = int(rng.integers(1, 6))
    lens = [int(x) for x in rng.integers(500 * scale, 120_000 * scale, nch)]
    skip = [bool(rng.random() < 0.15) for _ in lens]
    if all(skip):
        skip[0] = False
    beds = []
    for L, sk in zip(lens, skip):
        regs = []
        if not sk and rng.random() < 0.3:
            for _ in range(int(rng.integers(1, 4))):
                s = int(rng.integers(0, max(1, L - 5)))
                regs.append((s, min(L, s + int(rng.integers(1, 3000 * scale)))))
        regs.sort()
        merged = []
        for s, e in regs:  # merged, clipped, as saveXBed leaves them (Genrich.c:1144-1206)
            if merged and s <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], e)
            else:
                merged.append([s, e])
        beds.append([v for r in merged for v in r])
    reps = []
    for r in range(int(rng.choice([1, 1, 2, 3]))):
        n = int(rng.integers(200 * scale, 6000 * scale))
        tr = synth.make_fragments(lens, n, seed=seed * 7 + r, frac_peak=0.4, frac_tower=0.2)
        if rng.random() < 0.3:
            tr = synth.add_multimap(tr, lens, 0.25, seed=seed + 11)
        ct = None
        if rng.random() < 0.5:
            ct = synth.make_fragments(lens, int(rng.integers(200 * scale, 6000 * scale)), seed=seed * 7 + 3 + r, uniform_only=True)
        reps.append(dict(save=None, treat=tr, ctrl=ct))
    qval = bool(rng.random() < 0.5)
    params = B.make_params(pq=float(rng.choice([0.3, 0.1, 0.05])) if qval else float(rng.choice([0.05, 0.01, 0.001])),
                           qval=qval, min_auc=float(rng.choice([1, 10, 50])), min_len=int(rng.choice([0, 0, 80])),
                           max_gap=int(rng.choice([0, 100, 250])))
    return dict(lens=lens, skip=skip, beds=beds, replicates=reps), params


@pytest.mark.parametrize("block", range(4))
def test_random_runs_against_oracle(block):
    """40 random runs: 1-5 chromosomes (some skipped, some with -E regions), 1-3 replicates with or
    without control, multimapping, -p / -q, assorted -a / -l / -g."""
    for seed in range(block * 10, block * 10 + 10):
        case, params = _random_case(1000 + seed)
        try:
            o, h, so, sh = run_both(case, params)
        except RuntimeError as ex:  # both must refuse the same inputs (e.g. a sample without fragments)
            with pytest.raises(RuntimeError):
                B.run_case(B.Oracle(params), case)
            continue
        assert_same_run(o, h, so, sh, case)


def _saturating_case(seed, frac):
    """> 32,767 alignments starting on one base, ending on one base, ending at a chromosome's last
    position, and ending on the hot start base, in random order (optionally with multimapped reads on
    the hot bases): the reference's int16 difference array saturates and saveInterval starts dropping
    alignments (Genrich.c:2558-2573)."""
    rng = np.random.default_rng(seed)
    lens = [20_000, 8_000]

    def blk(n, c, s, e, cnt=1):
        a = np.zeros(n, dtype=B.EVENT_DTYPE)
        a["chrom"], a["start"], a["end"], a["count"] = c, s, e, cnt
        return a

    n1, n2, n3, n4 = (int(rng.integers(33_000, 42_000)), int(rng.integers(33_000, 42_000)),
                      int(rng.integers(33_000, 40_000)), int(rng.integers(1_000, 20_000)))
    parts = [synth.make_fragments(lens, 2000, seed=seed),
             blk(n1, 0, 5000, 5000 + rng.integers(100, 300, n1)), blk(n2, 0, 9000 - rng.integers(100, 300, n2), 9000),
             blk(n3, 1, 8000 - rng.integers(100, 300, n3), 8100), blk(n4, 0, 5000 - rng.integers(100, 300, n4), 5000)]
    if frac:
        k = int(rng.integers(5_000, 30_000))
        parts.append(blk(k, 0, 5000, 5000 + rng.integers(100, 300, k), cnt=rng.choice([2, 3, 4, 5, 6, 8, 10], k)))
        k = int(rng.integers(5_000, 30_000))
        parts.append(blk(k, 0, 9000 - rng.integers(100, 300, k), 9000, cnt=rng.choice([2, 3, 4, 5, 6, 8, 10], k)))
    ev = np.concatenate(parts)
    return lens, ev[rng.permutation(len(ev))]


@pytest.mark.parametrize("seed", range(4))
def test_int16_saturation_skips_as_in_the_reference(seed):
    """The device only finds out that some base can saturate (k_hot_check); which alignments are dropped
    depends on their order and is replayed on the host (gx_saturate.h) before the sample is built
    again.  Pileups, p-values and peaks must be the oracle's, which drops alignments as the reference does
    (pinned against the reference binary by tools/fuzz_oracle_vs_reference.py --saturate)."""
    lens, ev = _saturating_case(seed, frac=bool(seed % 2))
    ctrl = synth.make_fragments(lens, 3000, seed=seed + 50, uniform_only=True) if seed >= 2 else None
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=ctrl)])
    o, h, so, sh = run_both(case, B.make_params(pq=0.01, min_auc=20.0))
    assert_same_run(o, h, so, sh, case)


def test_window_net_is_the_open_samples_difference_array():
    """gx_window_net: the exact diff[] of saveInterval on a window, from the events pushed so far (several pushes, fractional
    weights, ends clamped to the chromosome's length, rejected events left out) -- what genrich-amd's read-by-read int16
    checks start from."""
    lens = [20_000, 8_000]
    _, ev = _saturating_case(1, frac=True)
    extra = np.zeros(3, dtype=B.EVENT_DTYPE)
    extra["chrom"], extra["start"], extra["end"], extra["count"] = [1, 1, 0], [7_900, 100, 19_000], [9_000, 200, 30_000], [1, 2, 4]
    ev = np.concatenate([ev, extra])
    h = hip_backend(B.make_params(pq=0.01))
    h.set_chroms(lens)
    h.sample_begin(0, None)
    w = np.array([0, 120, 60, 40, 30, 24, 20, 0, 15, 0, 12], dtype=np.int64)
    third = len(ev) // 3
    for part, upto in ((ev[:third], third), (ev[third:], len(ev))):
        h.push_events(part)
        seen = ev[:upto]
        for c, pos0, n in ((0, 4096, 4096), (0, 8192, 4096), (1, 4096, 8_001 - 4096), (0, 16384, 20_001 - 16384)):
            want = np.zeros(n, dtype=np.int64)
            m = seen[seen["chrom"] == c]
            end = np.minimum(m["end"], lens[c])
            ws = w[m["count"]]
            a = (m["start"] >= pos0) & (m["start"] < pos0 + n)
            np.add.at(want, m["start"][a] - pos0, ws[a])
            b = (end >= pos0) & (end < pos0 + n)
            np.add.at(want, end[b] - pos0, -ws[b])
            assert np.array_equal(h.window_net(c, pos0, n), want), (c, pos0)
    h.sample_end()
    with pytest.raises(Exception):
        h.window_net(0, 0, 16)  # (only while a sample is open)


def test_saturation_only_at_the_chromosome_end():
    """The reference's difference array has an entry at `len` as well (every fragment clamped to the end
    of its chromosome lands there); those events have no end record on the device and are counted per
    chromosome by k_convert."""
    lens = [50_000]
    n = 33_000
    st = np.random.default_rng(3).integers(49_000, 49_900, n).astype(np.uint32)
    ev = np.zeros(n, dtype=B.EVENT_DTYPE)
    ev["chrom"], ev["start"], ev["end"], ev["count"] = 0, st, 50_000, 1
    for m in (n, 32_767):
        case = dict(lens=lens, replicates=[dict(save=None, treat=ev[:m], ctrl=None)])
        o, h, so, sh = run_both(case, B.make_params(pq=0.01))
        assert_same_run(o, h, so, sh, case)


def test_intervals_that_end_before_they_start():
    """saveInterval adds +w at start and -w at end whatever their order (the reference's BAM reader produces
    such intervals from reverse reads without SEQ).  Covered by other fragments the run goes through, with
    the pileup lowered between end and start; uncovered it stops with "Invalid pileup value (< 0)"."""
    lens = [30_000]
    bg = synth.make_fragments(lens, 4000, seed=5)
    rng = np.random.default_rng(6)
    deep = np.zeros(3000, dtype=B.EVENT_DTYPE)   # a plateau over [10000, 12000)
    deep["chrom"], deep["start"], deep["end"], deep["count"] = 0, 10_000 - rng.integers(0, 50, 3000), 12_000 + rng.integers(0, 50, 3000), 1
    back = np.zeros(200, dtype=B.EVENT_DTYPE)     # 200 intervals running backwards inside the plateau
    st = rng.integers(10_600, 11_900, 200)
    back["chrom"], back["start"], back["end"], back["count"] = 0, st, st - rng.integers(1, 500, 200), 1
    ev = np.concatenate([bg, deep, back])
    ev = ev[rng.permutation(len(ev))]
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    o, h, so, sh = run_both(case, B.make_params(pq=0.01, min_auc=20.0))
    assert_same_run(o, h, so, sh, case)
    lone = np.zeros(1, dtype=B.EVENT_DTYPE)   # nothing else on its chromosome
    lone["chrom"], lone["start"], lone["end"], lone["count"] = 1, 3_000, 2_900, 1
    for backend in (B.Oracle, hip_backend):
        b = backend(B.make_params(pq=0.01))
        b.set_chroms(lens + [5_000])
        b.sample_begin(0, None)
        b.push_events(np.concatenate([bg[:50], lone]))
        with pytest.raises(RuntimeError, match="Invalid pileup"):
            b.sample_end()


@pytest.mark.parametrize("qval", [False, True])
def test_atac_geometry_with_multimap_weights(qval):
    """configs[3]'s combination: ATAC -j -d 100 cut-site intervals (two per fragment, saveFragAtac 2728-2749)
    carrying -s multimapping weights 1/k, k in {2,3,4,5,6,8,10} (10 % of the fragments): fractional pileups in
    nearly every tile (all tiles take the 32-bit LDS path), the general fragLen path, and p / q on fractional values."""
    lens = [9_000_000, 5_000_000, 1_200_000, 16_569]
    fr = synth.make_fragments(lens, 700_000, 31, peak_every=50_000, tower_every=2_000_000)
    ev = synth.atac_events(synth.add_multimap(fr, lens, 0.10, seed=32), lens, d=100)
    assert len(ev) > 1_500_000 and (ev["count"] > 1).sum() > 200_000
    case = dict(lens=lens, replicates=[dict(save=None, treat=ev, ctrl=None)])
    params = B.make_params(pq=0.05 if qval else 0.01, qval=qval, min_auc=50.0)
    o, h, so, sh = run_both(case, params)
    assert_same_run(o, h, so, sh, case)
    assert h.n_peaks > 50


def test_bh_table_grows_when_full(monkeypatch):
    """The table of distinct p-values starts small here (2^6 slots) and must grow until the run's few thousand
    distinct values fit -- same q-values, same peaks as the oracle."""
    monkeypatch.setenv("GX_BH_CAPLOG", "6")
    lens = [400_000, 150_000]
    tr = synth.make_fragments(lens, 120_000, 41, peak_every=20_000, tower_every=90_000, frac_tower=0.1)
    ct = synth.make_fragments(lens, 90_000, 42, uniform_only=True)
    case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=ct)])
    o, h, so, sh = run_both(case, B.make_params(pq=0.2, qval=True, min_auc=20.0))
    assert_same_run(o, h, so, sh, case)
    assert h.n_peaks > 0


def test_many_distinct_pvalues_take_the_chunked_bh_table(monkeypatch):
    """Three replicates combined by Fisher's method give tens of thousands of distinct p-values: the BH table
    is then built by the chunked kernels (k_qt_sums / k_qt_raw / k_qt_apply), forced here for any count --
    same q-values and peaks as the oracle."""
    monkeypatch.setenv("GX_QT_MULTI", "1")
    lens = [300_000, 120_000]
    reps = [dict(save=None, treat=synth.make_fragments(lens, 90_000, 50 + r, peak_every=15_000, tower_every=80_000,
                                                       frac_tower=0.1), ctrl=None) for r in range(3)]
    case = dict(lens=lens, replicates=reps)
    o, h, so, sh = run_both(case, B.make_params(pq=0.1, qval=True, min_auc=20.0))
    assert_same_run(o, h, so, sh, case)
    assert h.n_peaks > 0


def test_peak_sweep_repeats_when_its_guess_is_too_small():
    """The sweep sizes its run / candidate arrays by a guess and checks the true run count at the end; when the
    guess was too small the pass is thrown away and repeated with arrays that fit.  The first guess is 4 runs
    here (GX_RUN_CAP_MIN; 65,536 normally): same peaks as the oracle, and again on a second run of the same
    context (the guess is remembered)."""
    code = (
        "import os, sys\n"
        "os.environ['GX_RUN_CAP_MIN'] = '4'\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, backends as B, synth, genrich_amd\n"
        "lens = [400_000, 150_000]\n"
        "tr = synth.make_fragments(lens, 120_000, 77, peak_every=20_000, tower_every=90_000, frac_tower=0.1)\n"
        "case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=None)])\n"
        "par = B.make_params(pq=0.05, qval=False, min_auc=20.0)\n"
        "o = B.Oracle(par); B.run_case(o, case)\n"
        "h = genrich_amd.Genrich(par); B.run_case(h, case)\n"
        "assert o.n_peaks > 16, o.n_peaks\n"
        "assert o.get_peaks().tobytes() == h.get_peaks().tobytes()\n"
        "h.reset(); B.run_case(h, case)\n"
        "assert o.get_peaks().tobytes() == h.get_peaks().tobytes()\n"
        "print('OK', o.n_peaks)\n"
    ) % (os.path.join(os.path.dirname(__file__), ".."), os.path.dirname(__file__))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_phase_timers_follow_their_level():
    """gx_set_phase_timing: nothing by default, the tile stage at level 1, every phase at level 2."""
    lens = [200_000]
    tr = synth.make_fragments(lens, 30_000, 5)
    case = dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=None)])
    g = hip_backend(B.make_params(pq=0.01))
    B.run_case(g, case)
    assert g.phase_times() == []
    g.set_phase_timing(1)
    g.reset()
    B.run_case(g, case)
    assert [n for n, _ in g.phase_times()] == ["t.tile"]
    g.set_phase_timing(2)
    g.reset()
    B.run_case(g, case)
    names = [n for n, _ in g.phase_times()]
    assert "t.sort1" in names and "t.tile" in names and "sweep" in names
