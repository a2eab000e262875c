"""Pins the CPU oracle (oracle/genrich_oracle.c) to the reference:
 * byte-identical narrowPeak / -f / -k text on every committed golden fixture (the fixtures
   are the unmodified reference's own outputs, tests/golden/make_golden.py);
 * the calcPval known answers printed in the reference's README (README.md:243-249);
 * (build container only, marker `ref`) scalar functions and the fraction codec probed
   against the compiled reference in oracle/_ref/libgenrich_ref.so.
"""
import ctypes as C
import os

import numpy as np
import pytest

import backends as B
import golden_cases as G


@pytest.mark.parametrize("name", G.case_names())
def test_oracle_matches_reference_outputs(name, tmp_path):
    meta, case, params, names = G.load_case(name)
    o = B.Oracle(params)
    o.set_chroms(case["lens"], case["skip"], case["beds"])
    pile = str(tmp_path / "pile")
    lams, facs = [], []
    for r, (rep, rm) in enumerate(zip(case["replicates"], meta["replicates"])):
        o.sample_begin(0, rep["save"])
        o.push_events(rep["treat"])
        o.sample_end()
        if rep["ctrl"] is not None:
            o.sample_begin(1, None)
            o.push_events(rep["ctrl"])
            _, lam, fac = o.sample_end()
            facs.append(fac)
        else:
            lam = o.sample_no_control()
        lams.append(lam)
        cname = None if rm["control"] is None else (
            "null" if rm["control"] == "null" else meta["tmp_prefix"] + rm["ctrl_name"])
        o.pvalues_to(pile, r > 0, names, meta["tmp_prefix"] + rm["expt_name"], cname)
    peaks_opt = "-X" not in meta["args"]
    npk, glen, bp = o.find_peaks_to(str(tmp_path / "np"), str(tmp_path / "log"), names, peaks_opt)

    assert [f"{v:f}" for v in lams] == [f"{v:f}" for v in meta["ref_lambda"]]
    assert [f"{v:f}" for v in facs] == [f"{v:f}" for v in meta["ref_factor"]]
    assert glen == meta["ref_genome_len"][0]
    assert open(pile, "rb").read() == G.read_gz(name, "out.pile")
    assert open(tmp_path / "log", "rb").read() == G.read_gz(name, "out.log")
    if peaks_opt:
        assert [npk, bp] == meta["ref_peaks"][0]
        assert open(tmp_path / "np", "rb").read() == G.read_gz(name, "out.narrowPeak")
        if name not in ("nopeaks_log",):
            assert npk > 0, "fixture should exercise the peak sweep"


def test_calc_pval_readme_known_answers():
    """README.md:243-249: seven -f rows of a real hg19 run, control 2.477916."""
    lib = B.Oracle.lib()
    kat = {33: 3.183460, 34: 3.231466, 35: 3.278469, 36: 3.324516,
           39: 3.457329, 40: 3.499948, 41: 3.541798}
    for e, want in kat.items():
        got = lib.gxo_calc_pval(float(e), 2.477916)
        assert abs(got - want) <= 1.5e-6, (e, got, want)


def test_calc_pval_special_values():
    """calcPval's early exits (Genrich.c:1629-1634) and SURVEY Appendix B probes."""
    lib = B.Oracle.lib()
    fmax = float(np.finfo(np.float32).max)
    assert lib.gxo_calc_pval(5.0, -1.0) == -1.0
    assert lib.gxo_calc_pval(0.0, 0.0) == 0.0
    assert lib.gxo_calc_pval(1.0, 0.0) == fmax
    assert lib.gxo_calc_pval(0.0, 3.0) == 0.0
    assert abs(lib.gxo_calc_pval(30.0, 10.0) - 1.38409615) < 1e-6
    assert abs(lib.gxo_calc_pval(5.0, 10.0) - 0.179538146) < 1e-7
    assert abs(lib.gxo_calc_pval(1e6, 1.0) - 51.0932045) < 1e-5
    assert abs(lib.gxo_pchisq(2 * 7 / np.log10(np.e), 4) - 5.76654455) < 1e-7
    assert abs(lib.gxo_pchisq(2 * 0.6 / np.log10(np.e), 6) - 0.076787925) < 1e-8


def test_getval_fraction_text():
    """pileup floats of thirds print as the reference's."""
    lib = B.Oracle.lib()
    neg = C.c_int(0)
    assert f"{lib.gxo_getval(40, C.byref(neg)):f}" == "0.333333"
    assert f"{lib.gxo_getval(160, C.byref(neg)):f}" == "1.333333"
    assert f"{lib.gxo_getval(1160, C.byref(neg)):f}" == "9.666667"
    assert lib.gxo_getval(120 * 77, C.byref(neg)) == 77.0 and not neg.value


# ---- probes against the compiled reference (build container only) ----------------

def _ref():
    lib = C.CDLL(B.REF_SO)
    lib.calcPval.restype = C.c_float
    lib.calcPval.argtypes = [C.c_float, C.c_float]
    lib.pchisq.restype = C.c_double
    lib.pchisq.argtypes = [C.c_double, C.c_int]
    lib.updateVal.restype = C.c_float
    lib.updateVal.argtypes = [C.c_int16, C.c_uint8, C.POINTER(C.c_int32), C.POINTER(C.c_uint8)]
    lib.addFrac.argtypes = [C.POINTER(C.c_int16), C.POINTER(C.c_uint8), C.c_uint8]
    lib.subFrac.argtypes = [C.POINTER(C.c_int16), C.POINTER(C.c_uint8), C.c_uint8]
    return lib


@pytest.mark.ref
def test_calc_pval_bitexact_vs_reference():
    ref, lib = _ref(), B.Oracle.lib()
    rng = np.random.default_rng(5)
    expt = np.concatenate([rng.integers(0, 3000, 4000) / 120.0 * rng.integers(1, 121, 4000),
                           rng.random(2000) * 50, [0, 1, 7, 7.0000005, 1e6, 3e38]]).astype(np.float32)
    ctrl = np.concatenate([rng.random(4000) * 30, rng.random(2000) * 8,
                           [0, -1, 7, 7.0000005, 1e-30, 6.9999995]]).astype(np.float32)
    for e, c in zip(expt, ctrl):
        a, b = ref.calcPval(float(e), float(c)), lib.gxo_calc_pval(float(e), float(c))
        assert np.float32(a).tobytes() == np.float32(b).tobytes(), (e, c, a, b)


@pytest.mark.ref
def test_pchisq_bitexact_vs_reference():
    ref, lib = _ref(), B.Oracle.lib()
    rng = np.random.default_rng(6)
    for _ in range(5000):
        df = int(rng.integers(2, 201)) * 2
        x = float(rng.random() * (3 * df if rng.random() < 0.8 else 0.9))
        a, b = ref.pchisq(x, df), lib.gxo_pchisq(x, df)
        assert a == b or (np.isnan(a) and np.isnan(b)), (x, df, a, b)


@pytest.mark.ref
def test_fraction_codec_is_exact_rational():
    """The (cov, frac) state of the reference after any add/sub sequence, and the running
    state of updateVal, are pure functions of the exact sum in 1/120 units."""
    ref, lib = _ref(), B.Oracle.lib()
    rng = np.random.default_rng(7)
    counts = [2, 3, 4, 5, 6, 8, 10]
    neg = C.c_int(0)
    for _trial in range(300):
        cov, frac = C.c_int16(0), C.c_uint8(0)   # one diff element
        rcov, rfrac = C.c_int32(0), C.c_uint8(0)  # running pileup state
        v = 0
        run_v = int(rng.integers(0, 5)) * 120 * 50
        # seed the running state with an integer pileup so it never goes negative
        ref.updateVal(C.c_int16(run_v // 120), C.c_uint8(0), C.byref(rcov), C.byref(rfrac))
        for _step in range(40):
            c = counts[rng.integers(0, len(counts))]
            if rng.random() < 0.5:
                ref.addFrac(C.byref(cov), C.byref(frac), c)
                v += 120 // c
            else:
                ref.subFrac(C.byref(cov), C.byref(frac), c)
                v -= 120 // c
            # element value must equal v/120 exactly:  cov + e/8 + s/6 + t/10
            e, s, t = frac.value & 7, (frac.value >> 3) & 3, (frac.value >> 5) & 7
            assert s < 3 and t < 5
            assert 120 * cov.value + 15 * e + 20 * s + 12 * t == v
            assert (cov.value == 0 and frac.value == 0) == (v == 0)
        # the reference exit()s on a negative canonical integer part (ERRPILE, 1921/1969), which a
        # random +/- sequence can reach even for a non-negative total: probe only when it is >= 0
        want = lib.gxo_getval(run_v + v, C.byref(neg))
        if run_v + v >= 0 and not neg.value:
            got = ref.updateVal(cov, frac, C.byref(rcov), C.byref(rfrac))
            assert np.float32(got).tobytes() == np.float32(want).tobytes()
