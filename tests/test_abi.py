"""CPU-only checks of the drop-in boundary: the HIP library builds, loads, and exports every
symbol that include/genrich_amd.h declares (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "genrich_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(gx_[a-z_0-9]+)\s*\(", src)
    return sorted(set(n for n in names if not n.endswith("_fn")))


def test_library_exports_every_declared_symbol():
    from genrich_amd import build, lib

    path = build.build()
    dll = C.CDLL(path)
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(dll, n), f"{n} declared in include/genrich_amd.h but not exported"
    lib.load_library()


def test_header_cites_reference_for_every_entry_point():
    src = open(os.path.join(ROOT, "include", "genrich_amd.h")).read()
    assert src.count("Genrich.c:") + src.count(" :") > 20


def test_no_device_means_loud_failure():
    """Without a GPU the product must fail loudly, never fall back to a CPU path."""
    import genrich_amd

    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    with pytest.raises(RuntimeError):
        genrich_amd.Genrich(genrich_amd.GxParams(2.0, 0, 200.0, 0, 100, 0, 0))


def test_product_never_references_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "genrich_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".c", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "gxo_" not in txt and "libgenrich_oracle" not in txt, f
                # mentioning the checker in a comment (e.g. check_log10f.c) is fine; importing is not
                assert not re.search(r"^\s*(from|import)\s+.*oracle", txt, flags=re.M), f


def test_host_build_of_the_p_value_routines_equals_the_oracle():
    """The library re-evaluates p-values that lie next to a float rounding boundary with the host build of
    its own routines (gx_math.h is __host__ __device__), i.e. with this machine's libm: those must be
    the oracle's bits everywhere, or the patched values would not be the reference's."""
    import numpy as np
    import backends as B
    from genrich_amd.lib import selftest_host
    lib = B.Oracle.lib()
    rng = np.random.default_rng(12)
    n = 60_000
    expt = (rng.integers(0, 400_000, n) / 120.0).astype(np.float32)
    ctrl = np.where(rng.random(n) < 0.5, rng.random(n) * 7.5, rng.random(n) * 300).astype(np.float32)
    ctrl[:10] = [0, -1, 7, 7.0000005, 1e-30, 6.9999995, 3, 3, 3, 3]
    expt[:10] = [5, 5, 0, 1, 1, 1, 0, 1e6, 3e38, 1e-3]
    got, _ = selftest_host(1, expt, ctrl)
    want = np.array([lib.gxo_calc_pval(float(e), float(c)) for e, c in zip(expt, ctrl)], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # multPval's tail: pchisq of 2 sum / log10(e) over df = 2 x replicates
    sums = (rng.random(20_000) * 40).astype(np.float32)
    dfs = (2 * rng.integers(2, 17, 20_000)).astype(np.float32)
    got, _ = selftest_host(3, sums, dfs)
    want = np.array([min(lib.gxo_pchisq(2.0 * float(s) / 0.434294481903251827651, int(d)), 3.4028234663852886e38) if s else 0.0
                     for s, d in zip(sums, dfs)], dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_closed_form_of_the_fisher_tail_agrees_with_the_reference_algorithm_on_the_host():
    """gx_math.h fisher_fast_double (what the merge kernels evaluate) against fisher_double (multPval's pchisq through pgamma's
    series, Genrich.c:528-583), both compiled for the host: the doubles within 2^-41 of each other over sums in [1e-6, 1e38] and
    every even df up to 64 -- an eighth of the margin (RISK_B = 2^-38) inside which a value is not rounded on the device at all."""
    import numpy as np
    from genrich_amd.lib import selftest_host
    rng = np.random.default_rng(5)
    n = 300_000
    sums = np.concatenate([(rng.random(n // 3) * 60), 10.0 ** (-6 + 8 * rng.random(n // 3)), 10.0 ** (rng.random(n // 3) * 38.0)]).astype(np.float32)
    dfs = (2 * rng.integers(2, 33, n)).astype(np.float32)
    _, fast = selftest_host(4, sums, dfs)
    _, ref = selftest_host(3, sums, dfs)
    ok = (ref > 1e-40) & (ref < 1e30)
    rel = np.abs(fast[ok] - ref[ok]) / ref[ok]
    assert ok.sum() > n // 2 and rel.max() < 2.0 ** -41, rel.max()
