"""ctypes drivers used by the tests: the CPU oracle (checker) and the HIP library (product)
behind one small Python interface, so a parity test is `run(oracle) == run(hip)`.

The oracle is test infrastructure; the product never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libgenrich_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "Genrich")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libgenrich_ref.so")

# shared POD definitions come from the product's binding of include/genrich_amd.h
from genrich_amd.lib import EVENT_DTYPE, PEAK_DTYPE, GxParams, minus_log10f  # noqa: E402,F401


def make_params(pq=0.01, qval=False, min_auc=200.0, min_len=0, max_gap=100, genome_len=0,
                device=0):
    return GxParams(minus_log10f(pq), int(qval), float(min_auc), int(min_len), int(max_gap),
                    int(device), int(genome_len))


def build_oracle():
    if not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(ORACLE_DIR, "genrich_oracle.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libgenrich_oracle.so"])
    return ORACLE_SO


class _Backend:
    """Common driver over a `<prefix>_*` C API (gxo_ = oracle, gx_ = HIP library)."""

    prefix = ""

    def __init__(self, lib, params: GxParams):
        self.lib = lib
        self.ctx = C.c_void_p()
        self._check(self._f("create")(C.byref(self.ctx), C.byref(params)))
        self._keep = []

    def _f(self, name):
        return getattr(self.lib, self.prefix + name)

    def _check(self, rc):
        if rc != 0:
            msg = self._f("last_error")(self.ctx) if self.ctx else b""
            raise RuntimeError(f"{self.prefix}* failed rc={rc}: {(msg or b'').decode()}")

    def close(self):
        if self.ctx:
            self._f("destroy")(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- API mirror ----------------------------------------------------
    def set_chroms(self, lens, skip=None, beds=None):
        n = len(lens)
        self.n_chrom = n
        lens_a = np.ascontiguousarray(lens, dtype=np.uint32)
        skip_a = np.ascontiguousarray(skip if skip is not None else np.zeros(n), dtype=np.uint8)
        bed_ptrs = (C.POINTER(C.c_uint32) * n)()
        bed_len = np.zeros(n, dtype=np.int32)
        keep = []
        if beds is not None:
            for i, b in enumerate(beds):
                arr = np.ascontiguousarray(b, dtype=np.uint32).ravel()
                keep.append(arr)
                bed_len[i] = arr.size
                bed_ptrs[i] = arr.ctypes.data_as(C.POINTER(C.c_uint32))
        self._keep.append((lens_a, skip_a, keep, bed_len))
        self._check(self._f("set_chroms")(
            self.ctx, n, lens_a.ctypes.data_as(C.c_void_p), skip_a.ctypes.data_as(C.c_void_p),
            bed_ptrs if beds is not None else None,
            bed_len.ctypes.data_as(C.c_void_p) if beds is not None else None))

    def sample_begin(self, is_ctrl, save=None):
        sp = None
        if save is not None:
            sa = np.ascontiguousarray(save, dtype=np.uint8)
            self._keep.append(sa)
            sp = sa.ctypes.data_as(C.c_void_p)
        self._check(self._f("sample_begin")(self.ctx, int(is_ctrl), sp))

    def push_events(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self._check(self._f("push_events")(self.ctx, ev.ctypes.data_as(C.c_void_p),
                                           C.c_size_t(len(ev))))

    def sample_end(self):
        frag = C.c_double(0)
        lam = C.c_float(0)
        fac = C.c_float(0)
        self._check(self._f("sample_end")(self.ctx, C.byref(frag), C.byref(lam), C.byref(fac)))
        return frag.value, lam.value, fac.value

    def sample_no_control(self):
        lam = C.c_float(0)
        self._check(self._f("sample_no_control")(self.ctx, C.byref(lam)))
        return lam.value

    def pvalues(self):
        self._check(self._f("pvalues")(self.ctx))

    def find_peaks(self):
        n = C.c_size_t(0)
        g = C.c_uint64(0)
        bp = C.c_uint64(0)
        self._check(self._f("find_peaks")(self.ctx, C.byref(n), C.byref(g), C.byref(bp)))
        self.n_peaks, self.genome_len, self.peak_bp = n.value, g.value, bp.value
        return n.value, g.value, bp.value

    def get_peaks(self):
        out = np.zeros(self.n_peaks, dtype=PEAK_DTYPE)
        if self.n_peaks:
            self._check(self._f("get_peaks")(self.ctx, out.ctypes.data_as(C.c_void_p),
                                             C.c_size_t(self.n_peaks)))
        return out

    def get_intervals(self, which, chrom):
        n = C.c_size_t(0)
        self._check(self._f("interval_count")(self.ctx, int(which), int(chrom), C.byref(n)))
        n = n.value
        end = np.zeros(n, dtype=np.uint32)
        cols = {k: np.zeros(n, dtype=np.float32) for k in ("expt", "ctrl", "p", "q")}
        if n:
            self._check(self._f("get_intervals")(
                self.ctx, int(which), int(chrom), C.c_size_t(n), end.ctypes.data_as(C.c_void_p),
                *[cols[k].ctypes.data_as(C.c_void_p) for k in ("expt", "ctrl", "p", "q")]))
        return end, cols


class Oracle(_Backend):
    prefix = "gxo_"
    _lib = None

    def __init__(self, params):
        if Oracle._lib is None:
            lib = C.CDLL(build_oracle())
            lib.gxo_last_error.restype = C.c_char_p
            lib.gxo_last_error.argtypes = [C.c_void_p]
            lib.gxo_destroy.argtypes = [C.c_void_p]
            lib.gxo_calc_pval.restype = C.c_float
            lib.gxo_calc_pval.argtypes = [C.c_float, C.c_float]
            lib.gxo_pchisq.restype = C.c_double
            lib.gxo_pchisq.argtypes = [C.c_double, C.c_int]
            lib.gxo_getval.restype = C.c_float
            lib.gxo_getval.argtypes = [C.c_int64, C.POINTER(C.c_int)]
            for fn in ("create", "set_chroms", "sample_begin", "push_events", "sample_end",
                       "sample_no_control", "pvalues", "find_peaks", "get_peaks",
                       "interval_count", "get_intervals", "pvalues_path", "find_peaks_path"):
                getattr(lib, "gxo_" + fn).restype = C.c_int
            for fn in ("set_chroms", "sample_begin", "push_events", "sample_end",
                       "sample_no_control", "pvalues", "find_peaks", "get_peaks",
                       "interval_count", "get_intervals"):
                getattr(lib, "gxo_" + fn).argtypes = None
            lib.gxo_frag_len_exact.restype = C.c_double
            lib.gxo_frag_len_exact.argtypes = [C.c_void_p]
            lib.gxo_frag_inexact.restype = C.c_uint64
            lib.gxo_frag_inexact.argtypes = [C.c_void_p]
            Oracle._lib = lib
        super().__init__(Oracle._lib, params)
        self.exact = []  # per replicate (run_case): the treatment's fragLen as an exact sum, and how many of the reference's additions rounded

    def frag_exact(self):
        return float(self.lib.gxo_frag_len_exact(self.ctx)), int(self.lib.gxo_frag_inexact(self.ctx))

    @staticmethod
    def lib():
        if Oracle._lib is None:
            Oracle(make_params()).close()
        return Oracle._lib

    # text emitters of the oracle (checked against the reference's own output files)
    def _names(self, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._keep.append(arr)
        return arr

    def pvalues_to(self, pile_path, append, names, expt_name, ctrl_name):
        self._check(self.lib.gxo_pvalues_path(
            self.ctx, pile_path.encode() if pile_path else None, int(append), self._names(names),
            expt_name.encode(), ctrl_name.encode() if ctrl_name else None))

    def find_peaks_to(self, out_path, log_path, names, peaks_opt=True):
        n = C.c_size_t(0)
        g = C.c_uint64(0)
        bp = C.c_uint64(0)
        self._check(self.lib.gxo_find_peaks_path(
            self.ctx, out_path.encode() if out_path else None,
            log_path.encode() if log_path else None, self._names(names), int(peaks_opt),
            C.byref(n), C.byref(g), C.byref(bp)))
        self.n_peaks, self.genome_len, self.peak_bp = n.value, g.value, bp.value
        return n.value, g.value, bp.value


def assert_fraglen(o, k, fo, fh):
    """fragLen of replicate k: the device adds the reference's float products (Genrich.c:2246) EXACTLY -- two int64 parts,
    rounded once -- so it must equal the oracle's exact sum of the same products bit for bit; the reference's own double
    accumulation may have rounded `inexact` of its additions, each by at most half a unit in the last place (never, for
    unit weights or below 2^26: then all three are the same number)."""
    exact, inexact = o.exact[k]
    assert fh == exact, ("fragLen (device vs the exact sum)", fh, exact)
    assert abs(fo - exact) <= 0.5 * (inexact + 1) * np.spacing(max(fo, exact)), ("fragLen (reference's rounding)", fo, exact, inexact)
    if inexact == 0:
        assert fo == fh


def run_case(be, case, names=None):
    """Drive one backend through a whole run.  `case` = dict(lens, skip, beds, replicates=[
    dict(save, treat=events, ctrl=events|None)]).  Returns per-replicate scalars."""
    be.set_chroms(case["lens"], case.get("skip"), case.get("beds"))
    scal = []
    for rep in case["replicates"]:
        be.sample_begin(0, rep.get("save"))
        be.push_events(rep["treat"])
        frag, _, _ = be.sample_end()
        if hasattr(be, "frag_exact"):
            be.exact.append(be.frag_exact())
        if rep.get("ctrl") is not None:
            be.sample_begin(1, None)
            be.push_events(rep["ctrl"])
            _, lam, fac = be.sample_end()
        else:
            lam, fac = be.sample_no_control(), None
        be.pvalues()
        scal.append((frag, lam, fac))
    be.find_peaks()
    return scal
