"""ctypes binding of include/genrich_amd.h (the drop-in boundary of the hot path)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libgenrich_amd.so")

EVENT_DTYPE = np.dtype([("chrom", "<u4"), ("start", "<u4"), ("end", "<u4"), ("count", "<u4")])
PEAK_DTYPE = np.dtype(
    [("chrom", "<u4"), ("start", "<u4"), ("end", "<u4"), ("summit", "<u4"),
     ("auc", "<f4"), ("p", "<f4"), ("q", "<f4")]
)
GX_IV_FINAL = -1
EVENT8_DTYPE = np.dtype([("start", "<u4"), ("lcc", "<u4")])   # gx_event8: lcc = [15:0] end - start, [18:16] count class, [31:19] chromosome
_COUNT_CLASS = np.full(11, 8, dtype=np.uint32)
_COUNT_CLASS[[1, 2, 3, 4, 5, 6, 8, 10]] = np.arange(8, dtype=np.uint32)


def pack_events(ev):
    """gx_event records -> (gx_event8 records of the ones that fit, the others as they are): include/genrich_amd.h, gx_event8_pack
    (the same rule, vectorised)."""
    ev = np.ascontiguousarray(ev)
    ln = ev["end"].astype(np.int64) - ev["start"].astype(np.int64)
    cnt = ev["count"]
    cls = np.where(cnt <= 10, _COUNT_CLASS[np.minimum(cnt, 10)], 8)
    fits = (ln >= 0) & (ln < 0xFFFF) & (cls < 8) & (ev["chrom"] < (1 << 13))
    out = np.empty(int(fits.sum()), dtype=EVENT8_DTYPE)
    sel = ev[fits]
    out["start"] = sel["start"]
    out["lcc"] = ln[fits].astype(np.uint32) | (cls[fits].astype(np.uint32) << 16) | (sel["chrom"].astype(np.uint32) << 19)
    return out, ev[~fits]

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_int64), C.c_size_t, C.c_void_p)


class GxParams(C.Structure):
    """gx_params: the peak-calling tail of runProgram's arguments (Genrich.c:5390-5395)."""
    _fields_ = [
        ("thr", C.c_float), ("qval_opt", C.c_int32), ("min_auc", C.c_float),
        ("min_len", C.c_int32), ("max_gap", C.c_int32), ("device", C.c_int32),
        ("genome_len", C.c_uint64),
    ]


_libm = C.CDLL("libm.so.6")
_libm.log10f.restype = C.c_float
_libm.log10f.argtypes = [C.c_float]


def minus_log10f(x: float) -> float:
    """getArgs: pqvalue = -log10f(pqvalue) (Genrich.c:5817), with the host's libm."""
    return float(-_libm.log10f(C.c_float(x)))


_lib = None

_SIGS = {
    "gx_create": [C.POINTER(C.c_void_p), C.POINTER(GxParams)],
    "gx_destroy": [C.c_void_p],
    "gx_reset": [C.c_void_p],
    "gx_set_chroms": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "gx_set_owned": [C.c_void_p, C.c_void_p],
    "gx_set_keep_pileups": [C.c_void_p, C.c_int],
    "gx_expect_fractional": [C.c_void_p, C.c_int],
    "gx_set_knob": [C.c_void_p, C.c_char_p, C.c_char_p],
    "gx_set_collectives": [C.c_void_p, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p],
    "gx_rccl_unique_id": [C.c_void_p, C.c_size_t],
    "gx_set_rccl": [C.c_void_p, C.c_int, C.c_int, C.c_void_p],
    "gx_sample_begin": [C.c_void_p, C.c_int, C.c_void_p],
    "gx_push_events": [C.c_void_p, C.c_void_p, C.c_size_t],
    "gx_push_events_device": [C.c_void_p, C.c_void_p, C.c_size_t],
    "gx_push_events_pinned": [C.c_void_p, C.c_void_p, C.c_size_t],
    "gx_push_events_packed": [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int],
    "gx_event8_pack": [C.c_void_p, C.c_void_p],
    "gx_sample_end": [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_float)],
    "gx_sample_no_control": [C.c_void_p, C.POINTER(C.c_float)],
    "gx_saturation_dropped": [C.c_void_p, C.POINTER(C.c_longlong)],
    "gx_window_net": [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p],
    "gx_pvalues": [C.c_void_p],
    "gx_find_peaks": [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "gx_get_peaks": [C.c_void_p, C.c_void_p, C.c_size_t],
    "gx_peak_count": [C.c_void_p, C.POINTER(C.c_size_t)],
    "gx_write_narrowpeak_path": [C.c_void_p, C.c_void_p, C.c_char_p],
    "gx_write_pile_path": [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int],
    "gx_write_log_path": [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_char_p],
    "gx_total_intervals": [C.c_void_p, C.c_int, C.POINTER(C.c_size_t)],
    "gx_interval_count": [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_size_t)],
    "gx_get_intervals": [C.c_void_p, C.c_int, C.c_int, C.c_size_t] + [C.c_void_p] * 5,
    "gx_selftest": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "gx_selftest2": [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)],
    "gx_selftest_host": [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t],
    "gx_path_info": [C.c_void_p, C.POINTER(C.c_uint)],
    "gx_rccl_nranks": [C.c_void_p, C.POINTER(C.c_int)],
    "gx_set_phase_filter": [C.c_void_p, C.c_char_p],
    "gx_set_phase_timing": [C.c_void_p, C.c_int],
    "gx_phase_times": [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_float))],
}


def load_library(path: str = os.environ.get("GENRICH_AMD_LIB", LIB_PATH)):
    """Load libgenrich_amd.so and declare every entry point of include/genrich_amd.h.
    Raises if the library or a symbol is missing -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = args
        fn.restype = None if name == "gx_destroy" else C.c_int
    lib.gx_last_error.restype = C.c_char_p
    lib.gx_last_error.argtypes = [C.c_void_p]
    lib.gx_strerror.restype = C.c_char_p
    lib.gx_strerror.argtypes = [C.c_int]
    lib.gx_filter_saturation.restype = C.c_longlong
    lib.gx_filter_saturation.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def filter_saturation(events, lens):
    """keep flags (uint8) for `events` under the reference's int16 saturation rule (Genrich.c:2558-2573);
    host-only, needs no GPU."""
    import numpy as np
    lib = load_library()
    ev = np.ascontiguousarray(events)
    ln = np.ascontiguousarray(lens, dtype=np.uint32)
    keep = np.ones(len(ev), dtype=np.uint8)
    rc = lib.gx_filter_saturation(ev.ctypes.data, len(ev), len(ln), ln.ctypes.data, keep.ctypes.data)
    if rc < 0:
        raise RuntimeError(f"gx_filter_saturation: {rc}")
    return keep, int(rc)


def rccl_unique_id() -> bytes:
    lib = load_library()
    buf = C.create_string_buffer(128)
    rc = lib.gx_rccl_unique_id(buf, 128)
    if rc:
        raise RuntimeError(f"gx_rccl_unique_id: {rc}")
    return buf.raw


def selftest_host(what, a, b):
    """calcPval (what 1) / multPval's tail (what 3) by the host build of the library's routines: returns
    (float results, the doubles they were rounded from).  Needs no GPU."""
    import numpy as np
    lib = load_library()
    a = np.ascontiguousarray(a, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    out = np.zeros_like(a)
    dbl = np.zeros(a.size, dtype=np.float64)
    rc = lib.gx_selftest_host(int(what), a.ctypes.data, b.ctypes.data, out.ctypes.data, dbl.ctypes.data, a.size)
    if rc:
        raise RuntimeError(f"gx_selftest_host: {rc}")
    return out, dbl


class Genrich:
    """One run of the hot path on one GPU: the call order of runProgram (Genrich.c:5386-5607)."""

    def __init__(self, params: GxParams):
        self.lib = load_library()
        self.ctx = C.c_void_p()
        rc = self.lib.gx_create(C.byref(self.ctx), C.byref(params))
        if rc != 0:
            msg = self.lib.gx_last_error(self.ctx).decode() if self.ctx else ""
            raise RuntimeError(f"gx_create failed ({rc}): {self.lib.gx_strerror(rc).decode()} {msg}")
        self._keep = []
        self.n_peaks = 0

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"genrich_amd error {rc}: {self.lib.gx_strerror(rc).decode()} "
                               f"[{self.lib.gx_last_error(self.ctx).decode()}]")

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.gx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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
        self._keep.append((lens_a, skip_a, keep, bed_len, bed_ptrs))
        self._check(self.lib.gx_set_chroms(
            self.ctx, n, lens_a.ctypes.data, skip_a.ctypes.data,
            C.cast(bed_ptrs, C.c_void_p) if beds is not None else None,
            bed_len.ctypes.data if beds is not None else None))

    def reset(self):
        self._check(self.lib.gx_reset(self.ctx))

    def expect_fractional(self, on=True):
        """Hint: the run may hold fractional weights (Genrich's -s): pair records with a weight class from the first sample on."""
        self._check(self.lib.gx_expect_fractional(self.ctx, int(bool(on))))

    def set_knob(self, name, value=1):
        """A test / measurement switch (GX_NO_LOOSE, ...) on the live context; the environment is read once, in gx_create."""
        self._check(self.lib.gx_set_knob(self.ctx, name.encode(), str(int(value)).encode()))

    def window_net(self, chrom, pos0, n):
        """The open sample's difference array on [pos0, pos0 + n) of a chromosome (1/120 units), from the events pushed so far."""
        out = np.zeros(n, dtype=np.int64)
        self._check(self.lib.gx_window_net(self.ctx, chrom, pos0, n, out.ctypes.data))
        return out

    def set_keep_pileups(self, keep):
        """keep=False: the pileup floats of the p-value intervals (only the -f / -k emitters read them)
        are not materialised."""
        self._check(self.lib.gx_set_keep_pileups(self.ctx, int(bool(keep))))

    def set_owned(self, owned):
        a = np.ascontiguousarray(owned, dtype=np.uint8)
        self._check(self.lib.gx_set_owned(self.ctx, a.ctypes.data))

    def set_collectives(self, rank, world, allreduce):
        """Every exchange of the library through ONE host callback: an all-reduce (sum) of int64 words."""
        self._cb = ALLREDUCE_FN(allreduce)   # (kept alive with the object: the library calls it later)
        self._check(self.lib.gx_set_collectives(self.ctx, rank, world, self._cb, None))

    def set_rccl(self, rank, world, unique_id: bytes):
        """The library's own RCCL communicator (collective over all ranks); unique_id = rccl_unique_id()
        of one rank, handed to the others by the host program."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.gx_set_rccl(self.ctx, int(rank), int(world), buf))

    def sample_begin(self, is_ctrl, save=None):
        sp = None
        if save is not None:
            sa = np.ascontiguousarray(save, dtype=np.uint8)
            self._keep.append(sa)
            sp = sa.ctypes.data
        self._check(self.lib.gx_sample_begin(self.ctx, int(is_ctrl), sp))

    def push_events(self, ev):
        ev = np.ascontiguousarray(ev, dtype=EVENT_DTYPE)
        self._check(self.lib.gx_push_events(self.ctx, ev.ctypes.data, len(ev)))

    def push_events_ptr(self, host_ptr, n, pinned=False):
        """gx_push_events on a raw host pointer; pinned=True: page-locked memory that stays untouched until
        sample_end (gx_push_events_pinned: uploaded in place)."""
        f = self.lib.gx_push_events_pinned if pinned else self.lib.gx_push_events
        self._check(f(self.ctx, C.c_void_p(host_ptr), int(n)))

    def push_events_device(self, dev_ptr: int, n: int):
        """Events already resident in HBM (e.g. a torch tensor's data_ptr())."""
        self._check(self.lib.gx_push_events_device(self.ctx, C.c_void_p(dev_ptr), n))

    def push_events_packed(self, ev8, where=0, n=None):
        """8-byte events (gx_event8: pack_events): a numpy array of EVENT8_DTYPE in host memory (where=0), or a raw pointer
        with n -- pinned host memory (where=1) or device memory (where=2)."""
        if isinstance(ev8, np.ndarray):
            ev8 = np.ascontiguousarray(ev8, dtype=EVENT8_DTYPE)
            self._check(self.lib.gx_push_events_packed(self.ctx, ev8.ctypes.data, len(ev8), 0))
        else:
            self._check(self.lib.gx_push_events_packed(self.ctx, C.c_void_p(int(ev8)), int(n), int(where)))

    def sample_end(self):
        frag, lam, fac = C.c_double(0), C.c_float(0), C.c_float(0)
        self._check(self.lib.gx_sample_end(self.ctx, C.byref(frag), C.byref(lam), C.byref(fac)))
        return frag.value, lam.value, fac.value

    def sample_no_control(self):
        lam = C.c_float(0)
        self._check(self.lib.gx_sample_no_control(self.ctx, C.byref(lam)))
        return lam.value

    def pvalues(self):
        self._check(self.lib.gx_pvalues(self.ctx))

    def find_peaks(self):
        n, g, bp = C.c_size_t(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.gx_find_peaks(self.ctx, C.byref(n), C.byref(g), C.byref(bp)))
        self.n_peaks, self.genome_len, self.peak_bp = n.value, g.value, bp.value
        return n.value, g.value, bp.value

    def get_peaks(self):
        out = np.zeros(self.n_peaks, dtype=PEAK_DTYPE)
        if self.n_peaks:
            self._check(self.lib.gx_get_peaks(self.ctx, out.ctypes.data, self.n_peaks))
        return out

    def get_intervals(self, which, chrom, piles=True):
        """(ends, {"expt", "ctrl", "p", "q"}) of one chromosome; piles=False leaves the pileup
        columns out (the only choice after set_keep_pileups(False))."""
        n = C.c_size_t(0)
        self._check(self.lib.gx_interval_count(self.ctx, int(which), int(chrom), C.byref(n)))
        n = n.value
        end = np.zeros(n, dtype=np.uint32)
        cols = {k: np.zeros(n, dtype=np.float32) for k in ("expt", "ctrl", "p", "q")}
        if n:
            self._check(self.lib.gx_get_intervals(
                self.ctx, int(which), int(chrom), n, end.ctypes.data,
                *[cols[k].ctypes.data if piles or k in ("p", "q") else None for k in ("expt", "ctrl", "p", "q")]))
        return end, cols

    # -- text emitters (gx_emit.cpp) ----------------------------------------------------
    def _names(self, names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._keep.append(arr)
        return C.cast(arr, C.c_void_p)

    def write_narrowpeak(self, names, path):
        self._check(self.lib.gx_write_narrowpeak_path(self.ctx, self._names(names), path.encode()))

    def write_pile(self, rep, names, expt_name, ctrl_name, path, append=False):
        self._check(self.lib.gx_write_pile_path(
            self.ctx, rep, self._names(names), len(names), expt_name.encode(),
            ctrl_name.encode() if ctrl_name else None, path.encode(), int(append)))

    def write_log(self, n_rep, names, qval_opt, peaks_opt, thr, path):
        self._check(self.lib.gx_write_log_path(self.ctx, n_rep, self._names(names), len(names), int(qval_opt),
                                               int(peaks_opt), float(thr), path.encode()))

    def selftest(self, what, a, b=None):
        a = np.ascontiguousarray(a, dtype=np.float32)
        out = np.zeros_like(a)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float32)
            bp = b.ctypes.data
        self._check(self.lib.gx_selftest(self.ctx, int(what), a.ctypes.data, bp, out.ctypes.data, a.size))
        return out

    def selftest2(self, what, a, b):
        """selftest plus the doubles before rounding and the number of results the host re-evaluated."""
        a = np.ascontiguousarray(a, dtype=np.float32)
        b = np.ascontiguousarray(b, dtype=np.float32)
        out = np.zeros_like(a)
        dbl = np.zeros(a.size, dtype=np.float64)
        nr = C.c_size_t(0)
        self._check(self.lib.gx_selftest2(self.ctx, int(what), a.ctypes.data, b.ctypes.data, out.ctypes.data,
                                          dbl.ctypes.data, a.size, C.byref(nr)))
        return out, dbl, nr.value

    def interval_total(self, which=GX_IV_FINAL):
        n = C.c_size_t(0)
        self._check(self.lib.gx_total_intervals(self.ctx, int(which), C.byref(n)))
        return n.value

    def path_info(self):
        """Which device path the last calls took: GX_PATH_* bits (1 fused tile stage, 2 loose-slot sweep, 4 fell back, 8 page tables grew,
        16 pair records, 32 dense BH all-reduce, 64 range BH exchange, 128 fractional pair records, 256 pileup floats written, 512 8-byte
        events read in place, 1024 the control merge scored its intervals, 2048 BH's histogram from the pileup sums, 8192 q looked up
        where it is read, 16384 the loose slots swept with bits written late, 32768 -q on the loose slots)."""
        f = C.c_uint(0)
        self._check(self.lib.gx_path_info(self.ctx, C.byref(f)))
        return f.value

    def rccl_nranks(self):
        """Ranks of the library's own RCCL communicator as RCCL reports them (0: none)."""
        n = C.c_int(0)
        self._check(self.lib.gx_rccl_nranks(self.ctx, C.byref(n)))
        return n.value

    def set_phase_filter(self, name):
        """Time only the phase `name` (HIP events on the library's stream)."""
        self._check(self.lib.gx_set_phase_filter(self.ctx, name.encode()))

    def set_phase_timing(self, level):
        """0 none (default), 1 the tile stage only, 2 every phase (each event record costs the stream ~5 us)."""
        self._check(self.lib.gx_set_phase_timing(self.ctx, int(level)))

    def phase_times(self):
        names = C.c_char_p()
        ms = C.POINTER(C.c_float)()
        k = self.lib.gx_phase_times(self.ctx, C.byref(names), C.byref(ms))
        out = []
        # names are NUL-separated: walk the buffer
        addr = C.cast(names, C.c_void_p).value
        for i in range(k):
            s = C.string_at(addr)
            out.append((s.decode(), ms[i]))
            addr += len(s) + 1
        return out
