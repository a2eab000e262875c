"""BASELINE.json configs[0] at its stated size (1 Mbp, 100,000 paired fragments as SAM text, default -p 0.01):
the reference's own outputs for this workload (tests/golden/config1/, made by tests/golden/make_config1.py from
the unmodified reference binary) against
  * the CPU oracle on the fragment list (not-gpu: pins the oracle at this size),
  * genrich-amd's ingest: its -b event list has the reference's SHA-256 (not-gpu, --events-only),
  * genrich-amd end to end on the GPU: narrowPeak bytes, -f and -k SHA-256 (gpu).
The SAM text (11 MB) is regenerated from genrich_amd/synth.py; nothing here reads /root/reference."""
import gzip
import hashlib
import importlib.util
import json
import os
import subprocess

import numpy as np
import pytest

import backends as B
import golden_cases as G

D = os.path.join(G.GOLDEN, "config1")
META = json.load(open(os.path.join(D, "config1.json")))


def _mk():
    spec = importlib.util.spec_from_file_location("make_config1", os.path.join(G.GOLDEN, "make_config1.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _sha(path):
    h = hashlib.sha256()
    n = 0
    with open(path, "rb") as f:
        for line in f:
            h.update(line)
            n += 1
    return {"sha256": h.hexdigest(), "lines": n}


def _want_peaks():
    with gzip.open(os.path.join(D, "out.narrowPeak.gz"), "rb") as f:
        return f.read()


def _sam(mk):
    import synth
    tmp = META["tmp_prefix"]  # (the -k header prints the input's path: same prefix as when the fixture was made)
    os.makedirs(tmp, exist_ok=True)
    sam = os.path.join(tmp, "t0.sam")
    synth.write_sam(sam, mk.NAMES, mk.LENS, mk.fragments(), name_prefix="t0_")
    assert os.path.getsize(sam) == META["sam_bytes"]
    return sam


def test_config1_oracle_is_the_references_bytes(tmp_path):
    mk = _mk()
    ev = mk.fragments()
    o = B.Oracle(B.make_params(pq=0.01, min_auc=200.0, max_gap=100))
    o.set_chroms(mk.LENS)
    o.sample_begin(0, None)
    o.push_events(ev)
    o.sample_end()
    lam = o.sample_no_control()
    o.pvalues()
    out = str(tmp_path / "o.narrowPeak")
    n, g, bp = o.find_peaks_to(out, None, mk.NAMES)
    assert [n, bp] == META["ref_peaks"][0] and g == META["ref_genome_len"][0]
    assert f"{lam:.6f}" == f"{META['ref_lambda'][0]:.6f}"
    assert open(out, "rb").read() == _want_peaks()
    end, _ = o.get_intervals(-1, 0)
    assert len(end) == META["files"]["out.log"]["lines"] - 1   # (the -f log has a header line)


def test_config1_ingest_event_list_is_the_references(tmp_path):
    from genrich_amd import build
    build.build_host()
    mk = _mk()
    sam = _sam(mk)
    bed = str(tmp_path / "events.bed")
    subprocess.check_call([build.HOST_BIN, "--events-only", "-t", sam, "-b", bed] + META["args"], stderr=subprocess.DEVNULL)
    assert _sha(bed) == META["files"]["events.bed"]


@pytest.mark.gpu
def test_config1_sam_to_narrowpeak_is_the_references_bytes(tmp_path):
    from genrich_amd import build
    build.build()
    mk = _mk()
    sam = _sam(mk)
    tmp = META["tmp_prefix"]
    out = {k: os.path.join(tmp, k) for k in ("out.narrowPeak", "out.log", "out.pile", "events.bed")}
    res = subprocess.run([build.HOST_BIN, "-t", sam, "-v", "-o", out["out.narrowPeak"], "-f", out["out.log"], "-k", out["out.pile"],
                          "-b", out["events.bed"]] + META["args"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert open(out["out.narrowPeak"], "rb").read() == _want_peaks()
    for k in ("out.log", "out.pile", "events.bed"):
        assert _sha(out[k]) == META["files"][k], k
    assert f"Background pileup value: {META['ref_lambda'][0]:.6f}" in res.stderr
    assert f"Peaks identified: {META['ref_peaks'][0][0]} ({META['ref_peaks'][0][1]}bp)" in res.stderr


@pytest.mark.gpu
def test_config1_hip_path_vs_oracle_bits():
    import genrich_amd
    mk = _mk()
    ev = mk.fragments()
    case = dict(lens=mk.LENS, replicates=[dict(save=None, treat=ev, ctrl=None)])
    par = B.make_params(pq=0.01, min_auc=200.0, max_gap=100)
    h, o = genrich_amd.Genrich(par), B.Oracle(par)
    B.run_case(h, case)
    B.run_case(o, case)
    assert h.get_peaks().tobytes() == o.get_peaks().tobytes() and h.n_peaks == META["ref_peaks"][0][0]
    eh, ch = h.get_intervals(-1, 0)
    eo, co = o.get_intervals(-1, 0)
    assert np.array_equal(eh, eo)
    for k in ("expt", "p"):
        assert np.array_equal(ch[k].view(np.uint32), co[k].view(np.uint32)), k
