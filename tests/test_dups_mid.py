"""-r (PCR-duplicate removal, Genrich.c:3267-4042) at size: 550,000 templates = 1.1 M alignments with pairs,
multi-alignment sets, singletons and discordant pairs, a fifth of them copies of earlier ones.  The reference's own -R log,
-b event list and narrowPeak for this input (tests/golden/dups_mid/, made by tests/golden/make_dups_mid.py from the
unmodified reference binary; SHA-256 of the files, the counts it printed) against
  * genrich-amd's host tables (not-gpu, --events-only): -R and -b,
  * genrich-amd with the membership half of all three tables on the device (gpu: gx_dups_first for proper pairs,
    DISCORDANT pairs and singletons), and once more with GENRICH_DUPS_HOST=1; the wall times of the two are printed.
The SAM text (174 MB) is regenerated from genrich_amd/synth.py; nothing here reads /root/reference."""
import importlib.util
import json
import os
import subprocess
import time

import pytest

import golden_cases as G

D = os.path.join(G.GOLDEN, "dups_mid")
META = json.load(open(os.path.join(D, "dups_mid.json")))
TMP = "/tmp/genrich_test/dups_mid"


def _mk():
    spec = importlib.util.spec_from_file_location("make_dups_mid", os.path.join(G.GOLDEN, "make_dups_mid.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def sam():
    mk = _mk()
    path = os.path.join(TMP, "t0.sam")
    if not (os.path.exists(path) and os.path.getsize(path) == META["sam_bytes"]):
        assert mk.write_sam(path) == META["alignments"]
    assert os.path.getsize(path) == META["sam_bytes"]
    yield path, mk
    os.remove(path)


def _run(mk, sam, tag, extra_args=(), env=None):
    from genrich_amd import build
    build.build_host()
    out = {k: os.path.join(TMP, tag + "." + k) for k in ("out.narrowPeak", "out.dups", "events.bed")}
    cmd = [build.HOST_BIN, "-t", sam, "-v", "-R", out["out.dups"], "-b", out["events.bed"]] + list(extra_args) + META["args"]
    if "--events-only" not in extra_args:
        cmd += ["-o", out["out.narrowPeak"]]
    t0 = time.perf_counter()
    res = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, GENRICH_DUPS_REPORT="1", **(env or {})))
    dt = time.perf_counter() - t0
    assert res.returncode == 0, res.stderr[-2000:]
    assert mk.dup_lines(res.stderr) == META["ref_dups"]
    assert mk.sha(out["out.dups"], True) == META["files"]["out.dups"]
    assert mk.sha(out["events.bed"]) == META["files"]["events.bed"]
    return out, res.stderr, dt


def test_dups_mid_host_tables_are_the_references(sam):
    path, mk = sam
    _run(mk, path, "host", ["--events-only"])


@pytest.mark.gpu
def test_dups_mid_on_the_device_is_the_references(sam):
    path, mk = sam
    times = {}
    for mode, env in (("device", {}), ("host", {"GENRICH_DUPS_HOST": "1"})):
        best = None
        for _ in range(2):
            out, err, dt = _run(mk, path, mode, env=env)
            best = dt if best is None else min(best, dt)
            assert mk.sha(out["out.narrowPeak"]) == META["files"]["out.narrowPeak"]
            rep = [l for l in err.splitlines() if l.startswith("[dups] device:")]
            if mode == "device":
                # "... by table: P paired, D discordant, S single": all three tables went through gx_dups_first
                assert rep, err[-600:]
                by = rep[-1].split("by table:")[1].replace(",", " ").split()
                assert int(by[0]) > 300_000 and int(by[2]) > 100_000 and int(by[4]) > 80_000, rep[-1]
            else:
                assert not rep
        times[mode] = best
    line = (f"[dups_mid] {META['alignments']} alignments, SAM text -> narrowPeak with -r: device tables {times['device']:.2f} s, "
            f"host tables {times['host']:.2f} s (best of 2; wall, process start and HIP initialisation included; "
            f"{os.cpu_count()} host cores)")
    print("\n" + line)
    rep = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(rep):   # (on the GPU box: the only directory that travels back)
        open(os.path.join(rep, "dups_mid_times.txt"), "a").write(line + "\n")
