"""The two fixture-sized SAM writers of tools/ (C, 10 M lines a second) against the Python writers of genrich_amd/synth.py
they stand in for: same bytes.  bench.py's `e2e_cli` leg and tests/test_dups_mid.py depend on them."""
import os
import subprocess

import numpy as np

import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool(name):
    src, dst = os.path.join(ROOT, "tools", name + ".c"), os.path.join(ROOT, "tools", name)
    if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-o", dst, src])
    return dst


def _chroms(path, names, lens):
    open(path, "w").write("".join(f"{n} {l}\n" for n, l in zip(names, lens)))


def test_events_to_sam_writes_what_write_sam_writes(tmp_path):
    names, lens = ["chrA", "chrB", "chrM"], [400_000, 90_000, 16_569]
    ev = synth.make_fragments(lens, 20_000, 7, peak_every=20_000)
    ev = np.concatenate([ev, np.array([(2, 16_400, 16_569, 1), (0, 0, 37, 1)], dtype=ev.dtype)])   # a fragment at a chromosome's end, a short one
    a, b = str(tmp_path / "a.sam"), str(tmp_path / "b.sam")
    synth.write_sam(a, names, lens, ev)
    ev.tofile(str(tmp_path / "ev.bin"))
    _chroms(str(tmp_path / "ch.txt"), names, lens)
    subprocess.check_call([_tool("events_to_sam"), str(tmp_path / "ev.bin"), str(tmp_path / "ch.txt"), b])
    assert open(a, "rb").read() == open(b, "rb").read()


def test_events_to_sam_refuses_what_it_cannot_render(tmp_path):
    names, lens = ["chrA"], [10_000]
    ev = np.array([(0, 100, 300, 2)], dtype=synth.EVENT_DTYPE)   # a multimapped read: not its business
    ev.tofile(str(tmp_path / "ev.bin"))
    _chroms(str(tmp_path / "ch.txt"), names, lens)
    res = subprocess.run([_tool("events_to_sam"), str(tmp_path / "ev.bin"), str(tmp_path / "ch.txt"), str(tmp_path / "o.sam")],
                         capture_output=True, text=True)
    assert res.returncode != 0 and "not a unit-weight fragment" in res.stderr


def test_records_to_sam_renders_every_field(tmp_path):
    names, lens = ["chrA", "chrB"], [50_000, 20_000]
    recs = synth.make_dups_records(lens, 2_000, 5)
    assert recs.dtype.itemsize == 28
    recs.tofile(str(tmp_path / "r.bin"))
    _chroms(str(tmp_path / "ch.txt"), names, lens)
    out = str(tmp_path / "o.sam")
    subprocess.check_call([_tool("records_to_sam"), str(tmp_path / "r.bin"), str(tmp_path / "ch.txt"), out, "d"])
    lines = [l.rstrip("\n").split("\t") for l in open(out) if not l.startswith("@")]
    assert len(lines) == len(recs)
    for f, r in list(zip(lines, recs))[:400] + list(zip(lines, recs))[-50:]:
        assert f[0] == f"d{r['tmpl']}" and int(f[1]) == r["flag"] and f[2] == names[r["chrom"]] and int(f[3]) == r["pos"] + 1
        assert int(f[4]) == r["mapq"] and f[5] == f"{r['rl']}M"
        if r["rnext"] < 0:
            assert f[6] == "*" and f[7] == "0"
        else:
            assert f[6] == ("=" if r["rnext"] == r["chrom"] else names[r["rnext"]]) and int(f[7]) == r["pnext"] + 1
        assert int(f[8]) == r["tlen"] and f[9] == "A" * r["rl"]
        assert f[10] == ("*" if r["qual"] == 0xFF else chr(33 + r["qual"]) * r["rl"])
        assert f[11] == "NM:i:0" and f[12] == f"AS:i:{r['AS']}"
