#!/usr/bin/env python3
"""BASELINE.json configs[0] at its stated size, through the UNMODIFIED reference (oracle/_ref/Genrich):
1 chromosome x 1,000,000 bp, 100,000 paired fragments as SAM text (about 9 MB: regenerated from
genrich_amd/synth.py, never committed), 30 % of the fragments around 50 centres, default -p 0.01, -a 200 -g 100.
Run in the build container only:  python tests/golden/make_config1.py

Commits DATA only, under tests/golden/config1/: the reference's narrowPeak (gzip), the scalars it printed under
-v, and -- the -b / -f / -k outputs are megabytes -- their line counts and SHA-256.  The directory has no case.json
on purpose: the per-case parametrised tests skip it, tests/test_config1.py owns it.
"""
import gzip
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from genrich_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "Genrich")
NAMES, LENS = ["chr1"], [1_000_000]
ARGS = ["-p", "0.01", "-a", "200", "-g", "100"]
TMP = "/tmp/genrich_golden/config1"


def fragments():
    """100 k fragments, 30 % within +-150 bp of 50 centres (one every 20 kb), the rest uniform."""
    return synth.make_fragments(LENS, 100_000, seed=1, peak_every=20_000, tower_every=10**12, frac_peak=0.30, frac_tower=0.0)


def sha(path):
    h = hashlib.sha256()
    n = 0
    with open(path, "rb") as f:
        for line in f:
            h.update(line)
            n += 1
    return {"sha256": h.hexdigest(), "lines": n}


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/Genrich missing: run `make -C oracle` in the build container")
    shutil.rmtree(TMP, ignore_errors=True)
    os.makedirs(TMP)
    sam = os.path.join(TMP, "t0.sam")
    synth.write_sam(sam, NAMES, LENS, fragments(), name_prefix="t0_")
    out = {k: os.path.join(TMP, k) for k in ("out.narrowPeak", "out.log", "out.pile", "events.bed")}
    res = subprocess.run([REF, "-t", sam, "-v", "-o", out["out.narrowPeak"], "-f", out["out.log"], "-k", out["out.pile"],
                          "-b", out["events.bed"]] + ARGS, capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit("reference failed:\n" + res.stderr)
    err = res.stderr
    meta = dict(
        args=ARGS, names=NAMES, lens=LENS, sam_bytes=os.path.getsize(sam), tmp_prefix=TMP + "/",
        ref_lambda=[float(v) for v in re.findall(r"Background pileup value: ([0-9.]+)", err)],
        ref_genome_len=[int(v) for v in re.findall(r"Genome length: (\d+)bp", err)],
        ref_peaks=[[int(a), int(b)] for a, b in re.findall(r"Peaks identified: (\d+) \((\d+)bp\)", err)],
        files={k: sha(p) for k, p in out.items()},
    )
    dst = os.path.join(HERE, "config1")
    os.makedirs(dst, exist_ok=True)
    with open(out["out.narrowPeak"], "rb") as f, gzip.GzipFile(os.path.join(dst, "out.narrowPeak.gz"), "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    with open(os.path.join(dst, "config1.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps({k: meta[k] for k in ("ref_lambda", "ref_genome_len", "ref_peaks", "sam_bytes")}), meta["files"]["out.log"]["lines"], "intervals")
    shutil.rmtree(TMP)


if __name__ == "__main__":
    main()
