#!/usr/bin/env python3
"""-r (PCR duplicates, Genrich.c:3267-4042) at size, through the UNMODIFIED reference (oracle/_ref/Genrich): 550,000
templates = 1,127,663 alignments on four chromosomes -- proper pairs, pairs with a secondary pair (multi-alignment sets),
singletons, discordant pairs (also across chromosomes, also with the mates swapped); a fifth of the templates reuse the
coordinates of an earlier one.  The SAM text (174 MB) is regenerated from genrich_amd/synth.py + tools/records_to_sam,
never committed.  Run in the build container only:  python tests/golden/make_dups_mid.py

Commits DATA only, under tests/golden/dups_mid/: the counts the reference printed under -v and the line counts and
SHA-256 of its -R log and its -b event list (the directory has no case.json on purpose: tests/test_dups_mid.py owns it).
"""
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
NAMES, LENS = ["chrA", "chrB", "chrC", "chrD"], [3_000_000, 2_000_000, 1_200_000, 500_000]
N_TEMPLATES, SEED = 550_000, 17
ARGS = ["-r", "-y"]
TMP = "/tmp/genrich_golden/dups_mid"


def write_sam(path):
    """The fixture's SAM text; returns the number of alignment records."""
    tool = os.path.join(ROOT, "tools", "records_to_sam")
    if not os.path.exists(tool):
        subprocess.check_call(["gcc", "-O2", "-o", tool, tool + ".c"])
    recs = synth.make_dups_records(LENS, N_TEMPLATES, SEED)
    d = os.path.dirname(path)
    os.makedirs(d, exist_ok=True)
    rb, ch = os.path.join(d, "recs.bin"), os.path.join(d, "chroms.txt")
    recs.tofile(rb)
    open(ch, "w").write("".join(f"{n} {l}\n" for n, l in zip(NAMES, LENS)))
    subprocess.check_call([tool, rb, ch, path, "d"])
    os.remove(rb)
    os.remove(ch)
    return len(recs)


def sha(path, skip_hash_lines=False):
    h = hashlib.sha256()
    n = 0
    with open(path, "rb") as f:
        for line in f:
            if skip_hash_lines and line.startswith(b"#"):   # ("# experimental file #0: <path>": the path differs by machine)
                continue
            h.update(line)
            n += 1
    return {"sha256": h.hexdigest(), "lines": n}


def dup_lines(err):
    return [l.strip() for l in err.splitlines() if "aln sets:" in l or "duplicates:" in l]


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/Genrich missing: run `make -C oracle` in the build container")
    shutil.rmtree(TMP, ignore_errors=True)
    os.makedirs(TMP)
    sam = os.path.join(TMP, "t0.sam")
    nrec = write_sam(sam)
    out = {k: os.path.join(TMP, k) for k in ("out.narrowPeak", "out.dups", "events.bed")}
    res = subprocess.run([REF, "-t", sam, "-v", "-o", out["out.narrowPeak"], "-R", out["out.dups"], "-b", out["events.bed"]] + ARGS,
                         capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit("reference failed:\n" + res.stderr)
    meta = dict(args=ARGS, names=NAMES, lens=LENS, templates=N_TEMPLATES, seed=SEED, alignments=nrec, sam_bytes=os.path.getsize(sam),
                ref_dups=dup_lines(res.stderr),
                ref_lambda=[float(v) for v in re.findall(r"Background pileup value: ([0-9.]+)", res.stderr)],
                files={"out.dups": sha(out["out.dups"], True), "events.bed": sha(out["events.bed"]), "out.narrowPeak": sha(out["out.narrowPeak"])})
    with open(os.path.join(HERE, "dups_mid", "dups_mid.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps(meta, indent=1))
    shutil.rmtree(TMP)


if __name__ == "__main__":
    main()
