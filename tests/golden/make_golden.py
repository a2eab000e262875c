#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(oracle/_ref/Genrich, built by oracle/Makefile from /root/reference) on small synthetic
SAM files.  Run in the build container only:  python tests/golden/make_golden.py

Per case the fixture holds DATA only:
  case.json          chromosome table, -e/-E, per-replicate `save` flags, parameters, the
                     scalars the reference printed under -v (lambda, factor, genome length,
                     peak count / bp)
  events.bed.gz      the reference's own -b output = the post-clamp event list
                     (chr start end name_count_[E|C]_sample; Genrich.c:2497-2508)
  out.narrowPeak.gz, out.log.gz (-f), out.pile.gz (-k)   the reference's outputs
The SAM inputs are regenerated from tests/synth.py and are not committed.
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "Genrich")


def merged_bed(regions, clen):
    """saveXBed's sort + clamp + merge (Genrich.c:1144-1206) for one chromosome."""
    iv = sorted([list(r) for r in regions if r[0] < clen], key=lambda r: r[0])
    out = []
    for s, e in iv:
        e = min(e, clen)
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return [c for r in out for c in r]


def multimap_reads(lens, n_reads, seed, hot=0.75):
    """Reads with 1..12 equally scored alignments (exercises counts 2,3,4,5,6,8,10 and the
    7 / 9 / >10 subsampling, Genrich.c:3145-3146)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = []
    for _ in range(n_reads):
        k = int(rng.integers(1, 13))
        for _a in range(k):
            c = int(rng.integers(0, len(lens)))
            fl = int(100 + rng.integers(0, 150))
            s = int(rng.integers(0, lens[c] - fl))
            # pull most of the alignments towards a few hot spots so peaks exist
            if rng.random() < hot:
                s = int(min(max(0, (s // 10000) * 10000 + 5000 + rng.integers(-60, 60)), lens[c] - fl))
            rows.append((c, s, s + fl, k))
    return np.array(rows, dtype=synth.EVENT_DTYPE)


def mf(lens, n, seed, peak_every=8000, tower_every=30000, **kw):
    """Strongly enriched stream so that the tiny genomes still yield peaks and q < 1."""
    if kw.get("uniform_only"):
        return synth.make_fragments(lens, n, seed, **kw)
    return synth.make_fragments(lens, n, seed, peak_every, tower_every, frac_peak=0.4,
                                frac_tower=0.3, **kw)


def int16_towers(lens, n_start, n_end, seed):
    """n_start fragments [1000, 1200), n_end fragments that all end at 5400 (starts spread over 50 bases) and some
    background, in random order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    bg = mf(lens, 1500, seed + 1000)
    a = np.zeros(n_start, dtype=synth.EVENT_DTYPE)
    a["start"], a["end"], a["count"] = 1000, 1200, 1
    b = np.zeros(n_end, dtype=synth.EVENT_DTYPE)
    b["start"] = 5000 + 2 * rng.integers(0, 50, n_end)
    b["end"], b["count"] = 5400, 1
    ev = np.concatenate([bg, a, b])
    return ev[rng.permutation(len(ev))]


def int16_towers_frac(lens, n_start, n_end, seed):
    """As int16_towers, with a third of the tower reads multimapped (k = 2 .. 10 equally scored alignments, weight 1 / k):
    one alignment on the tower, the others elsewhere -- the int16 part of the reference's (cov, eighths, sixths, tenths)
    counters reaches its limits in between whole numbers.  Consecutive events with count = k are one read (write_sam)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    reads = [[(0, int(s), int(e), 1)] for _, s, e, _ in mf(lens, 1500, seed + 1000)]
    L = lens[0]

    def multi(first, k):
        out = [first + (k,)]
        for _ in range(k - 1):
            s = 10_000 + 1_000 * int(rng.integers(0, 40))
            out.append((0, s, s + 150 + 10 * int(rng.integers(0, 8)), k))
        perm = rng.permutation(k)
        return [out[j] for j in perm]

    for _ in range(n_start):
        k = int(rng.choice([1, 1, 1, 1, 2, 3, 4, 5, 6, 8, 10]))
        reads.append(multi((0, 1000, 1200), k) if k > 1 else [(0, 1000, 1200, 1)])
    for _ in range(n_end):
        k = int(rng.choice([1, 1, 1, 1, 2, 3, 4, 5, 6, 8, 10]))
        st = 5000 + 2 * int(rng.integers(0, 50))
        reads.append(multi((0, st, 5400), k) if k > 1 else [(0, st, 5400, 1)])
    order = rng.permutation(len(reads))
    rows = [a for i in order for a in reads[i]]
    return np.array(rows, dtype=synth.EVENT_DTYPE)


# CPU-only fixtures of tests/test_host_cli.py::test_cli_int16_decisions_more: the reference's -b list (hash + line count) and
# its read-by-read warnings.  Not `cases` (no case.json): the suites that walk every case leave them alone.
def int16_extra():
    L1 = [60_000]
    return {
        # a third of the piled-up reads multimapped
        "saturate16_frac": dict(ev=lambda: int16_towers_frac(L1, 100_000, 100_000, 71), mixed=None, args=["-a", "20"]),
        # ATAC cut sites (-j -d 40): two intervals per fragment, either of which can be the one that is dropped -- the
        # fragment's length towards the -x average is then the other one's; unpaired reads kept (-y)
        "saturate16_atac": dict(ev=lambda: int16_towers(L1, 47_600, 47_400, 81), mixed=dict(seed=6, bam=False),
                                args=["-j", "-d", "40", "-y", "-a", "20"]),
    }


def make_int16_extra(name):
    import hashlib
    spec = int16_extra()[name]
    L1 = [60_000]
    out_dir = os.path.join(HERE, name)
    os.makedirs(out_dir, exist_ok=True)
    tmp = os.path.join("/tmp/genrich_golden", name)
    shutil.rmtree(tmp, ignore_errors=True)
    os.makedirs(tmp)
    sam = os.path.join(tmp, "t0.sam")
    write_input(sam, ["chrA"], L1, spec["ev"](), spec["mixed"], 0, "t0_")
    res = subprocess.run([REF, "-t", sam, "-v", "-b", os.path.join(tmp, "events.bed"), "-o", os.path.join(tmp, "o.np")] + spec["args"],
                         capture_output=True, text=True)
    if res.returncode != 0:
        sys.exit(name + ": reference failed:\n" + res.stderr)
    bed = open(os.path.join(tmp, "events.bed"), "rb").read()
    with open(os.path.join(out_dir, "events.bed.sha256"), "w") as f:  # (megabytes of lines: their hash and count are kept)
        f.write(f"{hashlib.sha256(bed).hexdigest()} {bed.count(10)}\n")
    with gzip.GzipFile(os.path.join(out_dir, "out.int16.gz"), "wb", mtime=0) as g:
        g.write("".join(l + "\n" for l in res.stderr.splitlines() if "skipped due to" in l).encode())
    n = sum(1 for l in res.stderr.splitlines() if "skipped due to" in l)
    print(f"{name:18s} {n} alignments skipped by the reference")
    shutil.rmtree(tmp)


def cases():
    L1 = [60_000]
    yield dict(
        name="basic", names=["chrA"], args=[],
        reps=[dict(t=(["chrA"], L1, mf(L1, 3000, 11)), c=None)])

    L2 = [40_000, 25_000]
    N2 = ["chr1", "chr2"]
    yield dict(
        name="ctrl_q", names=N2, args=["-q", "0.3", "-a", "20"],
        reps=[dict(t=(N2, L2, mf(L2, 3000, 21)),
                   c=(N2, L2, mf(L2, 2500, 22, uniform_only=True)))])

    yield dict(
        name="multimap", names=N2, args=["-p", "0.05", "-a", "5", "-g", "50"],
        reps=[dict(t=(N2, L2, multimap_reads(L2, 1200, 31)),
                   c=(N2, L2, multimap_reads(L2, 600, 32, hot=0.0)))])

    yield dict(
        name="atac", names=N2, args=["-j", "-a", "50"],
        reps=[dict(t=(N2, L2, mf(L2, 2500, 41)), c=None)])

    yield dict(
        name="atac_odd", names=N2, args=["-j", "-d", "151", "-q", "0.5", "-a", "10", "-l", "40"],
        reps=[dict(t=(N2, L2, mf(L2, 2500, 42, min_len=30)),
                   c=(N2, L2, mf(L2, 2000, 43, uniform_only=True)))])

    yield dict(
        name="reps3", names=N2, args=["-q", "0.2", "-a", "30"],
        reps=[dict(t=(N2, L2, mf(L2, 2000, 51)),
                   c=(N2, L2, mf(L2, 2000, 52, uniform_only=True))),
              dict(t=(N2, L2, mf(L2, 2200, 53)), c="null"),
              dict(t=(N2, L2, mf(L2, 1800, 55)),
                   c=(N2, L2, mf(L2, 1500, 56, uniform_only=True)))])

    # three replicates, p-value mode, one replicate lacks chr2 (NULL padding, 1735-1750)
    yield dict(
        name="reps3_p_missing", names=N2, args=["-a", "100"],
        reps=[dict(t=(N2, L2, mf(L2, 2000, 61)), c=None),
              dict(t=(N2[:1], L2[:1], mf(L2[:1], 1500, 62)), c=None),
              dict(t=(N2, L2, mf(L2, 1800, 63)), c=None)])

    L3 = [30_000, 20_000, 8_000]
    N3 = ["chr1", "chr2", "chrX"]
    yield dict(
        name="bedx", names=N3, args=["-e", "chrX", "-a", "40", "-q", "0.2"],
        bed=[("chr1", 0, 1500), ("chr1", 9000, 9800), ("chr1", 9500, 11000),
             ("chr1", 29000, 31000), ("chr2", 5000, 5001), ("chr2", 19990, 20000),
             ("chr2", 25000, 26000)],
        reps=[dict(t=(N3, L3, mf(L3, 3000, 71)),
                   c=(N3, L3, mf(L3, 2500, 72, uniform_only=True)))])

    yield dict(
        name="bedx_noctrl", names=N3, args=["-a", "40"],
        bed=[("chr1", 0, 1500), ("chr1", 12000, 13000), ("chr2", 19000, 20000)],
        reps=[dict(t=(N3, L3, mf(L3, 3000, 73)), c=None)])

    # chrC appears only in the control header (save = false, Genrich.c:4244); chr2 has
    # treatment reads but no control reads
    NC = ["chr1", "chr2", "chrC"]
    LC = [30_000, 20_000, 10_000]
    ctrl = mf(LC, 2500, 82, uniform_only=True)
    ctrl = ctrl[ctrl["chrom"] != 1]
    yield dict(
        name="ctrl_only_chrom", names=NC, args=["-a", "40", "-L", "45000"],
        reps=[dict(t=(NC[:2], LC[:2], mf(LC[:2], 2500, 81)),
                   c=(NC, LC, ctrl))])

    # unpaired alignments: -y (as is), -w (fixed extension), -x (average fragment length), with a MAPQ
    # filter and assorted records the reader must skip; the same stream also as BAM input
    for nm, extra in (("unpaired_y", ["-y"]), ("unpaired_w", ["-w", "150", "-m", "10"]), ("unpaired_x", ["-x"]),
                      ("unpaired_bam_atac", ["-y", "-j", "-D", "-d", "80"])):
        yield dict(
            name=nm, names=N2, args=extra + ["-a", "20"], mixed=dict(seed=7, bam=nm.endswith("bam_atac")),
            reps=[dict(t=(N2, L2, mf(L2, 2500, 95)), c=(N2, L2, mf(L2, 2000, 96, uniform_only=True)))])

    # -r / -R: PCR-duplicate removal -- pairs only; with unpaired alignments kept (-y); BAM input with -x
    for nm, extra, isbam in (("dups_pairs", ["-r"], False), ("dups_y", ["-r", "-y", "-s", "1.5"], False),
                             ("dups_x_bam", ["-r", "-x"], True)):
        yield dict(
            name=nm, names=N2, args=extra + ["-a", "20"], mixed=dict(seed=21, bam=isbam, writer="dups"),
            reps=[dict(t=(N2, L2, mf(L2, 2500, 97)), c=(N2, L2, mf(L2, 2000, 98, uniform_only=True)))])

    # valid but unusual records: SAM lines without optional fields (QUAL keeps its line feed), records
    # without SEQ (the BAM reader takes l_seq = 0 at face value), soft clips -- with -r, where qualities count
    # (the BAM case keeps to proper pairs: on an unpaired reverse read without SEQ the reference's own
    # arithmetic ends in "Invalid pileup value (< 0)")
    for nm, extra, isbam in (("quirks_sam", ["-r", "-y"], False), ("quirks_bam", ["-r"], True)):
        yield dict(
            name=nm, names=N2, args=extra + ["-a", "20"], mixed=dict(seed=33, bam=isbam, writer="dups", quirks=0.3),
            reps=[dict(t=(N2, L2, mf(L2, 2000, 101)), c=(N2, L2, mf(L2, 1500, 102, uniform_only=True)))])

    # saveInterval's int16 limits (Genrich.c:2558-2573): > 32,767 alignments that start on one base ("skipped due to
    # overflow"), > 32,768 that end on one ("underflow"), in shuffled order among ordinary reads; 30 % of them unpaired
    # and extended to the average fragment length at the end of the file (-x: the dropped pairs count as length 0).
    # The control piles 20,000 reads on the same bases: nothing of the treatment's state may be left.
    yield dict(
        name="saturate16", names=["chrA"], args=["-x", "-a", "20"], mixed=dict(seed=5, bam=False), int16=True,
        reps=[dict(t=(["chrA"], L1, int16_towers(L1, 47_600, 47_400, 61)),
                   c=(["chrA"], L1, int16_towers(L1, 20_000, 20_000, 62)))])

    # -X: no peak calling, just the -f log (logIntervals, Genrich.c:837)
    yield dict(
        name="nopeaks_log", names=N2, args=["-X", "-q", "0.05"],
        reps=[dict(t=(N2, L2, mf(L2, 1500, 91)), c=None),
              dict(t=(N2, L2, mf(L2, 1500, 92)), c=None)])


# -P re-calls on the -f log of a case (callPeaksLog, Genrich.c:1277-1488): new thresholds, new -e / -E
P_RUNS = {
    "ctrl_q": [dict(args=["-q", "0.4", "-a", "15"]), dict(args=["-p", "0.001", "-a", "10", "-g", "20", "-l", "50"])],
    "reps3": [dict(args=["-p", "0.01", "-a", "50"]), dict(args=["-q", "0.3", "-a", "5", "-e", "chr2"])],
    "bedx": [dict(args=["-q", "0.25", "-a", "30"], bed=[("chr1", 3000, 3400), ("chr1", 12000, 12345), ("chr2", 0, 777)])],
    "basic": [dict(args=["-a", "100", "-L", "1000"], bed=[("chrA", 12500, 12520)])],
}


def write_input(path, names, lens, ev, mixed, seed_off, prefix):
    """The synthetic SAM / BAM input of one sample, by the writer the case asks for."""
    if mixed and mixed.get("writer") == "dups":
        synth.write_sam_dups(path, names, lens, ev, mixed["seed"] + seed_off, name_prefix=prefix, bam=mixed["bam"],
                             quirks=mixed.get("quirks", 0.0))
    elif mixed:
        synth.write_sam_mixed(path, names, lens, ev, mixed["seed"] + seed_off, name_prefix=prefix, bam=mixed["bam"])
    else:
        synth.write_sam(path, names, lens, ev, name_prefix=prefix)


def gz_copy(src, dst):
    with open(src, "rb") as f, gzip.GzipFile(dst, "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/Genrich missing: run `make -C oracle` in the build container")
    for extra in int16_extra():
        if not sys.argv[1:] or extra in sys.argv[1:]:
            make_int16_extra(extra)
    for case in cases():
        if sys.argv[1:] and case["name"] not in sys.argv[1:]:
            continue  # (python make_golden.py <case> ...: only those)
        out_dir = os.path.join(HERE, case["name"])
        os.makedirs(out_dir, exist_ok=True)
        tmp = os.path.join("/tmp/genrich_golden", case["name"])
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        tfiles, cfiles = [], []
        chrom_order, chrom_len = [], {}
        saves = []

        def seen(names, lens):
            for n, l in zip(names, lens):
                if n not in chrom_len:
                    chrom_order.append(n)
                    chrom_len[n] = l

        for r, rep in enumerate(case["reps"]):
            names, lens, ev = rep["t"]
            mixed = case.get("mixed")
            ext = "bam" if mixed and mixed["bam"] else "sam"
            p = os.path.join(tmp, f"t{r}.{ext}")
            write_input(p, names, lens, ev, mixed, 0, f"t{r}_")
            tfiles.append(p)
            seen(names, lens)
            saves.append(list(names))
            if rep["c"] is None:
                cfiles.append(None)
            elif rep["c"] == "null":
                cfiles.append("null")
            else:
                names, lens, ev = rep["c"]
                p = os.path.join(tmp, f"c{r}.{ext}")
                write_input(p, names, lens, ev, mixed, 1, f"c{r}_")
                cfiles.append(p)
                seen(names, lens)
        args = [REF, "-t", ",".join(tfiles), "-v",
                "-f", os.path.join(tmp, "out.log"), "-k", os.path.join(tmp, "out.pile"),
                "-b", os.path.join(tmp, "events.bed")]
        if "-X" not in case["args"]:
            args += ["-o", os.path.join(tmp, "out.narrowPeak")]
        if any(c is not None for c in cfiles):
            args += ["-c", ",".join(c if c else "null" for c in cfiles)]
        if case.get("bed"):
            bp = os.path.join(tmp, "x.bed")
            with open(bp, "w") as f:
                for c, s, e in case["bed"]:
                    f.write(f"{c}\t{s}\t{e}\n")
            args += ["-E", bp]
        args += case["args"]
        if "-r" in case["args"]:
            args += ["-R", os.path.join(tmp, "out.dups")]
        res = subprocess.run(args, capture_output=True, text=True)
        if res.returncode != 0:
            sys.exit(f"{case['name']}: reference failed:\n{res.stderr}")
        err = res.stderr
        skip = []
        if "-e" in case["args"]:
            skip = case["args"][case["args"].index("-e") + 1].split(",")
        beds = {}
        for n in chrom_order:
            regs = [(s, e) for c, s, e in case.get("bed", []) if c == n]
            beds[n] = merged_bed(regs, chrom_len[n]) if n not in skip else []
        meta = dict(
            name=case["name"], args=case["args"],
            chroms=[dict(name=n, len=chrom_len[n], skip=n in skip, bed=beds[n]) for n in chrom_order],
            replicates=[dict(save=[n in s for n in chrom_order],
                             control=(None if c is None else ("null" if c == "null" else "file")),
                             expt_name=os.path.basename(t),
                             ctrl_name=(None if c is None else os.path.basename(c)))
                        for s, c, t in zip(saves, cfiles, tfiles)],
            ref_lambda=[float(v) for v in re.findall(r"Background pileup value: ([0-9.]+)", err)],
            ref_factor=[float(v) for v in re.findall(r"Scaling factor for control pileup: ([0-9.]+)", err)],
            ref_genome_len=[int(v) for v in re.findall(r"Genome length: (\d+)bp", err)],
            ref_peaks=[[int(a), int(b)] for a, b in re.findall(r"Peaks identified: (\d+) \((\d+)bp\)", err)],
            tmp_prefix=tmp + "/",
            ref_dups=[l.strip() for l in err.splitlines() if "aln sets:" in l or "duplicates:" in l],
        )
        with open(os.path.join(out_dir, "case.json"), "w") as f:
            json.dump(meta, f, indent=1)
        if case.get("int16"):  # the reference's read-by-read warnings, in order
            with gzip.GzipFile(os.path.join(out_dir, "out.int16.gz"), "wb", mtime=0) as g:
                g.write("".join(l + "\n" for l in err.splitlines() if "skipped due to" in l).encode())
        for fn in ("events.bed", "out.narrowPeak", "out.log", "out.pile", "out.dups"):
            src = os.path.join(tmp, fn)
            if os.path.exists(src):
                gz_copy(src, os.path.join(out_dir, fn + ".gz"))
        p_meta = []
        for k, pr in enumerate(P_RUNS.get(case["name"], [])):
            pargs = [REF, "-P", "-f", os.path.join(tmp, "out.log"), "-o", os.path.join(tmp, f"out.P{k}.narrowPeak")] + pr["args"]
            if pr.get("bed"):
                bp = os.path.join(tmp, f"xP{k}.bed")
                with open(bp, "w") as f:
                    for c, s_, e in pr["bed"]:
                        f.write(f"{c}\t{s_}\t{e}\n")
                pargs += ["-E", bp]
            r2 = subprocess.run(pargs, capture_output=True, text=True)
            if r2.returncode != 0:
                sys.exit(f"{case['name']}: reference -P failed:\n{r2.stderr}")
            gz_copy(os.path.join(tmp, f"out.P{k}.narrowPeak"), os.path.join(out_dir, f"out.P{k}.narrowPeak.gz"))
            p_meta.append(dict(args=pr["args"], bed=pr.get("bed", [])))
        meta["p_runs"] = p_meta
        with open(os.path.join(out_dir, "case.json"), "w") as f:
            json.dump(meta, f, indent=1)
        n_np = sum(1 for _ in open(os.path.join(tmp, "out.narrowPeak"))) if "-X" not in case["args"] else 0
        print(f"{case['name']:18s} peaks={n_np:4d} lambda={meta['ref_lambda']} factor={meta['ref_factor']}")
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
