"""Build container only (needs oracle/_ref/Genrich): a table of small damaged or unusual SAM inputs with
what the reference makes of them -- exit status, its `Error!` line, and the -b event stream when it gets
that far.  Written to tests/golden/cli_errors.json.gz; replayed against the host program by
tests/test_host_cli.py::test_cli_damaged_sam_as_the_reference.  Data only: the inputs are synthetic, the
expected values are the reference's output.  usage: python tests/golden/make_cli_errors.py"""
import gzip, json, os, random, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from genrich_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "Genrich")
N2 = ["chr1", "chr2"]
L = [9000, 5000]


def load_mutator():
    src = open(os.path.join(ROOT, "tools", "fuzz_host_malformed.py")).read().split("def one(seed):")[0]
    ns = {"__file__": os.path.join(ROOT, "tools", "fuzz_host_malformed.py")}
    exec(compile(src, "fuzz_host_malformed", "exec"), ns)
    return ns["mutate_sam"], ns["mutate_bam"]


def main():
    mutate_sam, mutate_bam = load_mutator()
    out = []
    tmp = "/tmp/genrich_cli_errors"
    os.makedirs(tmp, exist_ok=True)
    seen = set()
    for seed in range(400):
        rng = random.Random(seed)
        ev = synth.make_fragments(L, 12, seed=seed)
        p = os.path.join(tmp, "t.sam")
        (synth.write_sam_mixed if seed % 2 else synth.write_sam_dups)(p, N2, L, ev, seed, name_prefix="t_")
        text, kind = mutate_sam(open(p).read(), rng)
        open(p, "w").write(text)
        args = rng.choice([[], ["-y"], ["-r"], ["-y", "-r"], ["-x"]])
        bed = os.path.join(tmp, "ref.bed")
        if os.path.exists(bed):
            os.remove(bed)
        r = subprocess.run([REF, "-t", p] + args + ["-b", bed, "-o", "/dev/null"], capture_output=True, text=True, errors="replace")
        if r.returncode not in (0, 1):
            continue
        err = next((l for l in r.stderr.splitlines() if l.startswith("Error!")), "")
        late = any(k in err for k in ("no analyzable fragments", "Experimental sample", "peak", "No analyzable", "Invalid pileup"))
        key = (kind, err.split(":")[-1], tuple(args))
        if key in seen and kind != "none":
            continue
        seen.add(key)
        out.append(dict(kind=kind, args=args, sam=text, rc=0 if (late or r.returncode == 0) else 1, error="" if late else err,
                        events=open(bed).read() if os.path.exists(bed) and (late or r.returncode == 0) else None))
        if len(out) >= 90:
            break
    nsam = len(out)
    import base64
    for seed in range(1000, 3000):   # the same for BAM: truncations and single flipped bits
        rng = random.Random(seed)
        ev = synth.make_fragments(L, 12, seed=seed)
        p = os.path.join(tmp, "t.bam")
        (synth.write_sam_mixed if seed % 2 else synth.write_sam_dups)(p, N2, L, ev, seed, name_prefix="t_", bam=True)
        data, kind = mutate_bam(gzip.decompress(open(p, "rb").read()), rng)
        z = gzip.compress(data, mtime=0)
        open(p, "wb").write(z)
        args = rng.choice([[], ["-y"], ["-r"], ["-y", "-r"], ["-x"]])
        bed = os.path.join(tmp, "ref.bed")
        if os.path.exists(bed):
            os.remove(bed)
        r = subprocess.run([REF, "-t", p] + args + ["-b", bed, "-o", "/dev/null"], capture_output=True, text=True, errors="replace")
        if r.returncode not in (0, 1):
            continue
        # (not the inputs on which the reference reads outside its buffer: a negative l_seq, a name length >= 128)
        err = next((l for l in r.stderr.splitlines() if l.startswith("Error!")), "")
        late = any(k in err for k in ("no analyzable fragments", "Experimental sample", "peak", "No analyzable", "Invalid pileup"))
        key = ("bam", kind, err.split(":")[-1], tuple(args))
        if r.returncode == 0 or late:
            key += (seed % 6,)
        if key in seen:
            continue
        seen.add(key)
        out.append(dict(kind="bam " + kind, args=args, bam=base64.b64encode(z).decode(), rc=0 if (late or r.returncode == 0) else 1,
                        error="" if late else err,
                        events=open(bed, errors="replace").read() if os.path.exists(bed) and (late or r.returncode == 0) else None))
        if len(out) >= nsam + 60:
            break
    # valid BAM with alignment sets of every shape, some records without SEQ (tools/fuzz_host_vs_reference.py
    # --wild), under the options that extend or shift unpaired alignments: positions that wrapped in the
    # reference's 32-bit arithmetic must wrap here
    wsrc = open(os.path.join(ROOT, "tools", "fuzz_host_vs_reference.py")).read().split("def one(seed):")[0]
    wns = {"__file__": os.path.join(ROOT, "tools", "fuzz_host_vs_reference.py")}
    sys.argv = [sys.argv[0], "0", "0"]
    exec(compile(wsrc, "fuzz_host_vs_reference", "exec"), wns)
    nbefore = len(out)
    for seed in range(5000, 5400):
        rng = random.Random(seed)
        p = os.path.join(tmp, "t.bam")
        wns["write_wild"](p, N2, L, seed, 25, "w_", bam=True)
        args = rng.choice([["-y"], ["-w", str(rng.randint(50, 400))], ["-y", "-j"], ["-w", "399", "-j", "-s", "1", "-r"], ["-x"],
                           ["-y", "-j", "-d", "183", "-D"]])
        bed = os.path.join(tmp, "ref.bed")
        if os.path.exists(bed):
            os.remove(bed)
        r = subprocess.run([REF, "-t", p] + args + ["-b", bed, "-o", "/dev/null"], capture_output=True, text=True, errors="replace")
        if r.returncode not in (0, 1):
            continue
        err = next((l for l in r.stderr.splitlines() if l.startswith("Error!")), "")
        late = any(k in err for k in ("no analyzable fragments", "Experimental sample", "peak", "No analyzable", "Invalid pileup"))
        if r.returncode != 0 and not late:
            continue
        out.append(dict(kind="bam wild", args=args, bam=base64.b64encode(open(p, "rb").read()).decode(), rc=0, error="",
                        events=open(bed, errors="replace").read()))
        if len(out) >= nbefore + 12:
            break
    with gzip.GzipFile(os.path.join(HERE, "cli_errors.json.gz"), "wb", mtime=0) as g:
        g.write(json.dumps(out, indent=0).encode())
    print(len(out), "cases;", sum(1 for c in out if c["rc"]), "failing;", nbefore - nsam, "damaged BAM;", len(out) - nbefore, "wild BAM")


main()
