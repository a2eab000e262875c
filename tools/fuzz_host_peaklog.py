"""Build container only (needs oracle/_ref/Genrich): -P (peak calling from a -f log, callPeaksLog
1277-1488) on the logs of the golden cases, intact and damaged, with random thresholds and -e / -E.  The
host program and the reference must agree on success / failure, on the `Error!` line and on the narrowPeak
output.  usage: fuzz_host_peaklog.py SEED0 SEED1"""
import sys, os, subprocess, random, gzip, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, 'oracle', '_ref', 'Genrich'); BIN = os.path.join(ROOT, 'genrich_amd', 'genrich-amd')
LOGS = sorted(glob.glob(os.path.join(ROOT, 'tests', 'golden', '*', 'out.log.gz')))


def mutate(text, rng):
    lines = text.split("\n")
    kind = rng.choice(["none", "none", "cell", "dropcol", "header", "trunc", "swap", "dupline", "na", "blank", "chrom"])
    body = [i for i, l in enumerate(lines) if l and not l.startswith("#")]
    if kind == "none" or not body:
        return text, "none"
    i = rng.choice(body)
    f = lines[i].split("\t")
    if kind == "cell":
        f[rng.randrange(len(f))] = rng.choice(["x", "", "-1", "1e9", "NA", "nan", "inf", "0.5x", "*"])
    elif kind == "dropcol":
        f.pop(rng.randrange(len(f)))
    elif kind == "header":
        lines[0] = rng.choice(["# chr\tstart\tend\texperimental\tcontrol\t-log(p)", "chr\tstart\tend", "", "# chr\tstart\tend\t-log(q)\t-log(p)",
                               lines[0].replace("-log(p)", "-log(P)"), lines[0] + "\t-log(p)"])
        return "\n".join(lines), kind
    elif kind == "trunc":
        return text[:rng.randint(1, len(text) - 1)], kind
    elif kind == "swap":
        j = rng.choice(body); lines[i], lines[j] = lines[j], lines[i]
        return "\n".join(lines), kind
    elif kind == "dupline":
        lines.insert(i, lines[i]); return "\n".join(lines), kind
    elif kind == "na":
        f = [("NA" if k >= 3 and rng.random() < 0.5 else v) for k, v in enumerate(f)]
    elif kind == "blank":
        lines.insert(i, ""); return "\n".join(lines), kind
    elif kind == "chrom":
        f[0] = rng.choice(["chrQ", "", f[0] + "x"])
    lines[i] = "\t".join(f)
    return "\n".join(lines), kind


def first_error(txt):
    for l in txt.splitlines():
        if l.startswith("Error!"):
            return l
    return ""


def one(seed):
    rng = random.Random(seed)
    d = f"/tmp/fuzz/p{seed}"; os.makedirs(d, exist_ok=True)
    src = rng.choice(LOGS)
    text, kind = mutate(gzip.open(src, "rt").read(), rng)
    log = f"{d}/in.log" + (".gz" if rng.random() < 0.3 else "")
    (gzip.open(log, "wt") if log.endswith(".gz") else open(log, "w")).write(text)
    args = ["-P", "-f", log]
    args += rng.choice([["-p", "0.01"], ["-q", "0.2"], ["-p", "0.2"], ["-q", "0.9"], []])
    if rng.random() < 0.6: args += ["-a", rng.choice(["1", "20", "200"])]
    if rng.random() < 0.3: args += ["-l", str(rng.randint(0, 300))]
    if rng.random() < 0.3: args += ["-g", str(rng.randint(0, 400))]
    if rng.random() < 0.3: args += ["-e", rng.choice(["chr2", "chr1", "chrX,chr2", "chrQ"])]
    if rng.random() < 0.3:
        bed = f"{d}/x.bed"
        open(bed, "w").write("".join(f"{rng.choice(['chr1', 'chr2', 'chrA'])}\t{(a := rng.randint(0, 30000))}\t{a + rng.randint(1, 5000)}\n" for _ in range(rng.randint(1, 4))))
        args += ["-E", bed]
    if rng.random() < 0.15: args += ["-v"]
    r = subprocess.run([REF] + args + ["-o", f"{d}/ref.np"], capture_output=True, text=True, errors="replace")
    h = subprocess.run([BIN] + args + ["-o", f"{d}/hip.np"], capture_output=True, text=True, errors="replace")
    if r.returncode not in (0, 1):
        subprocess.run(["rm", "-rf", d]); return None
    msg = None
    if (r.returncode != 0) != (h.returncode != 0):
        msg = f"rc ref={r.returncode} host={h.returncode}\n  REF: {first_error(r.stderr)}\n  HOST: {first_error(h.stderr)}"
    elif r.returncode != 0 and first_error(r.stderr) != first_error(h.stderr):
        msg = f"messages differ\n  REF: {first_error(r.stderr)}\n  HOST: {first_error(h.stderr)}"
    elif r.returncode == 0 and open(f"{d}/ref.np", "rb").read() != open(f"{d}/hip.np", "rb").read():
        msg = "narrowPeak differs"
    elif r.returncode == 0 and "-v" in args and r.stderr != h.stderr:
        msg = "-v output differs"
    if msg:
        return f"seed {seed} ({os.path.basename(os.path.dirname(src))}, {kind}): {' '.join(args)}\n  {msg}"
    subprocess.run(["rm", "-rf", d]); return None


bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    m = one(seed)
    if m: print(m); bad += 1
print("done, failures:", bad)
