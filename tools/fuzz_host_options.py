"""Build container only (needs oracle/_ref/Genrich): random command lines -- valid and invalid option
values, missing / unreadable files, unknown options, -h / -V -- on a small valid input.  The host program
(--events-only) and the reference must agree on success / failure, on the first message (program name
aside) and, when both run, on the -b event stream.  usage: fuzz_host_options.py SEED0 SEED1"""
import sys, os, subprocess, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genrich_amd import synth
REF=os.path.join(ROOT,'oracle','_ref','Genrich'); BIN=os.path.join(ROOT,'genrich_amd','genrich-amd')
N2=["chr1","chr2"]; L=[30000,20000]
d="/tmp/fuzz/opts"; os.makedirs(d,exist_ok=True)
ev=synth.make_fragments(L,400,seed=1); synth.write_sam_mixed(f"{d}/t.sam",N2,L,ev,1,name_prefix="t_")
ct=synth.make_fragments(L,300,seed=2,uniform_only=True); synth.write_sam_mixed(f"{d}/c.sam",N2,L,ct,2,name_prefix="c_")
open(f"{d}/x.bed","w").write("chr1\t100\t900\n")
def first_error(txt):
    out=[l for l in txt.splitlines() if l.startswith("Error!") or l.startswith("Usage") or "nvalid" in l or "equire" in l]
    out=[l.replace(REF,"PROG").replace(BIN,"PROG").replace("./Genrich","PROG").replace("genrich-amd","PROG") for l in out]
    return out[:1]
vals=["0","1","-1","0.5","1.5","x","","200","1e3","999999999999","0.0","-0.1","2x"]
bad=0
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    rng=random.Random(seed)
    args=[]
    if rng.random()<0.9: args+=["-t",rng.choice([f"{d}/t.sam",f"{d}/t.sam,{d}/t.sam","/nonexistent.sam",f"{d}/t.sam,"])]
    if rng.random()<0.4: args+=["-c",rng.choice([f"{d}/c.sam","null",f"{d}/c.sam,null","/nonexistent"])]
    for opt in rng.sample(["-a","-l","-g","-p","-q","-m","-s","-w","-d","-e","-E","-y","-x","-j","-D","-r","-z","-S","-X","-L","-N","-Z","--foo","-h","-V"], rng.randint(0,4)):
        if opt in ("-y","-x","-j","-D","-r","-z","-S","-X","-h","-V","--foo","-N","-Z"): args+=[opt]
        elif opt=="-e": args+=[opt, rng.choice(["chr2","chrZ","chr1,chr2",""])]
        elif opt=="-E":
            if rng.random()<0.6:   # a BED file of its own: unsorted, overlapping, off the end, damaged lines
                lines=[]
                for _ in range(rng.randint(0,6)):
                    c=rng.choice(["chr1","chr2","chrQ","chr1 "]); a=rng.randint(-5,31000); b=a+rng.randint(-10,4000)
                    lines.append(rng.choice([f"{c}\t{a}\t{b}", f"{c}\t{a}\t{b}\tname\t0\t+", f"{c}\t{a}", f"{c} {a} {b}", f"{c}\t{a}\tx", "", f"{c}\t\t{a}\t{b}", f"track name=x", f"{c}\t{a}\t{b}\r"]))
                open(f"{d}/y.bed","w").write("\n".join(lines)+rng.choice(["\n",""]))
                args+=[opt, rng.choice([f"{d}/y.bed", f"{d}/y.bed,{d}/x.bed"])]
            else: args+=[opt, rng.choice([f"{d}/x.bed","/nonexistent.bed"])]
        else: args+=[opt, rng.choice(vals)]
    r=subprocess.run([REF]+args+["-b",f"{d}/ref.bed","-o",f"{d}/ref.np"],capture_output=True,text=True,errors="replace")
    h=subprocess.run([BIN,"--events-only"]+args+["-b",f"{d}/hip.bed"],capture_output=True,text=True,errors="replace")
    if r.returncode not in (0,1): continue
    re_,he_=first_error(r.stderr),first_error(h.stderr)
    late=any(k in r.stderr for k in ("no analyzable fragments","Experimental sample","Invalid pileup","No analyzable"))
    msg=None
    if late:
        pass   # (the events-only host has no statistics to fail in and goes on to the next file)
    elif (r.returncode!=0)!=(h.returncode!=0): msg=f"rc ref={r.returncode} host={h.returncode} REF:{re_} HOST:{he_}"
    elif r.returncode!=0 and re_!=he_: msg=f"messages differ REF:{re_} HOST:{he_}"
    elif r.returncode==0 and "-h" not in args and "-V" not in args and os.path.exists(f"{d}/ref.bed") and os.path.exists(f"{d}/hip.bed") and open(f"{d}/ref.bed","rb").read()!=open(f"{d}/hip.bed","rb").read(): msg="-b differs"
    for f in ("ref.bed","hip.bed"):
        try: os.remove(f"{d}/{f}")
        except OSError: pass
    if msg: print(f"seed {seed}: {' '.join(args)}\n   {msg}"); bad+=1
print("done, failures:",bad)
