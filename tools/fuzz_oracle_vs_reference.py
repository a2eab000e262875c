"""Build container only (needs oracle/_ref/Genrich): the oracle (CPU restatement) against the
unmodified reference on random runs (replicates, controls incl. null, -p/-q, -a/-l/-g, -e, -E,
multimapping): narrowPeak / -f / -k byte for byte.  usage: fuzz_oracle_vs_reference.py SEED0 SEED1
[--saturate]   (--saturate: inputs that drive the reference's int16 difference array into its skip
rule, Genrich.c:2565-2573 -- > 32,767 starts on one base, ends on one base, ends at the chromosome's
last position, and ends on the hot start base, in random order)"""
import sys, os, subprocess, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, backends as B
from genrich_amd import synth
import importlib.util
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, 'tests', 'golden', 'make_golden.py'))
mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
REF=os.path.join(ROOT, 'oracle', '_ref', 'Genrich')
N=["chr1","chr2","chr3"]
def one(seed):
    rng=random.Random(seed)
    L=[rng.randint(30_000,90_000), rng.randint(10_000,50_000), rng.randint(3_000,12_000)]
    scale=40 if "--mid" in sys.argv else 1      # --mid: chromosomes of Mbases, 10^5 fragments per sample
    L=[x*scale for x in L]
    d=f"/tmp/fuzz/o{seed}"; os.makedirs(d,exist_ok=True)
    nrep=rng.choice([1,1,1,2,3])
    tf,cf,reps=[],[],[]
    for r in range(nrep):
        ev=synth.make_fragments(L, rng.randint(800,4000)*scale, seed=seed*10+r, frac_peak=0.4, frac_tower=0.2)
        if rng.random()<0.3: ev=synth.add_multimap(ev,L,rng.choice([0.1,0.3]),seed=seed+3)
        t=f"{d}/t{r}.sam"; synth.write_sam(t,N,L,ev,name_prefix=f"t{r}_"); tf.append(t)
        if rng.random()<0.6:
            ct=synth.make_fragments(L, rng.randint(800,4000)*scale, seed=seed*10+5+r, uniform_only=True)
            c=f"{d}/c{r}.sam"; synth.write_sam(c,N,L,ct,name_prefix=f"c{r}_"); cf.append(c)
        else: cf.append(None)
    args=["-t",",".join(tf)]
    if any(cf): args+=["-c",",".join(c if c else "null" for c in cf)]
    extra=[]
    if rng.random()<0.5: extra+=["-q",str(rng.choice([0.05,0.2,0.5,1,0.999,1e-30]))]
    else: extra+=["-p",str(rng.choice([0.01,0.05,0.001,1,0.999,1e-30,1e-300]))]
    extra+=["-a",str(rng.choice([5,20,50,200,0,0.001,1e6]))]
    if rng.random()<0.3: extra+=["-l",str(rng.choice([rng.randint(0,200),100000]))]
    if rng.random()<0.3: extra+=["-g",str(rng.choice([rng.randint(0,300),0,100000]))]
    if rng.random()<0.3: extra+=["-e",rng.choice(["chr3","chr2"])]
    bed=[]
    if rng.random()<0.35:
        for _ in range(rng.randint(1,4)):
            ci=rng.randrange(3); s=rng.randint(0,L[ci]-10); e=min(L[ci], s+rng.randint(1,4000)); bed.append((N[ci],s,e))
        bp=f"{d}/x.bed"; open(bp,"w").write("".join(f"{c}\t{s}\t{e}\n" for c,s,e in bed)); extra+=["-E",bp]
    if rng.random()<0.3: extra+=["-s",str(rng.choice([1,5]))]
    run=[REF]+args+extra+["-v","-o",f"{d}/ref.np","-f",f"{d}/ref.log","-k",f"{d}/ref.pile","-b",f"{d}/ev.bed"]
    r=subprocess.run(run,capture_output=True,text=True)
    if r.returncode!=0:
        subprocess.run(["rm","-rf",d]); return None
    # events -> oracle
    idx={n:i for i,n in enumerate(N)}
    rows={}
    for line in open(f"{d}/ev.bed"):
        c,s,e,nm=line.rstrip("\n").split("\t"); _,cnt,kind,smp=nm.rsplit("_",3)
        rows.setdefault((int(smp),kind),[]).append((idx[c],int(s),int(e),int(cnt)))
    skip=[]
    if "-e" in extra: skip=extra[extra.index("-e")+1].split(",")
    beds=[mg.merged_bed([(s,e) for c,s,e in bed if c==n], L[i]) if n not in skip else [] for i,n in enumerate(N)]
    def opt(flag,default,conv=float): return conv(extra[extra.index(flag)+1]) if flag in extra else default
    qval="-q" in extra
    params=B.make_params(pq=opt("-q",0.0) if qval else opt("-p",0.01), qval=qval, min_auc=opt("-a",200.0), min_len=opt("-l",0,int), max_gap=opt("-g",100,int))
    o=B.Oracle(params); o.set_chroms(L,[n in skip for n in N],beds)
    for rr in range(nrep):
        o.sample_begin(0,None); o.push_events(np.array(rows.get((rr,"E"),[]),dtype=B.EVENT_DTYPE)); o.sample_end()
        if cf[rr]:
            o.sample_begin(1,None); o.push_events(np.array(rows.get((rr,"C"),[]),dtype=B.EVENT_DTYPE)); o.sample_end(); cname=cf[rr]
        else:
            o.sample_no_control(); cname=("null" if any(cf) else None)
        o.pvalues_to(f"{d}/or.pile", rr>0, N, tf[rr], cname)
    o.find_peaks_to(f"{d}/or.np", f"{d}/or.log", N, True)
    for a,b in (("ref.np","or.np"),("ref.log","or.log"),("ref.pile","or.pile")):
        if open(f"{d}/{a}","rb").read()!=open(f"{d}/{b}","rb").read():
            return f"seed {seed}: {a} differs: {' '.join(args+extra)}"
    subprocess.run(["rm","-rf",d]); return None
def saturating(seed):
    rng=np.random.default_rng(seed)
    L=[20000,8000]; names=N[:2]
    d=f"/tmp/fuzz/s{seed}"; os.makedirs(d,exist_ok=True)
    def blk(n,c,s,e):
        a=np.zeros(n,dtype=B.EVENT_DTYPE); a["chrom"]=c; a["start"]=s; a["end"]=e; a["count"]=1; return a
    n1,n2,n3,n4=int(rng.integers(33000,42000)),int(rng.integers(33000,42000)),int(rng.integers(33000,40000)),int(rng.integers(1000,20000))
    ev=np.concatenate([synth.make_fragments(L,2000,seed=seed),
                       blk(n1,0,5000,5000+rng.integers(100,300,n1)), blk(n2,0,9000-rng.integers(100,300,n2),9000),
                       blk(n3,1,8000-rng.integers(100,300,n3),8000), blk(n4,0,5000-rng.integers(100,300,n4),5000)])
    ev=ev[rng.permutation(len(ev))]
    if seed % 2:   # multimapped reads too: k alignments of one read are k consecutive events of count k, one of them on a hot base
        groups=[]
        for _ in range(int(rng.integers(5000,30000))):
            k=int(rng.choice([2,3,4,5,6,8,10])); g=synth.make_fragments(L,k,seed=int(rng.integers(1<<30)),uniform_only=True); g["count"]=k
            hot=int(rng.integers(3))
            if hot==0: g[0]["chrom"],g[0]["start"],g[0]["end"]=0,5000,5000+int(rng.integers(100,300))
            elif hot==1: g[0]["chrom"],g[0]["end"]=0,9000; g[0]["start"]=9000-int(rng.integers(100,300))
            groups.append(g)
        # groups go in between the unit events, each group contiguous
        cut=np.sort(rng.integers(0,len(ev)+1,len(groups))); pieces=[]; last=0
        for c,g in zip(cut,groups): pieces+= [ev[last:c], g]; last=c
        ev=np.concatenate(pieces+[ev[last:]])
    t=f"{d}/t.sam"; synth.write_sam(t,names,L,ev,name_prefix="t_")
    run=[REF,"-t",t,"-p","0.01","-a","20","-v","-o",f"{d}/ref.np","-f",f"{d}/ref.log","-k",f"{d}/ref.pile","-b",f"{d}/ev.bed"]
    r=subprocess.run(run,capture_output=True,text=True)
    if r.returncode!=0: return f"seed {seed}: reference failed {r.stderr[-200:]}"
    idx={n:i for i,n in enumerate(names)}; rows=[]
    for line in open(f"{d}/ev.bed"):
        c,s,e,nm=line.rstrip("\n").split("\t"); _,cnt,kind,smp=nm.rsplit("_",3); rows.append((idx[c],int(s),int(e),int(cnt)))
    o=B.Oracle(B.make_params(pq=0.01, qval=False, min_auc=20.0)); o.set_chroms(L,[False,False],[[],[]])
    o.sample_begin(0,None); o.push_events(np.array(rows,dtype=B.EVENT_DTYPE)); o.sample_end(); o.sample_no_control()
    o.pvalues_to(f"{d}/or.pile", False, names, t, None)
    o.find_peaks_to(f"{d}/or.np", f"{d}/or.log", names, True)
    for a,b in (("ref.np","or.np"),("ref.log","or.log"),("ref.pile","or.pile")):
        if open(f"{d}/{a}","rb").read()!=open(f"{d}/{b}","rb").read(): return f"seed {seed}: {a} differs (saturating input)"
    subprocess.run(["rm","-rf",d]); return None
if "--saturate" in sys.argv: one=saturating
bad=0
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    try: m=one(seed)
    except Exception as ex: m=f"seed {seed}: exception {ex!r}"
    if m: print(m); bad+=1
print("done, failures:",bad)
