"""Build container only (needs oracle/_ref/Genrich): random SAM / BAM inputs and option sets; the
host program (--events-only, no GPU needed) against the unmodified reference: -b event stream, -R
duplicate log and the -v accounting must be identical.  usage: fuzz_host_vs_reference.py SEED0 SEED1"""
import sys, os, subprocess, random, gzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from genrich_amd import synth
REF=os.path.join(ROOT, 'oracle', '_ref', 'Genrich'); BIN=os.path.join(ROOT, 'genrich_amd', 'genrich-amd')
N2=["chr1","chr2","chrM"]; 
def bgzf(data, rng):
    import struct, zlib
    out=bytearray(); off=0
    while True:
        chunk=data[off:off+rng.choice([rng.randint(40,400), rng.randint(400,5000), rng.randint(5000,65000)])]
        cobj=zlib.compressobj(rng.choice([0,1,6]), zlib.DEFLATED, -15)
        payload=cobj.compress(chunk)+cobj.flush()
        out+=b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0"+struct.pack("<H",18+len(payload)+8-1)+payload
        out+=struct.pack("<II", zlib.crc32(chunk)&0xFFFFFFFF, len(chunk))
        if not chunk: break
        off+=len(chunk)
    return bytes(out)
def write_wild(path, names, lens, seed, nreads, prefix, bam=False):
    """Syntactically valid SAM with alignment sets of every shape: 1-6 alignments per read name (now and then
    more than the reference's 128-alignment buffer), free combinations of the paired / proper / mate-unmapped /
    strand / first / last / secondary flags, mates that point at each other or nowhere, any chromosome,
    assorted CIGARs, scores present or not, SEQ / QUAL present or '*'."""
    rng=random.Random(seed)
    cigars=["50M","20M2D30M","5S45M","45M5S","10M3I37M","25M100N25M","5H50M","50=","30M2X18M","1M","3S20M2D20M7S"]
    qlen={"50M":50,"20M2D30M":50,"5S45M":50,"45M5S":50,"10M3I37M":50,"25M100N25M":50,"5H50M":50,"50=":50,"30M2X18M":50,"1M":1,"3S20M2D20M7S":50}
    import io
    f=io.StringIO()
    if True:
        f.write("@HD\tVN:1.0\tSO:queryname\n")
        for n,l in zip(names,lens): f.write(f"@SQ\tSN:{n}\tLN:{l}\n")
        for i in range(nreads):
            nm=f"{prefix}{i}"
            k=rng.choice([1,1,2,2,2,3,4,6]) if rng.random()>0.01 else rng.randint(125,135)
            alns=[]
            for a in range(k):
                c=rng.randrange(len(names)); pos=rng.randint(1,max(1,lens[c]-60))
                alns.append([c,pos])
            npair=0
            if rng.random()<0.6:     # the set begins with one or two coherent proper pairs (the second one secondary)
                for sec in ([0] if rng.random()<0.6 else [0,256]):
                    c=rng.randrange(len(names)); p1=rng.randint(1,max(1,lens[c]-400)); p2=p1+rng.randint(0,300)
                    sc=rng.randint(-30,0)
                    order=[(99|sec,p1,p2),(147|sec,p2,p1)]
                    if rng.random()<0.3: order.reverse()
                    for fl,pa,pb in order:
                        tg=[f"AS:i:{sc+rng.choice([0,0,-1])}"] if rng.random()<0.9 else []
                        f.write("\t".join([nm,str(fl),names[c],str(pa),str(rng.choice([0,10,30,42])),"50M","=",str(pb),"0","A"*50,"I"*50]+tg)+"\n")
                    npair+=1
            for a,(c,pos) in enumerate(alns if not npair or rng.random()<0.5 else []):
                flag=0
                for bit,pr in ((1,0.8),(2,0.6),(8,0.15),(16,0.5),(32,0.5),(256,0.3),(4,0.03),(2048,0.03),(1024,0.02)):
                    if rng.random()<pr: flag|=bit
                if flag&1: flag|=rng.choice([64,128])          # (neither or both: the reference stops with an error)
                elif rng.random()<0.3: flag|=rng.choice([64,128,192])
                if rng.random()<0.6 and len(alns)>1:   # mate = another alignment of the set
                    m=alns[rng.randrange(len(alns))]; rnext="=" if m[0]==c else names[m[0]]; pnext=m[1]
                else:
                    rnext=rng.choice(["*","=",names[rng.randrange(len(names))]]); pnext=rng.randint(0,lens[c])
                cg=rng.choice(cigars)
                if rng.random()<0.15: seq,qual="*","*"
                else:
                    seq="A"*qlen[cg]; qual="".join(chr(33+rng.randint(2,40)) for _ in range(qlen[cg])) if rng.random()<0.8 else "*"
                tags=[]
                if rng.random()<0.85: tags.append(f"AS:i:{rng.randint(-40,0)}")
                if rng.random()<0.5: tags.insert(rng.randrange(len(tags)+1),"NM:i:1")
                if rng.random()<0.2: tags.append("YS:i:-3")
                f.write("\t".join([nm,str(flag),names[c],str(pos),str(rng.choice([0,1,10,30,42])),cg,rnext,str(pnext),str(rng.randint(-500,500)),seq,qual]+tags)+"\n")
    if bam: open(path,"wb").write(sam_to_bam(f.getvalue(),names))
    else: open(path,"w").write(f.getvalue())
def sam_to_bam(text, names):
    """the records of write_wild as BAM (integer tags as one signed / unsigned byte when they fit)"""
    import struct, re
    hdr="".join(l+"\n" for l in text.split("\n") if l.startswith("@"))
    raw=bytearray(b"BAM\1"+struct.pack("<i",len(hdr))+hdr.encode()+struct.pack("<i",len(names)))
    lens={}
    for l in hdr.split("\n"):
        if l.startswith("@SQ"):
            d=dict(x.split(":",1) for x in l.split("\t")[1:]); lens[d["SN"]]=int(d["LN"])
    for n in names: raw+=struct.pack("<i",len(n)+1)+n.encode()+b"\0"+struct.pack("<i",lens[n])
    idx={n:i for i,n in enumerate(names)}; ops="MIDNSHP=X"
    for l in text.split("\n"):
        if not l or l.startswith("@"): continue
        q,flag,rn,pos,mapq,cg,rnext,pnext,tlen,seq,qual,*tags=l.split("\t")
        cig=b"".join(struct.pack("<I",(int(n)<<4)|ops.index(o)) for n,o in re.findall(r"(\d+)([MIDNSHP=X])",cg))
        lseq=0 if seq=="*" else len(seq)
        sq=b"\x11"*((lseq+1)//2); ql=(b"\xff"*lseq) if qual=="*" else bytes(ord(c)-33 for c in qual)
        aux=b""
        for t in tags:
            tg,ty,v=t.split(":"); v=int(v)
            aux+=tg.encode()+((b"c"+struct.pack("<b",v)) if -128<=v<0 else (b"C"+struct.pack("<B",v)) if 0<=v<256 else (b"i"+struct.pack("<i",v)))
        nref=idx[rn] if rnext=="=" else (idx[rnext] if rnext in idx else -1)
        body=struct.pack("<iiBBHHHiiii",idx[rn],int(pos)-1,len(q)+1,int(mapq),0,len(cig)//4,int(flag),lseq,nref,int(pnext)-1,int(tlen))
        body+=q.encode()+b"\0"+cig+sq+ql+aux
        raw+=struct.pack("<i",len(body))+body
    return gzip.compress(bytes(raw))
def one(seed):
    rng=random.Random(seed)
    L=[rng.randint(20_000,60_000), rng.randint(10_000,40_000), rng.randint(2_000,8_000)]
    d=f"/tmp/fuzz/c{seed}"; os.makedirs(d,exist_ok=True)
    ev=synth.make_fragments(L, rng.randint(300,3000), seed=seed)
    ct=synth.make_fragments(L, rng.randint(300,3000), seed=seed+1, uniform_only=True)
    writer=rng.choice(["mixed","dups","plain"]) if "--wild" not in sys.argv else "wild"; bam=rng.random()<0.4 and writer!="plain"
    ext="bam" if bam else "sam"
    quirks=rng.choice([0.0,0.0,0.25])   # records without optional fields / without SEQ, soft clips
    def wr(p,e,s,pre):
        if writer=="mixed": synth.write_sam_mixed(p,N2,L,e,s,name_prefix=pre,bam=bam)
        elif writer=="wild": write_wild(p,N2,L,s,rng.randint(100,800),pre,bam=bam)
        elif writer=="dups": synth.write_sam_dups(p,N2,L,e,s,name_prefix=pre,bam=bam,quirks=quirks)
        else: synth.write_sam(p,N2,L,e,name_prefix=pre)
    t=f"{d}/t.{ext}"; c=f"{d}/c.{ext}"; wr(t,ev,seed,"t_"); wr(c,ct,seed+7,"c_")
    threads=[]
    if rng.random()<0.5:   # BGZF with small random blocks: the host's thread pool, records straddling blocks
        for p in (t,c):
            raw=open(p,"rb").read()
            if bam: raw=gzip.decompress(raw)
            open(p,"wb").write(bgzf(raw, rng))
        threads=["--threads",str(rng.randint(2,5))]
    piped = None
    if not threads and not bam and rng.random()<0.4: piped = t   # plain SAM text through stdin
    args=["-t","-" if piped else t]
    if rng.random()<0.7: args+=["-c",c]
    single=rng.choice([None,"-y","-w","-x"])
    if single=="-w": args+=["-w",str(rng.randint(50,400))]
    elif single: args+=[single]
    if rng.random()<0.3:
        args+=["-j"]
        if rng.random()<0.5: args+=["-d",str(rng.randint(20,200))]
        if rng.random()<0.5: args+=["-D"]
    if rng.random()<0.4: args+=["-m",str(rng.randint(1,40))]
    if rng.random()<0.4: args+=["-s",str(rng.choice([0.5,1,2,5,20]))]
    dups = rng.random()<0.5
    if dups: args+=["-r"]
    if rng.random()<0.3: args+=["-e",rng.choice(["chrM","chr2","chrM,chr2"])]
    if rng.random()<0.3:
        bp=f"{d}/x.bed"
        with open(bp,"w") as f:
            for _ in range(rng.randint(1,5)):
                ci=rng.randrange(3); s=rng.randint(0,L[ci]-10); e=min(L[ci], s+rng.randint(1,3000)); f.write(f"{N2[ci]}\t{s}\t{e}\n")
        args+=["-E",bp]
    ra=[REF]+args+["-b",f"{d}/ref.bed","-o",f"{d}/ref.np","-v"]+(["-R",f"{d}/ref.dups"] if dups else [])
    ha=[BIN,"--events-only"]+threads+args+["-b",f"{d}/hip.bed","-v"]+(["-R",f"{d}/hip.dups"] if dups else [])
    r=subprocess.run(ra,capture_output=True,text=True,stdin=open(piped) if piped else None); h=subprocess.run(ha,capture_output=True,text=True,stdin=open(piped) if piped else None)
    # the reference may fail legitimately (e.g. no fragments): then both must fail
    if r.returncode!=0 and ("no analyzable fragments" in r.stderr or "Experimental sample" in r.stderr):
        subprocess.run(["rm","-rf",d]); return None   # the reference stops after the treatment file; the events-only host goes on
    if r.returncode!=0 or h.returncode!=0:
        # events-only host stops before the statistics: compare only when the reference got past ingest
        if "Experimental sample" in r.stderr or "No analyzable" in r.stderr or "peak" in r.stderr.lower() or "Invalid pileup" in r.stderr:
            pass
        elif r.returncode!=h.returncode:
            return f"seed {seed}: rc ref={r.returncode} hip={h.returncode}\n{' '.join(args)}\nREF: {r.stderr[-300:]}\nHIP: {h.stderr[-300:]}"
    if os.path.exists(f"{d}/ref.bed") and os.path.exists(f"{d}/hip.bed"):
        rb,hb=open(f"{d}/ref.bed","rb").read(),open(f"{d}/hip.bed","rb").read()
        # (a reference that stopped in its statistics has written the events up to that point only)
        if (rb!=hb) if r.returncode==0 else (not hb.startswith(rb[:rb.rfind(b"\n")+1])):
            return f"seed {seed}: -b differs: {' '.join(args)}"
    if dups and os.path.exists(f"{d}/ref.dups"):
        rd,hd=open(f"{d}/ref.dups","rb").read(),open(f"{d}/hip.dups","rb").read()
        if (rd!=hd) if r.returncode==0 else (not hd.startswith(rd[:rd.rfind(b"\n")+1])):
            return f"seed {seed}: -R differs: {' '.join(args)}"
    # verbose accounting lines (up to the point the host stops)
    def acct(txt): return [l for l in txt.splitlines() if l.startswith("  ") or l.startswith("Processing") or "Warning" in l]
    ra_,ha_=acct(r.stderr),acct(h.stderr)
    ra_=[l for l in ra_ if "Background" not in l and "Scaling" not in l and "Genome length" not in l]
    if r.returncode!=0:   # the reference stopped in its statistics: its log ends there
        ra_=[l for l in ra_ if "internal error" not in l]; ha_=ha_[:len(ra_)]
    if ra_[:len(ha_)]!=ha_: 
        import difflib
        return f"seed {seed}: -v differs: {' '.join(args)}\n"+"\n".join(list(difflib.unified_diff(ra_,ha_,lineterm=''))[:20])
    subprocess.run(["rm","-rf",d])
    return None
bad=0
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    m=one(seed)
    if m: print(m); bad+=1
print("done, failures:",bad)
