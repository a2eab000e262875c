"""GPU diagnostic: two processes on ONE GPU through the library's RCCL path (gx_set_rccl), if this RCCL build allows
several ranks per device; prints what happens.  The unique id travels through a file."""
import os, subprocess, sys, tempfile, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CODE = r'''
import os, sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
import backends as B, synth, genrich_amd
from genrich_amd.lib import rccl_unique_id
from genrich_amd.dist import lpt_partition
rank, world, path = int(sys.argv[1]), 2, sys.argv[2]
if rank == 0:
    uid = rccl_unique_id()
    open(path + ".tmp", "wb").write(uid); os.rename(path + ".tmp", path)
else:
    while not os.path.exists(path): time.sleep(0.05)
    uid = open(path, "rb").read()
lens = [300_000, 200_000, 150_000]
tr = synth.make_fragments(lens, 60_000, 5, peak_every=20_000, tower_every=70_000, frac_tower=0.1)
par = B.make_params(pq=0.2, qval=True, min_auc=20.0)
owner = lpt_partition(lens, world)
owned = np.array([o == rank for o in owner], dtype=np.uint8)
g = genrich_amd.Genrich(par); g.set_chroms(lens); g.set_owned(owned)
try:
    g.set_rccl(rank, world, uid)
    mine = tr[owned[tr["chrom"]].astype(bool)]
    g.sample_begin(0, None); g.push_events(mine); s = g.sample_end(); g.sample_no_control(); g.pvalues(); r = g.find_peaks()
    print("rank", rank, "ok", s, r[0], flush=True)
except Exception as e:
    print("rank", rank, "FAILED:", repr(e), flush=True)
'''
d = tempfile.mkdtemp()
env = dict(os.environ)
for extra in ({}, {"RCCL_ENABLE_MULTI_RANK_PER_GPU": "1", "NCCL_MULTI_RANK_GPU_ENABLE": "1"}):
    path = os.path.join(d, "uid%d" % len(extra))
    e2 = dict(env); e2.update(extra)
    ps = [subprocess.Popen([sys.executable, "-c", CODE % (ROOT, ROOT), str(r), path], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, env=e2) for r in range(2)]
    t0 = time.time()
    for p in ps:
        try:
            out, _ = p.communicate(timeout=max(1, 90 - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill(); out = "TIMEOUT"
        print("==== env", extra, "\n", "\n".join(l for l in out.splitlines() if "amdgpu.ids" not in l)[-1500:], flush=True)
