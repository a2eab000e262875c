"""GPU diagnostic: the library's RCCL path (one-rank communicator) with and without torch loaded first."""
import os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CODE = r'''
import os, sys
sys.path.insert(0, "%s"); sys.path.insert(0, "%s/tests")
if %d:
    import torch
    torch.cuda.init(); x = torch.zeros(4, device="cuda"); print("torch first:", torch.__version__)
    if %d:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); print("torch nccl allreduce ok", t.tolist())
import numpy as np
import backends as B, synth, genrich_amd
from genrich_amd.lib import rccl_unique_id
os.environ["GX_FORCE_COLL"] = "1"
lens = [200_000, 150_000]
tr = synth.make_fragments(lens, 40_000, 5, peak_every=20_000, tower_every=70_000, frac_tower=0.1)
par = B.make_params(pq=0.3, qval=True, min_auc=20.0)
g = genrich_amd.Genrich(par); g.set_chroms(lens)
try:
    g.set_rccl(0, 1, rccl_unique_id())
    s = B.run_case(g, dict(lens=lens, replicates=[dict(save=None, treat=tr, ctrl=None)]))
    print("rccl path ok:", s, g.n_peaks)
except Exception as e:
    print("FAILED:", repr(e))
import subprocess
print(subprocess.run("grep -E 'rccl|amdhip|hsa-runtime' /proc/%%d/maps | awk '{print $6}' | sort -u" %% os.getpid(), shell=True, capture_output=True, text=True).stdout)
'''
for torch_first, pg in ((0, 0), (1, 0), (1, 1)):
    print(f"==== torch_first={torch_first} process_group={pg}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE % (ROOT, ROOT, torch_first, pg)], capture_output=True, text=True, timeout=280)
    print(r.stdout[-3000:]); print(r.stderr[-2500:])
