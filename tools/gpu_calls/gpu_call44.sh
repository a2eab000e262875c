cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout -s KILL 400 python bench.py --steps 20 --warmup 5 ) > gpurun_out/c44_line_config2.json 2> gpurun_out/c44_line_config2.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c44_line_config2.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), round(d["value"],1), d["gate"]["passed"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["whole_step"], d.get("h2d"), d.get("e2e_from_pinned",{}).get("ms_per_step"), d["cpu_baseline"])
PY
