timeout -s KILL 200 python -m pytest tests -m gpu -x -q --timeout 120 2>&1 | tail -6
timeout -s KILL 60 python bench.py --no-cpu 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('ms',round(d['ms_per_step'],3),{k:round(v,3) for k,v in d['phases_ms'].items()})"
