"""GPU box: the HIP library against the CPU oracle on random runs (the generator of
tests/test_hip_parity.py::test_random_runs_against_oracle).  usage: fuzz_hip_vs_oracle.py SEED0 SEED1"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_parity as T  # noqa: E402

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    case, params = T._random_case(seed)
    try:
        o, h, so, sh = T.run_both(case, params)
        T.assert_same_run(o, h, so, sh, case)
    except Exception as ex:  # noqa: BLE001
        bad += 1
        print("seed", seed, type(ex).__name__, str(ex)[:200])
print("done, failures:", bad)
