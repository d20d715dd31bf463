"""one-off: the randomized GPU-vs-oracle case with larger graphs (up to 40 000 nodes), deep probes and
small spill lists, to stress the frontier spill / overflow re-run paths"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_random as t
orig = t.random_case


def big_case(rng):
    c = orig(rng)
    c["n"] = int(rng.integers(5000, 40000))
    c["dim"] = int(rng.choice([2, 3, 8, 17, 32]))
    c["dup"] = int(rng.choice([1, 1, 2, 50]))
    c["pd"] = int(rng.choice([2, 9, 40, 200]))
    c["ef"] = int(rng.choice([1, 3, 17, 64, 300, 1024]))
    c["upper"] = c["ef"]
    return c


t.random_case = big_case
lo, hi = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:
    os.environ["PHNSW_OVF_CAP"] = sys.argv[3]
t0 = time.time(); bad = []
for s in range(lo, hi):
    try:
        t.test_random_build_and_search_parity(s)
    except AssertionError as e:
        bad.append(s); print("seed", s, "FAILED", str(e)[:300], flush=True)
    if (s - lo) % 10 == 9:
        print("done", s + 1 - lo, "cases in %.0f s, failures %s" % (time.time() - t0, bad), flush=True)
print("failures:", bad)
