"""one-off: more seeds of tests/test_gpu_random.py's randomized GPU-vs-oracle parity case"""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_random as t
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []
for s in range(lo, hi):
    try:
        t.test_random_build_and_search_parity(s)
    except AssertionError as e:
        bad.append(s); print("seed", s, "FAILED", str(e)[:300], flush=True)
    if (s - lo) % 20 == 19:
        print("done", s + 1 - lo, "cases in %.0f s, failures %s" % (time.time() - t0, bad), flush=True)
print("failures:", bad)
