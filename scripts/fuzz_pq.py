"""one-off: random shapes / table modes through tests/test_gpu_pq.py's build+search parity case"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import test_gpu_pq as t
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []
for s in range(lo, hi):
    rng = np.random.default_rng(5000 + s)
    m = int(rng.choice([4, 8, 12, 16, 24, 32, 48, 96]))
    dsub = int(rng.choice([1, 2, 3, 4, 8, 16]))
    dim = m * dsub
    if dim > 1536:
        continue
    n = int(rng.integers(300, 2500))
    ksub = int(rng.choice([2, 16, 37, 64, 100, 128, 255, 256]))
    mode = int(rng.choice([0, 1, 2]))
    try:
        t.test_pq_index_build_and_search_parity(n, dim, m, ksub, mode)
    except AssertionError as e:
        bad.append((s, n, dim, m, ksub, mode)); print("case", s, (n, dim, m, ksub, mode), "FAILED", str(e)[:300], flush=True)
    if (s - lo) % 20 == 19:
        print("done", s + 1 - lo, "cases in %.0f s, failures %s" % (time.time() - t0, bad), flush=True)
print("failures:", bad)
