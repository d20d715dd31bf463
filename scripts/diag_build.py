import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import parallel_hnsw_amd as ph
torch.cuda.set_device(0)
print("mem_get_info", [round(x / 2**30, 1) for x in torch.cuda.mem_get_info()], flush=True)
n = 1000000
store = ph.VectorStore.clustered(n, 768, seed=42, first=0, n_clusters=1000, noise=0.1 * 768 ** 0.5)
for i in range(3):
    if i == 2: os.environ["PHNSW_VERBOSE"] = "1"
    t = time.time()
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters())
    torch.cuda.synchronize()
    import ctypes
    out = (ctypes.c_uint64 * 3)(); ph.lib().phnsw_debug_alloc_stats(out)
    print("build %d: %.3f s; in hipMalloc/hipFree %.3f s over %d calls, %.1f GB" % (i, time.time() - t, out[0] * 1e-9, out[1], out[2] / 1e9), flush=True)
    del h
