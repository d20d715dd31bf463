"""ad-hoc probe: run the sharded build driver on ONE GPU (world 1) with every engine phase
timed, to split the build into the part that shards over ranks and the part every rank repeats
(the Amdahl fraction of the multi-GPU build).  Not part of the tests."""
import os
import sys
import time
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import parallel_hnsw_amd as ph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
kind = sys.argv[3] if len(sys.argv) > 3 else "clustered"

SHARDED = {"layer_init_search", "layer_seed", "link_search", "recall_hits", "discover_hits"}
REPL = {"plan", "layer_begin", "layer_finish", "link_apply", "promote_from_hits", "promote_at_layer"}


class Timed:
    def __init__(self, eng):
        self._e = eng
        self.t = defaultdict(float)
        self.c = defaultdict(int)

    def __getattr__(self, name):
        f = getattr(self._e, name)
        if name not in SHARDED and name not in REPL:
            return f

        def wrapped(*a, **k):
            torch.cuda.synchronize()
            t0 = time.time()
            r = f(*a, **k)
            torch.cuda.synchronize()
            self.t[name] += time.time() - t0
            self.c[name] += 1
            return r
        return wrapped


if kind == "clustered":
    store = ph.VectorStore.clustered(n, dim, seed=42, first=0, n_clusters=1000, noise=1.0)
elif kind == "survey":
    store = ph.VectorStore.clustered(n, dim, seed=42, first=0, n_clusters=max(1000, n // 1000), noise=0.1 * dim ** 0.5)
else:
    store = ph.VectorStore.synthetic(n, dim, seed=42)
bp = ph.BuildParameters()
eng = Timed(ph.GpuEngine(store, bp))
comm = ph.TorchComm()
gathered = [0]
orig = comm.all_gather


def counting(t):
    gathered[0] += t.numel() * t.element_size()
    return orig(t)


comm.all_gather = counting
torch.cuda.synchronize()
t0 = time.time()
ph.ShardedBuilder(eng, comm).generate(np.arange(n, dtype=np.uint64))
torch.cuda.synchronize()
total = time.time() - t0
sh = sum(v for k, v in eng.t.items() if k in SHARDED)
rp = sum(v for k, v in eng.t.items() if k in REPL)
print("total %.2f s  (%.0f vectors/s)   sharded phases %.2f s   replicated phases %.2f s   driver/other %.2f s" % (
    total, n / total, sh, rp, total - sh - rp))
for k in sorted(eng.t, key=lambda k: -eng.t[k]):
    print("  %-22s %8.3f s  x%d   %s" % (k, eng.t[k], eng.c[k], "sharded" if k in SHARDED else "replicated"))
print("all-gather payload per rank (full result, any world): %.1f MB" % (gathered[0] / 1e6))


# ---- emulated ranks: rank 0 of a world of w is timed, the other ranks' ranges are computed
# untimed on the same GPU so the replica stays correct; gives the per-rank critical path
# including the fixed launch cost of the small layers
class EmuEngine(Timed):
    def __init__(self, eng, w):
        super().__init__(eng)
        self.w = w
        self.queue = []
        self.hits_rest = 0

    def _run(self, name, total, outs, call):
        """outs: rank 0's output views (its whole range: the emulation runs without the sub-chunk pipeline);
        call(first, count, outs).  The other ranks' blocks are computed untimed and queued, packed the way
        ShardedBuilder._phase packs them, for EmuComm.all_gather"""
        whole = total < ph.ShardedBuilder.SHARD_MIN  # short lists run whole on every rank, no collective
        chunk = total if whole else -(-total // self.w)
        torch.cuda.synchronize()
        t0 = time.time()
        call(0, min(chunk, total), outs)
        torch.cuda.synchronize()
        self.t[name] += time.time() - t0
        self.c[name] += 1
        if whole:
            return
        blocks = []
        for r in range(1, self.w):
            f = min(total, r * chunk)
            cnt = min(total, f + chunk) - f
            o2 = [torch.zeros((chunk,) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device) for o in outs]
            if cnt:
                call(f, cnt, [o[:cnt] for o in o2])
            cols = [o.reshape(chunk, -1).contiguous().view(torch.uint8) for o in o2]
            blocks.append(cols[0] if len(cols) == 1 else torch.cat(cols, dim=1))
        self.queue.append(blocks)

    def layer_begin(self, vids, W):
        self.n_layer = len(vids)
        return Timed.__getattr__(self, "layer_begin")(vids, W)

    def layer_init_search(self, first, count, ids, d, ln):
        self._run("layer_init_search", self.n_layer, [ids, d, ln],
                  lambda f, c, o: self._e.layer_init_search(f, c, *o))

    def layer_seed(self, ids, d, ln, first, count, rows, rows_d):
        self._run("layer_seed", self.n_layer, [rows, rows_d],
                  lambda f, c, o: self._e.layer_seed(ids, d, ln, f, c, *o))

    def link_search(self, lft, sp, M, first, count, ids, d, ln):
        self._run("link_search", self._e.layer_nodes(lft), [ids, d, ln],
                  lambda f, c, o: self._e.link_search(lft, sp, M, f, c, *o))

    def discover_hits(self, lft, sp, first, count, hit):
        self._run("discover_hits", self._e.layer_nodes(lft), [hit],
                  lambda f, c, o: self._e.discover_hits(lft, sp, f, c, *o))

    def recall_hits(self, at, op, first, count):
        torch.cuda.synchronize()
        t0 = time.time()
        hits, sel = self._e.recall_hits(at, op, 0, count)
        torch.cuda.synchronize()
        self.t["recall_hits"] += time.time() - t0
        self.c["recall_hits"] += 1
        self.hits_rest = 0
        for r in range(1, self.w if sel >= ph.ShardedBuilder.SHARD_MIN else 1):
            f = min(sel, r * count)
            h2, _ = self._e.recall_hits(at, op, f, min(sel, f + count) - f)
            self.hits_rest += h2
        return hits, sel


class EmuComm:
    def __init__(self, eng, w):
        self.e, self.world, self.rank, self.bytes = eng, w, 0, 0

    def all_gather(self, t):
        self.bytes += t.numel() * t.element_size() * self.world
        return torch.cat([t] + self.e.queue.pop(0), 0)

    def all_reduce_sum(self, values, device):
        return [values[0] + self.e.hits_rest]


ref_layers = None
for w in [int(x) for x in os.environ.get("EMU_WORLDS", "1,2,4,8").split(",")]:
    e = EmuEngine(ph.GpuEngine(store, bp), w)
    c = EmuComm(e, w)
    torch.cuda.synchronize()
    sb = ph.ShardedBuilder(e, c)
    sb.SUBCHUNKS = 1
    sb.generate(np.arange(n, dtype=np.uint64))
    torch.cuda.synchronize()
    sh = sum(v for k, v in e.t.items() if k in SHARDED)
    rp = sum(v for k, v in e.t.items() if k in REPL)
    comm_s = c.bytes * (w - 1) / w / 50e9  # ring all-gather, one xGMI link's worth
    layers = [e.layer_nodes(l) for l in range(e.layer_count())]
    print("emulated rank 0 of %d: sharded %.3f s + replicated %.3f s + comm(model) %.3f s = %.3f s  -> %.0f vectors/s  layers %s" % (
        w, sh, rp, comm_s, sh + rp + comm_s, n / (sh + rp + comm_s), layers), flush=True)
    print("   " + "  ".join("%s %.3f" % (k, v) for k, v in sorted(e.t.items(), key=lambda kv: -kv[1])), flush=True)
    del e, c

