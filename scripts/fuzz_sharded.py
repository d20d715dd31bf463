"""one-off: the sharded build driver with an emulated world of w ranks on one GPU (every rank's
node range really computed, results concatenated as an all-gather would) against phnsw_build, on
random shapes, parameters and world sizes: the graphs must be bit-identical"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import parallel_hnsw_amd as ph


class EmuEngine:
    """runs every rank's share of a sharded phase on this GPU; rank 0's buffers get rank 0's share, the other
    ranks' blocks are queued packed the way ShardedBuilder._phase packs them (raw bytes side by side)"""
    def __init__(self, eng, w):
        self._e, self.w, self.queue, self.hits_rest, self.shard_min = eng, w, [], 0, 0

    def __getattr__(self, name):
        return getattr(self._e, name)

    def _run(self, total, outs, call):
        whole = total < self.shard_min  # short lists run whole on every rank, no collective
        chunk = total if whole else -(-total // self.w)
        call(0, min(chunk, total), outs)
        if whole:
            return
        blocks = []
        for r in range(1, self.w):
            f = min(total, r * chunk); cnt = min(total, f + chunk) - f
            o2 = [torch.zeros((chunk,) + tuple(o.shape[1:]), dtype=o.dtype, device=o.device) for o in outs]
            if cnt:
                call(f, cnt, [o[:cnt] for o in o2])
            cols = [o.reshape(chunk, -1).contiguous().view(torch.uint8) for o in o2]
            blocks.append(cols[0] if len(cols) == 1 else torch.cat(cols, dim=1))
        self.queue.append(blocks)

    def layer_begin(self, vids, W):
        self.n_layer = len(vids)
        return self._e.layer_begin(vids, W)

    def layer_init_search(self, first, count, ids, d, ln):
        self._run(self.n_layer, [ids, d, ln], lambda f, c, o: self._e.layer_init_search(f, c, *o))

    def layer_seed(self, ids, d, ln, first, count, rows, rows_d):
        self._run(self.n_layer, [rows, rows_d], lambda f, c, o: self._e.layer_seed(ids, d, ln, f, c, *o))

    def link_search(self, lft, sp, M, first, count, ids, d, ln):
        self._run(self._e.layer_nodes(lft), [ids, d, ln], lambda f, c, o: self._e.link_search(lft, sp, M, f, c, *o))

    def discover_hits(self, lft, sp, first, count, hit):
        self._run(self._e.layer_nodes(lft), [hit], lambda f, c, o: self._e.discover_hits(lft, sp, f, c, *o))

    def recall_hits(self, at, op, first, count):
        hits, sel = self._e.recall_hits(at, op, 0, count)
        self.hits_rest = 0
        if sel >= self.shard_min:
            for r in range(1, self.w):
                f = min(sel, r * count)
                self.hits_rest += self._e.recall_hits(at, op, f, min(sel, f + count) - f)[0]
        return hits, sel


class EmuComm:
    """no all_gather_async: the driver then runs a phase as one piece (the pipeline itself is unit-tested on CPU)"""
    def __init__(self, eng, w):
        self.e, self.world, self.rank = eng, w, 0

    def all_gather(self, t):
        return torch.cat([t] + self.e.queue.pop(0), 0)

    def all_reduce_sum(self, values, device):
        return [values[0] + self.e.hits_rest]


lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time(); bad = []
for s in range(lo, hi):
    rng = np.random.default_rng(3000 + s)
    n = int(rng.integers(300, 6000)); dim = int(rng.choice([4, 16, 48, 100])); w = int(rng.choice([2, 3, 5, 8]))
    dup = int(rng.choice([1, 1, 1, 25]))
    base = ph.VectorStore.synthetic(max(2, n // dup), dim, seed=s).read()
    rows = np.repeat(base, dup, axis=0)[:n].copy(); n = rows.shape[0]
    store = ph.VectorStore(rows)
    bp = ph.BuildParameters(order=int(rng.choice([3, 6, 12])), neighborhood_size=int(rng.integers(2, 16)), seed=s,
                            max_link_rounds=int(rng.choice([1, 2])), promote=int(rng.integers(0, 2)))
    bp.zero_layer_neighborhood_size = int(rng.integers(bp.neighborhood_size, 32))
    bp.optimization.search.number_of_candidates = bp.optimization.search.upper_layer_candidate_count = int(rng.choice([16, 40]))
    bp.optimization.recall_proportion = float(rng.choice([0.1, 1.0]))
    shard_min = int(rng.choice([0, 64, 700]))
    ref = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), bp)
    e = EmuEngine(ph.GpuEngine(store, bp), w); e.shard_min = shard_min
    h = ph.ShardedBuilder(e, EmuComm(e, w), shard_min=shard_min).generate(np.arange(n, dtype=np.uint64))
    ok = h.layer_count() == ref.layer_count()
    for x, y in zip(h.layers, ref.layers):
        ok = ok and np.array_equal(x.nodes, y.nodes) and np.array_equal(x.neighbors, y.neighbors)
    if not ok:
        bad.append(s); print("case", s, (n, dim, w, dup, shard_min), "MISMATCH", flush=True)
    if (s - lo) % 20 == 19:
        print("done", s + 1 - lo, "cases in %.0f s, failures %s" % (time.time() - t0, bad), flush=True)
print("failures:", bad)
