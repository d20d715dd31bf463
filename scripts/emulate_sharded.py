"""The scaling model of the sharded build on ONE GPU: phnsw_build_sharded with an emulated world (one process plays
every rank in turn through the driver's real split / block layout / reassembly, csrc/sharded.hip) for worlds
1, 2, 4, 8.  Rank 0's critical path = its share of the sharded phases + the replicated phases + the reassembly
copies (all measured) + the all-gather time (modelled: bytes received from the other ranks over ONE xGMI link at
50 GB/s, the pessimistic ring bound; RCCL's mesh algorithm over 7 links is faster) + the host cost per collective
(measured in the two-rank rehearsal, passed in as HOST_US_PER_COLLECTIVE).

usage: emulate_sharded.py [n] [dim] [survey|tight|iid] ; env EMU_WORLDS=1,2,4,8  HOST_US_PER_COLLECTIVE=60"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallel_hnsw_amd as ph  # noqa: E402
from parallel_hnsw_amd.sharded import EmulatedComm, build_sharded  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
kind = sys.argv[3] if len(sys.argv) > 3 else "survey"
host_us = float(os.environ.get("HOST_US_PER_COLLECTIVE", "60"))
link_gbs = float(os.environ.get("LINK_GBS", "50"))

if kind == "tight":
    store = ph.VectorStore.clustered(n, dim, seed=42, first=0, n_clusters=1000, noise=1.0)
elif kind == "survey":
    store = ph.VectorStore.clustered(n, dim, seed=42, first=0, n_clusters=1000, noise=0.1 * dim ** 0.5)
else:
    store = ph.VectorStore.synthetic(n, dim, seed=42)
bp = ph.BuildParameters()
vids = np.arange(n, dtype=np.uint64)

t0 = time.time()
ref = ph.Hnsw.generate(store, vids, bp)
single = time.time() - t0
layers = [ref._layer(l).node_count() for l in range(ref.layer_count())]
print("phnsw_build: %.2f s (%.0f vectors/s)  layers %s" % (single, n / single, layers), flush=True)
ref_nb = ref._layer(ref.layer_count() - 1).neighbors if n <= 2_000_000 else None
del ref

rows = []
for w in [int(x) for x in os.environ.get("EMU_WORLDS", "1,2,4,8").split(",")]:
    if w == 1:
        rows.append({"world": 1, "rank0_s": single, "speedup": 1.0})
        continue
    h, st = build_sharded(store, vids, bp, EmulatedComm(w, 0))
    same = None
    if ref_nb is not None:
        same = bool(np.array_equal(h._layer(h.layer_count() - 1).neighbors, ref_nb))
    recv_other = st["all_gather_bytes"] * (w - 1) / w
    comm_model = recv_other / (link_gbs * 1e9)
    host = (st["all_gather_calls"] + st["all_reduce_calls"]) * host_us * 1e-6
    crit = st["seconds_sharded"] + st["seconds_replicated"] + st["seconds_comm"] + comm_model + host
    rows.append({"world": w, "rank0_s": round(crit, 3), "speedup": round(single / crit, 2),
                 "sharded_s": round(st["seconds_sharded"], 3), "replicated_s": round(st["seconds_replicated"], 3),
                 "reassembly_s": round(st["seconds_comm"], 3), "all_gather_model_s": round(comm_model, 3),
                 "host_collective_s": round(host, 4), "all_gather_gb_per_rank": round(st["all_gather_bytes"] / 1e9, 3),
                 "collectives": st["all_gather_calls"] + st["all_reduce_calls"], "phases": st["phases"],
                 "phases_not_split": st["phases_whole"], "others_s": round(st["seconds_others"], 2),
                 "bottom_layer_identical_to_phnsw_build": same,
                 "rank0_seconds_by_phase": {k: round(v, 3) for k, v in st["seconds_by_phase"].items()}})
    print(json.dumps(rows[-1]), flush=True)
    del h
print(json.dumps({"n": n, "dim": dim, "dataset": kind, "single_gpu_s": round(single, 3), "link_gbs": link_gbs,
                  "host_us_per_collective": host_us, "worlds": rows}))
