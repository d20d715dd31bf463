"""The frontier spill list (the part of the reference's unbounded `visit_queue` that does not fit
the LDS queue, lib.rs:182-191) has a fixed capacity per resident wave; a query that outgrows it is
flagged and re-run with more room.  PHNSW_OVF_CAP forces a tiny list so that these paths run:
results must still equal the oracle's."""
import os

import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph

pytestmark = pytest.mark.gpu


class tiny_spill:
    def __init__(self, cap):
        self.cap = cap

    def __enter__(self):
        os.environ["PHNSW_OVF_CAP"] = str(self.cap)

    def __exit__(self, *a):
        del os.environ["PHNSW_OVF_CAP"]


def test_search_reruns_overflowing_queries():
    import torch
    n, dim = 3000, 32
    rows = oracle.synth_rows(0, n, dim)
    oix = oracle.Index.generate(rows, np.arange(n), oracle.default_build_params(seed=1), dim=dim,
                                sum_mode=oracle.SUM_BLOCKED64)
    store = ph.VectorStore(rows[:, :dim])
    q = oracle.synth_rows(2 ** 32, 300, dim)[:, :dim]
    sp = (16, 16, 8)
    ci, cd, cl, cs = oix.search(queries=q, sp=sp, stats=True)
    with tiny_spill(8):
        g = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
        # the device entry point reports the overflow per query (status 5) and leaves the retry to the caller
        dev = torch.device("cuda", 0)
        qd = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
        ids = torch.empty((300, 16), dtype=torch.int32, device=dev)
        d = torch.empty((300, 16), dtype=torch.float32, device=dev)
        ln = torch.empty(300, dtype=torch.int32, device=dev)
        status = torch.empty(300, dtype=torch.int32, device=dev)
        g.search_batch_device(300, ph.SearchParameters(*sp), ids.data_ptr(), d.data_ptr(), ln.data_ptr(),
                              status.data_ptr(), queries=qd.data_ptr(), ldq=dim)
        torch.cuda.synchronize()
        st = status.cpu().numpy()
        assert (st == 5).sum() > 0 and set(st.tolist()) <= {0, 5}
        ok = st == 0
        np.testing.assert_array_equal(ids.cpu().numpy()[ok].astype(np.uint64)[:, :1], ci[ok][:, :1])
        # the host entry point re-runs them with 8x the room until they fit
        g2 = ph.Hnsw.from_layers(store, [oix.layer(l) for l in range(oix.layer_count)])
        gi, gd, gl, gs = g2.search_batch(queries=q, sp=ph.SearchParameters(*sp), stats=True)
    np.testing.assert_array_equal(gi, ci)
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32))
    np.testing.assert_array_equal(gl, cl)
    np.testing.assert_array_equal(gs, cs)


def test_build_reruns_overflowing_rounds():
    n, dim = 2500, 24
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    bp = ph.BuildParameters(seed=5)
    ref = ph.Hnsw.generate(store, np.arange(n), bp)
    with tiny_spill(32):
        h = ph.Hnsw.generate(store, np.arange(n), bp)
    assert h.layer_count() == ref.layer_count()
    for a, b in zip(h.layers, ref.layers):
        np.testing.assert_array_equal(a.nodes, b.nodes)
        np.testing.assert_array_equal(a.neighbors, b.neighbors)
