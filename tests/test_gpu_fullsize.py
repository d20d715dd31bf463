"""BASELINE full size (configs[1]/[2]: 1M x 768 f32) through size-independent properties --
the oracle does not finish at this size in test time, so what is checked is what must hold for
any correct run: sorted (d, id) results without duplicates, distances that are exactly the
distance-batch kernel's values, determinism, the reference's layer invariants
(search.rs:142-171) and its self-recall assertion (lib.rs:2218-2224: >= 0.9 after generate)."""
import numpy as np
import pytest

import parallel_hnsw_amd as ph

pytestmark = pytest.mark.gpu

N, DIM = 1_000_000, 768


@pytest.fixture(scope="module")
def built():
    store = ph.VectorStore.clustered(N, DIM, seed=42)
    h = ph.Hnsw.generate(store, np.arange(N, dtype=np.uint64), ph.BuildParameters())
    return store, h


def test_layer_invariants_full_size(built):
    store, h = built
    layers = h.layers
    sizes = [l.node_count() for l in layers]
    assert sizes[-1] == N and sizes == sorted(sizes)
    for up, lo in zip(layers[:-1], layers[1:]):
        assert (np.diff(up.nodes.astype(np.int64)) > 0).all()           # strictly increasing
        assert np.isin(up.nodes, lo.nodes).all()                        # nested
    bottom = layers[-1]
    assert (bottom.nodes == np.arange(N, dtype=np.uint64)).all()
    nb = bottom.neighbors
    assert nb.shape == (N, 48)
    live = nb != ph.EMPTY
    assert ((~live[:, :-1]) <= (~live[:, 1:])).all()                    # sentinels trailing only
    assert (nb[live] < N).all()
    assert not (nb == np.arange(N, dtype=np.uint64)[:, None]).any()    # no self loops
    sample = nb[::5000]
    for row in sample:
        r = row[row != ph.EMPTY]
        assert len(set(r.tolist())) == len(r)                           # rows are duplicate free


def test_search_properties_full_size(built):
    store, h = built
    q = ph.VectorStore.clustered(4096, DIM, seed=42, first=2 ** 32).read()
    sp = ph.SearchParameters(128, 128, 2)
    ids, d, ln, st = h.search_batch(queries=q, sp=sp, stats=True)
    ids2, d2, ln2, st2 = h.search_batch(queries=q, sp=sp, stats=True)
    np.testing.assert_array_equal(ids, ids2)                            # idempotent / deterministic
    np.testing.assert_array_equal(d.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_array_equal(st, st2)
    assert (ln == 128).all()
    assert (np.diff(d, axis=1) >= 0).all()                              # sorted by distance
    ties = np.diff(d, axis=1) == 0
    assert (np.diff(ids.astype(np.int64), axis=1)[ties] > 0).all()      # ties by id
    for i in range(0, 4096, 256):
        assert len(set(ids[i].tolist())) == 128                         # no duplicates
        # every reported distance is exactly compare_vec(query, Stored(id))
        np.testing.assert_array_equal(store.compare_vec(ph.Unstored(q[i]), ids[i]).view(np.uint32),
                                      d[i].view(np.uint32))
    assert (st[:, 0] >= ln).all() and (st[:, 1] >= h.layer_count()).all()
    # a larger queue can only improve the k-th distance
    ids3, d3, ln3 = h.search_batch(queries=q[:512], sp=ph.SearchParameters(300, 300, 2))
    assert (d3[:, 9] <= d[:512, 9] + 1e-7).all()


def test_self_recall_full_size(built):
    """test_recall (lib.rs:2217-2231): stored vectors find themselves first"""
    store, h = built
    qids = np.arange(0, N, 97, dtype=np.uint64)
    ids, d, ln = h.search_batch(qids=qids, sp=ph.SearchParameters(300, 300, 2))
    recall = float(np.mean(ids[:, 0] == qids))
    assert recall >= 0.9, recall
    assert np.abs(d[ids[:, 0] == qids, 0]).max() < 1e-5
    assert h.stochastic_recall() >= 0.9


def test_search_sample_equals_oracle_full_size(built):
    """the oracle cannot build at this size in test time, but it can search: a sample of queries
    over the GPU-built 1M x 768 graph must come back bit-identical (ids, distances, counters), in
    the small-batch path and inside a 40 000-query batch (split, cell-ordered descent)"""
    import oracle
    store, h = built
    rows = store.read()
    ix = oracle.Index(rows, dim=DIM, metric=oracle.METRIC_COSINE_HALF, sum_mode=oracle.SUM_BLOCKED64)
    for l in h.layers:
        ix.push_layer(l.nodes, l.neighbors, l.neighborhood_size)
    qs = ph.VectorStore.clustered(40_000, DIM, seed=42, first=2 ** 34)
    q = qs.read()
    sp = ph.SearchParameters(128, 128, 8)
    m = 192
    ci, cd, cl, cs = ix.search(queries=q[:m], sp=(128, 128, 8), stats=True)
    for got in (h.search_batch(queries=q[:m], sp=sp, stats=True), h.search_batch(queries=q, sp=sp, stats=True)):
        np.testing.assert_array_equal(got[0][:m], ci)
        np.testing.assert_array_equal(got[1][:m].view(np.uint32), cd.view(np.uint32))
        np.testing.assert_array_equal(got[2][:m], cl)
        np.testing.assert_array_equal(got[3][:m], cs)


def test_link_round_searches_equal_oracle_full_size(built):
    """the search step of a link round (lib.rs:1107-1117: every node searches the stack with itself
    excluded, the first neighborhood_size results become proposals) for a slice of bottom-layer and
    of 83k-layer nodes on the 1M graph: the phase entry point of the ABI against the oracle's"""
    import ctypes as C
    import torch
    import oracle
    from parallel_hnsw_amd._lib import check, lib
    store, h = built
    ix = oracle.Index(store.read(), dim=DIM, metric=oracle.METRIC_COSINE_HALF, sum_mode=oracle.SUM_BLOCKED64)
    for l in h.layers:
        ix.push_layer(l.nodes, l.neighbors, l.neighborhood_size)
    L = oracle.lib()
    dev = torch.device("cuda", 0)
    M, count = 24, 96
    bp = ph.BuildParameters()
    sp = bp.optimization.search
    osp = oracle.SearchParams(sp.number_of_candidates, sp.upper_layer_candidate_count, sp.probe_depth)
    for lft, first in ((h.layer_count() - 1, 617_283), (h.layer_count() - 2, 40_000)):
        ids = torch.empty((count, M), dtype=torch.int32, device=dev)
        d = torch.empty((count, M), dtype=torch.float32, device=dev)
        ln = torch.empty(count, dtype=torch.int32, device=dev)
        check(lib().phnsw_link_search_device(h._h, lft, C.byref(sp), M, first, count, C.c_void_p(ids.data_ptr()),
                                             C.c_void_p(d.data_ptr()), C.c_void_p(ln.data_ptr())))
        torch.cuda.synchronize()
        oi = np.empty((count, M), dtype=np.uint64)
        od = np.empty((count, M), dtype=np.float32)
        ol = np.empty(count, dtype=np.uint64)
        assert L.orc_link_search(ix.h, lft, osp, M, first, count, oi.ctypes.data_as(C.c_void_p),
                                 od.ctypes.data_as(C.c_void_p), ol.ctypes.data_as(C.c_void_p), 8) == 0
        gi = ids.cpu().numpy().astype(np.uint64)
        gi[gi == 0xFFFFFFFF] = oracle.EMPTY
        np.testing.assert_array_equal(gi, oi)
        np.testing.assert_array_equal(d.cpu().numpy().view(np.uint32), od.view(np.uint32))
        np.testing.assert_array_equal(ln.cpu().numpy().astype(np.uint64), ol)
