"""Dense top layers (csrc/tiny.hip): the tile pass that evaluates every query against every node
of the small top layers, and the table-id traversal that consumes it, must reproduce the per-hop
path bit for bit -- ids, distance bits, lengths, hop and distance counters -- and the oracle."""
import os

import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph

pytestmark = pytest.mark.gpu


class per_hop_path:
    """PHNSW_NO_TINY=1: every layer through the per-hop distance batch"""

    def __enter__(self):
        os.environ["PHNSW_NO_TINY"] = "1"

    def __exit__(self, *a):
        del os.environ["PHNSW_NO_TINY"]


def same(a, b):
    for x, y in zip(a, b):
        if x.dtype == np.float32:
            x, y = x.view(np.uint32), y.view(np.uint32)
        np.testing.assert_array_equal(x, y)


CASES = [
    # n, dim, metric, sp
    (6000, 768, ph.METRIC_COSINE_HALF, (104, 104, 8)),   # 3 chunks per lane, exact
    (6000, 100, ph.METRIC_ONE_MINUS_DOT, (64, 20, 2)),   # one partly filled chunk
    (3000, 1536, ph.METRIC_COSINE_HALF, (128, 128, 2)),  # 6 chunks per lane: 4 queries per wave
    (5000, 200, ph.METRIC_L2, (300, 300, 2)),            # L2 chain, queue of 300
    (700, 64, ph.METRIC_COSINE_HALF, (32, 32, 3)),       # every layer is a dense one
    (30000, 128, ph.METRIC_COSINE_HALF, (64, 64, 2)),    # table layer of ~2500 nodes: past PH_TINY_LDS_NODES (rows in global memory; in LDS on the small-batch kernels)
    (8000, 256, ph.METRIC_ONE_MINUS_DOT, (64, 64, 2)),   # one whole chunk per lane: the matrix-core table, 4 steps per leaf
]


@pytest.mark.parametrize("n,dim,metric,sp", CASES)
def test_dense_top_layers_match_per_hop_path_and_oracle(n, dim, metric, sp):
    rows = oracle.synth_rows(0, n, dim)[:, :dim]
    store = ph.VectorStore(rows, metric=metric)
    bp = ph.BuildParameters(seed=3, max_link_rounds=1)
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), bp)
    assert h._layer(0).node_count() <= 1024  # the shape does have dense top layers
    q = oracle.synth_rows(2 ** 32, 333, dim)[:, :dim]
    spx = ph.SearchParameters(*sp)
    dense = h.search_batch(queries=q, sp=spx, stats=True)
    with per_hop_path():
        hop = h.search_batch(queries=q, sp=spx, stats=True)
    same(dense, hop)
    # partial tiles: 1, 5 and 33 queries
    for m in (1, 5, 33):
        same(h.search_batch(queries=q[:m], sp=spx, stats=True), [x[:m] for x in hop])
    # Stored queries with exclude (the link-round form)
    qid = np.arange(0, n, 7, dtype=np.uint64)
    d2 = h.search_batch(qids=qid, sp=spx, exclude=qid, stats=True)
    with per_hop_path():
        h2 = h.search_batch(qids=qid, sp=spx, exclude=qid, stats=True)
    same(d2, h2)
    # and the oracle, on the same graph
    ix = oracle.Index(rows, dim=dim, metric=metric, sum_mode=oracle.SUM_BLOCKED64)
    for l in h.layers:
        ix.push_layer(l.nodes, l.neighbors, l.neighborhood_size)
    ci, cd, cl, cs = ix.search(queries=q, sp=sp, stats=True)
    np.testing.assert_array_equal(dense[0], ci)
    np.testing.assert_array_equal(dense[1].view(np.uint32), cd.view(np.uint32))
    np.testing.assert_array_equal(dense[2], cl)
    np.testing.assert_array_equal(dense[3], cs)


def test_build_is_identical_with_and_without_dense_top_layers():
    n, dim = 20000, 96
    rows = oracle.synth_rows(0, n, dim)[:, :dim]
    store = ph.VectorStore(rows)
    bp = ph.BuildParameters(seed=5)
    a = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), bp)
    with per_hop_path():
        b = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), bp)
    assert a.layer_count() == b.layer_count()
    for x, y in zip(a.layers, b.layers):
        np.testing.assert_array_equal(x.nodes, y.nodes)
        np.testing.assert_array_equal(x.neighbors, y.neighbors)


def test_layers_that_are_not_nested_fall_back():
    """a top layer holding a vector the layer below lacks: the table cannot represent it; the
    per-hop path runs and reports the reference's panic (lib.rs:261) as PHNSW_E_MISSING_NODE"""
    n, dim = 64, 16
    rows = oracle.synth_rows(0, n, dim)[:, :dim]
    store = ph.VectorStore(rows)
    E = ph.EMPTY
    top = (np.array([5, 9], dtype=np.uint64), np.array([[1, E], [0, E]], dtype=np.uint64))
    low_nodes = np.array([3, 9, 20, 40], dtype=np.uint64)  # 5 is missing
    low = (low_nodes, np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], dtype=np.uint64))
    h = ph.Hnsw.from_layers(store, [top, low])
    q = oracle.synth_rows(2 ** 32, 4, dim)[:, :dim]
    with pytest.raises(ph.PhnswError) as e1:
        h.search_batch(queries=q, sp=ph.SearchParameters(8, 8, 2))
    with per_hop_path():
        with pytest.raises(ph.PhnswError) as e2:
            h.search_batch(queries=q, sp=ph.SearchParameters(8, 8, 2))
    assert e1.value.code == e2.value.code


def test_latency_kernels_equal_throughput_kernels():
    """batches of <= 1024 queries run kernels with 12 rows in flight per wave (PHNSW_NO_LAT=1: the 4-row
    throughput kernels): same rows, same arithmetic, same results"""
    n, dim = 20000, 768
    rows = oracle.synth_rows(0, n, dim)[:, :dim]
    store = ph.VectorStore(rows)
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters(seed=9, max_link_rounds=1))
    q = oracle.synth_rows(2 ** 32, 700, dim)[:, :dim]
    for sp in (ph.SearchParameters(100, 100, 4), ph.SearchParameters(300, 300, 2)):
        lat = h.search_batch(queries=q, sp=sp, stats=True)
        os.environ["PHNSW_NO_LAT"] = "1"
        try:
            thr = h.search_batch(queries=q, sp=sp, stats=True)
        finally:
            del os.environ["PHNSW_NO_LAT"]
        same(lat, thr)


def test_mfma_chain_is_an_fma_chain(tmp_path):
    """the premise of the matrix-core table kernel (csrc/tiny.hip): K = 1 MFMA steps are single fused multiply-adds,
    bit-identical to v_fma_f32 on unit-range, denormal and wide-exponent operands (scripts/micro/mfma_fma_exact.hip)"""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "micro", "mfma_fma_exact.hip")
    exe = str(tmp_path / "mfma_fma_exact")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-w", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "MFMA chain == FMA chain on every trial" in out.stdout, out.stdout


def test_matrix_core_table_equals_vector_unit_table():
    """same launch, both table kernels (PHNSW_TINY_VALU=1 forces the vector-unit one): identical results, including
    partial 64 x 64 tiles on both sides (333 queries, table layers that are not multiples of 64)"""
    n, dim = 9000, 768
    rows = oracle.synth_rows(0, n, dim)[:, :dim]
    store = ph.VectorStore(rows)
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters(seed=11, max_link_rounds=1))
    q = oracle.synth_rows(2 ** 32, 333, dim)[:, :dim]
    sp = ph.SearchParameters(128, 128, 4)
    a = h.search_batch(queries=q, sp=sp, stats=True)
    os.environ["PHNSW_TINY_VALU"] = "1"
    try:
        b = h.search_batch(queries=q, sp=sp, stats=True)
    finally:
        del os.environ["PHNSW_TINY_VALU"]
    same(a, b)


class no_dense_split:
    """PHNSW_NO_DENSE_SPLIT=1: the dense top layers stay in the full kernel's first launch"""

    def __enter__(self):
        os.environ["PHNSW_NO_DENSE_SPLIT"] = "1"

    def __exit__(self, *a):
        del os.environ["PHNSW_NO_DENSE_SPLIT"]


class split_bytes:
    """PHNSW_SPLIT_BYTES: layers above this many row bytes get a launch of their own in a split descent"""

    def __init__(self, n):
        self.n = str(n)

    def __enter__(self):
        os.environ["PHNSW_SPLIT_BYTES"] = self.n

    def __exit__(self, *a):
        del os.environ["PHNSW_SPLIT_BYTES"]


@pytest.mark.parametrize("ef,dim", [(104, 768), (256, 96), (300, 256), (600, 64)])
def test_dense_layers_in_a_launch_of_their_own(ef, dim):
    """split descents (batches of >= 32768 queries) walk their dense top layers in ph_search_kernel_dense (queues of
    128 / 256 / 512 / 1024 slots) and continue in the full kernel from the parked candidates: same results as with
    the dense layers in the full kernel's first launch, as the per-hop path, as the oracle"""
    n, nq = 9000, 33000
    rows = oracle.synth_rows(0, n, dim)[:, :dim]
    store = ph.VectorStore(rows)
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters(seed=13, max_link_rounds=1))
    q = oracle.synth_rows(2 ** 32, nq, dim)[:, :dim]
    sp = ph.SearchParameters(ef, ef, 3)
    with split_bytes(100000):
        two = h.search_batch(queries=q, sp=sp, stats=True)
        disp = h.dispatches()
        assert len(disp) >= 3 and disp[1]["layers"][0] == 0 and disp[2]["layers"][0] == disp[1]["layers"][1], disp
        with no_dense_split():
            one = h.search_batch(queries=q, sp=sp, stats=True)
        same(two, one)
        with per_hop_path():
            same(two, h.search_batch(queries=q, sp=sp, stats=True))
    same(two, h.search_batch(queries=q, sp=sp, stats=True))  # and the unsplit descent
    ix = oracle.Index(rows, dim=dim, sum_mode=oracle.SUM_BLOCKED64)
    for l in h.layers:
        ix.push_layer(l.nodes, l.neighbors, l.neighborhood_size)
    ci, cd, cl, cs = ix.search(queries=q[:500], sp=(ef, ef, 3), stats=True)
    np.testing.assert_array_equal(two[0][:500], ci)
    np.testing.assert_array_equal(two[1][:500].view(np.uint32), cd.view(np.uint32))
    np.testing.assert_array_equal(two[3][:500], cs)


def test_not_nested_layers_with_a_dense_first_launch():
    """the table's usable flag is only known on the device: with layers that are not nested the dense launch does
    nothing and the follow-up launch walks every layer per hop -- same error as the per-hop path"""
    n, dim = 64, 16
    rows = oracle.synth_rows(0, n, dim)[:, :dim]
    store = ph.VectorStore(rows)
    E = ph.EMPTY
    top = (np.array([5, 9], dtype=np.uint64), np.array([[1, E], [0, E]], dtype=np.uint64))
    low_nodes = np.array([3, 9, 20, 40], dtype=np.uint64)  # 5 is missing
    low = (low_nodes, np.array([[1, 2, 3], [0, 2, 3], [0, 1, 3], [0, 1, 2]], dtype=np.uint64))
    h = ph.Hnsw.from_layers(store, [top, low])
    q = oracle.synth_rows(2 ** 32, 33000, dim)[:, :dim]
    with split_bytes(1):
        with pytest.raises(ph.PhnswError) as e1:
            h.search_batch(queries=q, sp=ph.SearchParameters(8, 8, 2))
        with per_hop_path():
            with pytest.raises(ph.PhnswError) as e2:
                h.search_batch(queries=q, sp=ph.SearchParameters(8, 8, 2))
        assert e1.value.code == e2.value.code
        # the nested graph of the same shapes: dense launch + follow-up == one launch
        low2 = (np.array([5, 9, 20, 40], dtype=np.uint64), low[1])
        h2 = ph.Hnsw.from_layers(store, [top, low2])
        r2 = h2.search_batch(queries=q, sp=ph.SearchParameters(8, 8, 2), stats=True)
        with no_dense_split():
            same(r2, h2.search_batch(queries=q, sp=ph.SearchParameters(8, 8, 2), stats=True))


def test_chunked_descents_equal_unchunked(monkeypatch):
    """a query list longer than the dense table holds runs in consecutive chunks of the list (api.hip): with the table
    budget cut to a few hundred rows (PHNSW_TINY_TABLE_BYTES) the results -- ids, distance bits, lengths, counters --
    equal the one-piece run, for plain lists, for lists with a processing order (build rounds) and through the host
    path; the smallest budget still runs (tables of 64 rows)"""
    import ctypes as C
    n, dim = 30000, 128
    store = ph.VectorStore.synthetic(n, dim, seed=42)
    h = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters(seed=3, max_link_rounds=1))
    q = ph.VectorStore.synthetic(2000, dim, seed=42, first=2 ** 32).read()
    sp = ph.SearchParameters(64, 64, 2)
    whole = h.search_batch(queries=q, sp=sp, stats=True)
    qid = np.arange(0, n, 11, dtype=np.uint64)
    swhole = h.search_batch(qids=qid, sp=sp, exclude=qid, stats=True)
    L = ph.lib()
    L.phnsw_debug_last_search_chunks.restype = C.c_uint32
    L.phnsw_debug_last_search_chunks.argtypes = [C.c_void_p]
    assert L.phnsw_debug_last_search_chunks(h._h) == 1
    stride = (h._layer(2).node_count() + 63) // 64 * 64   # the table layer of this shape
    monkeypatch.setenv("PHNSW_TINY_TABLE_BYTES", str(300 * stride * 4))
    same(h.search_batch(queries=q, sp=sp, stats=True), whole)
    assert L.phnsw_debug_last_search_chunks(h._h) >= 5
    same(h.search_batch(qids=qid, sp=sp, exclude=qid, stats=True), swhole)
    # a build (its rounds carry a processing order and run Stored queries through the same chunk loop) is unchanged
    g = ph.Hnsw.generate(store, np.arange(n, dtype=np.uint64), ph.BuildParameters(seed=3, max_link_rounds=1))
    for l in range(h.layer_count()):
        np.testing.assert_array_equal(g._layer(l).neighbors, h._layer(l).neighbors)
    monkeypatch.setenv("PHNSW_TINY_TABLE_BYTES", "1")
    same(h.search_batch(queries=q[:500], sp=sp, stats=True), [x[:500] for x in whole])
