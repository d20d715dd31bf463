"""Randomised GPU-vs-oracle parity over shapes and parameters, with duplicate vectors to force
exact distance ties (the (d, id) order, merge()'s tail-tie rule, priority_queue.rs:109-144) and
tiny queues / large probe depths to force pops from the frontier spill list."""
import numpy as np
import pytest

import oracle
import parallel_hnsw_amd as ph

pytestmark = pytest.mark.gpu


def random_case(rng):
    n = int(rng.integers(40, 2500))
    dim = int(rng.choice([1, 2, 3, 5, 8, 17, 32, 100, 256, 260, 768]))
    metric = int(rng.integers(0, 3))
    order = int(rng.choice([2, 3, 6, 12, 40]))
    M = int(rng.integers(1, 33))
    M0 = int(rng.integers(M, 65))
    dup = int(rng.choice([1, 1, 1, 2, 7, 30]))
    ef = int(rng.choice([1, 2, 3, 6, 17, 64, 65, 128, 129, 300, 512, 513, 1024]))
    upper = int(rng.choice([1, 2, ef, ef, max(1, ef // 2), ef + 5]))
    pd = int(rng.choice([1, 2, 2, 3, 9, 40]))
    link_ef = int(rng.choice([M, 2 * M + 1, 40, 300]))
    return dict(n=n, dim=dim, metric=metric, order=order, M=M, M0=M0, dup=dup, ef=ef, upper=upper, pd=pd,
                link_ef=max(link_ef, M), seed=int(rng.integers(0, 1 << 30)))


@pytest.mark.parametrize("case_seed", range(24))
def test_random_build_and_search_parity(case_seed):
    rng = np.random.default_rng(1000 + case_seed)
    c = random_case(rng)
    base_n = max(2, c["n"] // c["dup"])
    rows = oracle.synth_rows(0, base_n, c["dim"], seed=c["seed"], normalize=(c["metric"] != 2))
    rows = np.repeat(rows, c["dup"], axis=0)[:c["n"]].copy()
    n = rows.shape[0]
    kw = dict(order=c["order"], neighborhood_size=c["M"], zero_layer_neighborhood_size=c["M0"], seed=c["seed"],
              promote=int(rng.integers(0, 2)), max_link_rounds=2)
    obp, gbp = oracle.default_build_params(**kw), ph.BuildParameters(**kw)
    for bp in (obp, gbp):
        s = bp.optimization.search
        s.number_of_candidates, s.upper_layer_candidate_count, s.probe_depth = c["link_ef"], c["link_ef"], 2
    oix = oracle.Index.generate(rows, np.arange(n), obp, dim=c["dim"], metric=c["metric"],
                                sum_mode=oracle.SUM_BLOCKED64, threads=4)
    store = ph.VectorStore(rows[:, :c["dim"]], metric=c["metric"])
    gix = ph.Hnsw.generate(store, np.arange(n), gbp)
    assert gix.layer_count() == oix.layer_count, c
    for l in range(oix.layer_count):
        onodes, onb = oix.layer(l)
        gl = gix._layer(l)
        np.testing.assert_array_equal(gl.nodes, onodes, err_msg=str(c))
        np.testing.assert_array_equal(gl.neighbors, onb, err_msg="layer %d %s" % (l, c))
    assert oix.check_layer_invariants() == 0
    q = np.concatenate([oracle.synth_rows(2 ** 32, 40, c["dim"], seed=c["seed"], normalize=(c["metric"] != 2)),
                        rows[:24]])[:, :c["dim"]]  # fresh queries and stored vectors (exact zero distances)
    sp = (c["ef"], c["upper"], c["pd"])
    gi, gd, gl_, gs = gix.search_batch(queries=q, sp=ph.SearchParameters(*sp), stats=True)
    ci, cd, cl, cs = oix.search(queries=q, sp=sp, stats=True, threads=4)
    np.testing.assert_array_equal(gl_, cl, err_msg=str(c))
    np.testing.assert_array_equal(gi, ci, err_msg=str(c))
    np.testing.assert_array_equal(gd.view(np.uint32), cd.view(np.uint32), err_msg=str(c))
    np.testing.assert_array_equal(gs, cs, err_msg=str(c))
    qids = rng.integers(0, n, size=32).astype(np.uint64)
    g2 = gix.search_batch(qids=qids, sp=ph.SearchParameters(*sp), exclude=qids)
    c2 = oix.search(qids=qids, sp=sp, exclude=qids, threads=4)
    np.testing.assert_array_equal(g2[0], c2[0], err_msg=str(c))
    np.testing.assert_array_equal(g2[2], c2[2], err_msg=str(c))
