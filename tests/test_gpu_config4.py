"""BASELINE configs[3] / SURVEY config 4 at FULL size: 10M x 768 f32, index construction sharded over 8 ranks.

A test box has one GPU, so the world of 8 is emulated: `phnsw_build_sharded` (the C-ABI driver, csrc/sharded.hip)
plays every rank in turn -- each rank's node range of every round runs as its own launches through the phase entry
points, the per-node results are laid out in the blocks an RCCL all-gather would deliver and reassembled by the
driver's own copies -- which is every line of the sharded build except the transport (covered by the RCCL self-test,
the two-rank gloo rehearsal and, on a multi-GPU node, test_sharded_build_over_rccl).  The oracle cannot build at this
size, so what is checked is what must hold for any correct build (tests/test_gpu_fullsize.py's properties): the
reference's layer invariants (search.rs:142-171), its self-recall assertion (lib.rs:2218-2224: >= 0.9 after
generate), distances that are exactly compare_vec's, determinism -- on SURVEY 8d's sigma = 0.1 clustered set and on
round 1's tight clusters."""
import time

import numpy as np
import pytest

import parallel_hnsw_amd as ph

pytestmark = pytest.mark.gpu

N, DIM, WORLD = 10_000_000, 768, 8


@pytest.fixture(scope="module", params=["survey_sigma_0.1", "tight"])
def built(request):
    noise = 0.1 * DIM ** 0.5 if request.param.startswith("survey") else 1.0
    store = ph.VectorStore.clustered(N, DIM, seed=42, n_clusters=1000, noise=noise)
    t0 = time.time()
    h, st = ph.build_sharded(store, np.arange(N, dtype=np.uint64), ph.BuildParameters(), ph.EmulatedComm(WORLD, 0))
    secs = time.time() - t0
    print("\n[config 4 / %s] 10M x 768 through phnsw_build_sharded, emulated world of %d: %.1f s in all; rank 0: sharded "
          "%.2f s + replicated %.2f s + reassembly %.3f s; %.2f GB all-gathered per rank in %d collectives"
          % (request.param, WORLD, secs, st["seconds_sharded"], st["seconds_replicated"], st["seconds_comm"],
             st["all_gather_bytes"] / 1e9, st["all_gather_calls"]))
    yield store, h, st
    del h, store


def test_config4_every_long_phase_was_split(built):
    store, h, st = built
    assert st["phases"] > st["phases_whole"] > 0          # the small top layers run whole, the rest is split
    assert st["all_gather_calls"] >= st["phases"] - st["phases_whole"] - st["all_reduce_calls"]
    # a link round of the bottom layer alone gathers n x M x 8 bytes + lengths
    assert st["all_gather_bytes"] > N * 24 * 8
    assert st["seconds_others"] > 3 * st["seconds_sharded"]  # seven of the eight shares were the other ranks'


def test_config4_layer_invariants(built):
    store, h, st = built
    count = h.layer_count()
    assert count >= 7                                                   # [3, 40, 482, 5 787, 69 444, 833 333, 10M] + promotions
    prev = None
    for lft in range(count - 1):                                        # upper layers: sorted, nested
        lay = h._layer(lft)
        assert (np.diff(lay.nodes.astype(np.int64)) > 0).all()
        if prev is not None:
            assert np.isin(prev, lay.nodes).all()
        nb = lay.neighbors
        live = nb != ph.EMPTY
        assert ((~live[:, :-1]) <= (~live[:, 1:])).all() and (nb[live] < lay.node_count()).all()
        prev = lay.nodes
    bottom = h._layer(count - 1)
    assert bottom.node_count() == N and bottom.neighbors.shape == (N, 48)
    assert (bottom.nodes[::1009] == np.arange(N, dtype=np.uint64)[::1009]).all() and np.isin(prev, bottom.nodes).all()
    nb = bottom.neighbors
    live = nb != ph.EMPTY
    assert ((~live[:, :-1]) <= (~live[:, 1:])).all()                    # sentinels trailing only
    assert (nb[live] < N).all()
    assert not (nb == np.arange(N, dtype=np.uint64)[:, None]).any()    # no self loops
    assert live.sum(1).min() >= 1                                       # nobody is isolated
    for row in nb[::50_000]:
        r = row[row != ph.EMPTY]
        assert len(set(r.tolist())) == len(r)                           # rows are duplicate free


def test_config4_self_recall_and_exact_distances(built):
    """test_recall (lib.rs:2217-2231): stored vectors find themselves first; every reported distance is exactly
    compare_vec(query, Stored(id)); searching twice gives the same bits"""
    store, h, st = built
    qids = np.arange(0, N, 1999, dtype=np.uint64)
    sp = ph.SearchParameters(300, 300, 2)
    ids, d, ln, stats = h.search_batch(qids=qids, sp=sp, stats=True)
    ids2, d2, ln2, stats2 = h.search_batch(qids=qids, sp=sp, stats=True)
    np.testing.assert_array_equal(ids, ids2)
    np.testing.assert_array_equal(d.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_array_equal(stats, stats2)
    recall = float(np.mean(ids[:, 0] == qids))
    assert recall >= 0.9, recall
    assert np.abs(d[ids[:, 0] == qids, 0]).max() < 1e-5
    assert (ln == 300).all() and (np.diff(d, axis=1) >= 0).all()
    for i in range(0, len(qids), 500):
        assert len(set(ids[i].tolist())) == 300
        np.testing.assert_array_equal(store.compare_vec(ph.Stored(int(qids[i])), ids[i]).view(np.uint32),
                                      d[i].view(np.uint32))
    assert h.stochastic_recall() >= 0.9                                 # the reference's own estimator, lib.rs:1463-1499
