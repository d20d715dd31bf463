// C++ rendition of the reference's own unit tests (src/lib.rs:1994-2068, 2154-2164, 2270-2284,
// 2358-2420) against include/phnsw.hpp: same fixture data, same calls, same assertions.
// Built with g++ and linked to libphnsw.so by tests/test_gpu_cpp.py; needs a GPU.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "phnsw.hpp"

using namespace phnsw;

static int failures = 0;
#define EXPECT(cond)                                                   \
  do {                                                                 \
    if (!(cond)) {                                                     \
      printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond);           \
      failures++;                                                      \
    }                                                                  \
  } while (0)

static bool close_to(float a, float b) { return std::fabs(a - b) <= 1e-5f * std::fabs(b) + 1e-7f; }

// make_simple_hnsw's vectors  lib.rs:1996-2006
static std::vector<float> simple_data() {
  const float s = 0.70710678118f;  // FRAC_1_SQRT_2
  return {1, 0, 0, 0, 1, 0, 0, 0, 1, s, s, 0, 0.5773f, 0.5773f, 0.5773f, -1, 0, 0, 0, -1, 0, 0, 0, -1, 0, s, s};
}
static BuildParameters simple_bp() {
  BuildParameters bp;  // lib.rs:2009-2012
  bp.order = 6;
  bp.neighborhood_size = 3;
  bp.zero_layer_neighborhood_size = 6;
  return bp;
}
// the reference's own test graph (test_generation's literal, lib.rs:2090-2151) under a 1-node top layer
static Hnsw fixture_hnsw(const Comparator &c, VectorId entry) {
  const uint64_t nb[54] = {3, 4, 1, 2, 6, 7, 3, 8, 4, 0, 2, 5, 8, 4, 0, 1, 3, 5, 4, 0, 1, 8, 2, 7, 3, 8, 0,
                           1, 2, 5, 1, 2, 6, 8, 4, 3, 0, 2, 5, 7, 4, 3, 0, 1, 3, 6, 4, 8, 4, 1, 2, 3, 0, 5};
  Layer top, bottom;
  top.neighborhood_size = 3;
  top.nodes = {entry};
  top.neighbors = {EMPTY, EMPTY, EMPTY};
  bottom.neighborhood_size = 6;
  for (uint64_t i = 0; i < 9; i++) bottom.nodes.push_back(i);
  bottom.neighbors.assign(nb, nb + 54);
  return Hnsw::from_layers(c, {top, bottom});
}

static void test_nearness_search(const Comparator &c) {  // lib.rs:2046-2068
  Hnsw hnsw = fixture_hnsw(c, 0);
  const float s = 0.70710678118f;
  std::vector<float> q = {0.0f, s, s};
  auto results = hnsw.search(AbstractVector::Unstored(q), hnsw.build_parameters.optimization.search);
  const std::pair<VectorId, float> expect[9] = {{8, 5.9604645e-8f}, {4, 0.1835745f}, {1, 0.29289323f}, {2, 0.29289323f},
                                                {3, 0.5f},          {0, 1.0f},       {5, 1.0f},        {6, 1.7071068f},
                                                {7, 1.7071068f}};
  EXPECT(results.size() == 9);
  for (size_t i = 0; i < results.size() && i < 9; i++) {
    EXPECT(results[i].first == expect[i].first);
    EXPECT(close_to(results[i].second, expect[i].second));
  }
}

static void test_knn(const Comparator &c) {  // lib.rs:2358-2377
  Hnsw hnsw = fixture_hnsw(c, 4);
  auto results = hnsw.knn(1, 1);
  const std::pair<VectorId, float> expect[9] = {{3, 0.29289323f}, {3, 0.29289323f}, {8, 0.29289323f}, {4, 0.1835745f},
                                                {3, 0.1835745f},  {1, 1.0f},        {0, 1.0f},        {0, 1.0f},
                                                {4, 0.1835745f}};
  EXPECT(results.size() == 9);
  for (size_t i = 0; i < 9 && i < results.size(); i++) {
    EXPECT(results[i].first == i);
    EXPECT(results[i].second.size() == 1);
    if (results[i].second.size() == 1) {
      EXPECT(results[i].second[0].first == expect[i].first);
      EXPECT(close_to(results[i].second[0].second, expect[i].second));
    }
  }
}

static void test_threshold_nn(const Comparator &c) {  // lib.rs:2379-2420
  Hnsw hnsw = fixture_hnsw(c, 2);
  auto results = hnsw.threshold_nn(0.3f, 1, hnsw.build_parameters.zero_layer_neighborhood_size > 6 ? 6 : 6);
  const std::vector<std::vector<VectorId>> expect = {{3}, {3, 8}, {8}, {4, 0, 1}, {3, 8}, {}, {}, {}, {4, 1, 2}};
  EXPECT(results.size() == 9);
  for (size_t i = 0; i < 9 && i < results.size(); i++) {
    EXPECT(results[i].second.size() == expect[i].size());
    for (size_t j = 0; j < expect[i].size() && j < results[i].second.size(); j++)
      EXPECT(results[i].second[j].first == expect[i][j]);
  }
}

static void test_generate_and_improve(const Comparator &c) {  // lib.rs:1994-2015, 2270-2284
  std::vector<VectorId> vs;
  for (uint64_t i = 0; i < 9; i++) vs.push_back(i);
  Hnsw hnsw = Hnsw::generate(c, vs, simple_bp());
  EXPECT(hnsw.layer_count() == 2);
  Layer bottom = hnsw.get_layer(0);
  EXPECT(bottom.nodes == vs);  // test_generation's first assertion  lib.rs:2073-2089
  hnsw.improve_index(hnsw.build_parameters);
  auto data = simple_data();
  for (uint64_t i = 0; i < 9; i++) {
    auto results = hnsw.search(AbstractVector::Unstored(data.data() + 3 * i), hnsw.build_parameters.optimization.search);
    EXPECT(!results.empty() && results[0].first == i);  // test_small_index_improvement  lib.rs:2280-2283
  }
  EXPECT(hnsw.entry_vector() == hnsw.get_layer_from_top(0).nodes[0]);
  {  // search_instrumented (lib.rs:667-673) returns search's results; top-k == the leading k of them
    auto sp = hnsw.build_parameters.optimization.search;
    auto plain = hnsw.search(AbstractVector::Unstored(data.data()), sp);
    auto inst = hnsw.search_instrumented(AbstractVector::Unstored(data.data()), sp);
    EXPECT(inst.first == plain);
    auto stored = hnsw.search_instrumented(AbstractVector::Stored(3), sp);
    EXPECT(!stored.first.empty() && stored.first[0].first == 3);
    auto top = hnsw.search_many_topk({data.data(), data.data() + 3}, sp, 2);
    EXPECT(top.size() == 2 && top[0].size() == 2 && top[0][0] == plain[0] && top[0][1] == plain[1]);
    // the sharded build through the C ABI with one process playing two ranks: the graph phnsw_build gives
    phnsw_comm comm = {};
    comm.rank = 0;
    comm.world = 2;
    comm.emulate = 1;
    phnsw_sharded_tuning(1, 1, 1);
    phnsw_sharded_stats st;
    Hnsw sh = Hnsw::generate_sharded(c, vs, simple_bp(), comm, &st);
    phnsw_sharded_tuning(4096, 4, 65536);
    Hnsw ref = Hnsw::generate(c, vs, simple_bp());
    EXPECT(sh.layer_count() == ref.layer_count() && st.all_gather_calls > 0);
    EXPECT(sh.get_layer(0).neighbors == ref.get_layer(0).neighbors);
  }
  // panics of the reference surface as exceptions
  bool threw = false;
  try {
    Hnsw::generate(c, {}, simple_bp());  // assert!(total_size > 0)  lib.rs:837
  } catch (const Error &) {
    threw = true;
  }
  EXPECT(threw);
}

int main() {
  try {
    auto data = simple_data();
    Comparator c(data.data(), 9, 3, OneMinusDot);  // SillyComparator  lib.rs:1971-1992
    test_nearness_search(c);
    test_knn(c);
    test_threshold_nn(c);
    test_generate_and_improve(c);
    auto d = c.compare_vec(AbstractVector::Stored(3), {4, 0});
    EXPECT(close_to(d[0], 0.1835745f) && close_to(d[1], 0.29289323f));
  } catch (const Error &e) {
    printf("phnsw::Error %d: %s\n", e.code, e.what());
    return 2;
  }
  printf(failures ? "%d FAILURES\n" : "ALL OK%.0d\n", failures);
  return failures ? 1 : 0;
}
