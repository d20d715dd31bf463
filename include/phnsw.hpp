// phnsw.hpp -- C++ mirror of the reference crate's public surface over the C ABI (phnsw.h).
//
// Names, argument meaning and result ordering follow terminusdb-labs/parallel-hnsw so that
// code (and tests) written against the crate read the same:
//
//   Hnsw::generate(c, vs, bp)                 src/lib.rs:825-830
//   hnsw.search(AbstractVector, sp)           src/lib.rs:663-665  -> Vec<(VectorId, f32)>, (d, id) order
//   hnsw.search_upto(v, sp, upto)             src/lib.rs:654-661
//   hnsw.improve_index(bp) / improve_neighbors / stochastic_recall   src/lib.rs:1501-1513, 1664-1686
//   hnsw.knn(k, probe_depth) / threshold_nn(threshold, probe_depth, initial_search_depth)   :905-962
//   hnsw.layer_count() / entry_vector() / get_layer(i)                src/lib.rs:591-650
//   SearchParameters / BuildParameters with the defaults of src/parameters.rs
//   AbstractVector::Stored(id) / Unstored(&v)                         src/types.rs:40-43
//   QuantizedHnsw::new(centroids, comparator, bp) / search            src/pq.rs:287-364
//
// Where the crate panics (unwrap / assert), these wrappers throw phnsw::Error.
#pragma once
#include <algorithm>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "phnsw.h"

namespace phnsw {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  if (rc != 0) throw Error(rc, phnsw_last_error());
}

using VectorId = uint64_t;  // types.rs:3-4
using NodeId = uint64_t;    // types.rs:5-6
constexpr uint64_t EMPTY = PHNSW_EMPTY;

struct SearchParameters : phnsw_search_params {  // parameters.rs:3-18
  SearchParameters() { phnsw_default_search_params(this); }
  SearchParameters(uint64_t candidates, uint64_t upper, uint64_t probe) {
    number_of_candidates = candidates;
    upper_layer_candidate_count = upper;
    probe_depth = probe;
  }
  SearchParameters(const phnsw_search_params &p) : phnsw_search_params(p) {}  // e.g. bp.optimization.search
};
struct BuildParameters : phnsw_build_params {  // parameters.rs:42-64
  BuildParameters() { phnsw_default_build_params(this); }
};

// AbstractVector<'a, T>  types.rs:40-43
struct AbstractVector {
  bool stored;
  VectorId id;
  const float *vec;
  static AbstractVector Stored(VectorId v) { return {true, v, nullptr}; }
  static AbstractVector Unstored(const float *v) { return {false, 0, v}; }
  static AbstractVector Unstored(const std::vector<float> &v) { return {false, 0, v.data()}; }
};

enum Metric { CosineHalf = PHNSW_METRIC_COSINE_HALF, OneMinusDot = PHNSW_METRIC_ONE_MINUS_DOT, L2 = PHNSW_METRIC_L2 };

// the Comparator of the GPU path: vectors + metric (BigComparator bigvec.rs:38-57)
class Comparator {
 public:
  Comparator(const float *rows, uint64_t n, uint32_t dim, Metric metric = CosineHalf, int device = 0) : dim_(dim), n_(n) {
    check(phnsw_store_create(rows, n, dim, metric, device, &s_));
  }
  explicit Comparator(phnsw_store *adopt) : s_(adopt) {
    check(phnsw_store_info(s_, &n_, &dim_, nullptr, nullptr, nullptr));
  }
  Comparator(const Comparator &) = delete;
  Comparator &operator=(const Comparator &) = delete;
  ~Comparator() { phnsw_store_destroy(s_); }
  // compare_vec for a list of stored vectors in one launch  lib.rs:69-73
  std::vector<float> compare_vec(const AbstractVector &v, const std::vector<VectorId> &ids) const {
    std::vector<float> out(ids.size());
    check(phnsw_distance_batch(s_, v.stored ? nullptr : v.vec, v.id, ids.data(), ids.size(), out.data()));
    return out;
  }
  // more vectors behind the same comparator; returns the first new VectorId
  VectorId append(const float *rows, uint64_t count) {
    uint64_t first = 0;
    check(phnsw_store_append(s_, rows, count, &first));
    n_ += count;
    return first;
  }
  // exact k nearest (ground truth for recall@k)
  std::vector<std::vector<std::pair<VectorId, float>>> bruteforce(const std::vector<float> &queries, uint32_t k) const {
    uint64_t nq = queries.size() / dim_;
    std::vector<uint64_t> ids(nq * k);
    std::vector<float> d(nq * k);
    check(phnsw_bruteforce_topk(s_, queries.data(), nq, k, ids.data(), d.data()));
    std::vector<std::vector<std::pair<VectorId, float>>> r(nq);
    for (uint64_t q = 0; q < nq; q++)
      for (uint32_t j = 0; j < k; j++)
        if (ids[q * k + j] != EMPTY) r[q].push_back({ids[q * k + j], d[q * k + j]});
    return r;
  }
  uint32_t dim() const { return dim_; }
  uint64_t len() const { return n_; }
  phnsw_store *handle() const { return s_; }

 private:
  phnsw_store *s_ = nullptr;
  uint32_t dim_ = 0;
  uint64_t n_ = 0;
};

// Layer { neighborhood_size, nodes, neighbors }  lib.rs:85-91 (host copy)
struct Layer {
  uint64_t neighborhood_size = 0;
  std::vector<VectorId> nodes;
  std::vector<NodeId> neighbors;
  uint64_t node_count() const { return nodes.size(); }
  VectorId get_vector(NodeId n) const { return nodes[n]; }
  std::vector<NodeId> get_neighbors(NodeId n) const {  // trailing sentinels trimmed  lib.rs:114-148
    std::vector<NodeId> r;
    for (uint64_t k = 0; k < neighborhood_size; k++) {
      NodeId x = neighbors[n * neighborhood_size + k];
      if (x == EMPTY) break;
      r.push_back(x);
    }
    return r;
  }
};

using SearchResult = std::vector<std::pair<VectorId, float>>;

class Hnsw {
 public:
  BuildParameters build_parameters;

  // Hnsw::generate(c, vs, bp, progress)  lib.rs:825-893
  static Hnsw generate(const Comparator &c, const std::vector<VectorId> &vs, const BuildParameters &bp = BuildParameters()) {
    phnsw_index *ix = nullptr;
    check(phnsw_build(c.handle(), vs.data(), vs.size(), &bp, nullptr, nullptr, &ix));
    return Hnsw(ix, &c, bp);
  }
  // Hnsw::generate with every per-node phase split over the ranks of `comm` (BASELINE config 4: one process per
  // GPU; phnsw_comm_rccl_create for the built-in RCCL transport, or two callbacks of the host's own)
  static Hnsw generate_sharded(const Comparator &c, const std::vector<VectorId> &vs, const BuildParameters &bp,
                               const phnsw_comm &comm, phnsw_sharded_stats *stats = nullptr) {
    phnsw_index *ix = nullptr;
    check(phnsw_build_sharded(c.handle(), vs.data(), vs.size(), &bp, &comm, nullptr, nullptr, &ix, stats));
    return Hnsw(ix, &c, bp);
  }
  // adopt layers built elsewhere (top first), e.g. deserialised by the crate
  static Hnsw from_layers(const Comparator &c, const std::vector<Layer> &layers) {
    std::vector<uint64_t> counts, widths;
    std::vector<const uint64_t *> pn, pb;
    for (auto &l : layers) {
      counts.push_back(l.nodes.size());
      widths.push_back(l.neighborhood_size);
      pn.push_back(l.nodes.data());
      pb.push_back(l.neighbors.data());
    }
    phnsw_index *ix = nullptr;
    check(phnsw_index_from_layers(c.handle(), (uint32_t)layers.size(), counts.data(), widths.data(), pn.data(), pb.data(), &ix));
    return Hnsw(ix, &c, BuildParameters());
  }
  static Hnsw deserialize(const std::string &path, const Comparator &c) {  // lib.rs:1693-1698
    phnsw_index *ix = nullptr;
    check(phnsw_index_deserialize(c.handle(), path.c_str(), &ix));
    BuildParameters bp;
    check(phnsw_index_build_params(ix, &bp));
    return Hnsw(ix, &c, bp);
  }
  void serialize(const std::string &path) const { check(phnsw_index_serialize(ix_, path.c_str())); }

  Hnsw(Hnsw &&o) noexcept : build_parameters(o.build_parameters), ix_(o.ix_), c_(o.c_) { o.ix_ = nullptr; }
  Hnsw(const Hnsw &) = delete;
  ~Hnsw() {
    if (ix_) phnsw_index_destroy(ix_);
  }

  // Hnsw::search(v, sp)  lib.rs:663-665
  SearchResult search(const AbstractVector &v, const SearchParameters &sp = SearchParameters()) const {
    return search_upto(v, sp, 0);
  }
  // Hnsw::search_upto(v, sp, upto_layer_from_top)  lib.rs:654-661 (0 = all layers)
  SearchResult search_upto(const AbstractVector &v, const SearchParameters &sp, uint32_t upto) const {
    return search_many({v}, sp, upto)[0];
  }
  // the batched form every GPU caller should use
  std::vector<SearchResult> search_many(const std::vector<AbstractVector> &vs, const SearchParameters &sp,
                                        uint32_t upto = 0) const {
    const uint64_t nq = vs.size(), ef = sp.number_of_candidates;
    std::vector<uint64_t> ids(nq * ef), len(nq);
    std::vector<float> d(nq * ef);
    bool stored = !vs.empty() && vs[0].stored;
    if (stored) {
      std::vector<uint64_t> q(nq);
      for (uint64_t i = 0; i < nq; i++) q[i] = vs[i].id;
      check(phnsw_search_batch_stored(ix_, q.data(), nq, &sp, upto, nullptr, ids.data(), d.data(), len.data(), nullptr));
    } else {
      std::vector<float> q(nq * c_->dim());
      for (uint64_t i = 0; i < nq; i++) std::copy(vs[i].vec, vs[i].vec + c_->dim(), q.begin() + i * c_->dim());
      check(phnsw_search_batch(ix_, q.data(), nq, &sp, upto, nullptr, ids.data(), d.data(), len.data(), nullptr));
    }
    std::vector<SearchResult> out(nq);
    for (uint64_t i = 0; i < nq; i++)
      for (uint64_t j = 0; j < len[i]; j++) out[i].push_back({ids[i * ef + j], d[i * ef + j]});
    return out;
  }
  // raw queries, the best k results of each: the truncation of lib.rs:1118 done before the transfer
  std::vector<SearchResult> search_many_topk(const std::vector<const float *> &queries, const SearchParameters &sp,
                                             uint64_t k) const {
    const uint64_t nq = queries.size(), dim = c_->dim();
    std::vector<float> q(nq * dim);
    for (uint64_t i = 0; i < nq; i++) std::copy(queries[i], queries[i] + dim, q.begin() + i * dim);
    std::vector<uint64_t> ids(nq * k), len(nq);
    std::vector<float> d(nq * k);
    check(phnsw_search_batch_topk(ix_, q.data(), nullptr, nq, &sp, 0, nullptr, k, ids.data(), d.data(), len.data()));
    std::vector<SearchResult> out(nq);
    for (uint64_t i = 0; i < nq; i++)
      for (uint64_t j = 0; j < len[i]; j++) out[i].push_back({ids[i * k + j], d[i * k + j]});
    return out;
  }
  // Hnsw::search_instrumented(v, sp) -> (results, index_distance)  lib.rs:667-673
  std::pair<SearchResult, uint64_t> search_instrumented(const AbstractVector &v, const SearchParameters &sp) const {
    const uint64_t ef = sp.number_of_candidates;
    std::vector<uint64_t> ids(ef);
    std::vector<float> d(ef);
    uint64_t len = 0, index = 0;
    check(phnsw_search_instrumented(ix_, v.stored ? nullptr : v.vec, v.stored ? &v.id : nullptr, 1, &sp, ids.data(), d.data(),
                                    &len, &index));
    SearchResult r;
    for (uint64_t j = 0; j < len; j++) r.push_back({ids[j], d[j]});
    return {r, index};
  }
  // improve_index(bp, last_recall: Option<f32>, progress)  lib.rs:1664-1686; NaN = None
  float improve_index(const BuildParameters &bp, float last_recall = __builtin_nanf("")) {
    float r = 0;
    check(phnsw_improve_index(ix_, &bp, last_recall, nullptr, nullptr, &r));
    return r;
  }
  float improve_neighbors(const BuildParameters &bp) {  // lib.rs:1507-1513
    float r = 0;
    check(phnsw_improve_neighbors_upto(ix_, layer_count(), &bp, __builtin_nanf(""), &r));
    return r;
  }
  float stochastic_recall(const phnsw_optimization_params &op) {  // lib.rs:1501-1505
    float r = 0;
    check(phnsw_stochastic_recall_at(ix_, layer_count() - 1, &op, &r));
    return r;
  }
  bool promote_at_layer(uint32_t layer_from_top, const BuildParameters &bp) {  // lib.rs:1273-1427
    int p = 0;
    check(phnsw_promote_at_layer(ix_, layer_from_top, &bp, &p));
    return p != 0;
  }
  void extend_layer(uint32_t layer_from_top, const std::vector<VectorId> &vecs) {  // lib.rs:1039-1068
    check(phnsw_extend_layer(ix_, layer_from_top, vecs.data(), vecs.size()));
  }
  // distance evaluations / hops of every search launched on this index (SearchStats summed)
  std::pair<uint64_t, uint64_t> counters() const {
    uint64_t a = 0, b = 0;
    check(phnsw_index_counters(ix_, &a, &b));
    return {a, b};
  }
  // Hnsw::knn(k, probe_depth)  lib.rs:905-928
  std::vector<std::pair<VectorId, SearchResult>> knn(uint64_t k, uint64_t probe_depth) const {
    Layer bottom = get_layer(0);
    uint64_t n = bottom.node_count();
    std::vector<uint64_t> ids(n * k), len(n);
    std::vector<float> d(n * k);
    check(phnsw_knn(ix_, k, probe_depth, ids.data(), d.data(), len.data()));
    return pack(bottom, ids, d, len, k);
  }
  // Hnsw::threshold_nn(threshold, probe_depth, initial_search_depth)  lib.rs:930-962
  std::vector<std::pair<VectorId, SearchResult>> threshold_nn(float threshold, uint64_t probe_depth,
                                                              uint64_t initial_search_depth, uint64_t max_out = 64) const {
    Layer bottom = get_layer(0);
    uint64_t n = bottom.node_count();
    std::vector<uint64_t> ids(n * max_out), len(n);
    std::vector<float> d(n * max_out);
    check(phnsw_threshold_nn(ix_, threshold, probe_depth, initial_search_depth, max_out, ids.data(), d.data(), len.data()));
    return pack(bottom, ids, d, len, max_out);
  }
  uint32_t layer_count() const { return phnsw_index_layer_count(ix_); }  // lib.rs:643-645
  // get_layer(i): counted from the BOTTOM  lib.rs:604-606
  Layer get_layer(uint32_t i) const { return get_layer_from_top(layer_count() - i - 1); }
  Layer get_layer_from_top(uint32_t i) const {  // lib.rs:617-624
    Layer l;
    uint64_t n = 0;
    check(phnsw_index_layer_info(ix_, i, &n, &l.neighborhood_size));
    l.nodes.resize(n);
    l.neighbors.resize(n * l.neighborhood_size);
    check(phnsw_index_layer_read(ix_, i, l.nodes.data(), l.neighbors.data()));
    return l;
  }
  VectorId entry_vector() const { return get_layer_from_top(0).nodes[0]; }  // lib.rs:638-641
  uint64_t vector_count() const {                                           // lib.rs:592-594
    uint64_t n = 0;
    check(phnsw_index_layer_info(ix_, layer_count() - 1, &n, nullptr));
    return n;
  }
  phnsw_index *handle() const { return ix_; }

 private:
  Hnsw(phnsw_index *ix, const Comparator *c, const BuildParameters &bp) : build_parameters(bp), ix_(ix), c_(c) {}
  static std::vector<std::pair<VectorId, SearchResult>> pack(const Layer &bottom, const std::vector<uint64_t> &ids,
                                                             const std::vector<float> &d, const std::vector<uint64_t> &len,
                                                             uint64_t stride) {
    std::vector<std::pair<VectorId, SearchResult>> out;
    for (uint64_t i = 0; i < bottom.node_count(); i++) {
      SearchResult r;
      for (uint64_t j = 0; j < len[i]; j++) r.push_back({ids[i * stride + j], d[i * stride + j]});
      out.push_back({bottom.nodes[i], r});
    }
    return out;
  }
  phnsw_index *ix_ = nullptr;
  const Comparator *c_ = nullptr;
};

// QuantizedHnsw  pq.rs:120-131, 287-364 (per-sub-space codebooks, u8 codes)
class QuantizedHnsw {
 public:
  // QuantizedHnsw::new(number_of_centroids, comparator, bp)
  QuantizedHnsw(uint32_t number_of_centroids, const Comparator &full, uint32_t m, BuildParameters bp = no_promotion(),
                uint64_t seed = 0, bool table_f16 = false)
      : full_(&full) {
    phnsw_store *ps = nullptr;
    check(phnsw_store_create_pq(full.handle(), m, number_of_centroids, seed, &ps));
    codes_.reset(new Comparator(ps));
    if (table_f16) check(phnsw_pq_set_table_f16(ps, 1));
    std::vector<VectorId> vs(full.len());
    for (uint64_t i = 0; i < vs.size(); i++) vs[i] = i;
    hnsw_.reset(new Hnsw(Hnsw::generate(*codes_, vs, bp)));
  }
  // QuantizedHnsw::search(v, sp)  pq.rs:346-364 (quantize_query = the reference's symmetric form)
  SearchResult search(const float *v, const SearchParameters &sp, bool quantize_query = false) const {
    const uint64_t ef = sp.number_of_candidates;
    std::vector<uint64_t> ids(ef), len(1);
    std::vector<float> d(ef);
    check(phnsw_pq_search_batch(hnsw_->handle(), full_->handle(), v, 1, &sp, quantize_query, ids.data(), d.data(), len.data(), nullptr));
    SearchResult r;
    for (uint64_t j = 0; j < len[0]; j++) r.push_back({ids[j], d[j]});
    return r;
  }
  uint64_t vector_count() const { return hnsw_->vector_count(); }
  static BuildParameters no_promotion() {
    BuildParameters bp;
    bp.promote = 0;
    return bp;
  }

 private:
  const Comparator *full_;
  std::unique_ptr<Comparator> codes_;
  std::unique_ptr<Hnsw> hnsw_;
};

}  // namespace phnsw
