// C ABI of libphnsw (include/phnsw.h): handles, transfers, launches.  Host-side logic
// only -- all arithmetic on vectors and all graph traversal happens in the gfx950 kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <stdexcept>
#include <vector>

#include "phnsw_internal.h"

static thread_local std::string g_err;

void ph_set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
int ph_hip_fail(hipError_t e, const char *what, const char *file, int line) {
  ph_set_error("HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? PHNSW_E_NO_DEVICE : PHNSW_E_HIP;
}

int ph_caught() noexcept {
  try {
    throw;
  } catch (const std::bad_alloc &) {
    try {
      ph_set_error("out of host memory");
    } catch (...) {
    }
    return PHNSW_E_NOMEM;
  } catch (const std::exception &e) {
    try {
      ph_set_error("internal error: %s", e.what());
    } catch (...) {
    }
    return PHNSW_E_INVALID;
  } catch (...) {
    return PHNSW_E_INVALID;
  }
}

extern "C" const char *phnsw_last_error(void) { return g_err.c_str(); }

extern "C" int phnsw_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
} catch (...) { return ph_caught(); }

extern "C" void phnsw_default_search_params(phnsw_search_params *sp) {
  // SearchParameters::default  src/parameters.rs:10-18
  sp->number_of_candidates = 300;
  sp->upper_layer_candidate_count = 300;
  sp->probe_depth = 2;
}

extern "C" void phnsw_default_build_params(phnsw_build_params *bp) {
  // BuildParameters::default  src/parameters.rs:50-64
  bp->order = 12;
  bp->zero_layer_neighborhood_size = 48;
  bp->neighborhood_size = 24;
  bp->optimization.promotion_threshold = 0.01f;
  bp->optimization.neighborhood_threshold = 0.01f;
  bp->optimization.recall_proportion = 0.1f;
  bp->optimization.promotion_proportion = 1.0f;
  phnsw_default_search_params(&bp->optimization.search);
  bp->initial_partition_search.number_of_candidates = 6;
  bp->initial_partition_search.upper_layer_candidate_count = 6;
  bp->initial_partition_search.probe_depth = 2;
  bp->seed = 0;
  bp->max_link_rounds = 0;
  bp->promote = 1;  // the reference always tries promote_at_layer (lib.rs:1575-1580)
}

static int use_device(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) {
    ph_set_error("no HIP device available (libphnsw has no CPU fallback)");
    return PHNSW_E_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    ph_set_error("device %d out of range (%d devices)", device, n);
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(device));
  return 0;
}

// ------------------------------------------------------------------ store

static int store_check_nan(phnsw_store *s) {
  uint32_t *cnt = nullptr;
  PH_HIP(hipMalloc(&cnt, 4));
  PH_HIP(hipMemset(cnt, 0, 4));
  int rc = ph_count_nan(s->rows, s->n * (uint64_t)s->ld, cnt, 0);
  uint32_t h = 0;
  if (!rc) {
    hipError_t e = hipMemcpy(&h, cnt, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = ph_hip_fail(e, "memcpy nan count", __FILE__, __LINE__);
  }
  hipFree(cnt);
  if (rc) return rc;
  if (h) {
    ph_set_error("%u NaN components in the vector store (OrderedFloat would panic, types.rs:86)", h);
    return PHNSW_E_NAN;
  }
  return 0;
}

extern "C" int phnsw_store_create(const float *rows, uint64_t n, uint32_t dim, int metric, int device,
                                  phnsw_store **out) try {
  if (!out || !rows || dim == 0 || n == 0 || metric < 0 || metric > 2 || n >= 0x7FFFFFFFull) {
    ph_set_error("phnsw_store_create: invalid argument");
    return PHNSW_E_INVALID;
  }
  int rc = use_device(device);
  if (rc) return rc;
  phnsw_store *s = new phnsw_store();
  s->device = device;
  s->n = n;
  s->dim = dim;
  s->ld = (dim + 3) / 4 * 4;
  s->metric = metric;
  hipError_t e = hipMalloc(&s->rows, (size_t)n * s->ld * 4);
  if (e != hipSuccess) {
    delete s;
    return ph_hip_fail(e, "hipMalloc store", __FILE__, __LINE__);
  }
  if (s->ld == dim)
    e = hipMemcpy(s->rows, rows, (size_t)n * dim * 4, hipMemcpyHostToDevice);
  else {
    e = hipMemset(s->rows, 0, (size_t)n * s->ld * 4);
    if (e == hipSuccess)
      e = hipMemcpy2D(s->rows, (size_t)s->ld * 4, rows, (size_t)dim * 4, (size_t)dim * 4, n, hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    hipFree(s->rows);
    delete s;
    return ph_hip_fail(e, "copy store", __FILE__, __LINE__);
  }
  rc = store_check_nan(s);
  if (rc) {
    hipFree(s->rows);
    delete s;
    return rc;
  }
  *out = s;
  return 0;
} catch (...) { return ph_caught(); }

// more rows for a store that owns its array (the reference grows `Vec<Vec<f32>>` behind the
// comparator and calls generate / extend on the new ids): the array is reallocated, so no search
// or build may be running on an index over this store; existing VectorIds keep their rows
extern "C" int phnsw_store_append(phnsw_store *s, const float *rows, uint64_t count, uint64_t *out_first_id) try {
  if (!s || !rows || !s->rows || !s->owns_rows || s->n + count >= 0x7FFFFFFFull) {
    ph_set_error("phnsw_store_append: needs an f32 store created by phnsw_store_create* that owns its rows, n < 2^31");
    return PHNSW_E_INVALID;
  }
  if (out_first_id) *out_first_id = s->n;
  if (count == 0) return 0;
  int rc = use_device(s->device);
  if (rc) return rc;
  float *grown = nullptr;
  const size_t ldb = (size_t)s->ld * 4;
  hipError_t e = hipMalloc(&grown, (size_t)(s->n + count) * ldb);
  if (e == hipSuccess) e = hipMemcpy(grown, s->rows, (size_t)s->n * ldb, hipMemcpyDeviceToDevice);
  if (e == hipSuccess && s->ld != s->dim) e = hipMemset(grown + (size_t)s->n * s->ld, 0, (size_t)count * ldb);
  if (e == hipSuccess)
    e = hipMemcpy2D(grown + (size_t)s->n * s->ld, ldb, rows, (size_t)s->dim * 4, (size_t)s->dim * 4, count,
                    hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    if (grown) hipFree(grown);
    return ph_hip_fail(e, "store append", __FILE__, __LINE__);
  }
  phnsw_store tail = *s;  // NaN check over the new rows only
  tail.rows = grown + (size_t)s->n * s->ld;
  tail.n = count;
  rc = store_check_nan(&tail);
  if (rc) {
    hipFree(grown);
    return rc;
  }
  hipFree(s->rows);
  s->rows = grown;
  s->n += count;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_create_device(const float *rows_dev, uint64_t n, uint32_t dim, uint32_t ld, int metric,
                                         int device, phnsw_store **out) try {
  if (!out || !rows_dev || dim == 0 || n == 0 || ld < dim || (ld % 4) || metric < 0 || metric > 2 ||
      n >= 0x7FFFFFFFull || ((uintptr_t)rows_dev % 16)) {
    ph_set_error("phnsw_store_create_device: invalid argument (ld must be a multiple of 4 >= dim, base 16-byte aligned)");
    return PHNSW_E_INVALID;
  }
  int rc = use_device(device);
  if (rc) return rc;
  phnsw_store *s = new phnsw_store();
  s->device = device;
  s->rows = const_cast<float *>(rows_dev);
  s->owns_rows = false;
  s->n = n;
  s->dim = dim;
  s->ld = ld;
  s->metric = metric;
  rc = store_check_nan(s);
  if (rc) {
    delete s;
    return rc;
  }
  *out = s;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_create_synthetic(uint64_t first, uint64_t n, uint32_t dim, uint64_t seed, int normalize,
                                            int metric, int device, phnsw_store **out) try {
  if (!out || dim == 0 || n == 0 || metric < 0 || metric > 2 || n >= 0x7FFFFFFFull) {
    ph_set_error("phnsw_store_create_synthetic: invalid argument");
    return PHNSW_E_INVALID;
  }
  int rc = use_device(device);
  if (rc) return rc;
  phnsw_store *s = new phnsw_store();
  s->device = device;
  s->n = n;
  s->dim = dim;
  s->ld = (dim + 3) / 4 * 4;
  s->metric = metric;
  hipError_t e = hipMalloc(&s->rows, (size_t)n * s->ld * 4);
  if (e != hipSuccess) {
    delete s;
    return ph_hip_fail(e, "hipMalloc store", __FILE__, __LINE__);
  }
  rc = ph_synth_rows(s->rows, first, n, dim, s->ld, seed, normalize, 0);
  if (!rc) {
    e = hipDeviceSynchronize();
    if (e != hipSuccess) rc = ph_hip_fail(e, "synth sync", __FILE__, __LINE__);
  }
  if (rc) {
    hipFree(s->rows);
    delete s;
    return rc;
  }
  *out = s;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_create_clustered(uint64_t first, uint64_t n, uint32_t dim, uint64_t seed,
                                            uint32_t n_clusters, float noise, int metric, int device,
                                            phnsw_store **out) try {
  if (!out || dim == 0 || n == 0 || n_clusters == 0 || metric < 0 || metric > 2 || n >= 0x7FFFFFFFull) {
    ph_set_error("phnsw_store_create_clustered: invalid argument");
    return PHNSW_E_INVALID;
  }
  int rc = use_device(device);
  if (rc) return rc;
  phnsw_store *s = new phnsw_store();
  s->device = device;
  s->n = n;
  s->dim = dim;
  s->ld = (dim + 3) / 4 * 4;
  s->metric = metric;
  hipError_t e = hipMalloc(&s->rows, (size_t)n * s->ld * 4);
  if (e != hipSuccess) {
    delete s;
    return ph_hip_fail(e, "hipMalloc store", __FILE__, __LINE__);
  }
  rc = ph_synth_clustered_rows(s->rows, first, n, dim, s->ld, seed, n_clusters, noise, 0);
  if (rc) {
    hipFree(s->rows);
    delete s;
    return rc;
  }
  *out = s;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_info(const phnsw_store *s, uint64_t *n, uint32_t *dim, uint32_t *ld, int *metric,
                                const float **rows_dev) try {
  if (!s) return PHNSW_E_INVALID;
  if (n) *n = s->n;
  if (dim) *dim = s->dim;
  if (ld) *ld = s->ld;
  if (metric) *metric = s->metric;
  if (rows_dev) *rows_dev = s->rows;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_store_read(const phnsw_store *s, uint64_t first, uint64_t count, float *out) try {
  if (!s || !out || !s->rows || first + count > s->n) {
    ph_set_error("phnsw_store_read: range out of bounds (or a product-quantised store: use phnsw_pq_read)");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  if (count == 0) return 0;
  PH_HIP(hipMemcpy2D(out, (size_t)s->dim * 4, s->rows + first * s->ld, (size_t)s->ld * 4, (size_t)s->dim * 4, count,
                     hipMemcpyDeviceToHost));
  return 0;
} catch (...) { return ph_caught(); }

extern "C" void phnsw_store_destroy(phnsw_store *s) {
  if (!s) return;
  if (--s->refcount > 0) return;
  hipSetDevice(s->device);
  if (s->owns_rows && s->rows) hipFree(s->rows);
  if (s->codes) hipFree(s->codes);
  if (s->codes16) hipFree(s->codes16);
  if (s->centroid_index) phnsw_index_destroy(s->centroid_index);  // (releases its reference to centroid_store)
  if (s->centroid_store) phnsw_store_destroy(s->centroid_store);
  if (s->codebook) hipFree(s->codebook);
  ph_store_anchors_free(s);
  delete s;
}

extern "C" int phnsw_distance_batch(const phnsw_store *s, const float *query, uint64_t query_id, const uint64_t *ids,
                                    uint64_t k, float *out) try {
  if (!s || !ids || !out || (!query && query_id >= s->n) || k > 0xFFFFFFFFull) {
    ph_set_error("phnsw_distance_batch: invalid argument");
    return PHNSW_E_INVALID;
  }
  if (k == 0) return 0;
  PH_HIP(hipSetDevice(s->device));
  std::vector<uint32_t> ids32(k);
  for (uint64_t i = 0; i < k; i++) ids32[i] = ids[i] >= s->n ? PH_EMPTY32 : (uint32_t)ids[i];
  float *qd = nullptr, *od = nullptr;
  uint32_t *idd = nullptr;
  int rc = 0;
  hipError_t e;
  std::vector<float> qpad;
  const float *qsrc = nullptr;
  if (query) {
    qpad.assign(s->ld, 0.f);
    memcpy(qpad.data(), query, (size_t)s->dim * 4);
    if ((e = hipMalloc(&qd, (size_t)s->ld * 4)) != hipSuccess) return ph_hip_fail(e, "malloc", __FILE__, __LINE__);
    e = hipMemcpy(qd, qpad.data(), (size_t)s->ld * 4, hipMemcpyHostToDevice);
    qsrc = qd;
  } else {
    e = hipSuccess;
  }
  if (e == hipSuccess) e = hipMalloc(&idd, k * 4);
  if (e == hipSuccess) e = hipMalloc(&od, k * 4);
  if (e == hipSuccess) e = hipMemcpy(idd, ids32.data(), k * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) rc = ph_hip_fail(e, "distance_batch setup", __FILE__, __LINE__);
  if (!rc) rc = ph_distance_batch(s, qsrc, (uint32_t)query_id, idd, (uint32_t)k, od, 0);
  if (!rc) {
    e = hipMemcpy(out, od, k * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = ph_hip_fail(e, "distance_batch copy", __FILE__, __LINE__);
  }
  if (qd) hipFree(qd);
  if (idd) hipFree(idd);
  if (od) hipFree(od);
  return rc;
} catch (...) { return ph_caught(); }

// ------------------------------------------------------------------ index

void ph_layer_free(PhLayerHost &l) {
  if (l.nodes) hipFree(l.nodes);
  if (l.neighbors) hipFree(l.neighbors);
  if (l.nbr_dist) hipFree(l.nbr_dist);
  if (l.vec2node) hipFree(l.vec2node);
  if (l.recall_q) hipFree(l.recall_q);
  if (l.pos) hipFree(l.pos);
  if (l.ord) hipFree(l.ord);
  l = PhLayerHost();
}

// processing order of the node range [first, first + count) of a layer: argsort of the nodes'
// positions, kept with the layer (build rounds repeat the same range).  *out = nullptr when
// the layer has no positions or the range is too short to matter.
int ph_layer_range_order(PhLayerHost &L, uint32_t first, uint32_t count, const uint32_t **out) {
  *out = nullptr;
  if (!L.pos || count < PH_ORDER_MIN || getenv("PHNSW_NO_LOCALITY")) return 0;
  if (!L.ord || L.ord_first != first || L.ord_count != count) {
    if (L.ord) hipFree(L.ord);
    L.ord = nullptr;
    PH_HIP(hipMalloc(&L.ord, (size_t)count * 4));
    int rc = ph_order_by_keys_device(L.pos + first, count, L.ord, 0);
    if (rc) return rc;
    L.ord_first = first;
    L.ord_count = count;
  }
  *out = L.ord;
  return 0;
}

// upload one layer (u32 device form) and derive the VectorId -> NodeId map that replaces
// Layer::get_node's binary search (src/lib.rs:129-131)
int ph_layer_upload(phnsw_index *ix, const uint32_t *nodes, const uint32_t *neighbors, uint32_t n, uint32_t W,
                    PhLayerHost *out) {
  if (ix) ix->nodes_epoch++;  // a node list comes into being: tables keyed by node lists are void (tiny.hip)
  PhLayerHost l;
  l.n_nodes = n;
  l.W = W;
  hipError_t e = hipMalloc(&l.nodes, (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc(&l.neighbors, (size_t)n * W * 4);
  if (e == hipSuccess) e = hipMemcpy(l.nodes, nodes, (size_t)n * 4, hipMemcpyHostToDevice);
  if (e == hipSuccess && neighbors)
    e = hipMemcpy(l.neighbors, neighbors, (size_t)n * W * 4, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    ph_layer_free(l);
    return ph_hip_fail(e, "layer upload", __FILE__, __LINE__);
  }
  if (!neighbors) {  // all rows empty
    int rc = ph_fill_u32(l.neighbors, PH_EMPTY32, (uint64_t)n * W, 0);
    if (rc) {
      ph_layer_free(l);
      return rc;
    }
  }
  bool identity = true;
  for (uint32_t i = 0; i < n; i++)
    if (nodes[i] != i) {
      identity = false;
      break;
    }
  l.identity = identity;
  if (!identity) {
    e = hipMalloc(&l.vec2node, (size_t)ix->store->n * 4);
    if (e != hipSuccess) {
      ph_layer_free(l);
      return ph_hip_fail(e, "vec2node alloc", __FILE__, __LINE__);
    }
    int rc = ph_fill_u32(l.vec2node, PH_EMPTY32, ix->store->n, 0);
    if (!rc) rc = ph_scatter_vec2node(l.nodes, n, l.vec2node, 0);
    if (!rc) {
      e = hipDeviceSynchronize();
      if (e != hipSuccess) rc = ph_hip_fail(e, "vec2node sync", __FILE__, __LINE__);
    }
    if (rc) {
      ph_layer_free(l);
      return rc;
    }
  }
  *out = l;
  return 0;
}

static std::atomic<uint64_t> g_import_dups{0};
extern "C" uint64_t phnsw_debug_import_duplicates(void) { return g_import_dups.load(); }

// u64 layer arrays of the ABI -> the device's u32 form, checked.  The crate's link step drops its read
// lock before it takes the write lock (lib.rs:1123-1147), so an index it built may hold a NodeId twice
// in one row.  The kernels' visited handling assumes distinct ids per row (the reference itself would
// evaluate such an id twice and burn a probe on the second, empty, expansion), so the LATER occurrences
// are dropped here and the row closed up -- the one documented deviation for imported graphs;
// PHNSW_STRICT_IMPORT=1 refuses such rows instead.
static int validate_layer(const phnsw_store *s, const uint64_t *nodes, const uint64_t *neighbors, uint64_t n,
                          uint64_t W, std::vector<uint32_t> &nodes32, std::vector<uint32_t> &nb32) {
  if (n == 0 || W == 0 || W > 64 || n >= 0x7FFFFFFFull) {
    ph_set_error("layer shape unsupported: node_count=%llu neighborhood_size=%llu (1..64)", (unsigned long long)n,
                 (unsigned long long)W);
    return PHNSW_E_UNSUPPORTED;
  }
  const bool strict = getenv("PHNSW_STRICT_IMPORT") != nullptr;
  nodes32.resize(n);
  nb32.resize(n * W);
  for (uint64_t i = 0; i < n; i++) {
    if (nodes[i] >= s->n || (i > 0 && nodes[i] <= nodes[i - 1])) {
      ph_set_error("layer nodes must be strictly increasing VectorIds below the store size (search.rs:150-157)");
      return PHNSW_E_INVALID;
    }
    nodes32[i] = (uint32_t)nodes[i];
  }
  for (uint64_t i = 0; i < n; i++) {
    bool ended = false;
    uint64_t o = 0;  // next free slot of the cleaned row
    for (uint64_t k = 0; k < W; k++) {
      uint64_t v = neighbors[i * W + k];
      if (v == PHNSW_EMPTY) {
        ended = true;
        continue;
      }
      if (ended || v >= n) {
        ph_set_error("neighbor row %llu: ids must be < node_count and sentinels trailing only (lib.rs:108-125)",
                     (unsigned long long)i);
        return PHNSW_E_INVALID;
      }
      bool dup = false;
      for (uint64_t j = 0; j < o; j++)
        if (nb32[i * W + j] == (uint32_t)v) dup = true;
      if (dup) {
        if (strict) {
          ph_set_error("neighbor row %llu holds node %llu twice (PHNSW_STRICT_IMPORT)", (unsigned long long)i,
                       (unsigned long long)v);
          return PHNSW_E_INVALID;
        }
        g_import_dups++;
        continue;
      }
      nb32[i * W + o++] = (uint32_t)v;
    }
    for (; o < W; o++) nb32[i * W + o] = PH_EMPTY32;
  }
  return 0;
}

extern "C" int phnsw_index_from_layers(phnsw_store *s, uint32_t layer_count, const uint64_t *node_counts,
                                       const uint64_t *neighborhood_sizes, const uint64_t *const *nodes,
                                       const uint64_t *const *neighbors, phnsw_index **out) try {
  if (!s || !out || layer_count == 0 || layer_count > PH_MAX_LAYERS || !node_counts || !neighborhood_sizes ||
      !nodes || !neighbors) {
    ph_set_error("phnsw_index_from_layers: invalid argument");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(s->device));
  phnsw_index *ix = new phnsw_index();
  ix->store = s;
  s->refcount++;
  phnsw_default_build_params(&ix->bp);
  for (uint32_t l = 0; l < layer_count; l++) {
    std::vector<uint32_t> n32, nb32;
    int rc = validate_layer(s, nodes[l], neighbors[l], node_counts[l], neighborhood_sizes[l], n32, nb32);
    PhLayerHost lh;
    if (!rc) rc = ph_layer_upload(ix, n32.data(), nb32.data(), (uint32_t)node_counts[l], (uint32_t)neighborhood_sizes[l], &lh);
    if (rc) {
      phnsw_index_destroy(ix);
      return rc;
    }
    ix->layers.push_back(lh);
  }
  *out = ix;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" void phnsw_index_destroy(phnsw_index *ix) {
  if (!ix) return;
  hipSetDevice(ix->store->device);
  for (auto &l : ix->layers) ph_layer_free(l);
  ph_pending_free(ix);
  ph_workspace_free(ix->ws[0]);
  ph_workspace_free(ix->ws[1]);
  ph_host_stages_free(ix);
  ph_build_table_free(ix);
  if (ix->totals) hipFree(ix->totals);
  phnsw_store_destroy(ix->store);
  delete ix;
  ph_pool_trim();
}

extern "C" uint32_t phnsw_index_layer_count(const phnsw_index *ix) { return ix ? (uint32_t)ix->layers.size() : 0; }

extern "C" int phnsw_index_layer_info(const phnsw_index *ix, uint32_t lft, uint64_t *node_count, uint64_t *W) try {
  if (!ix || lft >= ix->layers.size()) {
    ph_set_error("layer %u out of range", lft);
    return PHNSW_E_INVALID;
  }
  if (node_count) *node_count = ix->layers[lft].n_nodes;
  if (W) *W = ix->layers[lft].W;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_index_layer_read(const phnsw_index *ix, uint32_t lft, uint64_t *nodes, uint64_t *neighbors) try {
  if (!ix || lft >= ix->layers.size()) {
    ph_set_error("layer %u out of range", lft);
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(ix->store->device));
  const PhLayerHost &l = ix->layers[lft];
  if (nodes) {
    std::vector<uint32_t> t(l.n_nodes);
    PH_HIP(hipMemcpy(t.data(), l.nodes, (size_t)l.n_nodes * 4, hipMemcpyDeviceToHost));
    for (uint32_t i = 0; i < l.n_nodes; i++) nodes[i] = t[i];
  }
  if (neighbors) {
    std::vector<uint32_t> t((size_t)l.n_nodes * l.W);
    PH_HIP(hipMemcpy(t.data(), l.neighbors, t.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < t.size(); i++) neighbors[i] = t[i] == PH_EMPTY32 ? PHNSW_EMPTY : t[i];
  }
  return 0;
} catch (...) { return ph_caught(); }

// ------------------------------------------------------------------ search

static int check_sp(const phnsw_index *ix, const phnsw_search_params *sp) {
  if (!ix || !sp || ix->layers.empty()) {
    ph_set_error("search: null index/params or index without layers");
    return PHNSW_E_INVALID;
  }
  if (sp->number_of_candidates == 0 || sp->number_of_candidates > 1024 || sp->probe_depth == 0 ||
      sp->probe_depth > 0xFFFFFFFFull) {
    ph_set_error("search: number_of_candidates must be 1..1024 and probe_depth >= 1 (got %llu, %llu)",
                 (unsigned long long)sp->number_of_candidates, (unsigned long long)sp->probe_depth);
    return PHNSW_E_INVALID;
  }
  return 0;
}

static void fill_args(const phnsw_index *ix, const phnsw_search_params *sp, uint32_t upto, PhSearchArgs &a) {
  memset(&a, 0, sizeof(a));
  const phnsw_store *s = ix->store;
  a.dist = ph_dist_args(s);
  uint32_t nl = (upto == 0 || upto > ix->layers.size()) ? (uint32_t)ix->layers.size() : upto;
  a.n_layers = nl;
  for (uint32_t l = 0; l < nl; l++) {
    const PhLayerHost &h = ix->layers[l];
    a.layers[l].n_nodes = h.n_nodes;
    a.layers[l].W = h.W;
    a.layers[l].nodes = h.nodes;
    a.layers[l].neighbors = h.neighbors;
    a.layers[l].vec2node = h.identity ? nullptr : h.vec2node;
  }
  a.ef = (uint32_t)sp->number_of_candidates;
  a.upper = (uint32_t)std::min<uint64_t>(sp->upper_layer_candidate_count, 0xFFFFFFFFull);
  a.probe_depth = (uint32_t)sp->probe_depth;
}

// spill-list entries per resident wave; PHNSW_OVF_CAP (tests) forces a small list so that the
// overflow -> rerun paths run
uint32_t ph_default_ovf_cap(uint32_t ef) {
  if (const char *e = getenv("PHNSW_OVF_CAP"))
    if (atoi(e) > 0) return (uint32_t)atoi(e);
  return std::max<uint32_t>(8192u, ef * 64u);
}
static uint32_t default_ovf_cap(uint32_t ef) { return ph_default_ovf_cap(ef); }

// batches at least this large descend in several launches (PHNSW_TWO_LAUNCH_MIN overrides: tuning knob)
static uint64_t two_launch_min() {
  const char *e = getenv("PHNSW_TWO_LAUNCH_MIN");
  return (e && atoll(e) > 0) ? (uint64_t)atoll(e) : (uint64_t)PH_TWO_LAUNCH_MIN;
}
static std::atomic<uint64_t> g_two_launch_count{0};  // tests check that the split path really ran
extern "C" uint64_t phnsw_debug_two_launch_count(void) { return g_two_launch_count.load(); }
// chunks of the last descent on this index (phnsw_last_search_dispatches then describes the last one)
extern "C" uint32_t phnsw_debug_last_search_chunks(const phnsw_index *ix) { return ix ? ix->ws[ix->ws_last].n_chunks : 0; }

extern "C" int phnsw_stream_create_beside(int device, void *other_stream, void **out_stream) try {
  if (!out_stream) {
    ph_set_error("phnsw_stream_create_beside: out_stream is NULL");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(device));
  hipStream_t st = nullptr;
  int rc = ph_stream_beside((hipStream_t)other_stream, &st);
  if (rc) return rc;
  *out_stream = (void *)st;
  return 0;
} catch (...) { return ph_caught(); }

// enqueue one search launch; caller owns all device buffers
int ph_search_device(const phnsw_index *ix, const float *queries_dev, uint32_t ldq, const uint32_t *qids_dev,
                     uint64_t nq, const phnsw_search_params *sp, uint32_t upto, const uint32_t *exclude_dev,
                     uint32_t *out_ids, float *out_d, uint32_t *out_len, uint32_t *out_stats, uint32_t *status,
                     uint32_t ovf_cap, uint32_t knn_mode, hipStream_t stream, uint32_t out_stride,
                     uint32_t *out_hit, float threshold, uint32_t first_node, float hit_eps, const uint32_t *order,
                     uint32_t *out_index, const PhRowHint *hint) {
  phnsw_index *mix = const_cast<phnsw_index *>(ix);
  PhSearchArgs a;
  fill_args(ix, sp, upto, a);
  a.queries = queries_dev;
  a.ldq = ldq;
  a.qids = qids_dev;
  a.exclude = exclude_dev;
  a.nq = (uint32_t)nq;
  a.out_ids = out_ids;
  a.out_d = out_d;
  a.out_len = out_len;
  a.out_stats = out_stats;
  a.status = status;
  a.knn_mode = knn_mode;
  a.out_stride = out_stride;
  a.out_hit = out_hit;
  a.threshold = threshold;
  a.first_node = first_node;
  a.hit_eps = hit_eps;
  a.cap_max = knn_mode == 2 ? out_stride : 0;
  a.out_index = out_index;  // Hnsw::search_instrumented: the INSTR kernels, every layer on the per-hop path, one launch
  a.order = (ix->dbg_order && ix->dbg_order_n == nq) ? ix->dbg_order : order;
  std::lock_guard<std::mutex> g(mix->ws_mutex);
  if (!mix->totals) {
    PH_HIP(hipMalloc(&mix->totals, 16));
    PH_HIP(hipMemset(mix->totals, 0, 16));
  }
  a.totals = mix->totals;
  PhWorkspace &ws = mix->ws[mix->ws_next & 1];
  mix->ws_last = mix->ws_next & 1;
  mix->ws_next++;
  int rc = ph_workspace_ensure(ix, ws, std::max(a.ef, a.cap_max), ovf_cap ? ovf_cap : default_ovf_cap(a.ef));
  if (rc) return rc;
  // Large batches of independent queries descend in several launches: the small top layers in
  // the caller's order, then every layer whose rows exceed one XCD's L2 in a launch of its own with
  // the queries sorted by where they landed in the layer above (cell of the best candidate), one
  // contiguous eighth of that order per XCD.  Same arithmetic per query, so the results are
  // identical; neighbouring queries now share rows in L2 / the Infinity Cache.  Between launches
  // the running candidates are parked in the output rows.
  // (layers evaluated densely, tiny.hip, never read a vector row during the traversal: they stay in
  // the first launch whatever their size)
  const uint32_t T = (knn_mode || out_index) ? 0u : ph_tiny_layer_count(ix, a.n_layers, a.ef);
  uint32_t first_big = a.n_layers;
  for (uint32_t l = std::max(1u, T); l < a.n_layers; l++)
    if (ph_layer_own_launch(ix->layers[l].n_nodes, ix->store->ld)) {
      first_big = l;
      break;
    }
  if (!a.order && knn_mode && nq >= PH_ORDER_MIN) {
    // knn / threshold_nn: the queries are the bottom layer's nodes first_node .. first_node + nq
    PhLayerHost &B = mix->layers[a.n_layers - 1];
    rc = ph_layer_anchor_pos(ix->store, B);
    if (!rc && first_node + nq <= B.n_nodes) rc = ph_layer_range_order(B, first_node, (uint32_t)nq, &a.order);
    if (rc) return rc;
  }
  // (not for PQ stores: their searches are not bound by where rows come from, and every launch
  // would build the per-query table again)
  const bool split = ix->store->rows && !a.order && !knn_mode && !out_stride && !out_index && first_big < a.n_layers &&
                     nq >= two_launch_min() &&
                     !getenv("PHNSW_NO_LOCALITY");
  if (split) {
    g_two_launch_count++;
    rc = ph_workspace_order_ensure(ws, a.nq);
    for (uint32_t l = first_big; l < a.n_layers && !rc; l++)
      rc = ph_layer_anchor_pos(ix->store, mix->layers[l - 1]);  // first use on a loaded index; no-op afterwards
    if (rc) return rc;
  }
#ifdef PH_CELL_PROBE
  {
    static unsigned long long *probe_dev = nullptr;
    if (!probe_dev) {
      PH_HIP(hipMalloc(&probe_dev, 12 * sizeof(unsigned long long)));
      PH_HIP(hipMemset(probe_dev, 0, 12 * sizeof(unsigned long long)));
    } else {
      unsigned long long h[12];
      PH_HIP(hipDeviceSynchronize());
      PH_HIP(hipMemcpy(h, probe_dev, sizeof(h), hipMemcpyDeviceToHost));
      if (h[11]) {
        fprintf(stderr, "[cell probe] %llu queries, %llu bottom-layer evaluations; within r chain ranks of the landing cell:", h[11], h[10]);
        for (int k = 0; k < 10; k++) fprintf(stderr, " r<=%u: %.3f", k ? (1u << (k - 1)) : 0u, (double)h[k] / (double)h[10]);
        fprintf(stderr, "\n");
      }
      PH_HIP(hipMemset(probe_dev, 0, 12 * sizeof(unsigned long long)));
    }
    a.probe_out = probe_dev;
    a.probe_pos = nullptr;
    if (!knn_mode && ix->store->rows && nq >= 1000) {
      rc = ph_layer_anchor_pos(ix->store, mix->layers[a.n_layers - 1]);
      if (rc) return rc;
      a.probe_pos = ix->layers[a.n_layers - 1].pos;
    }
  }
#endif
  rc = ph_search_begin(ws, stream);
  if (rc) return rc;
  // The dense top layers (tiny.hip) keep one table row per launch position; a query list longer
  // than the table may hold runs in consecutive chunks of the list (same workspace, same stream).
  // a build's searches first try the table kept across its rounds (tiny.hip): no table pass, and no chunking either
  bool kept_table = false;
  if (hint && T && !split) {
    rc = ph_build_table_prepare(mix, ws, a, *hint, T, stream, &kept_table);
    if (rc) return rc;
  }
  const uint64_t tmax = (T && !kept_table) ? ph_tiny_max_positions(ix, a.n_layers, a.ef) : 0;
  const uint64_t chunk = (tmax && nq > tmax) ? tmax : nq;
  for (uint64_t c0 = 0; c0 < nq; c0 += chunk) {
    PhSearchArgs b = a;
    const uint64_t cnt = std::min<uint64_t>(chunk, nq - c0);
    const bool last_chunk = c0 + cnt == nq;
    b.nq = (uint32_t)cnt;
    if (b.order) {
      b.order += c0;  // positions [c0, c0 + cnt) of the processing order; query indices stay global
    } else if (c0) {
      const uint32_t ostride = out_stride ? out_stride : b.ef;
      if (b.queries) b.queries += c0 * ldq;
      if (b.qids) b.qids += c0;
      if (b.exclude) b.exclude += c0;
      b.out_ids += c0 * ostride;
      b.out_d += c0 * ostride;
      b.out_len += c0;
      if (b.out_stats) b.out_stats += 2 * c0;
      b.status += c0;
      if (b.out_hit) b.out_hit += c0;
      b.first_node += (uint32_t)c0;
    }
    ws.n_chunks = (uint32_t)(c0 / chunk) + 1;
    PH_HIP(hipEventRecord(ws.evc, stream));  // the dispatch record describes ONE chunk (the last): its clock ...
    if (c0) PH_HIP(hipMemsetAsync(ws.dtotals, 0, sizeof(unsigned long long) * 3 * PH_MAX_DISPATCH, stream));  // ... and counters
    if (!kept_table) {
      rc = ph_tiny_prepare(ix, ws, b, T, stream);
      if (rc) return rc;
    }
    ws.d_tiny = b.tiny_layers != 0;
    PH_HIP(hipEventRecord(ws.evd[0], stream));
    if (!split) {
      b.launch_totals = ws.dtotals;
      b.launch_tab = ws.dtotals + 2 * PH_MAX_DISPATCH;
      rc = ph_search_launch(ix, ws, b, stream, last_chunk);
      if (rc) return rc;
      PH_HIP(hipEventRecord(ws.evd[1], stream));
      ws.n_dispatch = 1;
      ws.d_lo[0] = knn_mode ? b.n_layers - 1 : 0;
      ws.d_hi[0] = b.n_layers;
      continue;
    }
    // In a split descent the dense top layers are walked by a launch of their own (ph_search_kernel_dense: twice the
    // resident waves; 17.3 -> 10.7 ms per 100 000 queries); the layers below follow in the full kernel.  Whether the
    // table is usable is a device-side flag: if not, the dense launch does nothing and the follow-up starts from
    // layer 0 (search.hip).  A descent that runs as ONE launch keeps its dense layers there: their walk (latency
    // bound, no HBM traffic) overlaps other queries' gathers, and taking it out costs more than it saves (measured:
    // 11.76 -> 12.53 ms per 10 000 queries).
    const uint32_t Td = b.tiny_layers;
    const bool dense_first = Td && Td < b.n_layers && b.nq >= PH_DENSE_SPLIT_MIN && !getenv("PHNSW_NO_DENSE_SPLIT");
    if (dense_first) b.dense_flag = ws.tiny_member + b.tiny_n;
    uint32_t di = 0;
    uint32_t *const out_hit_final = b.out_hit;
    const uint32_t first_hi = dense_first ? Td : first_big;
    for (uint32_t lo = 0, hi = first_hi; lo < b.n_layers; lo = hi, hi = (lo == Td && dense_first && first_big > Td) ? first_big : hi + 1) {
      PhSearchArgs p = b;
      const bool last = hi == b.n_layers;
      p.layer_lo = lo;
      p.layer_hi = hi;
      if (lo) p.tiny_layers = 0;
      if (dense_first) {
        p.dense_only = lo == 0 ? 1u : 0u;
        p.after_dense = lo == Td ? Td : 0u;
      }
      p.order = lo ? ws.oorder : nullptr;
      p.out_hit = last ? out_hit_final : nullptr;
      p.out_key = last ? nullptr : ws.okey;
      p.key_pos = last ? nullptr : ix->layers[hi - 1].pos;
      p.launch_totals = ws.dtotals + 2 * di;
      p.launch_tab = ws.dtotals + 2 * PH_MAX_DISPATCH + di;
      rc = ph_search_launch(ix, ws, p, stream, last && last_chunk);
      if (!rc && !last) rc = ph_workspace_order_sort(ws, b.nq, stream);
      if (rc) return rc;
      PH_HIP(hipEventRecord(ws.evd[1 + di], stream));
      ws.d_lo[di] = lo;
      ws.d_hi[di] = hi;
      ws.n_dispatch = ++di;
    }
  }
  return 0;
}

extern "C" int phnsw_search_batch_device(const phnsw_index *ix, const float *queries_dev, uint32_t ldq,
                                         const uint32_t *qids_dev, uint64_t nq, const phnsw_search_params *sp,
                                         uint32_t upto_layers, const uint32_t *exclude_dev, uint32_t *out_ids_dev,
                                         float *out_d_dev, uint32_t *out_len_dev, uint32_t *out_stats_dev,
                                         uint32_t *status_dev, void *stream) try {
  int rc = check_sp(ix, sp);
  if (rc) return rc;
  if ((!queries_dev && !qids_dev) || !out_ids_dev || !out_d_dev || !out_len_dev || !status_dev ||
      nq > 0xFFFFFFFFull || (queries_dev && (ldq < ix->store->ld || (ldq % 4) || ((uintptr_t)queries_dev % 16)))) {
    ph_set_error("phnsw_search_batch_device: invalid argument (queries need ldq >= store ld, multiple of 4, 16-byte base)");
    return PHNSW_E_INVALID;
  }
  if (nq == 0) return 0;
  PH_HIP(hipSetDevice(ix->store->device));
  return ph_search_device(ix, queries_dev, ldq, qids_dev, nq, sp, upto_layers, exclude_dev, out_ids_dev, out_d_dev,
                          out_len_dev, out_stats_dev, status_dev, 0, 0, (hipStream_t)stream);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_debug_layer_pos(phnsw_index *ix, uint32_t lft, uint32_t *out_host) try {
  if (lft >= ix->layers.size() || !ix->layers[lft].pos) return PHNSW_E_INVALID;
  PH_HIP(hipMemcpy(out_host, ix->layers[lft].pos, (size_t)ix->layers[lft].n_nodes * 4, hipMemcpyDeviceToHost));
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_debug_set_order(phnsw_index *ix, const uint32_t *order_dev, uint64_t n) try {
  ix->dbg_order = order_dev;
  ix->dbg_order_n = n;
  return 0;
} catch (...) { return ph_caught(); }

// distance evaluations and hops of every search launched on this index since it was created --
// build rounds included: the basis of the build's algorithmic bytes
extern "C" int phnsw_index_counters(const phnsw_index *ix, uint64_t *n_dist, uint64_t *n_hops) try {
  if (!ix) return PHNSW_E_INVALID;
  unsigned long long h[2] = {0, 0};
  if (ix->totals) {
    PH_HIP(hipSetDevice(ix->store->device));
    PH_HIP(hipDeviceSynchronize());
    PH_HIP(hipMemcpy(h, ix->totals, 16, hipMemcpyDeviceToHost));
  }
  if (n_dist) *n_dist = h[0];
  if (n_hops) *n_hops = h[1];
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_last_search_kernel_ms(const phnsw_index *ix, float *ms) try {
  if (!ix || !ms) return PHNSW_E_INVALID;
  phnsw_index *mix = const_cast<phnsw_index *>(ix);
  std::lock_guard<std::mutex> g(mix->ws_mutex);
  PhWorkspace &ws = mix->ws[mix->ws_last];
  if (!ws.timed) {
    ph_set_error("no search has been launched on this index");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipEventSynchronize(ws.ev1));
  PH_HIP(hipEventElapsedTime(ms, ws.ev0, ws.ev1));
  return 0;
} catch (...) { return ph_caught(); }

// the last descent on this index, dispatch by dispatch: entry 0 = the dense-top-layer kernels
// (layer_lo = layer_hi = 0, no counters; 0 ms when the descent had none), then one entry per launch
// of the search kernel with the layers it covered, its time (HIP events on its stream; a launch's
// time includes the key sort in front of it) and the distance evaluations / hops it performed
extern "C" int phnsw_last_search_dispatches(const phnsw_index *ix, uint32_t cap, uint32_t *count, float *ms,
                                            uint64_t *n_dist, uint64_t *n_hops, uint32_t *layer_lo,
                                            uint32_t *layer_hi) try {
  if (!ix || !count) return PHNSW_E_INVALID;
  phnsw_index *mix = const_cast<phnsw_index *>(ix);
  std::lock_guard<std::mutex> g(mix->ws_mutex);
  PhWorkspace &ws = mix->ws[mix->ws_last];
  if (!ws.timed) {
    ph_set_error("no search has been launched on this index");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(ix->store->device));
  PH_HIP(hipEventSynchronize(ws.ev1));
  unsigned long long h[2 * PH_MAX_DISPATCH];
  PH_HIP(hipMemcpy(h, ws.dtotals, sizeof(h), hipMemcpyDeviceToHost));
  const uint32_t n = ws.n_dispatch + 1;
  *count = n;
  for (uint32_t i = 0; i < n && i < cap; i++) {
    float t = 0.f;
    PH_HIP(hipEventElapsedTime(&t, i == 0 ? ws.evc : ws.evd[i - 1], ws.evd[i]));
    if (ms) ms[i] = t;
    if (n_dist) n_dist[i] = i ? h[2 * (i - 1)] : 0;
    if (n_hops) n_hops[i] = i ? h[2 * (i - 1) + 1] : 0;
    if (layer_lo) layer_lo[i] = i ? ws.d_lo[i - 1] : 0;
    if (layer_hi) layer_hi[i] = i ? ws.d_hi[i - 1] : 0;
  }
  return 0;
} catch (...) { return ph_caught(); }

// of each search launch's distance evaluations, those served by the dense tables (entry 0, the table pass, is 0):
// n_dist - n_table are the gathered rows -- what the HBM roofline of a launch is computed from
extern "C" int phnsw_last_search_table_evals(const phnsw_index *ix, uint32_t cap, uint32_t *count, uint64_t *n_table) try {
  if (!ix || !count) return PHNSW_E_INVALID;
  phnsw_index *mix = const_cast<phnsw_index *>(ix);
  std::lock_guard<std::mutex> g(mix->ws_mutex);
  PhWorkspace &ws = mix->ws[mix->ws_last];
  if (!ws.timed) {
    ph_set_error("no search has been launched on this index");
    return PHNSW_E_INVALID;
  }
  PH_HIP(hipSetDevice(ix->store->device));
  PH_HIP(hipEventSynchronize(ws.ev1));
  unsigned long long h[PH_MAX_DISPATCH];
  PH_HIP(hipMemcpy(h, ws.dtotals + 2 * PH_MAX_DISPATCH, sizeof(h), hipMemcpyDeviceToHost));
  const uint32_t n = ws.n_dispatch + 1;
  *count = n;
  for (uint32_t i = 0; i < n && i < cap; i++)
    if (n_table) n_table[i] = i ? h[i - 1] : 0;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_dense_top_layers(const phnsw_index *ix, uint64_t number_of_candidates, uint32_t *layers, uint64_t *nodes,
                                      uint32_t *matrix_cores) try {
  if (!ix || !ix->store) {
    ph_set_error("phnsw_dense_top_layers: null index");
    return PHNSW_E_INVALID;
  }
  const uint32_t ef = (uint32_t)std::min<uint64_t>(number_of_candidates, 0xFFFFFFFFull);
  const uint32_t T = ph_tiny_layer_count(ix, (uint32_t)ix->layers.size(), ef);
  if (layers) *layers = T;
  if (nodes) *nodes = T ? ix->layers[T - 1].n_nodes : 0;
  if (matrix_cores) *matrix_cores = (T && ph_tiny_matrix_cores(ix)) ? 1u : 0u;
  return 0;
} catch (...) { return ph_caught(); }

// host-pointer searches live in hostpath.hip (persistent staging, pipelined chunks)
int ph_search_host(const phnsw_index *ix, const float *queries, const uint64_t *qids, uint64_t nq,
                   const phnsw_search_params *sp, uint32_t upto, const uint64_t *exclude, uint64_t out_k,
                   uint64_t *out_ids, float *out_d, uint64_t *out_len, uint64_t *out_stats, uint32_t knn_mode,
                   uint64_t *out_index);
static int search_host(const phnsw_index *ix, const float *queries, const uint64_t *qids, uint64_t nq,
                       const phnsw_search_params *sp, uint32_t upto, const uint64_t *exclude, uint64_t *out_ids,
                       float *out_d, uint64_t *out_len, uint64_t *out_stats, uint32_t knn_mode) {
  return ph_search_host(ix, queries, qids, nq, sp, upto, exclude, 0, out_ids, out_d, out_len, out_stats, knn_mode, nullptr);
}

// Hnsw::knn  src/lib.rs:905-928: queue of 3k seeded with (self, 0.0), closest_nodes on the
// bottom layer, drop self, take k
extern "C" int phnsw_knn(const phnsw_index *ix, uint64_t k, uint64_t probe_depth, uint64_t *out_ids, float *out_d,
                         uint64_t *out_len) try {
  if (!ix || ix->layers.empty() || k == 0 || k * 3 > 1024) {
    ph_set_error("phnsw_knn: k must be 1..341");
    return PHNSW_E_INVALID;
  }
  const PhLayerHost &L = ix->layers.back();
  phnsw_search_params sp = {k * 3, k * 3, probe_depth};
  uint64_t n = L.n_nodes, cap = k * 3;
  std::vector<uint64_t> ids(n * cap), len(n);
  std::vector<float> d(n * cap);
  int rc = search_host(ix, nullptr, nullptr, n, &sp, 0, nullptr, ids.data(), d.data(), len.data(), nullptr, 1);
  if (rc) return rc;
  std::vector<uint64_t> nodes(n);
  rc = phnsw_index_layer_read(ix, (uint32_t)ix->layers.size() - 1, nodes.data(), nullptr);
  if (rc) return rc;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t o = 0;
    for (uint64_t j = 0; j < len[i] && o < k; j++) {
      if (ids[i * cap + j] == nodes[i]) continue;  // filter(|(n,_)| *n != node)
      out_ids[i * k + o] = ids[i * cap + j];
      out_d[i * k + o] = d[i * cap + j];
      o++;
    }
    out_len[i] = o;
    for (; o < k; o++) {
      out_ids[i * k + o] = PHNSW_EMPTY;
      out_d[i * k + o] = PH_FMAX;
    }
  }
  return 0;
} catch (...) { return ph_caught(); }

namespace {
// device scratch that grows to the largest request and is released with its owner
template <class T>
struct GrowBuf {
  T *p = nullptr;
  size_t have = 0;
  int alloc(size_t count) {
    count = std::max<size_t>(count, 1);
    if (have >= count) return 0;
    if (p) hipFree(p);
    p = nullptr;
    have = 0;
    hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess) return ph_hip_fail(e, "hipMalloc (threshold_nn)", __FILE__, __LINE__);
    have = count;
    return 0;
  }
  ~GrowBuf() {
    if (p) hipFree(p);
  }
  GrowBuf() = default;
  GrowBuf(const GrowBuf &) = delete;
  GrowBuf &operator=(const GrowBuf &) = delete;
};
}  // namespace
#define PH_TRY_(x)         \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)

// threshold_nn for the nodes whose queue outgrew LDS: the same search from the start with the queue in global memory
// (ph_search_kernel_big), its capacity doubled from launch to launch until nobody asks for more.  A queue never
// holds more than the layer's nodes, so a capacity of twice that is final.
static int threshold_nn_big(const phnsw_index *ix, const phnsw_search_params *sp, float threshold,
                            const std::vector<uint32_t> &all_nodes, const std::vector<uint32_t> &h_nodes, uint64_t max_out,
                            uint64_t *out_ids, float *out_d, uint64_t *out_len, uint32_t lds_cap) {
  const PhLayerHost &L = ix->layers.back();
  const uint32_t n = L.n_nodes;
  const uint64_t isd = sp->number_of_candidates;
  const uint32_t os = (uint32_t)std::min<uint64_t>(max_out + 2, (uint64_t)n + 1);  // self + max_out + one more to tell "too many"
  const uint64_t words = ((uint64_t)n + 31) / 32 + 1;
  uint64_t cap0 = isd * 2;  // capacities follow the queue's own doubling sequence
  while (cap0 <= lds_cap) cap0 *= 2;  // what the LDS queues held has been tried
  GrowBuf<uint32_t> d_nodes, d_vis, d_q, d_oid, d_len, d_status, d_counter;
  GrowBuf<float> d_od;
  GrowBuf<uint2> d_ovf;
  std::vector<uint32_t> h_ids, h_len, h_status;
  std::vector<float> h_d;
  // the output rows of a launch are [nodes][os]: the list goes through in pieces that keep them below 1 GiB
  size_t piece = std::max<size_t>(1, std::min<size_t>(16384, (1ull << 30) / ((size_t)os * 8)));
  if (const char *e = getenv("PHNSW_THRESHOLD_BIG_PIECE"))  // tests
    if (atoll(e) > 0) piece = (size_t)atoll(e);
  for (size_t c0 = 0; c0 < all_nodes.size(); c0 += piece) {
  std::vector<uint32_t> nodes(all_nodes.begin() + c0, all_nodes.begin() + std::min(all_nodes.size(), c0 + piece));
  uint64_t cap = cap0;
  while (!nodes.empty()) {
    if (cap > 0x7FFFFFFFull) {
      ph_set_error("threshold_nn: queue capacity beyond 2^31");
      return PHNSW_E_UNSUPPORTED;
    }
    const bool final_cap = cap >= 2ull * n;
    // resident waves: what 4 GiB of queues, spill lists (a layer's nodes at most) and visited bits allow
    const uint64_t per_slot = cap * 8 + (uint64_t)n * 8 + words * 4;
    uint64_t budget = 4ull << 30;
    if (const char *e = getenv("PHNSW_THRESHOLD_BIG_BYTES")) budget = strtoull(e, nullptr, 10);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(nodes.size(), 1024), std::max<uint64_t>(1, budget / per_slot));
    const size_t cnt = nodes.size();
    PH_TRY_(d_nodes.alloc(cnt));
    PH_TRY_(d_oid.alloc(cnt * os));
    PH_TRY_(d_od.alloc(cnt * os));
    PH_TRY_(d_len.alloc(cnt));
    PH_TRY_(d_status.alloc(cnt));
    PH_TRY_(d_counter.alloc(128));
    PH_TRY_(d_vis.alloc((size_t)grid * words));
    PH_TRY_(d_ovf.alloc((size_t)grid * n));
    PH_TRY_(d_q.alloc((size_t)grid * 2 * cap));
    PH_HIP(hipMemset(d_vis.p, 0, (size_t)grid * words * 4));
    PH_HIP(hipMemcpy(d_nodes.p, nodes.data(), cnt * 4, hipMemcpyHostToDevice));
    PhSearchArgs a;
    fill_args(ix, sp, 0, a);
    a.nq = (uint32_t)cnt;
    a.out_ids = d_oid.p;
    a.out_d = d_od.p;
    a.out_len = d_len.p;
    a.status = d_status.p;
    a.knn_mode = 2;
    a.threshold = threshold;
    a.out_stride = os;
    a.cap_max = (uint32_t)cap;
    a.knn_nodes = d_nodes.p;
    a.big_q = d_q.p;
    a.visited = d_vis.p;
    a.visited_words = words;
    a.ovf = d_ovf.p;
    a.ovf_cap = n;
    a.counter = d_counter.p;
    PH_TRY_(ph_search_launch_big(ix, a, grid, 0));
    PH_HIP(hipDeviceSynchronize());
    h_ids.resize(cnt * os);
    h_d.resize(cnt * os);
    h_len.resize(cnt);
    h_status.resize(cnt);
    PH_HIP(hipMemcpy(h_ids.data(), d_oid.p, cnt * os * 4, hipMemcpyDeviceToHost));
    PH_HIP(hipMemcpy(h_d.data(), d_od.p, cnt * os * 4, hipMemcpyDeviceToHost));
    PH_HIP(hipMemcpy(h_len.data(), d_len.p, cnt * 4, hipMemcpyDeviceToHost));
    PH_HIP(hipMemcpy(h_status.data(), d_status.p, cnt * 4, hipMemcpyDeviceToHost));
    std::vector<uint32_t> again;
    for (size_t x = 0; x < cnt; x++) {
      const uint64_t i = nodes[x];
      if (h_status[x] == 6 && !final_cap) {  // ST_CAPACITY: once more with twice the queue
        again.push_back(nodes[x]);
        continue;
      }
      if (h_status[x]) {
        ph_set_error("threshold_nn: node %llu failed with status %u at a queue capacity of %llu", (unsigned long long)i,
                     h_status[x], (unsigned long long)cap);
        return PHNSW_E_OVERFLOW;
      }
      uint64_t o = 0;
      for (uint32_t j = 0; j < h_len[x]; j++) {
        const uint32_t v = h_ids[x * os + j];
        const float d = h_d[x * os + j];
        if (v == h_nodes[i]) continue;  // filter(|(n,_)| *n != node)
        if (!(d < threshold)) break;    // take_while(distance < threshold)
        if (o == max_out) {
          ph_set_error("threshold_nn: node %llu has more than max_out=%llu results", (unsigned long long)i,
                       (unsigned long long)max_out);
          return PHNSW_E_OVERFLOW;
        }
        out_ids[i * max_out + o] = v;
        out_d[i * max_out + o] = d;
        o++;
      }
      out_len[i] = o;
      for (; o < max_out; o++) {
        out_ids[i * max_out + o] = PHNSW_EMPTY;
        out_d[i * max_out + o] = PH_FMAX;
      }
    }
    nodes.swap(again);
    cap *= 2;
  }
  }
  return 0;
}

// Hnsw::threshold_nn  src/lib.rs:930-962: per bottom-layer node a queue seeded with
// (self, 0.0) that doubles (resize_capacity) until its last entry reaches the threshold;
// result = entries below the threshold without self.  Nodes are processed in chunks with the
// queue in LDS (up to 1024 entries); a node that needs more goes through threshold_nn_big.
extern "C" int phnsw_threshold_nn(const phnsw_index *ix, float threshold, uint64_t probe_depth,
                                  uint64_t initial_search_depth, uint64_t max_out, uint64_t *out_ids, float *out_d,
                                  uint64_t *out_len) try {
  if (!ix || ix->layers.empty() || !out_ids || !out_d || !out_len || initial_search_depth == 0 ||
      initial_search_depth > 0x40000000ull || probe_depth == 0 || max_out == 0) {
    ph_set_error("phnsw_threshold_nn: initial_search_depth and probe_depth must be >= 1");
    return PHNSW_E_INVALID;
  }
  const phnsw_store *s = ix->store;
  PH_HIP(hipSetDevice(s->device));
  const PhLayerHost &L = ix->layers.back();
  const uint32_t n = L.n_nodes, CAPMAX = 1024, CHUNK = 16384;  // the LDS queues hold 1024 entries
  phnsw_search_params sp = {initial_search_depth, initial_search_depth, probe_depth};
  std::vector<uint32_t> h_nodes(n);
  PH_HIP(hipMemcpy(h_nodes.data(), L.nodes, (size_t)n * 4, hipMemcpyDeviceToHost));
  std::vector<uint32_t> big;  // nodes whose queue has to grow past CAPMAX
  if (initial_search_depth > CAPMAX || getenv("PHNSW_THRESHOLD_ALL_BIG")) {
    big.resize(n);
    for (uint32_t i = 0; i < n; i++) big[i] = i;
    return threshold_nn_big(ix, &sp, threshold, big, h_nodes, max_out, out_ids, out_d, out_len, 0);
  }
  uint32_t *oid = nullptr, *olen = nullptr, *ostat = nullptr;
  float *od = nullptr;
  uint32_t cn = std::min(n, CHUNK);
  hipError_t e = hipMalloc(&oid, (size_t)cn * CAPMAX * 4);
  if (e == hipSuccess) e = hipMalloc(&od, (size_t)cn * CAPMAX * 4);
  if (e == hipSuccess) e = hipMalloc(&olen, (size_t)cn * 4);
  if (e == hipSuccess) e = hipMalloc(&ostat, (size_t)cn * 4);
  int rc = e == hipSuccess ? 0 : ph_hip_fail(e, "threshold_nn staging", __FILE__, __LINE__);
  std::vector<uint32_t> h_ids((size_t)cn * CAPMAX), h_len(cn), h_status(cn);
  std::vector<float> h_d((size_t)cn * CAPMAX);
  for (uint32_t first = 0; !rc && first < n; first += CHUNK) {
    uint32_t cnt = std::min(CHUNK, n - first);
    rc = ph_search_device(ix, nullptr, 0, nullptr, cnt, &sp, 0, nullptr, oid, od, olen, nullptr, ostat, 0, 2, 0, CAPMAX,
                          nullptr, threshold, first);
    if (rc) break;
    e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h_ids.data(), oid, (size_t)cnt * CAPMAX * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_d.data(), od, (size_t)cnt * CAPMAX * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_len.data(), olen, (size_t)cnt * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(h_status.data(), ostat, (size_t)cnt * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      rc = ph_hip_fail(e, "threshold_nn readback", __FILE__, __LINE__);
      break;
    }
    for (uint32_t x = 0; x < cnt && !rc; x++) {
      uint64_t i = first + x;
      if (h_status[x] == 6) {  // ST_CAPACITY
        big.push_back((uint32_t)i);
        continue;
      }
      if (h_status[x]) {
        ph_set_error("threshold_nn: node %llu overflowed the frontier workspace", (unsigned long long)i);
        rc = PHNSW_E_OVERFLOW;
        break;
      }
      uint64_t o = 0;
      for (uint32_t j = 0; j < h_len[x]; j++) {
        uint32_t v = h_ids[(size_t)x * CAPMAX + j];
        float d = h_d[(size_t)x * CAPMAX + j];
        if (v == h_nodes[i]) continue;   // filter(|(n,_)| *n != node)
        if (!(d < threshold)) break;     // take_while(distance < threshold)
        if (o == max_out) {
          ph_set_error("threshold_nn: node %llu has more than max_out=%llu results", (unsigned long long)i,
                       (unsigned long long)max_out);
          rc = PHNSW_E_OVERFLOW;
          break;
        }
        out_ids[i * max_out + o] = v;
        out_d[i * max_out + o] = d;
        o++;
      }
      out_len[i] = o;
      for (; o < max_out; o++) {
        out_ids[i * max_out + o] = PHNSW_EMPTY;
        out_d[i * max_out + o] = PH_FMAX;
      }
    }
  }
  if (oid) hipFree(oid);
  if (od) hipFree(od);
  if (olen) hipFree(olen);
  if (ostat) hipFree(ostat);
  if (!rc && !big.empty()) rc = threshold_nn_big(ix, &sp, threshold, big, h_nodes, max_out, out_ids, out_d, out_len, CAPMAX);
  return rc;
} catch (...) { return ph_caught(); }
