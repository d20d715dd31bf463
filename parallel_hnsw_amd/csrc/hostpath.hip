// The host-pointer search entry points: what `Hnsw::search` binds to through the Rust shim
// (/root/reference/src/lib.rs:654-673; rust/parallel-hnsw-gpu).  Queries arrive in host memory, results leave to
// host memory as u64 ids -- the boundary BASELINE's drop-in story is about, so it gets the same care as the kernels:
//
//  * staging is PERSISTENT per index (device buffers and pinned mirrors sized to the high-water mark, two slots
//    with a stream each): no hipMalloc, no hipFree, no device-wide synchronisation per call;
//  * a long query list is cut into chunks that alternate between the two slots: the upload of chunk k+1 and the
//    download of chunk k-1 run beside the search of chunk k (the two search workspaces of the index alternate the
//    same way), and the first chunk is small so that the GPU starts after a fraction of the upload;
//  * the results are narrowed to the caller's top-k and widened to u64 ON THE DEVICE (ph_take_kernel): the
//    reference returns the whole queue and lets the caller truncate (lib.rs:1118), here the truncation saves the
//    transfer -- phnsw_search_batch_topk; k = number_of_candidates is phnsw_search_batch;
//  * queries whose frontier outgrew the spill workspace are re-run alone with 8x the room, as before.
#include <hip/hip_runtime.h>

#include <chrono>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "phnsw_internal.h"

#define PH_TRY(x)          \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)

namespace {

struct Slot {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr;  // after the slot's last download was enqueued
  float *q = nullptr;
  size_t q_bytes = 0;
  uint32_t *small = nullptr;  // device: qid | excl | len | status | index | stats(2)   [7][n_cap]
  size_t n_cap = 0;
  uint32_t *ids = nullptr;  // device [cnt][ef]
  float *d = nullptr;
  size_t out_cap = 0;
  uint64_t *ids64 = nullptr;  // device [cnt][k]
  float *dk = nullptr;
  size_t take_cap = 0;
  uint32_t *h_small = nullptr;  // pinned mirror of `small`
  size_t h_cap = 0;
  uint64_t c0 = 0, cnt = 0;  // the chunk in flight
  bool busy = false;
};

}  // namespace

struct PhHostStage {
  Slot slot[2];
  bool in_use = false;
  bool streams_checked = false;  // the two slots' streams were seen to run side by side (pair_streams)
};

namespace {

void slot_free(Slot &s) {
  if (s.q) hipFree(s.q);
  if (s.small) hipFree(s.small);
  if (s.ids) hipFree(s.ids);
  if (s.d) hipFree(s.d);
  if (s.ids64) hipFree(s.ids64);
  if (s.dk) hipFree(s.dk);
  if (s.h_small) hipHostFree(s.h_small);
  if (s.done) hipEventDestroy(s.done);
  if (s.stream) hipStreamDestroy(s.stream);
  s = Slot();
}

template <class T>
int grow(T **p, size_t *have, size_t need, const char *what) {
  if (*have >= need) return 0;
  if (*p) hipFree(*p);
  *p = nullptr;
  *have = 0;
  need += need / 4;  // headroom: lists that grow a little do not reallocate every call
  hipError_t e = hipMalloc((void **)p, need);
  if (e != hipSuccess) return ph_hip_fail(e, what, __FILE__, __LINE__);
  *have = need;
  return 0;
}

// the pipeline needs its two streams on different hardware queues (ph_stream_beside, misc.hip): tried once per staging set
int pair_streams(PhHostStage &st) {
  if (st.streams_checked) return 0;
  Slot &a = st.slot[0], &b = st.slot[1];
  if (!a.stream) return 0;
  PH_TRY(ph_stream_beside(a.stream, &b.stream));
  st.streams_checked = true;
  return 0;
}

int slot_ensure(Slot &s, uint64_t cnt, uint32_t ld, bool has_q, uint32_t ef, uint32_t k) {
  if (!s.stream) {
    PH_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    PH_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  }
  if (has_q) PH_TRY(grow(&s.q, &s.q_bytes, (size_t)cnt * ld * 4, "host path: query staging"));
  if (s.n_cap < cnt) {
    if (s.small) hipFree(s.small);
    s.small = nullptr;
    s.n_cap = 0;
    const size_t n = cnt + cnt / 4 + 64;
    hipError_t e = hipMalloc(&s.small, n * 7 * 4);
    if (e != hipSuccess) return ph_hip_fail(e, "host path: per-query staging", __FILE__, __LINE__);
    s.n_cap = n;
  }
  if (s.h_cap < s.n_cap) {
    if (s.h_small) hipHostFree(s.h_small);
    s.h_small = nullptr;
    s.h_cap = 0;
    hipError_t e = hipHostMalloc((void **)&s.h_small, s.n_cap * 7 * 4, hipHostMallocDefault);
    if (e != hipSuccess) return ph_hip_fail(e, "host path: pinned staging", __FILE__, __LINE__);
    s.h_cap = s.n_cap;
  }
  const size_t out = (size_t)cnt * ef;
  if (s.out_cap < out) {
    if (s.ids) hipFree(s.ids);
    if (s.d) hipFree(s.d);
    s.ids = nullptr;
    s.d = nullptr;
    s.out_cap = 0;
    const size_t n = out + out / 4;
    hipError_t e = hipMalloc(&s.ids, n * 4);
    if (e == hipSuccess) e = hipMalloc(&s.d, n * 4);
    if (e != hipSuccess) return ph_hip_fail(e, "host path: result staging", __FILE__, __LINE__);
    s.out_cap = n;
  }
  const size_t take = (size_t)cnt * k;
  if (s.take_cap < take) {
    if (s.ids64) hipFree(s.ids64);
    if (s.dk) hipFree(s.dk);
    s.ids64 = nullptr;
    s.dk = nullptr;
    s.take_cap = 0;
    const size_t n = take + take / 4;
    hipError_t e = hipMalloc(&s.ids64, n * 8);
    if (e == hipSuccess) e = hipMalloc(&s.dk, n * 4);
    if (e != hipSuccess) return ph_hip_fail(e, "host path: top-k staging", __FILE__, __LINE__);
    s.take_cap = n;
  }
  return 0;
}

// rows of `ef` results -> the leading k of each as u64 ids (0xFFFFFFFF -> PHNSW_EMPTY) + distances
__global__ void ph_take_kernel(const uint32_t *ids, const float *d, uint32_t ef, uint32_t k, uint64_t n, uint64_t *ids64,
                               float *dk) {
  const uint64_t total = n * k;
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = x / k, c = x - r * k;
    const uint32_t id = ids[r * ef + c];
    ids64[x] = id == PH_EMPTY32 ? PHNSW_EMPTY : (uint64_t)id;
    dk[x] = d[r * ef + c];
  }
}

PhHostStage *stage_acquire(phnsw_index *ix) {
  std::lock_guard<std::mutex> g(ix->stage_mutex);
  for (PhHostStage *st : ix->stages)
    if (!st->in_use) {
      st->in_use = true;
      return st;
    }
  PhHostStage *st = new PhHostStage();
  st->in_use = true;
  ix->stages.push_back(st);
  return st;
}
void stage_release(phnsw_index *ix, PhHostStage *st) {
  std::lock_guard<std::mutex> g(ix->stage_mutex);
  st->in_use = false;
}
struct StageGuard {
  phnsw_index *ix;
  PhHostStage *st;
  ~StageGuard() { stage_release(ix, st); }
};

// chunk plan: short lists whole; long ones start with a small chunk (the GPU starts early), then even pieces
void plan_chunks(uint64_t nq, std::vector<uint64_t> &bounds) {
  uint64_t pipe_min = 6144, first = 1024, piece = 4096;
  if (const char *e = getenv("PHNSW_HOST_CHUNKS")) {  // "pipe_min,first,piece" -- tuning knob
    unsigned long long a = 0, b = 0, c = 0;
    if (sscanf(e, "%llu,%llu,%llu", &a, &b, &c) == 3 && b > 0 && c > 0) pipe_min = a, first = b, piece = c;
  }
  bounds.clear();
  bounds.push_back(0);
  if (nq < pipe_min) {
    bounds.push_back(nq);
    return;
  }
  uint64_t at = std::min(first, nq);
  bounds.push_back(at);
  const uint64_t rest = nq - at;
  uint64_t pieces = std::max<uint64_t>(1, (rest + piece - 1) / piece);
  // very long lists: pieces of at least PH_TWO_LAUNCH_MIN queries, so that each still descends in launches of
  // its own per large layer, its queries ordered by where they landed (api.hip; the upload of 100 MB per piece
  // hides behind the previous piece's search all the same)
  if (nq >= 2 * (uint64_t)PH_TWO_LAUNCH_MIN && !getenv("PHNSW_HOST_CHUNKS")) pieces = std::max<uint64_t>(1, rest / PH_TWO_LAUNCH_MIN);
  for (uint64_t p = 1; p <= pieces; p++) bounds.push_back(at + rest * p / pieces);
}

}  // namespace

void ph_host_stages_free(phnsw_index *ix) {
  for (PhHostStage *st : ix->stages) {
    slot_free(st->slot[0]);
    slot_free(st->slot[1]);
    delete st;
  }
  ix->stages.clear();
}

// out_k == 0: the whole queue (number_of_candidates entries per query)
int ph_search_host(const phnsw_index *ix, const float *queries, const uint64_t *qids, uint64_t nq,
                   const phnsw_search_params *sp, uint32_t upto, const uint64_t *exclude, uint64_t out_k,
                   uint64_t *out_ids, float *out_d, uint64_t *out_len, uint64_t *out_stats, uint32_t knn_mode,
                   uint64_t *out_index) {
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
  const uint32_t ef = (uint32_t)sp->number_of_candidates;
  if ((!queries && !qids && !knn_mode) || !out_ids || !out_d || !out_len || nq > 0xFFFFFFFFull || out_k > ef) {
    ph_set_error("search: invalid argument (queries or ids, outputs, k <= number_of_candidates)");
    return PHNSW_E_INVALID;
  }
  if (nq == 0) return 0;
  const uint32_t k = out_k ? (uint32_t)out_k : ef;
  const phnsw_store *s = ix->store;
  PH_HIP(hipSetDevice(s->device));
  if (qids)
    for (uint64_t i = 0; i < nq; i++)
      if (qids[i] >= s->n) {
        ph_set_error("search: stored query id %llu out of range", (unsigned long long)qids[i]);
        return PHNSW_E_INVALID;
      }
  phnsw_index *mix = const_cast<phnsw_index *>(ix);
  PhHostStage *st = stage_acquire(mix);
  StageGuard guard{mix, st};
  std::vector<uint64_t> bounds;
  plan_chunks(nq, bounds);
  const size_t n_chunks = bounds.size() - 1;
  std::vector<uint32_t> redo;
  int rc = 0;

  // download of the chunk a slot holds: per-query words through the pinned mirror, result rows straight into the
  // caller's arrays; returns after everything of that chunk has landed
  auto finish = [&](Slot &sl) -> int {
    if (!sl.busy) return 0;
    sl.busy = false;
    const uint64_t c0 = sl.c0, cnt = sl.cnt, N = sl.cnt;  // the per-query words are laid out [7][cnt]
    hipLaunchKernelGGL(ph_take_kernel, dim3((uint32_t)std::min<uint64_t>((cnt * k + 255) / 256, 4096)), dim3(256), 0, sl.stream,
                       sl.ids, sl.d, ef, k, cnt, sl.ids64, sl.dk);
    PH_HIP(hipGetLastError());
    // len | status | index | stats sit contiguously at rows 2..6 of `small`
    PH_HIP(hipMemcpyAsync(sl.h_small + 2 * N, sl.small + 2 * N, 5 * N * 4, hipMemcpyDeviceToHost, sl.stream));
    PH_HIP(hipMemcpyAsync(out_ids + c0 * k, sl.ids64, (size_t)cnt * k * 8, hipMemcpyDeviceToHost, sl.stream));
    PH_HIP(hipMemcpyAsync(out_d + c0 * k, sl.dk, (size_t)cnt * k * 4, hipMemcpyDeviceToHost, sl.stream));
    PH_HIP(hipStreamSynchronize(sl.stream));
    const uint32_t *h_len = sl.h_small + 2 * N, *h_status = sl.h_small + 3 * N, *h_index = sl.h_small + 4 * N,
                   *h_stats = sl.h_small + 5 * N;
    for (uint64_t i = 0; i < cnt; i++) {
      if (h_status[i] == 4) {
        ph_set_error("search: a candidate vector is missing from a lower layer (layers not nested, lib.rs:261)");
        return PHNSW_E_MISSING_NODE;
      }
      if (h_status[i] == 5) redo.push_back((uint32_t)(c0 + i));
      out_len[c0 + i] = std::min<uint32_t>(h_len[i], k);
    }
    if (out_stats)
      for (uint64_t i = 0; i < 2 * cnt; i++) out_stats[2 * c0 + i] = h_stats[i];
    if (out_index)
      for (uint64_t i = 0; i < cnt; i++) out_index[c0 + i] = h_index[i] == 0xFFFFFFFFu ? UINT64_MAX : h_index[i];
    return 0;
  };

  if (n_chunks >= 2 && !st->streams_checked) {  // the first pipelined call of this staging set: see pair_streams
    for (Slot &sl : st->slot)
      if (!sl.stream) {
        PH_HIP(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        PH_HIP(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
      }
    rc = pair_streams(*st);
  }
  for (size_t c = 0; c < n_chunks && !rc; c++) {
    Slot &sl = st->slot[c & 1];
    rc = finish(sl);  // the slot's previous chunk (two chunks back) comes home first
    if (rc) break;
    const uint64_t c0 = bounds[c], cnt = bounds[c + 1] - c0;
    rc = slot_ensure(sl, cnt, s->ld, queries != nullptr, ef, k);
    if (rc) break;
    const uint64_t N = cnt;
    uint32_t *d_qid = sl.small, *d_excl = sl.small + N, *d_len = sl.small + 2 * N, *d_status = sl.small + 3 * N,
             *d_index = sl.small + 4 * N, *d_stats = sl.small + 5 * N;
    if (queries) {
      if (s->ld != s->dim) {
        PH_HIP(hipMemsetAsync(sl.q, 0, (size_t)cnt * s->ld * 4, sl.stream));
        PH_HIP(hipMemcpy2DAsync(sl.q, (size_t)s->ld * 4, queries + c0 * s->dim, (size_t)s->dim * 4, (size_t)s->dim * 4, cnt,
                                hipMemcpyHostToDevice, sl.stream));
      } else {
        PH_HIP(hipMemcpyAsync(sl.q, queries + c0 * s->dim, (size_t)cnt * s->dim * 4, hipMemcpyHostToDevice, sl.stream));
      }
    }
    if (qids || exclude) {
      if (qids)
        for (uint64_t i = 0; i < cnt; i++) sl.h_small[i] = (uint32_t)qids[c0 + i];
      if (exclude)
        for (uint64_t i = 0; i < cnt; i++) sl.h_small[N + i] = exclude[c0 + i] >= s->n ? PH_EMPTY32 : (uint32_t)exclude[c0 + i];
      PH_HIP(hipMemcpyAsync(sl.small, sl.h_small, 2 * N * 4, hipMemcpyHostToDevice, sl.stream));
    }
    rc = ph_search_device(ix, queries ? sl.q : nullptr, s->ld, qids ? d_qid : nullptr, cnt, sp, upto, exclude ? d_excl : nullptr,
                          sl.ids, sl.d, d_len, d_stats, d_status, 0, knn_mode, sl.stream, 0, nullptr, 0.f, (uint32_t)c0, 0.f,
                          nullptr, out_index ? d_index : nullptr);
    if (rc) break;
    sl.c0 = c0;
    sl.cnt = cnt;
    sl.busy = true;
  }
  // the (up to two) chunks still in flight, oldest first
  if (n_chunks >= 2) {
    int r2 = finish(st->slot[n_chunks & 1]);
    if (!rc) rc = r2;
  }
  {
    int r2 = finish(st->slot[(n_chunks - 1) & 1]);
    if (!rc) rc = r2;
  }
  if (rc) {
    for (Slot &sl : st->slot) {
      if (sl.stream) hipStreamSynchronize(sl.stream);
      sl.busy = false;
    }
    return rc;
  }

  // queries whose frontier spill outgrew the workspace: rerun them alone with 8x the room (rare)
  uint32_t ovf_cap = ph_default_ovf_cap(ef);
  for (int attempt = 0; !redo.empty(); attempt++) {
    if (knn_mode) {
      ph_set_error("knn: frontier spill exceeded %u entries", ovf_cap);
      return PHNSW_E_OVERFLOW;
    }
    if (attempt == 3) {
      ph_set_error("search: frontier spill exceeded %u entries for %zu queries", ovf_cap, redo.size());
      return PHNSW_E_OVERFLOW;
    }
    ovf_cap *= 8;
    std::vector<uint32_t> todo;
    todo.swap(redo);
    Slot &sl = st->slot[0];
    for (uint32_t qi : todo) {
      PH_TRY(slot_ensure(sl, 1, s->ld, queries != nullptr, ef, k));
      const uint64_t N = 1;
      if (queries) {
        PH_HIP(hipMemsetAsync(sl.q, 0, (size_t)s->ld * 4, sl.stream));
        PH_HIP(hipMemcpyAsync(sl.q, queries + (uint64_t)qi * s->dim, (size_t)s->dim * 4, hipMemcpyHostToDevice, sl.stream));
      }
      sl.h_small[0] = qids ? (uint32_t)qids[qi] : 0u;
      sl.h_small[N] = (exclude && exclude[qi] < s->n) ? (uint32_t)exclude[qi] : PH_EMPTY32;
      PH_HIP(hipMemcpyAsync(sl.small, sl.h_small, 2 * N * 4, hipMemcpyHostToDevice, sl.stream));
      PH_TRY(ph_search_device(ix, queries ? sl.q : nullptr, s->ld, qids ? sl.small : nullptr, 1, sp, upto,
                              exclude ? sl.small + N : nullptr, sl.ids, sl.d, sl.small + 2 * N, sl.small + 5 * N, sl.small + 3 * N,
                              ovf_cap, 0, sl.stream, 0, nullptr, 0.f, 0, 0.f, nullptr, out_index ? sl.small + 4 * N : nullptr));
      sl.c0 = qi;
      sl.cnt = 1;
      sl.busy = true;
      PH_TRY(finish(sl));
    }
  }
  return 0;
}

// ------------------------------------------------------------------ C ABI

// Hnsw::search for a batch of AbstractVector::Unstored queries  lib.rs:663-665
extern "C" int phnsw_search_batch(const phnsw_index *ix, const float *queries, uint64_t nq,
                                  const phnsw_search_params *sp, uint32_t upto_layers, const uint64_t *exclude,
                                  uint64_t *out_ids, float *out_d, uint64_t *out_len, uint64_t *out_stats) try {
  if (!queries && nq) {
    ph_set_error("phnsw_search_batch: queries is NULL");
    return PHNSW_E_INVALID;
  }
  return ph_search_host(ix, queries, nullptr, nq, sp, upto_layers, exclude, 0, out_ids, out_d, out_len, out_stats, 0, nullptr);
} catch (...) { return ph_caught(); }

extern "C" int phnsw_search_batch_stored(const phnsw_index *ix, const uint64_t *qids, uint64_t nq,
                                         const phnsw_search_params *sp, uint32_t upto_layers, const uint64_t *exclude,
                                         uint64_t *out_ids, float *out_d, uint64_t *out_len, uint64_t *out_stats) try {
  if (!qids && nq) {
    ph_set_error("phnsw_search_batch_stored: qids is NULL");
    return PHNSW_E_INVALID;
  }
  return ph_search_host(ix, nullptr, qids, nq, sp, upto_layers, exclude, 0, out_ids, out_d, out_len, out_stats, 0, nullptr);
} catch (...) { return ph_caught(); }

// the same keeping only the best k results of every query: the truncation callers of Hnsw::search do themselves
// (lib.rs:1118 takes neighborhood_size of them, a k-NN service takes k) happens before the transfer
extern "C" int phnsw_search_batch_topk(const phnsw_index *ix, const float *queries, const uint64_t *qids, uint64_t nq,
                                       const phnsw_search_params *sp, uint32_t upto_layers, const uint64_t *exclude,
                                       uint64_t k, uint64_t *out_ids, float *out_d, uint64_t *out_len) try {
  if (((!queries) == (!qids)) && nq) {
    ph_set_error("phnsw_search_batch_topk: pass queries or qids (exactly one)");
    return PHNSW_E_INVALID;
  }
  if (k == 0) {
    ph_set_error("phnsw_search_batch_topk: k must be 1..number_of_candidates");
    return PHNSW_E_INVALID;
  }
  return ph_search_host(ix, queries, qids, nq, sp, upto_layers, exclude, k, out_ids, out_d, out_len, nullptr, 0, nullptr);
} catch (...) { return ph_caught(); }

// Hnsw::search_instrumented  lib.rs:667-673: results + the index_distance of search_layers_instrumented
// (search.rs:93-140; usize::MAX = UINT64_MAX when no layer ran)
extern "C" int phnsw_search_instrumented(const phnsw_index *ix, const float *queries, const uint64_t *qids, uint64_t nq,
                                         const phnsw_search_params *sp, uint64_t *out_ids, float *out_d,
                                         uint64_t *out_len, uint64_t *out_index_distance) try {
  if ((((!queries) == (!qids)) && nq) || !out_index_distance) {
    ph_set_error("phnsw_search_instrumented: pass queries or qids (exactly one) and out_index_distance");
    return PHNSW_E_INVALID;
  }
  return ph_search_host(ix, queries, qids, nq, sp, 0, nullptr, 0, out_ids, out_d, out_len, nullptr, 0, out_index_distance);
} catch (...) { return ph_caught(); }
