// Index construction sharded over the GPUs of one node (BASELINE config 4): the driver.
//
// Reference semantics: every build round of the crate is "each node of a layer runs a search against a
// snapshot, then proposes edges" (/root/reference/src/lib.rs:1097-1153; the seeding steps of
// generate_layer lib.rs:700-787 likewise), so nodes are independent within a round.  One process per GPU
// holds a replica of store and graph;
//
//     rank r runs the phase for the node range [r*chunk, (r+1)*chunk)      (K2 / K3, build.hip)
//     all-gather of the per-node results   (u32 ids + f32 distances, n x M x 8 bytes per link round)
//     every rank applies ALL results to its replica                        (K5, deterministic)
//
// so the replicas stay bit-identical and the only data-path collective is one all-gather per phase (plus
// an all-reduce of one integer for the recall estimate).  A rank's share of a long phase is cut into
// pieces: the all-gather of piece k is enqueued on the communicator's stream and travels over xGMI while
// piece k+1 is being searched.  Control flow: generate lib.rs:825-893, improve_neighbors_upto :1515-1544,
// improve_index_at :1546-1603, improve_index :1664-1686 -- the same as build.hip's single-GPU loops.
//
// The driver is written against phnsw_shard_engine (include/phnsw.h): libphnsw's own phases are the engine
// of phnsw_build_sharded; the CPU tests plug the oracle's phases in under gloo, so the split, the block
// layout, the pipeline and the reassembly below are what both run.
//
// Block layout of one piece (pc items per rank, padded to the same pc on every rank):
//     [ids: pc x w0 x id_bytes][d: pc x w1 x 4][len: pc x id_bytes] ...      each array 256-byte aligned
// A phase writes its outputs straight into the send block (no packing); after the all-gather, array a of
// piece [lo, hi) lands in the full array with ONE 2-D copy: `world` rows of pc*w*elem bytes, source pitch =
// block bytes, destination pitch = chunk*w*elem.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "phnsw_internal.h"

#define PH_TRY(x)          \
  do {                     \
    int rc__ = (x);        \
    if (rc__) return rc__; \
  } while (0)

namespace {

uint64_t g_shard_min = 4096;  // shorter work lists run whole on every rank: a launch over a few thousand queries
                              // takes one query latency however few of them a rank keeps
uint32_t g_subchunks = 4;
// A piece is a launch of its own, and a launch costs ~1.6 ms beyond its queries (ramp-up and tail of the persistent
// grid, probe_tail.py): pieces pay off only where the transfer they hide is longer than that -- shares of a quarter
// of a million nodes and more (a layer of 2M nodes at 8 ranks).  At 1M x 768 the whole build gathers 1.5 GB per rank.
uint64_t g_sub_min = 65536;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Spec {
  uint32_t width;  // elements per item
  uint32_t elem;   // bytes per element
};

struct Driver {
  const phnsw_shard_engine *e;
  phnsw_comm comm;  // a copy; world == 1 when the caller passed none
  phnsw_build_params bp;
  phnsw_sharded_stats st;
  hipStream_t stream = nullptr;  // device-buffer comms: the collectives' stream
  void *stage_send = nullptr, *stage_recv = nullptr;  // device engine + host comm: pinned staging
  size_t stage_send_bytes = 0, stage_recv_bytes = 0;
  std::vector<void *> owned;

  Driver(const phnsw_shard_engine *e_, const phnsw_comm *c, const phnsw_build_params *bp_) : e(e_), bp(*bp_) {
    memset(&st, 0, sizeof(st));
    if (c)
      comm = *c;
    else {
      memset(&comm, 0, sizeof(comm));
      comm.world = 1;
    }
  }
  ~Driver() {
    for (void *p : owned) e->release(e->ctx, p);
    if (stage_send) hipHostFree(stage_send);
    if (stage_recv) hipHostFree(stage_recv);
    if (stream) hipStreamDestroy(stream);
  }

  int check() {
    if (comm.world == 0 || comm.rank >= comm.world) {
      ph_set_error("sharded build: rank %u outside world %u", comm.rank, comm.world);
      return PHNSW_E_INVALID;
    }
    if (comm.world > 1) {
      if (!comm.all_gather && !comm.emulate) {
        ph_set_error("sharded build: a world of %u needs all_gather (or emulate)", comm.world);
        return PHNSW_E_INVALID;
      }
      if (!comm.emulate && !comm.all_reduce_sum) {
        ph_set_error("sharded build: all_reduce_sum missing");
        return PHNSW_E_INVALID;
      }
      if (comm.all_gather && e->host_buffers && !comm.host_buffers) {
        ph_set_error("sharded build: an engine with host buffers needs a communicator with host_buffers = 1");
        return PHNSW_E_INVALID;
      }
    }
    if (e->id_bytes != 4 && e->id_bytes != 8) {
      ph_set_error("sharded build: id_bytes must be 4 or 8");
      return PHNSW_E_INVALID;
    }
    return 0;
  }

  void *alloc(uint64_t bytes) {
    void *p = e->alloc(e->ctx, std::max<uint64_t>(bytes, 16));
    if (p) owned.push_back(p);
    return p;
  }
  void release(void *p) {
    if (!p) return;
    auto it = std::find(owned.begin(), owned.end(), p);
    if (it != owned.end()) owned.erase(it);
    e->release(e->ctx, p);
  }

  // rank r's share of a list of n items: (chunk, first, count)
  void range(uint64_t n, uint32_t r, uint64_t *chunk, uint64_t *first, uint64_t *count) const {
    if (n < g_shard_min || comm.world == 1) {
      *chunk = n, *first = 0, *count = n;
      return;
    }
    uint64_t c = (n + comm.world - 1) / comm.world;
    uint64_t f = std::min<uint64_t>(n, (uint64_t)r * c);
    *chunk = c, *first = f, *count = std::min<uint64_t>(n, f + c) - f;
  }
  bool whole(uint64_t n) const { return n < g_shard_min || comm.world == 1; }

  int gather(const void *send, void *recv, uint64_t bytes) {
    double t0 = now_s();
    int rc = 0;
    if (!e->host_buffers && comm.host_buffers) {  // device engine, host transport: stage through pinned memory
      if (stage_send_bytes < bytes) {
        if (stage_send) hipHostFree(stage_send);
        stage_send = nullptr;
        PH_HIP(hipHostMalloc(&stage_send, bytes, hipHostMallocDefault));
        stage_send_bytes = bytes;
      }
      if (stage_recv_bytes < bytes * comm.world) {
        if (stage_recv) hipHostFree(stage_recv);
        stage_recv = nullptr;
        PH_HIP(hipHostMalloc(&stage_recv, bytes * comm.world, hipHostMallocDefault));
        stage_recv_bytes = bytes * comm.world;
      }
      PH_HIP(hipMemcpy(stage_send, send, bytes, hipMemcpyDeviceToHost));
      rc = comm.all_gather(comm.ctx, stage_send, stage_recv, bytes, nullptr);
      if (!rc) PH_HIP(hipMemcpy(recv, stage_recv, bytes * comm.world, hipMemcpyHostToDevice));
    } else if (comm.host_buffers) {
      rc = comm.all_gather(comm.ctx, send, recv, bytes, nullptr);
    } else {
      if (!stream) PH_TRY(ph_stream_beside(nullptr, &stream));  // beside the engine's launches (the null stream), not behind them
      // the phase that filled `send` has synchronised the device (every phase of the engine ends with the
      // host reading its status words), so the transfer may start at once on the collectives' stream
      rc = comm.all_gather(comm.ctx, send, recv, bytes, stream);
    }
    st.seconds_comm += now_s() - t0;
    st.all_gather_bytes += bytes * comm.world;
    st.all_gather_calls++;
    if (rc > 0) {  // a host callback's own code; the library's transports have set their message
      ph_set_error("sharded build: all_gather failed with %d", rc);
      return PHNSW_E_INVALID;
    }
    return rc;
  }
  int gather_wait() {
    if (stream) {
      double t0 = now_s();
      PH_HIP(hipStreamSynchronize(stream));
      st.seconds_comm += now_s() - t0;
    }
    return 0;
  }

  using Run = std::function<int(uint64_t first, uint64_t count, void *const *outs)>;

  // One sharded phase over a list of n work items.  run(first, count, outs) fills rows [0, count) of the
  // given arrays with the results of items [first, first + count).  full[a] receives an engine buffer of
  // >= n items of array a, identical on every rank (owned by the driver until release()).
  int phase(int kind, uint64_t n, const std::vector<Spec> &specs, const Run &run, std::vector<void *> &full) {
    const size_t A = specs.size();
    full.assign(A, nullptr);
    st.phases++;
    if (whole(n)) {
      st.phases_whole++;
      for (size_t a = 0; a < A; a++) {
        full[a] = alloc(n * specs[a].width * specs[a].elem);
        if (!full[a]) return PHNSW_E_NOMEM;
      }
      double t0 = now_s();
      int rc = run(0, n, full.data());
      // every rank repeats a whole list; it counts as this rank's share of the phase all the same
      st.seconds_sharded += now_s() - t0;
      st.seconds_by_phase[kind] += now_s() - t0;
      return rc;
    }
    uint64_t chunk, first, count;
    range(n, comm.rank, &chunk, &first, &count);
    const bool overlap = comm.all_gather && !comm.host_buffers;  // only a stream-ordered transport overlaps anything
    uint32_t nsub = ((overlap || comm.emulate) && chunk >= (uint64_t)g_subchunks * g_sub_min) ? g_subchunks : 1;
    for (size_t a = 0; a < A; a++) {
      full[a] = alloc((uint64_t)comm.world * chunk * specs[a].width * specs[a].elem);
      if (!full[a]) return PHNSW_E_NOMEM;
    }
    struct Piece {
      uint64_t lo, hi, block;
      void *send, *recv;
      std::vector<uint64_t> off;
    };
    std::vector<Piece> pieces(nsub);
    for (uint32_t k = 0; k < nsub; k++) {
      Piece &p = pieces[k];
      p.lo = chunk * k / nsub, p.hi = chunk * (k + 1) / nsub;
      const uint64_t pc = p.hi - p.lo;
      p.off.resize(A);
      uint64_t at = 0;
      for (size_t a = 0; a < A; a++) {
        p.off[a] = at;
        at += (pc * specs[a].width * specs[a].elem + 255) & ~(uint64_t)255;
      }
      p.block = at;
      p.recv = alloc(p.block * comm.world);
      p.send = comm.emulate && !comm.all_gather ? nullptr : alloc(p.block);
      if (!p.recv || (!p.send && comm.all_gather)) return PHNSW_E_NOMEM;
      std::vector<void *> outs(A);
      if (comm.all_gather) {
        const uint64_t cnt = count > p.lo ? std::min(count, p.hi) - p.lo : 0;
        if (cnt) {
          for (size_t a = 0; a < A; a++) outs[a] = (char *)p.send + p.off[a];
          double t0 = now_s();
          int rc = run(first + p.lo, cnt, outs.data());
          st.seconds_sharded += now_s() - t0;
          st.seconds_by_phase[kind] += now_s() - t0;
          if (rc) return rc;
        }
        PH_TRY(gather(p.send, p.recv, p.block));
      } else {  // emulate: this process plays every rank in turn, straight into the receive blocks
        for (uint32_t r = 0; r < comm.world; r++) {
          uint64_t c2, f2, n2;
          range(n, r, &c2, &f2, &n2);
          const uint64_t cnt = n2 > p.lo ? std::min(n2, p.hi) - p.lo : 0;
          if (!cnt) continue;
          for (size_t a = 0; a < A; a++) outs[a] = (char *)p.recv + (uint64_t)r * p.block + p.off[a];
          double t0 = now_s();
          int rc = run(f2 + p.lo, cnt, outs.data());
          (r == comm.rank ? st.seconds_sharded : st.seconds_others) += now_s() - t0;
          if (r == comm.rank) st.seconds_by_phase[kind] += now_s() - t0;
          if (rc) return rc;
        }
        st.all_gather_bytes += p.block * comm.world;
        st.all_gather_calls++;
      }
    }
    PH_TRY(gather_wait());
    double t0 = now_s();
    for (Piece &p : pieces) {
      const uint64_t pc = p.hi - p.lo;
      for (size_t a = 0; a < A; a++) {
        const uint64_t item = (uint64_t)specs[a].width * specs[a].elem;
        int rc = e->copy2d(e->ctx, (char *)full[a] + p.lo * item, chunk * item, (const char *)p.recv + p.off[a], p.block,
                           pc * item, comm.world);
        if (rc) return rc;
      }
    }
    for (Piece &p : pieces) {
      release(p.send);
      release(p.recv);
    }
    st.seconds_comm += now_s() - t0;
    return 0;
  }

  template <class F>
  int replicated(int kind, F &&f) {
    double t0 = now_s();
    int rc = f();
    st.seconds_replicated += now_s() - t0;
    st.seconds_by_phase[kind] += now_s() - t0;
    return rc;
  }

  // generate_layer  lib.rs:675-823
  int generate_layer(const uint64_t *vids, uint64_t n, uint64_t W) {
    int needs = 0;
    uint32_t K = 0;
    PH_TRY(replicated(6, [&] { return e->layer_begin(e->ctx, vids, n, W, &needs, &K); }));
    if (!needs) return 0;
    const uint32_t ib = e->id_bytes;
    if ((needs & 2) && e->layer_cells && e->layer_set_cells) {  // the cells of the locality schedule, range by range
      std::vector<void *> pos;
      PH_TRY(phase(6, n, {{1, 4}}, [&](uint64_t f, uint64_t c, void *const *o) { return e->layer_cells(e->ctx, f, c, o[0]); }, pos));
      PH_TRY(replicated(6, [&] { return e->layer_set_cells(e->ctx, pos[0]); }));
      release(pos[0]);
    }
    std::vector<void *> init, rows;
    PH_TRY(phase(0, n, {{K, ib}, {K, 4}, {1, ib}},
                 [&](uint64_t f, uint64_t c, void *const *o) {
                   return e->layer_init_search(e->ctx, f, c, o[0], (float *)o[1], o[2]);
                 },
                 init));
    PH_TRY(phase(1, n, {{(uint32_t)W, ib}, {(uint32_t)W, 4}},
                 [&](uint64_t f, uint64_t c, void *const *o) {
                   return e->layer_seed(e->ctx, init[0], (const float *)init[1], init[2], f, c, o[0], (float *)o[1]);
                 },
                 rows));
    PH_TRY(replicated(7, [&] { return e->layer_finish(e->ctx, rows[0], (const float *)rows[1]); }));
    for (void *p : init) release(p);
    for (void *p : rows) release(p);
    return 0;
  }

  // link_layer_to_better_neighbors  lib.rs:1070-1154
  int link_layer(uint32_t lft, const phnsw_search_params *sp, uint64_t M) {
    const uint64_t n = e->layer_nodes(e->ctx, lft);
    const uint32_t ib = e->id_bytes;
    std::vector<void *> res;
    PH_TRY(phase(2, n, {{(uint32_t)M, ib}, {(uint32_t)M, 4}, {1, ib}},
                 [&](uint64_t f, uint64_t c, void *const *o) {
                   return e->link_search(e->ctx, lft, sp, M, f, c, o[0], (float *)o[1], o[2]);
                 },
                 res));
    uint64_t added = 0;
    PH_TRY(replicated(8, [&] { return e->link_apply(e->ctx, lft, M, res[0], (const float *)res[1], res[2], &added); }));
    for (void *p : res) release(p);
    return 0;
  }

  // stochastic_recall_at  lib.rs:1463-1499
  int recall_at(uint32_t at, float *out) {
    const phnsw_optimization_params *op = &bp.optimization;
    const uint64_t total = e->layer_nodes(e->ctx, at);
    uint64_t selection = (uint64_t)((float)total * op->recall_proportion);
    selection = std::min<uint64_t>(std::max<uint64_t>(selection, 1), total);
    uint64_t hits = 0;
    st.phases++;
    if (whole(selection)) st.phases_whole++;
    const uint32_t ranks = (comm.emulate && !comm.all_gather && !whole(selection)) ? comm.world : 1;
    for (uint32_t k = 0; k < ranks; k++) {
      const uint32_t r = ranks > 1 ? k : comm.rank;
      uint64_t chunk, first, count, h = 0, sel = 0;
      range(selection, r, &chunk, &first, &count);
      double t0 = now_s();
      PH_TRY(e->recall_hits(e->ctx, at, op, first, count, &h, &sel));
      (r == comm.rank ? st.seconds_sharded : st.seconds_others) += now_s() - t0;
      if (r == comm.rank) st.seconds_by_phase[3] += now_s() - t0;
      if (sel != selection) {
        ph_set_error("sharded build: the engine samples %llu vectors, the driver %llu", (unsigned long long)sel,
                     (unsigned long long)selection);
        return PHNSW_E_INVALID;
      }
      hits += h;
    }
    if (!whole(selection) && comm.all_gather) {
      double t0 = now_s();
      int rc = comm.all_reduce_sum(comm.ctx, &hits, 1);
      st.seconds_comm += now_s() - t0;
      st.all_reduce_calls++;
      if (rc) {
        ph_set_error("sharded build: all_reduce_sum failed with %d", rc);
        return rc < 0 ? rc : PHNSW_E_INVALID;
      }
    } else if (ranks > 1)
      st.all_reduce_calls++;
    *out = (float)hits / (float)selection;
    return 0;
  }

  // improve_neighbors_upto  lib.rs:1515-1544
  int improve_neighbors_upto(uint32_t upto, float last_recall, float *out) {
    const phnsw_optimization_params *op = &bp.optimization;
    float last = (last_recall != last_recall) ? 0.0f : last_recall;
    float improvement = 1.0f;
    uint64_t rounds = 0;
    while (improvement >= op->neighborhood_threshold && last < 1.0f) {
      for (uint32_t lft = 0; lft < upto; lft++) PH_TRY(link_layer(lft, &op->search, bp.neighborhood_size));
      float recall = 0.f;
      PH_TRY(recall_at(upto - 1, &recall));
      improvement = recall - last;
      last = recall;
      rounds++;
      if (bp.max_link_rounds && rounds >= bp.max_link_rounds) break;
    }
    *out = last;
    return 0;
  }

  // promote_at_layer  lib.rs:1273-1427: its n searches (discover_unreachable_vectors) are sharded like a
  // link round, the (integer, sequential) promotion itself runs replicated
  int promote_at_layer(uint32_t lft, int *promoted) {
    const uint64_t n = e->layer_nodes(e->ctx, lft);
    std::vector<void *> hit;
    PH_TRY(phase(4, n, {{1, e->id_bytes}},
                 [&](uint64_t f, uint64_t c, void *const *o) {
                   return e->discover_hits(e->ctx, lft, &bp.optimization.search, f, c, o[0]);
                 },
                 hit));
    PH_TRY(replicated(9, [&] { return e->promote_from_hits(e->ctx, lft, hit[0], promoted); }));
    release(hit[0]);
    return 0;
  }

  // improve_index_at  lib.rs:1546-1603; *lft_io grows when promotion adds layers
  int improve_index_at(uint32_t *lft_io, float *out) {
    const phnsw_optimization_params *op = &bp.optimization;
    uint32_t lft = *lft_io;
    float recall = 0.f;
    PH_TRY(recall_at(lft, &recall));
    float improvement = 1.0f;
    int bailout = 1;
    while (improvement >= op->promotion_threshold && recall < 1.0f && bailout != 0) {
      const float last = recall;
      uint32_t cur = 0;
      while (cur <= lft && bailout != 0) {
        const uint32_t layer_count = e->layer_count(e->ctx);
        PH_TRY(improve_neighbors_upto(cur + 1, NAN, &recall));
        if (recall == 1.0f) {  // :1569-1572
          cur++;
          continue;
        }
        if (bp.promote) {
          int promoted = 0;
          PH_TRY(promote_at_layer(cur, &promoted));  // :1575
          if (promoted) {
            const uint32_t delta = e->layer_count(e->ctx) - layer_count;
            cur += delta;
            lft += delta;
            PH_TRY(improve_neighbors_upto(cur + 1, recall, &recall));  // :1586-1587
          }
        }
        cur++;
      }
      bailout--;
      improvement = recall - last;
    }
    *lft_io = lft;
    *out = recall;
    return 0;
  }

  // improve_index  lib.rs:1664-1686
  int improve_index(float last_recall, float *out) {
    if (e->layer_count(e->ctx) == 0) {
      ph_set_error("improve_index: index has no layers");
      return PHNSW_E_INVALID;
    }
    float recall = last_recall;
    if (last_recall != last_recall) PH_TRY(recall_at(e->layer_count(e->ctx) - 1, &recall));
    uint32_t lft = 0;
    while (lft < e->layer_count(e->ctx)) {
      PH_TRY(improve_index_at(&lft, &recall));
      lft++;
    }
    if (out) *out = recall;
    return 0;
  }

  // Hnsw::generate  lib.rs:825-893
  int generate(const uint64_t *vids, uint64_t n, phnsw_progress_cb cb, void *user) {
    std::vector<uint64_t> vs(n), sizes(PH_MAX_LAYERS + 8);
    uint32_t cnt = 0;
    PH_TRY(replicated(5, [&] { return e->plan(e->ctx, vids, n, vs.data(), sizes.data(), (uint32_t)sizes.size(), &cnt); }));
    sizes.resize(cnt);
    size_t i = 0;
    while (i != sizes.size()) {  // lib.rs:854-890
      const uint64_t length = std::min<uint64_t>(sizes[i], n);
      const size_t level = sizes.size() - i - 1;
      const uint64_t W = level == 0 ? bp.zero_layer_neighborhood_size : bp.neighborhood_size;
      PH_TRY(generate_layer(vs.data(), length, W));
      const uint32_t old_count = e->layer_count(e->ctx);
      PH_TRY(improve_index(NAN, nullptr));  // improve_index(bp, None, progress)  lib.rs:877
      if (cb && cb(user, "generate", i + 1, sizes.size())) {
        ph_set_error("interrupted by the progress callback");
        return PHNSW_E_INVALID;
      }
      const uint32_t delta = e->layer_count(e->ctx) - old_count;
      if (delta > 0) {  // promotion added layers: fix the partitions  lib.rs:880-887
        std::vector<uint64_t> np;
        for (uint32_t l = 0; l < e->layer_count(e->ctx); l++) np.push_back(e->layer_nodes(e->ctx, l));
        np.insert(np.end(), sizes.begin() + i + 1, sizes.end());
        sizes = np;
        i += delta;
      }
      i++;
    }
    return 0;
  }
};

// ------------------------------------------------------------------ libphnsw's own phases as the engine

struct GpuCtx {
  phnsw_index *ix;
  phnsw_build_params bp;
};

void *gpu_alloc(void *, uint64_t bytes) {
  void *p = nullptr;
  if (ph_pool_alloc(&p, bytes) != hipSuccess) {
    ph_set_error("sharded build: device allocation of %llu bytes failed", (unsigned long long)bytes);
    return nullptr;
  }
  return p;
}
void gpu_release(void *, void *p) { ph_pool_free(p); }
// `height` runs of `width` bytes from pitched source to pitched destination (everything a multiple of 4 bytes;
// pitches of hundreds of megabytes at 10M nodes, beyond what hipMemcpy2D takes)
__global__ void ph_copy_rows_kernel(uint32_t *dst, uint64_t dpitch_w, const uint32_t *src, uint64_t spitch_w,
                                    uint64_t width_w, uint32_t height) {
  const uint64_t total = width_w * height;
  for (uint64_t x = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; x < total; x += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t r = x / width_w, c = x - r * width_w;
    dst[r * dpitch_w + c] = src[r * spitch_w + c];
  }
}
int gpu_copy2d(void *, void *dst, uint64_t dpitch, const void *src, uint64_t spitch, uint64_t width, uint64_t height) {
  if (((uintptr_t)dst | (uintptr_t)src | dpitch | spitch | width) & 3) {
    ph_set_error("sharded build: unaligned reassembly copy");
    return PHNSW_E_INVALID;
  }
  if (!width || !height) return 0;
  const uint64_t total = width / 4 * height;
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 256u * 16u);
  hipLaunchKernelGGL(ph_copy_rows_kernel, dim3(blocks), dim3(256), 0, 0, (uint32_t *)dst, dpitch / 4, (const uint32_t *)src,
                     spitch / 4, width / 4, (uint32_t)height);
  PH_HIP(hipGetLastError());
  return 0;
}
int gpu_plan(void *c, const uint64_t *vids, uint64_t n, uint64_t *sh, uint64_t *sizes, uint32_t max_layers, uint32_t *cnt) {
  return phnsw_build_plan(vids, n, &((GpuCtx *)c)->bp, sh, sizes, max_layers, cnt);
}
int gpu_layer_begin(void *c, const uint64_t *vids, uint64_t n, uint64_t W, int *needs, uint32_t *K) {
  GpuCtx *g = (GpuCtx *)c;
  *K = (uint32_t)g->bp.initial_partition_search.number_of_candidates;
  int cells = 0;
  int rc = phnsw_layer_begin_sharded(g->ix, vids, n, W, &g->bp, needs, &cells);
  if (!rc && *needs && cells) *needs |= 2;
  return rc;
}
int gpu_layer_cells(void *c, uint64_t first, uint64_t count, void *pos) {
  return phnsw_layer_cells_device(((GpuCtx *)c)->ix, first, count, (uint32_t *)pos);
}
int gpu_layer_set_cells(void *c, const void *pos) { return phnsw_layer_set_cells_device(((GpuCtx *)c)->ix, (const uint32_t *)pos); }
int gpu_layer_init_search(void *c, uint64_t first, uint64_t count, void *ids, float *d, void *len) {
  GpuCtx *g = (GpuCtx *)c;
  return phnsw_layer_init_search_device(g->ix, &g->bp, first, count, (uint32_t *)ids, d, (uint32_t *)len);
}
int gpu_layer_seed(void *c, const void *ii, const float *id, const void *il, uint64_t first, uint64_t count, void *rows,
                   float *rows_d) {
  GpuCtx *g = (GpuCtx *)c;
  return phnsw_layer_seed_device(g->ix, &g->bp, (const uint32_t *)ii, id, (const uint32_t *)il, first, count,
                                 (uint32_t *)rows, rows_d);
}
int gpu_layer_finish(void *c, const void *rows, const float *rows_d) {
  return phnsw_layer_finish_device(((GpuCtx *)c)->ix, (const uint32_t *)rows, rows_d);
}
uint32_t gpu_layer_count(void *c) { return phnsw_index_layer_count(((GpuCtx *)c)->ix); }
uint64_t gpu_layer_nodes(void *c, uint32_t lft) {
  uint64_t n = 0;
  phnsw_index_layer_info(((GpuCtx *)c)->ix, lft, &n, nullptr);
  return n;
}
int gpu_link_search(void *c, uint32_t lft, const phnsw_search_params *sp, uint64_t M, uint64_t first, uint64_t count,
                    void *ids, float *d, void *len) {
  return phnsw_link_search_device(((GpuCtx *)c)->ix, lft, sp, M, first, count, (uint32_t *)ids, d, (uint32_t *)len);
}
int gpu_link_apply(void *c, uint32_t lft, uint64_t M, const void *ids, const float *d, const void *len, uint64_t *added) {
  return phnsw_link_apply_device(((GpuCtx *)c)->ix, lft, M, (const uint32_t *)ids, d, (const uint32_t *)len, added);
}
int gpu_recall_hits(void *c, uint32_t at, const phnsw_optimization_params *op, uint64_t first, uint64_t count,
                    uint64_t *hits, uint64_t *sel) {
  return phnsw_recall_hits(((GpuCtx *)c)->ix, at, op, first, count, hits, sel);
}
int gpu_discover_hits(void *c, uint32_t lft, const phnsw_search_params *sp, uint64_t first, uint64_t count, void *hit) {
  return phnsw_discover_hits_device(((GpuCtx *)c)->ix, lft, sp, first, count, (uint32_t *)hit);
}
int gpu_promote_from_hits(void *c, uint32_t lft, const void *hit, int *promoted) {
  GpuCtx *g = (GpuCtx *)c;
  return phnsw_promote_at_layer_hits_device(g->ix, lft, &g->bp, (const uint32_t *)hit, promoted);
}

phnsw_shard_engine gpu_engine(GpuCtx *g) {
  phnsw_shard_engine e;
  memset(&e, 0, sizeof(e));
  e.ctx = g;
  e.id_bytes = 4;
  e.host_buffers = 0;
  e.alloc = gpu_alloc;
  e.release = gpu_release;
  e.copy2d = gpu_copy2d;
  e.plan = gpu_plan;
  e.layer_begin = gpu_layer_begin;
  e.layer_init_search = gpu_layer_init_search;
  e.layer_seed = gpu_layer_seed;
  e.layer_finish = gpu_layer_finish;
  e.layer_count = gpu_layer_count;
  e.layer_nodes = gpu_layer_nodes;
  e.link_search = gpu_link_search;
  e.link_apply = gpu_link_apply;
  e.recall_hits = gpu_recall_hits;
  e.discover_hits = gpu_discover_hits;
  e.promote_from_hits = gpu_promote_from_hits;
  e.layer_cells = gpu_layer_cells;
  e.layer_set_cells = gpu_layer_set_cells;
  return e;
}

bool engine_complete(const phnsw_shard_engine *e) {
  return e && e->alloc && e->release && e->copy2d && e->plan && e->layer_begin && e->layer_init_search && e->layer_seed &&
         e->layer_finish && e->layer_count && e->layer_nodes && e->link_search && e->link_apply && e->recall_hits &&
         e->discover_hits && e->promote_from_hits;
}

}  // namespace

// ------------------------------------------------------------------ shared with pq.hip (sharded encode)

// rank r's share of n items under `comm` (every list is split, however short: callers decide)
void ph_comm_range(const phnsw_comm *comm, uint32_t r, uint64_t n, uint64_t *chunk, uint64_t *first, uint64_t *count) {
  const uint32_t w = comm ? std::max<uint32_t>(comm->world, 1) : 1;
  const uint64_t c = (n + w - 1) / w, f = std::min<uint64_t>(n, (uint64_t)r * c);
  *chunk = c, *first = f, *count = std::min<uint64_t>(n, f + c) - f;
}

// synchronous all-gather of device blocks through any transport of phnsw_comm (not for emulated worlds)
int ph_comm_all_gather_device(const phnsw_comm *comm, const void *send_dev, void *recv_dev, uint64_t bytes) {
  if (!comm || !comm->all_gather) {
    ph_set_error("all_gather: no transport");
    return PHNSW_E_INVALID;
  }
  int rc = 0;
  if (comm->host_buffers) {
    void *hs = nullptr, *hr = nullptr;
    PH_HIP(hipHostMalloc(&hs, bytes, hipHostMallocDefault));
    hipError_t e = hipHostMalloc(&hr, bytes * comm->world, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemcpy(hs, send_dev, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) rc = comm->all_gather(comm->ctx, hs, hr, bytes, nullptr);
    if (e == hipSuccess && !rc) e = hipMemcpy(recv_dev, hr, bytes * comm->world, hipMemcpyHostToDevice);
    hipHostFree(hs);
    if (hr) hipHostFree(hr);
    if (e != hipSuccess) return ph_hip_fail(e, "all_gather staging", __FILE__, __LINE__);
  } else {
    hipStream_t st = nullptr;
    PH_HIP(hipDeviceSynchronize());
    PH_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    rc = comm->all_gather(comm->ctx, send_dev, recv_dev, bytes, st);
    hipError_t e = hipStreamSynchronize(st);
    hipStreamDestroy(st);
    if (e != hipSuccess) return ph_hip_fail(e, "all_gather", __FILE__, __LINE__);
  }
  if (rc > 0) {
    ph_set_error("all_gather failed with %d", rc);
    return PHNSW_E_INVALID;
  }
  return rc;
}

// ------------------------------------------------------------------ C ABI

extern "C" int phnsw_sharded_tuning(uint64_t shard_min, uint32_t subchunks, uint64_t sub_min) try {
  if (shard_min) g_shard_min = shard_min;
  if (subchunks) g_subchunks = std::min<uint32_t>(subchunks, 64);
  if (sub_min) g_sub_min = sub_min;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_build_sharded_engine(const phnsw_shard_engine *e, const uint64_t *vids, uint64_t n,
                                          const phnsw_build_params *bp, const phnsw_comm *comm,
                                          phnsw_sharded_stats *stats) try {
  if (!engine_complete(e) || !vids || !bp || n == 0 || bp->order < 2) {
    ph_set_error("phnsw_build_sharded_engine: invalid argument (a complete engine, n > 0, order >= 2)");
    return PHNSW_E_INVALID;
  }
  Driver d(e, comm, bp);
  PH_TRY(d.check());
  double t0 = now_s();
  int rc = d.generate(vids, n, nullptr, nullptr);
  d.st.seconds_total = now_s() - t0;
  if (stats) *stats = d.st;
  return rc;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_build_sharded(phnsw_store *s, const uint64_t *vids, uint64_t n, const phnsw_build_params *bp,
                                   const phnsw_comm *comm, phnsw_progress_cb cb, void *user, phnsw_index **out,
                                   phnsw_sharded_stats *stats) try {
  if (!s || !vids || !bp || !out || n == 0 || bp->order < 2) {
    ph_set_error("phnsw_build_sharded: invalid argument (need n > 0, order >= 2)");
    return PHNSW_E_INVALID;
  }
  if (!comm || comm->world <= 1) {
    double t0 = now_s();
    int rc = phnsw_build(s, vids, n, bp, cb, user, out);
    if (stats) {
      memset(stats, 0, sizeof(*stats));
      stats->seconds_total = now_s() - t0;
    }
    return rc;
  }
  GpuCtx g;
  g.bp = *bp;
  g.ix = nullptr;
  PH_TRY(phnsw_index_create(s, bp, &g.ix));
  g.ix->bt_enabled = true;  // the dense table of a layer's rounds is kept for this rank's node range (tiny.hip)
  phnsw_shard_engine e = gpu_engine(&g);
  int rc;
  {
    Driver d(&e, comm, bp);
    rc = d.check();
    double t0 = now_s();
    if (!rc) rc = d.generate(vids, n, cb, user);
    hipDeviceSynchronize();
    d.st.seconds_total = now_s() - t0;
    if (stats) *stats = d.st;
  }
  g.ix->bt_enabled = false;
  ph_build_table_free(g.ix);
  ph_pool_trim();
  if (rc) {
    phnsw_index_destroy(g.ix);
    return rc;
  }
  *out = g.ix;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_improve_index_sharded(phnsw_index *ix, const phnsw_build_params *bp, float last_recall,
                                           const phnsw_comm *comm, float *out_recall, phnsw_sharded_stats *stats) try {
  if (!ix || !bp) {
    ph_set_error("phnsw_improve_index_sharded: null index or parameters");
    return PHNSW_E_INVALID;
  }
  if (!comm || comm->world <= 1) return phnsw_improve_index(ix, bp, last_recall, nullptr, nullptr, out_recall);
  GpuCtx g;
  g.bp = *bp;
  g.ix = ix;
  phnsw_shard_engine e = gpu_engine(&g);
  Driver d(&e, comm, bp);
  PH_TRY(d.check());
  double t0 = now_s();
  ix->bt_enabled = true;
  int rc = d.improve_index(last_recall, out_recall);
  ix->bt_enabled = false;
  ph_build_table_free(ix);
  d.st.seconds_total = now_s() - t0;
  if (stats) *stats = d.st;
  return rc;
} catch (...) { return ph_caught(); }

// ------------------------------------------------------------------ the built-in transport: RCCL over xGMI
//
// librccl is half a gigabyte and only a multi-GPU build needs it, so it is loaded on first use, not linked.
// In a process that already holds an RCCL (e.g. one that imported torch) that copy is used.

namespace {

typedef struct {
  char internal[128];
} PhNcclUniqueId;
typedef void *PhNcclComm;

struct Rccl {
  void *h = nullptr;
  int (*GetUniqueId)(PhNcclUniqueId *) = nullptr;
  int (*CommInitRank)(PhNcclComm *, int, PhNcclUniqueId, int) = nullptr;
  int (*CommDestroy)(PhNcclComm) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, PhNcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, PhNcclComm, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

int rccl_load() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.h) return 0;
  void *h = nullptr;
  const char *env = getenv("PHNSW_RCCL_LIB");
  if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  const char *names[] = {"librccl.so", "librccl.so.1"};
  for (int i = 0; !h && i < 2; i++) h = dlopen(names[i], RTLD_NOW | RTLD_NOLOAD);  // a copy the process already holds
  for (int i = 1; !h && i >= 0; i--) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    ph_set_error("RCCL not found (%s); set PHNSW_RCCL_LIB", dlerror());
    return PHNSW_E_UNSUPPORTED;
  }
  Rccl r;
  r.h = h;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce) {
    ph_set_error("the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclAllReduce");
    return PHNSW_E_UNSUPPORTED;
  }
  g_rccl = r;
  return 0;
}

int rccl_fail(int rc, const char *what) {
  ph_set_error("%s: RCCL error %d (%s)", what, rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  return PHNSW_E_HIP;
}

struct RcclCtx {
  PhNcclComm comm = nullptr;
  int device = 0;
  hipStream_t stream = nullptr;  // all-reduce of the recall counts
  uint64_t *dev_vals = nullptr;
};

enum { PH_NCCL_INT8 = 0, PH_NCCL_UINT64 = 5, PH_NCCL_SUM = 0 };  // ncclDataType_t / ncclRedOp_t  (rccl.h)

int rccl_all_gather(void *ctx, const void *send, void *recv, uint64_t bytes, void *stream) {
  RcclCtx *c = (RcclCtx *)ctx;
  int rc = g_rccl.AllGather(send, recv, (size_t)bytes, PH_NCCL_INT8, c->comm, (hipStream_t)stream);
  return rc ? rccl_fail(rc, "ncclAllGather") : 0;
}

int rccl_all_reduce_sum(void *ctx, uint64_t *values, uint32_t count) {
  RcclCtx *c = (RcclCtx *)ctx;
  if (count > 16) return PHNSW_E_INVALID;
  PH_HIP(hipMemcpyAsync(c->dev_vals, values, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
  int rc = g_rccl.AllReduce(c->dev_vals, c->dev_vals, count, PH_NCCL_UINT64, PH_NCCL_SUM, c->comm, c->stream);
  if (rc) return rccl_fail(rc, "ncclAllReduce");
  PH_HIP(hipMemcpyAsync(values, c->dev_vals, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
  PH_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

}  // namespace

extern "C" int phnsw_comm_rccl_unique_id(uint8_t *out_id128) try {
  if (!out_id128) return PHNSW_E_INVALID;
  PH_TRY(rccl_load());
  PhNcclUniqueId id;
  memset(&id, 0, sizeof(id));
  int rc = g_rccl.GetUniqueId(&id);
  if (rc) return rccl_fail(rc, "ncclGetUniqueId");
  memcpy(out_id128, id.internal, 128);
  return 0;
} catch (...) { return ph_caught(); }

extern "C" int phnsw_comm_rccl_create(const uint8_t *id128, uint32_t rank, uint32_t world, int device,
                                      phnsw_comm **out) try {
  if (!id128 || !out || world == 0 || rank >= world) {
    ph_set_error("phnsw_comm_rccl_create: invalid argument");
    return PHNSW_E_INVALID;
  }
  PH_TRY(rccl_load());
  PH_HIP(hipSetDevice(device));
  RcclCtx *c = new RcclCtx();
  c->device = device;
  PhNcclUniqueId id;
  memcpy(id.internal, id128, 128);
  int rc = g_rccl.CommInitRank(&c->comm, (int)world, id, (int)rank);
  if (rc) {
    delete c;
    return rccl_fail(rc, "ncclCommInitRank");
  }
  hipError_t he = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipMalloc(&c->dev_vals, 16 * 8);
  if (he != hipSuccess) {
    g_rccl.CommDestroy(c->comm);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return ph_hip_fail(he, "phnsw_comm_rccl_create", __FILE__, __LINE__);
  }
  phnsw_comm *pc = new phnsw_comm();
  memset(pc, 0, sizeof(*pc));
  pc->rank = rank;
  pc->world = world;
  pc->host_buffers = 0;
  pc->ctx = c;
  pc->all_gather = rccl_all_gather;
  pc->all_reduce_sum = rccl_all_reduce_sum;
  *out = pc;
  return 0;
} catch (...) { return ph_caught(); }

// A host wiring its own transport (or the first RCCL call on a new node) can check it before trusting a build to
// it: rank r contributes the pattern f(r, i), every rank verifies all `world` blocks, then the all-reduce.
extern "C" int phnsw_comm_selftest(const phnsw_comm *comm, uint64_t bytes) try {
  if (!comm || !comm->all_gather || !comm->all_reduce_sum || comm->world == 0 || comm->rank >= comm->world || bytes == 0) {
    ph_set_error("phnsw_comm_selftest: need a communicator with both collectives and bytes > 0");
    return PHNSW_E_INVALID;
  }
  const uint32_t w = comm->world;
  auto f = [](uint32_t r, uint64_t i) { return (uint8_t)(ph_mix64(((uint64_t)r << 40) ^ i) >> 17); };
  std::vector<uint8_t> send(bytes), recv(bytes * w, 0);
  for (uint64_t i = 0; i < bytes; i++) send[i] = f(comm->rank, i);
  if (comm->host_buffers) {
    int rc = comm->all_gather(comm->ctx, send.data(), recv.data(), bytes, nullptr);
    if (rc) return rc < 0 ? rc : PHNSW_E_INVALID;
  } else {
    void *ds = nullptr, *dr = nullptr;
    hipStream_t st = nullptr;
    PH_HIP(hipMalloc(&ds, bytes));
    hipError_t he = hipMalloc(&dr, bytes * w);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (he == hipSuccess) he = hipMemcpy(ds, send.data(), bytes, hipMemcpyHostToDevice);
    int rc = he == hipSuccess ? comm->all_gather(comm->ctx, ds, dr, bytes, st) : 0;
    if (he == hipSuccess && !rc) he = hipStreamSynchronize(st);
    if (he == hipSuccess && !rc) he = hipMemcpy(recv.data(), dr, bytes * w, hipMemcpyDeviceToHost);
    hipFree(ds);
    if (dr) hipFree(dr);
    if (st) hipStreamDestroy(st);
    if (he != hipSuccess) return ph_hip_fail(he, "phnsw_comm_selftest", __FILE__, __LINE__);
    if (rc) return rc < 0 ? rc : PHNSW_E_INVALID;
  }
  for (uint32_t r = 0; r < w; r++)
    for (uint64_t i = 0; i < bytes; i++)
      if (recv[(uint64_t)r * bytes + i] != f(r, i)) {
        ph_set_error("phnsw_comm_selftest: all_gather block of rank %u differs at byte %llu", r, (unsigned long long)i);
        return PHNSW_E_INVALID;
      }
  uint64_t v[2] = {comm->rank + 1ull, 1ull};
  int rc = comm->all_reduce_sum(comm->ctx, v, 2);
  if (rc) return rc < 0 ? rc : PHNSW_E_INVALID;
  if (v[0] != (uint64_t)w * (w + 1) / 2 || v[1] != w) {
    ph_set_error("phnsw_comm_selftest: all_reduce_sum gave {%llu, %llu} for a world of %u", (unsigned long long)v[0],
                 (unsigned long long)v[1], w);
    return PHNSW_E_INVALID;
  }
  return 0;
} catch (...) { return ph_caught(); }

// What a collective costs the host and the wire: `iters` all-gathers of `bytes` per rank back to back on the
// collectives' stream -- *host_us = host time per call to ENQUEUE it (what the build's critical path pays even when
// the transfer hides behind the next piece's searches), *total_us = wall time per call until the last has landed.
// Collective: every rank calls it.  Device-buffer transports only.
extern "C" int phnsw_comm_benchmark(const phnsw_comm *comm, uint64_t bytes, uint32_t iters, double *host_us, double *total_us) try {
  if (!comm || !comm->all_gather || comm->host_buffers || !bytes || !iters || !host_us || !total_us) {
    ph_set_error("phnsw_comm_benchmark: needs a device-buffer communicator, bytes > 0, iters > 0");
    return PHNSW_E_INVALID;
  }
  void *ds = nullptr, *dr = nullptr;
  hipStream_t st = nullptr;
  PH_HIP(hipMalloc(&ds, bytes));
  hipError_t he = hipMalloc(&dr, bytes * comm->world);
  if (he == hipSuccess) he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  if (he == hipSuccess) he = hipMemset(ds, 1, bytes);
  int rc = 0;
  double host = 0.0, t0 = 0.0;
  for (uint32_t i = 0; he == hipSuccess && !rc && i < iters + 2; i++) {
    if (i == 2) {  // two warm-up calls
      he = hipStreamSynchronize(st);
      t0 = now_s();
      host = 0.0;
    }
    const double a = now_s();
    rc = comm->all_gather(comm->ctx, ds, dr, bytes, st);
    host += now_s() - a;
  }
  if (he == hipSuccess && !rc) he = hipStreamSynchronize(st);
  const double total = now_s() - t0;
  hipFree(ds);
  if (dr) hipFree(dr);
  if (st) hipStreamDestroy(st);
  if (he != hipSuccess) return ph_hip_fail(he, "phnsw_comm_benchmark", __FILE__, __LINE__);
  if (rc) return rc < 0 ? rc : PHNSW_E_INVALID;
  *host_us = host / iters * 1e6;
  *total_us = total / iters * 1e6;
  return 0;
} catch (...) { return ph_caught(); }

extern "C" void phnsw_comm_destroy(phnsw_comm *pc) {
  if (!pc) return;
  if (pc->all_gather == rccl_all_gather && pc->ctx) {
    RcclCtx *c = (RcclCtx *)pc->ctx;
    hipSetDevice(c->device);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->stream) hipStreamDestroy(c->stream);
    if (c->dev_vals) hipFree(c->dev_vals);
    delete c;
  }
  delete pc;
}
