// K2: batched greedy HNSW search for gfx950 -- one 64-lane wavefront per query.
//
// Replaces, for a whole batch of queries at once,
//   search_layers_instrumented   /root/reference/src/search.rs:93-140
//   Layer::closest_vectors       /root/reference/src/lib.rs:250-277
//   Layer::closest_nodes         /root/reference/src/lib.rs:175-248   (the hot loop)
//   PriorityQueue::merge         /root/reference/src/priority_queue.rs:109-144
//   Comparator::compare_vec      /root/reference/src/lib.rs:69-73 (+ bigvec.rs:47-53)
//
// Mapping to the hardware (DESIGN.md has the long form):
//  * a wave owns a query for its whole descent; waves pull queries from one atomic counter
//    (persistent grid, no tail of idle CUs);
//  * the query vector lives in registers (NV float4 per lane); a candidate row is read as
//    NV fully coalesced 1 KiB wave-loads (global_load_dwordx4), 4 rows in flight per wave
//    (4 / 12 / 24 in the small-batch kernels), fma-accumulated per lane and reduced with an xor
//    butterfly -- this fixed order is the oracle's ORC_SUM_BLOCKED64, so results are bit-comparable;
//  * the small top layers are not gathered at all: their distances come from a table built per
//    launch on the matrix cores (tiny.hip) and the walk there runs in table ids, visited set in LDS;
//  * the result queue (PriorityQueue, cap = number_of_candidates) is a sorted (d,id) array
//    in LDS; a hop's <=64 new candidates are compared with the queue two 64-entry chunks at a
//    time, top chunks first: one scalar loop over the entering keys (v_readlane + 64-bit compare)
//    yields every queue slot's shift, every entering key's insertion point (s_bcnt1 of the same
//    masks -- no search of the queue) and the entering keys' ranks among themselves; then
//    everything moves to its final slot in place (no per-element shifting);
//  * the reference's unbounded `visit_queue` (every evaluated node, re-sorted each hop,
//    pop = smallest (d,id)) is represented exactly as: an "expanded" bit on queue entries
//    + an append-only spill list in HBM for entries that fell out of / never entered the
//    queue.  All spilled entries are worse than every queue entry, so pop = first
//    unexpanded queue entry, else the minimum unexpanded spill entry (rare, linear scan);
//  * `visited` (HashSet) is a per-wave bitmap in HBM driven by returning atomicOr
//    (test-and-set in one L2 round trip), cleared after each layer by walking queue+spill.
//
// Hand-written for CDNA4: wave64 ballots/readlane, LDS queue, no portability layer.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "phnsw_internal.h"

#include "phnsw_device.h"

// -DPH_HOP_PROFILE: a debugging build that prints, per layer of every query, where the hops' time went
// (100 MHz ticks of s_memrealtime; each phase ends with a forced wait).  Never part of libphnsw.so proper.
#ifdef PH_HOP_PROFILE
#define PH_TICK(k)                                   \
  {                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");   \
    const uint64_t t_now = wall_clock64();           \
    tprof[k] += t_now - t_last;                      \
    t_last = t_now;                                  \
  }
#else
#define PH_TICK(k)
#endif

// INSTR: Hnsw::search_instrumented (lib.rs:667-673).  Every visit_queue entry of the reference carries the
// index_sum of its discovery path (lib.rs:211-220: parent's sum + 1-based rank in the parent's sorted batch); the
// value returned is the index_sum of the node expanded at the last hop of the bottom layer that changed
// candidates.first() (lib.rs:225-231).  The sums ride along in a third queue array (the prefix scratch S, idle
// during the hops) and a parallel spill array; the plain kernels compile none of it.
// BIG (threshold_nn only, ph_search_kernel_big): the layer queue lives in global memory with a capacity chosen at
// launch (a.cap_max), so resize_capacity can go on doubling past what LDS holds.  The queue is then shared between
// the lanes of the wave through L2: every barrier of the body also orders and invalidates (queue_sync), and the loops
// over the queue's 64-entry chunks run to the live length instead of being unrolled CAPC times.
template <bool BIG>
__device__ __forceinline__ void queue_sync() {
  if constexpr (BIG) __threadfence();
  __syncthreads();
}

template <int CAPC, class Dist, bool INSTR = false, bool BIG = false>
__device__ __forceinline__ void ph_search_body(const PhSearchArgs &a) {
  extern __shared__ uint32_t smem[];
  constexpr int CAP = CAPC * 64;
  uint32_t *Cid = smem;                    // running candidates: VectorIds (search.rs:110)
  float *Cd = (float *)(smem + CAP);       //
  uint32_t *Qid = smem + 2 * CAP;          // layer queue: NodeIds | EXPF (lib.rs:264)
  float *Qd = (float *)(smem + 3 * CAP);   //
  if constexpr (BIG) {
    Qid = a.big_q + (uint64_t)blockIdx.x * 2u * a.cap_max;
    Qd = (float *)(Qid + a.cap_max);
  }
  uint32_t *S = smem + 4 * CAP;            // prefix scratch [CAP + 64]
  float *dist_lds = (float *)(smem + 5 * CAP + 64);  // DistPQ: the query's lookup table
  if (Dist::GLOBAL_TABLE) dist_lds = (float *)((char *)a.pq_tables + (size_t)blockIdx.x * a.pq_table_bytes);
  // dense top layers (tiny.hip): this query's row of the distance table and the visited bits of the
  // table ids, both in LDS; T = 0 when the launch has none (or its layers turned out not to be nested)
  const uint32_t T = (a.tiny_layers && a.tiny_member[a.tiny_n] == 0u) ? a.tiny_layers : 0u;
  const bool tiny_lds_row = a.tiny_n <= a.tiny_lds_nodes;
  float *Dl = (float *)(smem + 5 * CAP + 64);
  uint32_t *Vl = smem + 5 * CAP + 64 + (tiny_lds_row ? a.tiny_stride : 0u);
  const uint32_t tiny_words = (a.tiny_n + 31u) / 32u;

  const uint32_t lane = threadIdx.x;
  const uint64_t lt = lanemask_lt(lane);
  constexpr bool DENSE_ONLY = dist_is_none<Dist>::value;
  // the dense-only launch and its follow-up agree through the table's device-side "usable" flag (phnsw_internal.h)
  const bool table_ok = (a.dense_only || a.after_dense) ? a.dense_flag[0] == 0u : true;
  if (DENSE_ONLY && !table_ok) return;  // layers not nested: the follow-up launch walks everything per hop
  const uint32_t layer_lo = a.after_dense ? (table_ok ? a.after_dense : 0u) : a.layer_lo;
  uint32_t *vis = a.visited + (uint64_t)blockIdx.x * a.visited_words;
  uint2 *ovf = DENSE_ONLY ? a.dense_ovf + (uint64_t)blockIdx.x * a.dense_ovf_cap : a.ovf + (uint64_t)blockIdx.x * a.ovf_cap;
  const uint32_t ovf_cap = DENSE_ONLY ? a.dense_ovf_cap : a.ovf_cap;
  uint32_t *const Qs = S;
  uint32_t *const ovf_s = INSTR ? a.ovf_s + (uint64_t)blockIdx.x * a.ovf_cap : nullptr;

  // locality schedule: with an `order` the query list is cut into 8 consecutive segments, one
  // per XCD, so that the queries one L2 serves together are neighbours in `order`; a wave
  // whose segment is exhausted moves on to the next one (no idle tail)
  uint32_t seg_cur = 0, seg_done = 0;
  if (a.order) {
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    seg_cur = (xcc & 15u) & 7u;
  }

  for (;;) {
    uint32_t q = 0, qpos = 0;  // qpos: position in the launch's processing order (row of the dense table)
    if (!a.order) {
      if (lane == 0) q = atomicAdd(a.counter, 1u);
      q = rfl32(q);
      if (q >= a.nq) break;
      qpos = q;
    } else {
      for (;;) {
        uint32_t p = 0;
        if (lane == 0) p = atomicAdd(a.counter + seg_cur * 16u, 1u);
        p = rfl32(p);
        const uint32_t base = seg_cur * a.seg;
        const uint32_t len = base >= a.nq ? 0u : min(a.seg, a.nq - base);
        if (p < len) {
          qpos = base + p;
          q = a.order[qpos];
          break;
        }
        seg_cur = (seg_cur + 1u) & 7u;
        if (++seg_done == 8u) {
          q = PH_EMPTY32;
          break;
        }
      }
      if (q == PH_EMPTY32) break;
    }

    const uint32_t last_layer = a.n_layers - 1;
    // a descent may run as two launches (upper layers; then the bottom layer with the queries
    // re-ordered by where they landed): layers [layer_lo, layer_hi) of this launch, the running
    // candidates parked in the output rows in between
    const uint32_t layer_hi = a.layer_hi ? a.layer_hi : a.n_layers;
    if (layer_lo && a.status[q] != ST_OK) continue;  // failed in the first launch: keep its status
    // knn modes: the query is a node of the bottom layer
    const uint32_t qnode = (BIG && a.knn_nodes) ? a.knn_nodes[q] : a.first_node + q;
    uint32_t qvec = a.knn_mode ? a.layers[last_layer].nodes[qnode] : (a.qids ? a.qids[q] : 0u);
    // row of this query in the dense table: its launch position, or (the build's kept table) its NodeId in layer X
    uint64_t trow = qpos;
    if (a.tiny_rows) trow = (uint64_t)((a.tiny_row_map ? a.tiny_row_map[qvec] : qvec) - a.tiny_row_first);
    Dist dist;
    if (a.queries && !a.knn_mode)
      dist.prepare_raw(a.dist, a.queries + (uint64_t)q * a.ldq, dist_lds, lane);
    else
      dist.prepare_stored(a.dist, qvec, dist_lds, lane);
    const uint32_t excl = a.exclude ? a.exclude[q] : PH_EMPTY32;
    uint32_t n_dist = 0, n_hops = 0, err = ST_OK;
    uint32_t index_distance = 0xFFFFFFFFu;  // usize::MAX until a layer has run  search.rs:112
    uint32_t n_tab = 0;  // of n_dist: evaluations served by the dense tables (measurement: the rest are gathered rows)
    uint32_t n_dist0 = 0, n_hops0 = 0;  // counters a split descent brought in from its earlier launches
    uint32_t clen = 0;
    uint32_t ef = a.ef;  // queue capacity; grows in threshold_nn mode (resize_capacity)
    bool big_written = false;

    if (layer_lo) {
      clen = a.out_len[q];
#pragma unroll
      for (int c = 0; c < CAPC; c++) {
        uint32_t i = lane + 64u * c;
        if (i < clen) {
          Cid[i] = a.out_ids[(uint64_t)q * a.ef + i];
          Cd[i] = a.out_d[(uint64_t)q * a.ef + i];
        }
      }
      if (a.out_stats) {
        n_dist = a.out_stats[2 * (uint64_t)q];
        n_hops = a.out_stats[2 * (uint64_t)q + 1];
      }
      n_dist0 = n_dist;
      n_hops0 = n_hops;
    } else if (!a.knn_mode) {
      // entry_vector + distance_from_entry  search.rs:101-111
      uint32_t entry = a.layers[0].nodes[0];
      float d0;
      if constexpr (DENSE_ONLY) {  // the table holds it
        const PhLayerDev TLy = a.layers[a.tiny_layers - 1];
        const uint32_t tid = TLy.vec2node ? TLy.vec2node[entry] : entry;
        d0 = (a.tiny_d + trow * a.tiny_stride)[tid < a.tiny_n ? tid : 0u];
      } else {
        d0 = dist.batch(a.dist, 1ull, entry, lane);
      }
      d0 = __uint_as_float(rl32(__float_as_uint(d0), 0));
      n_dist = 1;
      if (lane == 0) {
        Cid[0] = entry;
        Cd[0] = d0;
      }
      clen = 1;
    }
    const float *Dg = a.tiny_d + trow * a.tiny_stride;  // this query's row of the table
    if (T) {
      if (tiny_lds_row)
        for (uint32_t i = lane; i < a.tiny_n; i += 64) Dl[i] = Dg[i];
      for (uint32_t i = lane; i < tiny_words; i += 64) Vl[i] = 0u;
    }
    queue_sync<BIG>();

    for (uint32_t li = a.knn_mode ? last_layer : layer_lo; li < layer_hi && err == ST_OK; li++) {
      PhLayerDev L = a.layers[li];
      // a dense top layer is walked in table ids: ids, id maps and neighbour rows of the table layer
      const bool tl = li < T;
      if (tl) {
        const PhLayerDev TLy = a.layers[T - 1];
        L.n_nodes = a.tiny_n;
        L.nodes = TLy.nodes;
        L.vec2node = TLy.vec2node;
        L.neighbors = a.tiny_nbr + a.tiny_off[li];
      }
      const bool identity = L.vec2node == nullptr;
      // ---- closest_vectors: VectorId -> NodeId, queue = new(cap); merge_pairs  lib.rs:258-266
      uint32_t qlen;
      if (a.knn_mode) {
        // pq.merge_pairs(&[(node, 0.0)])  lib.rs:917-918
        if (lane == 0) {
          Qid[0] = qnode;
          Qd[0] = 0.0f;
          atomicOr(&vis[qnode >> 5], 1u << (qnode & 31));
        }
        qlen = 1;
      } else {
        bool miss = false;
#pragma unroll
        for (int c = 0; c < CAPC; c++) {
          uint32_t i = lane + 64u * c;
          if (i < clen) {
            uint32_t vid = Cid[i];
            uint32_t nid = identity ? vid : L.vec2node[vid];
            if (nid >= L.n_nodes || (tl && !((a.tiny_member[nid] >> li) & 1u))) {  // get_node(v).unwrap() would panic  lib.rs:261
              miss = true;
              nid = 0;
            }
            Qid[i] = nid;
            Qd[i] = Cd[i];
          }
        }
        if (__ballot(miss)) {
          err = ST_MISSING;
          break;
        }
        // visited = candidates ids  lib.rs:187
#pragma unroll
        for (int c = 0; c < CAPC; c++) {
          uint32_t i = lane + 64u * c;
          if (i < clen) {
            uint32_t nid = Qid[i];
            if (tl)
              atomicOr(&Vl[nid >> 5], 1u << (nid & 31));
            else
              atomicOr(&vis[nid >> 5], 1u << (nid & 31));
          }
        }
        qlen = clen;
      }
      queue_sync<BIG>();

      // Hnsw::threshold_nn (lib.rs:930-962, knn_mode == 2) calls closest_nodes repeatedly on a
      // growing queue; everything else runs this block once
      float thr_last = 0.0f;
      uint32_t thr_last_size = 0;
      for (;;) {
      if (a.knn_mode == 2) {
        if (!(thr_last < a.threshold && qlen > thr_last_size)) break;  // lib.rs:945
        thr_last_size = qlen;
        if (thr_last_size > 1 || n_hops > 0) {
          // a fresh closest_nodes call: every queue entry is a seed again (lib.rs:182-187)
#pragma unroll
          for (int c = 0; c < (BIG ? (int)((qlen + 63u) >> 6) : CAPC); c++) {
            uint32_t i = lane + 64u * c;
            if (i < qlen) {
              uint32_t nid = Qid[i] & IDM;
              Qid[i] = nid;
              atomicOr(&vis[nid >> 5], 1u << (nid & 31));
            }
          }
          queue_sync<BIG>();
        }
      }
      // ---- closest_nodes  lib.rs:175-248
      uint32_t ovf_n = 0;
      uint32_t pd = a.probe_depth;
      uint32_t highest = 0;  // highest_improvement  lib.rs:190
      if constexpr (INSTR) {
#pragma unroll
        for (int c = 0; c < CAPC; c++)
          if (lane + 64u * c < qlen) Qs[lane + 64u * c] = 0u;  // seeds: NodeDistance::ZERO  lib.rs:182-185
        queue_sync<BIG>();
      }
      // every queue entry below scan_from has been expanded: the pop scan starts at its 64-entry chunk,
      // and a hop's merge touches only the chunks from its first insertion point on
      uint32_t scan_from = 0;
#ifdef PH_CELL_PROBE
      const bool probing = a.probe_pos && li == last_layer && !tl;
      uint32_t probe_p0 = 0, probe_cnt[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (probing) probe_p0 = a.probe_pos[Qid[0] & IDM];
#endif
#ifdef PH_HOP_PROFILE
      uint64_t tprof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      uint64_t t_last = wall_clock64();
      const uint32_t hops_before = n_hops, dist_before = n_dist;
#endif
      for (;;) {
        // visit_queue.pop(): smallest (d,id) among not yet expanded nodes  lib.rs:191,243-244
        int pop = -1;
        uint32_t cur = 0;
        for (uint32_t c = scan_from >> 6; 64u * c < qlen; c++) {
          const uint32_t i = lane + 64u * c;
          const uint32_t e = i < qlen ? Qid[i] : EXPF;
          const uint64_t b = __ballot(!(e & EXPF));
          if (b) {
            const int pl = __builtin_ctzll(b);
            pop = (int)(64u * c) + pl;
            cur = rl32(e, pl);
            break;
          }
        }
        if (pop >= 0) {
          if (lane == 0) Qid[pop] = cur | EXPF;
          if constexpr (BIG) __threadfence();  // the merge below reads the slot back from another lane
        } else {
          if (ovf_n == 0) break;
          wait_vm0();  // spill stores of this wave have reached L2
          uint64_t best = KEY_NONE;
          uint32_t bi = 0;
          for (uint32_t i = lane; i < ovf_n; i += 64) {
            uint32_t id = __hip_atomic_load(&ovf[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!(id & EXPF)) {
              uint32_t db = __hip_atomic_load(&ovf[i].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              uint64_t k = mkkey(__uint_as_float(db), id);
              if (k < best) {
                best = k;
                bi = i;
              }
            }
          }
#pragma unroll
          for (int s = 32; s >= 1; s >>= 1) {
            uint64_t o = ((uint64_t)__shfl_xor((uint32_t)(best >> 32), s) << 32) | __shfl_xor((uint32_t)best, s);
            uint32_t oi = __shfl_xor(bi, s);
            if (o < best) {
              best = o;
              bi = oi;
            }
          }
          if (best == KEY_NONE) break;  // frontier exhausted
          cur = (uint32_t)best & IDM;
          if (lane == 0)
            __hip_atomic_store(&ovf[bi].x, cur | EXPF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pop = -1 - (int)bi;  // INSTR reads the spilled entry's index_sum below
        }
        uint32_t cur_s = 0;
        if constexpr (INSTR) cur_s = pop >= 0 ? Qs[pop] : ovf_s[(uint32_t)(-1 - pop)];
        cur &= IDM;
        n_hops++;
        PH_TICK(0)

        // get_neighbors(next) + filter(!visited)  lib.rs:195-198
        uint32_t nb = PH_EMPTY32;
        if (lane < L.W) nb = L.neighbors[(uint64_t)cur * L.W + lane];
        if constexpr (Dist::EARLY) {
          // the candidates' rows are requested before the visited test-and-set below returns
          const bool valid = nb < L.n_nodes;
          uint32_t v = 0;
          if (valid) v = identity ? nb : L.nodes[nb];
          dist.prefetch(a.dist, valid, v, lane);
        }
        PH_TICK(1)
        bool fresh = false;
        if (nb < L.n_nodes) {
          uint32_t bit = 1u << (nb & 31);
          uint32_t old = tl ? atomicOr(&Vl[nb >> 5], bit) : atomicOr(&vis[nb >> 5], bit);
          fresh = !(old & bit);
        }
        const uint64_t fm = __ballot(fresh);
        const uint32_t m = __popcll(fm);
        n_dist += m;
        if (tl) n_tab += m;
#ifdef PH_CELL_PROBE
        if (probing) {
          uint32_t dd = 0xFFFFFFFFu;
          if (fresh) {
            const int d0 = (int)a.probe_pos[nb] - (int)probe_p0;
            dd = (uint32_t)(d0 < 0 ? -d0 : d0);
          }
#pragma unroll
          for (int k = 0; k < 10; k++) probe_cnt[k] += __popcll(__ballot(fresh && dd <= (k ? (1u << (k - 1)) : 0u)));
          probe_cnt[10] += m;
        }
#endif
        PH_TICK(2)

        // distance batch: compare_vec(v, Stored(get_vector(n)))  lib.rs:200-202 -- in a dense top
        // layer the value was computed by the tile pass (same bits) and is looked up
        float myd;
        if (tl) {
          myd = 0.f;
          if (fresh) myd = tiny_lds_row ? Dl[nb] : Dg[nb];
        } else if constexpr (Dist::EARLY) {
          myd = dist.finish(a.dist, fm, lane);
        } else {
          uint32_t vid = 0;
          if (fresh) vid = identity ? nb : L.nodes[nb];
          myd = dist.batch(a.dist, fm, vid, lane);
        }

        PH_TICK(3)
        // candidates.merge_pairs(sorted batch)  lib.rs:206,226 / priority_queue.rs:109-144,
        // as one parallel rank-merge.  Batch keys are distinct and absent from the queue
        // (visited), so final slot = (#queue keys below) + (#batch keys below).
        const uint64_t key = fresh ? mkkey(myd, nb) : KEY_NONE;
        uint32_t my_s = 0;
        if constexpr (INSTR) {  // (ix, (n, d)) of the sorted batch: index_sum + ix + 1  lib.rs:211-220
          uint32_t rank_all = 0;
          uint64_t remf = fm;
          while (remf) {
            const int j = __builtin_ctzll(remf);
            remf &= remf - 1;
            rank_all += (rl64(key, j) < key) ? 1u : 0u;
          }
          my_s = cur_s + rank_all + 1u;
        }
        // An element that is worse than the tail of a FULL queue cannot enter it: it goes straight to
        // the spill list and takes no part in the merge.  Late in a layer most hops bring nothing
        // else; those skip the merge altogether.
        const bool full = qlen == ef;
        const float dtail = full ? Qd[ef - 1] : PH_FMAX;
        const uint64_t tailkey = full ? mkkey(dtail, Qid[ef - 1]) : KEY_NONE;
        const bool ins = fresh && key < tailkey;
        const uint64_t im = __ballot(ins);
        uint32_t pos = 0, pos_min = 0xFFFFFFFFu, newpos = 0xFFFFFFFFu;
        // merge()'s return value (priority_queue.rs:109-144), closed form for a sorted,
        // duplicate-free batch e_0 < e_1 < ...: the first element decides.  It is inserted
        // (true) unless it ranks past a full queue; then Err(i>=cap) => break => false,
        // except when its priority ties the queue's tail (Ok branch returns cap, false) and
        // a second element follows (Err(0) on the empty slice => true without writing).
        // e_0 enters the queue exactly when some element does (im != 0); otherwise every distance is
        // >= the tail's, so e_0 ties the tail exactly when some element's distance equals it.
        bool did = im != 0;
        if (!im && m >= 2) did = __ballot(fresh && myd == dtail) != 0;
        if (im) {
          // One pass handles two 64-entry chunks of the queue, TOP chunks first.  For every entering key k_j (a
          // scalar loop over the bits of `im`) the pass counts, per queue slot, the entering keys below it (the
          // distance the slot's entry moves up: into its own chunk or the one above, both already in registers
          // or already rewritten -- a wave's LDS reads and writes execute in program order) and, per entering
          // key, the processed slots holding a greater key (s_bcnt1 of the same compare mask), which gives its
          // insertion point without a search.  The first pass also ranks the entering keys among themselves.
          // The passes stop at the first chunk whose head is below every entering key: everything under it
          // stays where it is and is neither read nor rewritten.  What falls past `ef` is spilled.
          uint32_t rank = 0, greater = 0;
          bool first_pass = true;
          const int c_top = (int)((qlen - 1u) >> 6);
          int c = c_top, c_low;
          for (;;) {
            const bool two = c > 0;  // chunk c-1 belongs to this pass (it is a full chunk)
            const uint32_t i1 = lane + 64u * (uint32_t)c, i0 = i1 - 64u;
            const bool has1 = i1 < qlen;
            const uint32_t qi1 = has1 ? Qid[i1] : PH_EMPTY32;
            const float qd1 = has1 ? Qd[i1] : PH_FMAX;
            const uint64_t qk1 = has1 ? mkkey(qd1, qi1) : KEY_NONE;
            uint32_t qi0 = PH_EMPTY32;
            float qd0 = PH_FMAX;
            if (two) {
              qi0 = Qid[i0];
              qd0 = Qd[i0];
            }
            const uint64_t qk0 = two ? mkkey(qd0, qi0) : KEY_NONE;
            uint32_t qs1 = 0, qs0 = 0;
            if constexpr (INSTR) {
              qs1 = has1 ? Qs[i1] : 0u;
              qs0 = two ? Qs[i0] : 0u;
            }
            uint32_t sh1 = 0, sh0 = 0;
            uint64_t rem = im;
            if (first_pass) {
              while (rem) {
                const int j = __builtin_ctzll(rem);
                rem &= rem - 1;
                const uint64_t kj = rl64(key, j);
                rank += (kj < key) ? 1u : 0u;
                const bool g1 = kj < qk1, g0 = two && kj < qk0;
                sh1 += g1 ? 1u : 0u;
                sh0 += g0 ? 1u : 0u;
                const uint32_t g = (uint32_t)__popcll(__ballot(g1)) + (uint32_t)__popcll(__ballot(g0));
                greater += lane == (uint32_t)j ? g : 0u;
              }
              first_pass = false;
            } else {
              while (rem) {
                const int j = __builtin_ctzll(rem);
                rem &= rem - 1;
                const uint64_t kj = rl64(key, j);
                const bool g1 = kj < qk1, g0 = two && kj < qk0;
                sh1 += g1 ? 1u : 0u;
                sh0 += g0 ? 1u : 0u;
                const uint32_t g = (uint32_t)__popcll(__ballot(g1)) + (uint32_t)__popcll(__ballot(g0));
                greater += lane == (uint32_t)j ? g : 0u;
              }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // both chunks are in registers before either is overwritten
            if constexpr (BIG) wait_vm0();
            const uint32_t np1 = i1 + sh1, np0 = i0 + sh0;
            if (has1 && np1 < ef) {
              Qid[np1] = qi1;
              Qd[np1] = qd1;
              if constexpr (INSTR) Qs[np1] = qs1;
            }
            if (two && np0 < ef) {
              Qid[np0] = qi0;
              Qd[np0] = qd0;
              if constexpr (INSTR) Qs[np0] = qs0;
            }
            if (qlen + 64u > ef) {  // only a queue within 64 entries of its capacity can push anything out
              const bool spill1 = has1 && np1 >= ef;
              const uint64_t sm1 = __ballot(spill1);
              if (sm1) {
                const uint32_t at = ovf_n + __popcll(sm1 & lt);
                if (spill1 && at < ovf_cap) ovf[at] = make_uint2(qi1, __float_as_uint(qd1));
                if constexpr (INSTR)
                  if (spill1 && at < ovf_cap) ovf_s[at] = qs1;
                ovf_n += __popcll(sm1);
              }
              const bool spill0 = two && np0 >= ef;
              const uint64_t sm0 = __ballot(spill0);
              if (sm0) {
                const uint32_t at = ovf_n + __popcll(sm0 & lt);
                if (spill0 && at < ovf_cap) ovf[at] = make_uint2(qi0, __float_as_uint(qd0));
                if constexpr (INSTR)
                  if (spill0 && at < ovf_cap) ovf_s[at] = qs0;
                ovf_n += __popcll(sm0);
              }
            }
            c_low = two ? c - 1 : c;
            // the head of the lowest chunk done (its lane 0) is below every entering key: so is all the rest
            if (c_low == 0 || rl32(two ? sh0 : sh1, 0) == 0u) break;
            c -= 2;
          }
          // slots of the chunks done (the empty ones of the top chunk count as greater) + all of the chunks below
          pos = 64u * (uint32_t)(c_top + 1) - greater;
          if (ins) newpos = pos + rank;
          pos_min = rl32(pos, __builtin_ctzll(__ballot(ins && rank == 0u)));  // the smallest entering key's
          PH_TICK(5)
          queue_sync<BIG>();
          PH_TICK(6)
        }
        {
          if (fresh && newpos < ef) {
            Qid[newpos] = nb;
            Qd[newpos] = myd;
            if constexpr (INSTR) Qs[newpos] = my_s;
          }
          bool spill = fresh && newpos >= ef;
          uint64_t sm = __ballot(spill);
          if (sm) {
            uint32_t at = ovf_n + __popcll(sm & lt);
            if (spill && at < ovf_cap) ovf[at] = make_uint2(nb, __float_as_uint(myd));
            if constexpr (INSTR)
              if (spill && at < ovf_cap) ovf_s[at] = my_s;
            ovf_n += __popcll(sm);
          }
          if constexpr (INSTR)  // current_best != candidates.first(): a new entry took slot 0  lib.rs:225-231
            if (__ballot(fresh && newpos == 0u)) highest = cur_s;
        }
        qlen = min(ef, qlen + m);
        scan_from = min(pop >= 0 ? (uint32_t)pop + 1u : scan_from, pos_min);
        queue_sync<BIG>();
        PH_TICK(4)
        if (ovf_n > ovf_cap) {
          err = ST_OVERFLOW;
          break;
        }
        if (!did) {  // lib.rs:233-238
          pd -= 1;
          if (pd == 0) break;
        }
      }
      if (err != ST_OK) break;
      if constexpr (INSTR) index_distance = highest;  // last_index_distance = index_distance  search.rs:135
#ifdef PH_CELL_PROBE
      if (probing && lane == 0) {
        for (int k = 0; k < 11; k++) atomicAdd(&a.probe_out[k], (unsigned long long)probe_cnt[k]);
        atomicAdd(&a.probe_out[11], 1ull);
      }
#endif
#ifdef PH_HOP_PROFILE
      if (lane == 0 && q == 0)
        printf("hop profile q0 layer %u%s: hops %u evals %u | us: pop %.1f nbr %.1f visited %.1f dist %.1f merge %.1f (+ search/rank %.1f, shift %.1f)\n", li,
               tl ? " (dense)" : "", n_hops - hops_before, n_dist - dist_before, tprof[0] * 0.01, tprof[1] * 0.01, tprof[2] * 0.01,
               tprof[3] * 0.01, tprof[4] * 0.01, tprof[5] * 0.01, tprof[6] * 0.01);
#endif

      // ---- clear this layer's visited bits (queue + spill hold every evaluated node)
      if (tl) {
        for (uint32_t i = lane; i < tiny_words; i += 64) Vl[i] = 0u;
      } else {
#pragma unroll
        for (int c = 0; c < (BIG ? (int)((qlen + 63u) >> 6) : CAPC); c++) {
          uint32_t i = lane + 64u * c;
          if (i < qlen) vis[(Qid[i] & IDM) >> 5] = 0u;
        }
        if (ovf_n) {
          wait_vm0();
          for (uint32_t i = lane; i < ovf_n; i += 64) {
            uint32_t id = __hip_atomic_load(&ovf[i].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & IDM;
            vis[id >> 5] = 0u;
          }
        }
      }
      wait_vm0();
      if (a.knn_mode != 2) break;
      thr_last = Qd[qlen - 1];  // pq.last().1  lib.rs:948
      if (thr_last < a.threshold && qlen == ef) {  // pq.resize_capacity(capacity * 2)  lib.rs:949-951
        if (ef * 2 > (BIG ? a.cap_max : (uint32_t)CAP)) {
          err = ST_CAPACITY;
          break;
        }
        ef *= 2;
      }
      }  // closest_nodes call loop
      if (err != ST_OK) break;

      if constexpr (BIG) {
        // the knn modes start from no running candidates, so closest_vectors' tail (lib.rs:268-276) followed by
        // candidates.merge_pairs (search.rs:136) is the queue itself: it goes straight to the output row, of which
        // the caller reads out_stride entries at most
        const uint32_t os = a.out_stride ? a.out_stride : a.ef;
        uint32_t kept = 0;
        for (uint32_t base = 0; base < qlen && kept < os; base += 64) {
          const uint32_t i = base + lane;
          const bool has = i < qlen;
          const uint32_t nid = has ? (Qid[i] & IDM) : 0u;
          const uint32_t v = has ? (identity ? nid : L.nodes[nid]) : PH_EMPTY32;
          const bool keep = has && v != excl;
          const uint64_t km = __ballot(keep);
          const uint32_t at = kept + __popcll(km & lt);
          if (keep && at < os) {
            a.out_ids[(uint64_t)q * os + at] = v;
            a.out_d[(uint64_t)q * os + at] = Qd[i];
          }
          kept += __popcll(km);
        }
        clen = min(kept, os);
        for (uint32_t i = clen + lane; i < os; i += 64) {
          a.out_ids[(uint64_t)q * os + i] = PH_EMPTY32;
          a.out_d[(uint64_t)q * os + i] = PH_FMAX;
        }
        big_written = true;
        continue;
      }
      // ---- closest_vectors tail: NodeId -> VectorId, filter(include), take(count)  lib.rs:268-276
      const uint32_t candidate_count = (a.n_layers == 1 || li == last_layer) ? ef : a.upper;  // search.rs:122-126
      uint32_t bv[CAPC];
      float bd[CAPC];
      uint32_t bpos[CAPC];
      uint32_t kept = 0;
#pragma unroll
      for (int c = 0; c < CAPC; c++) {
        uint32_t i = lane + 64u * c;
        bool has = i < qlen;
        uint32_t nid = has ? (Qid[i] & IDM) : 0u;
        bd[c] = has ? Qd[i] : PH_FMAX;
        bv[c] = has ? (identity ? nid : L.nodes[nid]) : PH_EMPTY32;
        bool keep = has && bv[c] != excl;
        uint64_t km = __ballot(keep);
        uint32_t at = kept + __popcll(km & lt);
        bpos[c] = (keep && at < candidate_count) ? at : PH_EMPTY32;
        kept += __popcll(km);
      }
      const uint32_t blen = min(kept, candidate_count);
      queue_sync<BIG>();
#pragma unroll
      for (int c = 0; c < CAPC; c++) {
        if (bpos[c] != PH_EMPTY32) {
          Qid[bpos[c]] = bv[c];
          Qd[bpos[c]] = bd[c];
        }
      }
      queue_sync<BIG>();

      // ---- candidates.merge_pairs(&closest)  search.rs:136 : sorted set union, cap ef.
      // An element present in both lists (same id => same distance) is kept once.
      uint32_t ci[CAPC];
      float cd[CAPC];
      uint32_t cpos[CAPC];
      uint32_t dups = 0;
#pragma unroll
      for (int c = 0; c < CAPC; c++) {
        uint32_t i = lane + 64u * c;
        bool has = i < clen;
        ci[c] = has ? Cid[i] : PH_EMPTY32;
        cd[c] = has ? Cd[i] : PH_FMAX;
        bool dup = false;
        uint32_t lb = 0;
        if (has) {
          uint64_t k = mkkey(cd[c], ci[c]);
          lb = lds_lower_bound(Qid, Qd, blen, k);
          dup = lb < blen && mkkey(Qd[lb], Qid[lb]) == k;
        }
        uint64_t dm = __ballot(dup);
        uint32_t pre = dups + __popcll(dm & lt);  // duplicates among C[0..i)
        S[i] = pre;
        cpos[c] = (has && !dup) ? (i - pre) + lb : PH_EMPTY32;
        dups += __popcll(dm);
      }
      if (lane == 0) S[CAP] = dups;
      queue_sync<BIG>();
#pragma unroll
      for (int c = 0; c < CAPC; c++) {
        uint32_t j = lane + 64u * c;
        bool has = j < blen;
        bv[c] = has ? Qid[j] : PH_EMPTY32;
        bd[c] = has ? Qd[j] : PH_FMAX;
        bpos[c] = PH_EMPTY32;
        if (has) {
          uint32_t la = lds_lower_bound(Cid, Cd, clen, mkkey(bd[c], bv[c]));
          uint32_t dupb = la < clen ? S[la] : S[CAP];  // S[clen] when la == clen
          if (la == clen) dupb = dups;
          bpos[c] = j + (la - dupb);
        }
      }
      queue_sync<BIG>();
#pragma unroll
      for (int c = 0; c < CAPC; c++) {
        if (cpos[c] < ef) {
          Cid[cpos[c]] = ci[c];
          Cd[cpos[c]] = cd[c];
        }
        if (bpos[c] < ef) {
          Cid[bpos[c]] = bv[c];
          Cd[bpos[c]] = bd[c];
        }
      }
      clen = min(ef, clen - dups + blen);
      queue_sync<BIG>();
    }

    if (err != ST_OK) {
      // leave the slot clean for the next query: wipe the whole bitmap (rare path; a dense-only launch has none)
      if (!DENSE_ONLY)
        for (uint64_t w = lane; w < a.visited_words; w += 64) vis[w] = 0u;
      wait_vm0();
      clen = 0;
    }
    // (candidates.iter().collect(), ..)  search.rs:139
    const uint32_t ostride = a.out_stride ? a.out_stride : a.ef;
    for (uint32_t i = lane; i < ostride && !big_written; i += 64) {
      a.out_ids[(uint64_t)q * ostride + i] = i < clen ? Cid[i] : PH_EMPTY32;
      a.out_d[(uint64_t)q * ostride + i] = i < clen ? Cd[i] : PH_FMAX;
    }
    if (a.out_hit) {
      // res.iter().any(|v| v == *vid)  lib.rs:1492
      // hit_eps > 0: match_within_epsilon (search.rs:173-187): only the leading results with
      // |d| < eps count
      uint32_t cut = clen;
      if (a.hit_eps > 0.f) {
#pragma unroll
        for (int c = 0; c < CAPC; c++) {
          uint32_t i = lane + 64u * c;
          bool far = i < clen && !(fabsf(Cd[i]) < a.hit_eps);
          uint64_t fmk = __ballot(far);
          if (fmk && cut == clen) cut = 64u * c + __builtin_ctzll(fmk);
        }
      }
      bool hit = false;
#pragma unroll
      for (int c = 0; c < CAPC; c++) {
        uint32_t i = lane + 64u * c;
        hit |= (i < cut && Cid[i] == qvec);
      }
      uint64_t hm = __ballot(hit);
      if (lane == 0) a.out_hit[q] = hm ? 1u : 0u;
    }
    if (a.out_key && lane == 0) {
      // where the query landed: the cell (or node) of its best candidate in the last layer done
      uint32_t key = PH_EMPTY32;
      if (clen) {
        const PhLayerDev KL = a.layers[layer_hi - 1];
        uint32_t nid = KL.vec2node ? KL.vec2node[Cid[0]] : Cid[0];
        if (nid < KL.n_nodes) key = a.key_pos ? a.key_pos[nid] : nid;
      }
      a.out_key[q] = key;
    }
    if (lane == 0 && a.totals) {
      atomicAdd(&a.totals[0], (unsigned long long)(n_dist - n_dist0));
      atomicAdd(&a.totals[1], (unsigned long long)(n_hops - n_hops0));
    }
    if (lane == 0 && a.launch_totals) {
      atomicAdd(&a.launch_totals[0], (unsigned long long)(n_dist - n_dist0));
      atomicAdd(&a.launch_totals[1], (unsigned long long)(n_hops - n_hops0));
      if (a.launch_tab && n_tab) atomicAdd(a.launch_tab, (unsigned long long)n_tab);
    }
    if constexpr (INSTR)
      if (lane == 0) a.out_index[q] = index_distance;
    if (lane == 0) {
      a.out_len[q] = clen;
      a.status[q] = err;
      if (a.out_stats) {
        a.out_stats[2 * (uint64_t)q] = n_dist;
        a.out_stats[2 * (uint64_t)q + 1] = n_hops;
      }
    }
    queue_sync<BIG>();
  }
}

template <int CAPC, class Dist>
__global__ __launch_bounds__(64) void ph_search_kernel(PhSearchArgs a) {
  ph_search_body<CAPC, Dist>(a);
}

// The dense top layers in a launch of their own: no distance policy state, so half the registers and (queues of
// ef <= 256 in 256 slots) half the LDS of the full kernel -- twice the resident waves on a walk that is pure
// latency and instruction issue.  The running candidates are parked in the output rows for the follow-up launch.
template <int CAPC>
__global__ __launch_bounds__(64) void ph_search_kernel_dense(PhSearchArgs a) {
  ph_search_body<CAPC, DistNone>(a);
}

// Small batches leave most of the chip idle and finish with their slowest query: their kernels keep up to 24
// rows in flight per wave (one load round per hop instead of up to twelve) at one wave per SIMD.  Same
// arithmetic per row, so the same results.
template <int CAPC, int NV>
__global__ __launch_bounds__(64, 1) void ph_search_kernel_lat(PhSearchArgs a) {
  ph_search_body<CAPC, DistF32<NV, 0>>(a);
}

// Hnsw::search_instrumented: the same body carrying index sums (f32 stores; queues of 512 or 1024 slots)
template <int CAPC, int NV>
__global__ __launch_bounds__(64) void ph_search_kernel_instr(PhSearchArgs a) {
  ph_search_body<CAPC, DistF32<NV>, true>(a);
}

// Hnsw::threshold_nn past the LDS queues: the BIG body (queue in global memory, capacity a.cap_max)
template <class Dist>
__global__ __launch_bounds__(64) void ph_search_kernel_big(PhSearchArgs a) {
  ph_search_body<2, Dist, false, true>(a);
}

// the register-table policy keeps a whole lookup table in VGPRs: two waves per SIMD is its register budget
template <int CAPC, int M>
__global__ __launch_bounds__(64, 2) void ph_search_kernel_pqr(PhSearchArgs a) {
  ph_search_body<CAPC, DistPQR<M>>(a);
}

// ------------------------------------------------------------------ host side

typedef void (*ph_search_fn)(PhSearchArgs);

static int pick_capc(uint32_t ef) { return ef <= 128 ? 2 : (ef <= 512 ? 8 : (ef <= 1024 ? 16 : 0)); }
static int pick_capc_dense(uint32_t ef) { return ef <= 128 ? 2 : (ef <= 256 ? 4 : (ef <= 512 ? 8 : (ef <= 1024 ? 16 : 0))); }
static int pick_nv(uint32_t nv4) { return nv4 <= 64 ? 1 : (nv4 <= 192 ? 3 : (nv4 <= 384 ? 6 : 0)); }

// the register-table policy serves 8-bit tables (mode 2) of exactly 32 / 64 / 96 / 128 sub-spaces with at most
// 256 centroids and queues of up to 512 entries; PHNSW_PQ_TABLE=lds|global selects the older table placements
static int pick_pqr(const phnsw_store *s, int capc) {
  if (!s->codes || s->pq_table_f16 != 2 || (capc != 2 && capc != 8) || getenv("PHNSW_PQ_TABLE")) return 0;
  const uint32_t m = s->pq_m;
  return (s->pq_ksub == 256 && (m == 32 || m == 64 || m == 96 || m == 128)) ? (int)m : 0;
}
static ph_search_fn pick_kernel_pqr(int capc, int m) {
#define PH_KR(C, M) \
  if (capc == C && m == M) return (ph_search_fn)ph_search_kernel_pqr<C, M>;
  PH_KR(2, 32) PH_KR(2, 64) PH_KR(2, 96) PH_KR(2, 128)
  PH_KR(8, 32) PH_KR(8, 64) PH_KR(8, 96) PH_KR(8, 128)
#undef PH_KR
  return nullptr;
}

// shared-codebook PQ stores (u16 codes): the query layout is DistF32's, the candidate rows come through the codes
static ph_search_fn pick_kernel_pqs(int capc, int nv) {
#define PH_KS(C, N) \
  if (capc == C && nv == N) return (ph_search_fn)ph_search_kernel<C, DistPQS<N>>;
  PH_KS(2, 1) PH_KS(2, 3) PH_KS(2, 6) PH_KS(8, 1) PH_KS(8, 3) PH_KS(8, 6) PH_KS(16, 1) PH_KS(16, 3) PH_KS(16, 6)
#undef PH_KS
  return nullptr;
}

static ph_search_fn pick_kernel_dense(int capc) {
  switch (capc) {
    case 2: return (ph_search_fn)ph_search_kernel_dense<2>;
    case 4: return (ph_search_fn)ph_search_kernel_dense<4>;
    case 8: return (ph_search_fn)ph_search_kernel_dense<8>;
    case 16: return (ph_search_fn)ph_search_kernel_dense<16>;
  }
  return nullptr;
}

static ph_search_fn pick_kernel_instr(int capc, int nv) {
#define PH_KI(C, N) \
  if (capc == C && nv == N) return (ph_search_fn)ph_search_kernel_instr<C, N>;
  PH_KI(8, 1) PH_KI(8, 3) PH_KI(8, 6) PH_KI(16, 1) PH_KI(16, 3) PH_KI(16, 6)
#undef PH_KI
  return nullptr;
}

#define PH_LATENCY_MAX 1024u  // batches up to this many queries run the latency kernels (PHNSW_NO_LAT=1: never)
static ph_search_fn pick_kernel_lat(int capc, int nv) {
#define PH_KL(C, N) \
  if (capc == C && nv == N) return (ph_search_fn)ph_search_kernel_lat<C, N>;
  PH_KL(2, 1) PH_KL(2, 3) PH_KL(8, 1) PH_KL(8, 3)
#undef PH_KL
  return nullptr;
}

// nv == 0 selects the product-quantised policy
static ph_search_fn pick_kernel(int capc, int nv) {
#define PH_K(C, N) \
  if (capc == C && nv == N) return (ph_search_fn)ph_search_kernel<C, DistF32<N>>;
#define PH_KQ(C) \
  if (capc == C && nv == 0)   \
    return ph_pq_global_tables() ? (ph_search_fn)ph_search_kernel<C, DistPQG> : (ph_search_fn)ph_search_kernel<C, DistPQ>;
  PH_K(2, 1) PH_K(2, 3) PH_K(2, 6)
  PH_K(8, 1) PH_K(8, 3) PH_K(8, 6)
  PH_K(16, 1) PH_K(16, 3) PH_K(16, 6)
  PH_KQ(2) PH_KQ(8) PH_KQ(16)
#undef PH_K
#undef PH_KQ
  return nullptr;
}

// pq_lds: bytes behind the queues -- the PQ lookup table, or the dense-top-layer table row + visited bits
static size_t lds_bytes(int capc, size_t pq_lds) { return (size_t)(5 * capc * 64 + 64) * 4 + pq_lds; }

uint32_t ph_search_slots(uint32_t ef, uint32_t nv4, bool pq, size_t pq_lds, int pqr_m) {
  int capc = pick_capc(ef), nv = pq ? 0 : pick_nv(nv4);
  if (!capc || (!pq && !nv)) return 0;
  ph_search_fn fn = pqr_m == -1 ? pick_kernel_pqs(capc, nv) : (pqr_m ? pick_kernel_pqr(capc, pqr_m) : pick_kernel(capc, nv));
  if (pqr_m == -2) fn = pick_kernel_lat(capc, nv);
  if (pqr_m == -3) {
    capc = std::max(capc, 8);
    fn = pick_kernel_instr(capc, nv);
  }
  if (!fn) return 0;
  int dev = 0;
  hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  size_t lds = lds_bytes(capc, pq_lds);
  if (lds > 64 * 1024) {
    if (lds > 160 * 1024) return 0;
    if (hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
  }
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, 64, lds) != hipSuccess || per_cu <= 0)
    per_cu = 1;
  int cap = 16;
  per_cu = std::min(per_cu, cap);
  if (const char *e = getenv("PHNSW_WAVES_PER_CU")) {  // tuning knob: trusts the caller over the occupancy query
    if (atoi(e) > 0) per_cu = atoi(e);
    if (getenv("PHNSW_VERBOSE")) fprintf(stderr, "[phnsw] search grid: %d waves per CU (forced), lds %zu\n", per_cu, lds);
  }
  return (uint32_t)(per_cu * prop.multiProcessorCount);
}

void ph_workspace_free(PhWorkspace &ws) {
  if (ws.visited) hipFree(ws.visited);
  if (ws.ovf) hipFree(ws.ovf);
  if (ws.counter) hipFree(ws.counter);
  if (ws.pq_tables) hipFree(ws.pq_tables);
  if (ws.ev0) hipEventDestroy(ws.ev0);
  if (ws.ev1) hipEventDestroy(ws.ev1);
  if (ws.evc) hipEventDestroy(ws.evc);
  for (auto &e : ws.evd)
    if (e) hipEventDestroy(e);
  if (ws.dtotals) hipFree(ws.dtotals);
  if (ws.dense_ovf) hipFree(ws.dense_ovf);
  if (ws.ovf_s) hipFree(ws.ovf_s);
  ph_workspace_order_free(ws);
  ph_tiny_free(ws);
  ws = PhWorkspace();
}

int ph_workspace_ensure(const phnsw_index *ix, PhWorkspace &ws, uint32_t ef, uint32_t ovf_cap) {
  uint64_t max_nodes = 0;
  for (auto &l : ix->layers) max_nodes = std::max<uint64_t>(max_nodes, l.n_nodes);
  uint64_t words = (max_nodes + 31) / 32 + 1;
  const int pqr = ix->store->codes16 ? -1 : pick_pqr(ix->store, pick_capc(ef));  // -1: shared-codebook store
  const bool pqg = ix->store->codes != nullptr && (ph_pq_global_tables() || pqr);  // (the register policy stages its table there)
  uint32_t slots = ph_search_slots(ef, ix->store->ld / 4, ix->store->codes != nullptr,
                                   pqg ? 0 : ph_pq_lds_bytes(ix->store), pqr);
  if (slots == 0) {
    ph_set_error("unsupported search shape: ef=%u dim=%u (ef <= 1024, dim <= 1536; PQ table + queue <= 160 KB LDS)", ef,
                 ix->store->dim);
    return PHNSW_E_UNSUPPORTED;
  }
  if (!ws.counter) {
    PH_HIP(hipMalloc(&ws.counter, 512));  // 8 work counters, 64 B apart
    PH_HIP(hipEventCreate(&ws.ev0));
    PH_HIP(hipEventCreate(&ws.ev1));
    PH_HIP(hipEventCreate(&ws.evc));
    for (auto &e : ws.evd) PH_HIP(hipEventCreate(&e));
    PH_HIP(hipMalloc(&ws.dtotals, sizeof(unsigned long long) * 3 * PH_MAX_DISPATCH));
  }
  if (ws.n_slots < slots || ws.visited_words < words) {
    if (ws.visited) PH_HIP(hipFree(ws.visited));
    ws.visited = nullptr;
    uint32_t ns = std::max(ws.n_slots, slots);
    uint64_t nw = std::max(ws.visited_words, words);
    PH_HIP(hipMalloc(&ws.visited, (size_t)ns * nw * 4));
    PH_HIP(hipMemset(ws.visited, 0, (size_t)ns * nw * 4));
    if (ws.ovf && ws.n_slots < ns) {
      PH_HIP(hipFree(ws.ovf));
      ws.ovf = nullptr;
    }
    ws.n_slots = ns;
    ws.visited_words = nw;
  }
  if (pqg && ws.pq_tables_bytes < (size_t)ws.n_slots * ph_pq_lds_bytes(ix->store)) {
    if (ws.pq_tables) PH_HIP(hipFree(ws.pq_tables));
    ws.pq_tables = nullptr;
    ws.pq_tables_bytes = (size_t)ws.n_slots * ph_pq_lds_bytes(ix->store);
    PH_HIP(hipMalloc(&ws.pq_tables, ws.pq_tables_bytes));
  }
  if (!ws.ovf || ws.ovf_cap < ovf_cap) {
    if (ws.ovf) PH_HIP(hipFree(ws.ovf));
    ws.ovf = nullptr;
    ws.ovf_cap = std::max(ws.ovf_cap, ovf_cap);
    PH_HIP(hipMalloc(&ws.ovf, (size_t)ws.n_slots * ws.ovf_cap * sizeof(uint2)));
  }
  return 0;
}

// the wait that orders a descent behind the previous user of this workspace, and the event that
// starts its clock; the launches of the descent follow on the same stream
int ph_search_begin(PhWorkspace &ws, hipStream_t stream) {
  if (ws.timed) PH_HIP(hipStreamWaitEvent(stream, ws.ev1, 0));
  PH_HIP(hipEventRecord(ws.ev0, stream));
  PH_HIP(hipMemsetAsync(ws.dtotals, 0, sizeof(unsigned long long) * 3 * PH_MAX_DISPATCH, stream));
  ws.n_dispatch = 0;
  ws.d_tiny = false;
  return 0;
}

template <class T>
static hipError_t grow_buf(T **p, size_t *have, size_t need) {
  if (*have >= need) return hipSuccess;
  if (*p) hipFree(*p);
  *p = nullptr;
  *have = 0;
  hipError_t e = hipMalloc((void **)p, need);
  if (e == hipSuccess) *have = need;
  return e;
}

// the dense-only launch of a descent (PhSearchArgs::dense_only): its own kernel, grid and spill lists
static int search_launch_dense(PhWorkspace &ws, PhSearchArgs &a, hipStream_t stream) {
  const int capc = pick_capc_dense(a.ef);
  ph_search_fn fn = pick_kernel_dense(capc);
  if (!fn) {
    ph_set_error("unsupported search shape: ef=%u", a.ef);
    return PHNSW_E_UNSUPPORTED;
  }
  const size_t lds = lds_bytes(capc, ph_tiny_lds_bytes(a));
  if (!ws.cus) {  // per workspace, hence per index and device (a process may drive several devices)
    int dev = 0;
    PH_HIP(hipGetDevice(&dev));
    PH_HIP(hipDeviceGetAttribute(&ws.cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int cus = ws.cus;
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)fn, 64, lds) != hipSuccess || per_cu <= 0) per_cu = 8;
  per_cu = std::min(per_cu, 32);
  if (const char *e = getenv("PHNSW_DENSE_WAVES_PER_CU"))
    if (atoi(e) > 0) per_cu = atoi(e);
  const uint32_t grid = (uint32_t)std::min<uint64_t>((uint64_t)per_cu * cus, a.nq);
  if (grid == 0) return 0;
  // a layer of the table cannot evaluate more nodes than the table has: spill lists of tiny_stride entries never overflow
  a.dense_ovf_cap = a.tiny_stride;
  PH_HIP(grow_buf(&ws.dense_ovf, &ws.dense_ovf_bytes, (size_t)grid * a.dense_ovf_cap * sizeof(uint2)));
  a.dense_ovf = ws.dense_ovf;
  a.visited = ws.visited;  // never touched: the visited set of a dense layer is in LDS
  a.visited_words = 0;
  a.ovf = ws.ovf;
  a.ovf_cap = ws.ovf_cap;
  a.counter = ws.counter;
  PH_HIP(hipMemsetAsync(ws.counter, 0, 512, stream));
  a.seg = a.order ? (a.nq + 7u) / 8u : 0u;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, a);
  PH_HIP(hipGetLastError());
  return 0;
}

int ph_search_launch_big(const phnsw_index *ix, PhSearchArgs &a, uint32_t grid, hipStream_t stream) {
  const phnsw_store *s = ix->store;
  const bool pq = s->codes != nullptr;
  const int nv = pq ? 0 : pick_nv(a.dist.nv4);
  ph_search_fn fn = nullptr;
  size_t pq_lds = 0;
  if (s->codes16) {
    fn = nv == 1 ? (ph_search_fn)ph_search_kernel_big<DistPQS<1>>
                 : (nv == 3 ? (ph_search_fn)ph_search_kernel_big<DistPQS<3>> : (nv == 6 ? (ph_search_fn)ph_search_kernel_big<DistPQS<6>> : nullptr));
  } else if (pq) {
    fn = (ph_search_fn)ph_search_kernel_big<DistPQ>;  // the per-query table in LDS
    pq_lds = ph_pq_lds_bytes(s);
  } else {
    fn = nv == 1 ? (ph_search_fn)ph_search_kernel_big<DistF32<1>>
                 : (nv == 3 ? (ph_search_fn)ph_search_kernel_big<DistF32<3>> : (nv == 6 ? (ph_search_fn)ph_search_kernel_big<DistF32<6>> : nullptr));
  }
  const size_t lds = lds_bytes(2, pq_lds);
  if (!fn || lds > 160 * 1024 || a.knn_mode != 2 || !a.big_q || a.cap_max < 2 * a.ef || grid == 0) {
    ph_set_error("unsupported search shape: threshold_nn with a queue of %u entries, dim %u", a.cap_max, s->dim);
    return PHNSW_E_UNSUPPORTED;
  }
  if (lds > 64 * 1024)
    PH_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  a.pq_tables = nullptr;
  a.pq_table_bytes = 0;
  a.seg = 0;
  a.order = nullptr;
  PH_HIP(hipMemsetAsync(a.counter, 0, 512, stream));
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds, stream, a);
  PH_HIP(hipGetLastError());
  return 0;
}

int ph_search_launch(const phnsw_index *ix, PhWorkspace &ws, PhSearchArgs &a, hipStream_t stream, bool mark_end) {
  if (a.dense_only) return search_launch_dense(ws, a, stream);
  const bool pq = ix->store->codes != nullptr;
  int capc = pick_capc(std::max(a.ef, a.cap_max)), nv = pq ? 0 : pick_nv(a.dist.nv4);
  const int pqr = ix->store->codes16 ? -1 : pick_pqr(ix->store, capc);
  a.pq_tables = ws.pq_tables;
  a.pq_table_bytes = (uint32_t)ph_pq_lds_bytes(ix->store);
  ph_search_fn fn = nullptr;
  if (capc && (pq || nv)) fn = pqr == -1 ? pick_kernel_pqs(capc, nv) : (pqr ? pick_kernel_pqr(capc, pqr) : pick_kernel(capc, nv));
  // small batches of f32 queries: the latency kernels, when the shape has one
  int lat = 0;
  if (!pq && !pqr && a.nq <= PH_LATENCY_MAX && !getenv("PHNSW_NO_LAT") && pick_kernel_lat(capc, nv)) {
    fn = pick_kernel_lat(capc, nv);
    lat = -2;
    // one wave per SIMD leaves each wave a quarter of the CU's LDS: room for the table row of a far larger table layer
    // (7 331 nodes = 29 KB at 1M vectors), whose look-ups are then LDS reads instead of a global round trip per hop --
    // as long as three blocks still fit a CU (a batch of up to 768 queries loses no slot)
    if (a.tiny_layers && a.tiny_n > a.tiny_lds_nodes && !getenv("PHNSW_NO_LAT_LDS_ROW") &&
        lds_bytes(capc, (size_t)a.tiny_stride * 4u + (size_t)((a.tiny_n + 31u) / 32u + 1u) * 4u) <= (160u * 1024u) / 3u)
      a.tiny_lds_nodes = a.tiny_n;
  }
  const size_t pq_lds = pq ? ((pqr || ph_pq_global_tables()) ? 0 : ph_pq_lds_bytes(ix->store)) : ph_tiny_lds_bytes(a);
  if (a.out_index) {  // Hnsw::search_instrumented
    if (pq || pqr) {
      ph_set_error("search_instrumented: f32 stores only");
      return PHNSW_E_UNSUPPORTED;
    }
    capc = std::max(capc, 8);
    fn = pick_kernel_instr(capc, nv);
    lat = -3;
    if (ws.ovf_s_cap < (size_t)ws.n_slots * ws.ovf_cap) {
      if (ws.ovf_s) PH_HIP(hipFree(ws.ovf_s));
      ws.ovf_s = nullptr;
      ws.ovf_s_cap = 0;
      PH_HIP(hipMalloc(&ws.ovf_s, (size_t)ws.n_slots * ws.ovf_cap * 4));
      ws.ovf_s_cap = (size_t)ws.n_slots * ws.ovf_cap;
    }
    a.ovf_s = ws.ovf_s;
  }
  if (!fn) {
    ph_set_error("unsupported search shape: ef=%u nv4=%u", a.ef, a.dist.nv4);
    return PHNSW_E_UNSUPPORTED;
  }
  a.visited = ws.visited;
  a.visited_words = ws.visited_words;
  a.ovf = ws.ovf;
  a.ovf_cap = ws.ovf_cap;
  a.counter = ws.counter;
  uint32_t slots = std::min<uint32_t>(ph_search_slots(std::max(a.ef, a.cap_max), a.dist.nv4, pq, pq_lds, lat ? lat : pqr), ws.n_slots);
  uint32_t grid = (uint32_t)std::min<uint64_t>(slots, a.nq);
  if (grid == 0) return 0;
  PH_HIP(hipMemsetAsync(ws.counter, 0, 512, stream));
  a.seg = a.order ? (a.nq + 7u) / 8u : 0u;
  hipLaunchKernelGGL(fn, dim3(grid), dim3(64), lds_bytes(capc, pq_lds), stream, a);
  PH_HIP(hipGetLastError());
  if (mark_end) {
    PH_HIP(hipEventRecord(ws.ev1, stream));
    ws.timed = true;
  }
  return 0;
}
