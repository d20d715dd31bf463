/* ORACLE (test infrastructure): PriorityQueue restated from src/priority_queue.rs:28-223.
 * Quirks are kept on purpose -- merge()'s return value drives closest_nodes' probe_depth
 * accounting (src/lib.rs:226,233-238), so they are observable in search results. */
#include "orc.h"

/* OrderedFloat comparisons (src/types.rs:78-88): partial_cmp().unwrap(); NaN never enters
 * (rejected before any queue is built). */

/* partition_point(|d| d != f32::MAX)   src/priority_queue.rs:56-59 */
uint64_t orc_pq_len(const orc_pq *q) {
  uint64_t lo = 0, hi = q->cap;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (q->prio[mid] != ORC_FMAX)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

/* PriorityQueueIter: stops at the first empty *id*   src/priority_queue.rs:207-222 */
uint64_t orc_pq_iter_len(const orc_pq *q) {
  uint64_t i = 0;
  while (i < q->cap && q->data[i] != ORC_EMPTY) i++;
  return i;
}

/* insert_at   src/priority_queue.rs:70-100 */
static uint64_t pq_insert_at(orc_pq *q, uint64_t idx, uint64_t elt, float priority) {
  if (idx < q->cap && q->data[idx] != elt) {
    /* walk through all elements with exactly the same priority as us */
    while (q->prio[idx] == priority && q->data[idx] <= elt) {
      if (q->data[idx] == elt) return idx;
      idx++;
      if (idx == q->cap) return idx;
    }
    uint64_t swap_start = orc_pq_len(q);
    /* for i in (idx+1 ..= swap_start).rev() { if i == cap {continue}; x[i] = x[i-1] } */
    for (uint64_t i = swap_start + 1; i-- > idx + 1;) {
      if (i == q->cap) continue;
      q->data[i] = q->data[i - 1];
      q->prio[i] = q->prio[i - 1];
    }
    q->data[idx] = elt;
    q->prio[idx] = priority;
  }
  return idx;
}

/* insert   src/priority_queue.rs:102-107 : partition_point(|d| d < priority) */
uint64_t orc_pq_insert(orc_pq *q, uint64_t elt, float priority) {
  uint64_t lo = 0, hi = q->cap;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (q->prio[mid] < priority)
      lo = mid + 1;
    else
      hi = mid;
  }
  return pq_insert_at(q, lo, elt, priority);
}

/* slice.binary_search_by(cmp to target): Ok(i) for any equal element, else Err(insertion
 * point).  Which equal element is reported does not matter: the caller rewinds to the
 * start of the equal run (src/priority_queue.rs:121-129). */
static int bsearch_f32(const float *p, uint64_t n, float target, uint64_t *pos) {
  uint64_t lo = 0, hi = n;
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (p[mid] == target) {
      *pos = mid;
      return 1;
    }
    if (p[mid] < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  *pos = lo;
  return 0;
}

/* merge   src/priority_queue.rs:109-144 */
int orc_pq_merge(orc_pq *q, const uint64_t *ids, const float *prios, uint64_t m) {
  int did_something = 0;
  uint64_t last_idx = 0;
  for (uint64_t k = 0; k < m; k++) {
    float other_distance = prios[k];
    if (last_idx > q->cap) break;
    uint64_t i;
    int ok = bsearch_f32(q->prio + last_idx, q->cap - last_idx, other_distance, &i);
    if (ok) {
      /* walk to the beginning of the match (may rewind before last_idx) */
      uint64_t start_idx = i + last_idx;
      while (start_idx != 0) {
        if (q->prio[start_idx - 1] != other_distance) break;
        start_idx--;
      }
      last_idx = pq_insert_at(q, start_idx, ids[k], other_distance);
      did_something |= (last_idx != q->cap);
    } else {
      /* NB: i is relative to last_idx but compared with the full capacity */
      if (i >= q->cap) break;
      last_idx = pq_insert_at(q, i + last_idx, ids[k], other_distance);
      did_something = 1; /* unconditionally, even when nothing was written */
    }
  }
  return did_something;
}
