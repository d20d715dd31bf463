//! `GpuComparator` and `GpuHnsw`: the reference crate's `Comparator` / `Hnsw` surface
//! (parallel-hnsw `src/lib.rs:53-74, 585-1686`) with every batch of distance evaluations and every
//! greedy layer search running in libphnsw's gfx950 kernels.  Method names, argument meaning and
//! result ordering follow the crate; what the crate does per element through Rayon
//! (`lib.rs:1107-1117, 2169-2184`) is one launch here.
//!
//! Written against `include/phnsw.h` through `phnsw-sys`; not compiled in this repository's build
//! image (no Rust toolchain).  `tests/test_rust_shim.py` checks every `phnsw_*` call made here
//! against the header (name and argument count).
use std::ffi::{CStr, CString};
use std::os::raw::{c_char, c_int, c_void};
use std::path::Path;
use std::sync::Arc;

use parallel_hnsw::parameters::{BuildParameters, OptimizationParameters, SearchParameters};
use parallel_hnsw::progress::{ProgressMonitor, ProgressUpdate};
use parallel_hnsw::{AbstractVector, Comparator, Layer, NodeId, VectorId};
use phnsw_sys as sys;

fn check(rc: c_int) {
    if rc != 0 {
        // the crate panics where this library returns a status (lib.rs:261,683,837; types.rs:86)
        let msg = unsafe { CStr::from_ptr(sys::phnsw_last_error()) }.to_string_lossy().into_owned();
        panic!("phnsw error {rc}: {msg}");
    }
}

fn sp_c(s: SearchParameters) -> sys::phnsw_search_params {
    sys::phnsw_search_params {
        number_of_candidates: s.number_of_candidates as u64,
        upper_layer_candidate_count: s.upper_layer_candidate_count as u64,
        probe_depth: s.probe_depth as u64,
    }
}
fn op_c(o: OptimizationParameters) -> sys::phnsw_optimization_params {
    sys::phnsw_optimization_params {
        promotion_threshold: o.promotion_threshold,
        neighborhood_threshold: o.neighborhood_threshold,
        recall_proportion: o.recall_proportion,
        promotion_proportion: o.promotion_proportion,
        search: sp_c(o.search),
    }
}
/// parameters.rs:42-64 field for field; seed / max_link_rounds / promote are libphnsw's additions
/// (`thread_rng` of lib.rs:832, the unbounded loop of lib.rs:1527, promote_at_layer on/off)
fn bp_c(bp: BuildParameters, seed: u64) -> sys::phnsw_build_params {
    sys::phnsw_build_params {
        order: bp.order as u64,
        zero_layer_neighborhood_size: bp.zero_layer_neighborhood_size as u64,
        neighborhood_size: bp.neighborhood_size as u64,
        optimization: op_c(bp.optimization),
        initial_partition_search: sp_c(bp.initial_partition_search),
        seed,
        max_link_rounds: 0,
        promote: 1,
    }
}

struct StoreHandle(*mut sys::phnsw_store);
unsafe impl Send for StoreHandle {}
unsafe impl Sync for StoreHandle {}
impl Drop for StoreHandle {
    fn drop(&mut self) {
        unsafe { sys::phnsw_store_destroy(self.0) }
    }
}

/// `BigComparator` (bigvec.rs:38-57) with its vectors also resident in HBM.  The scalar
/// `compare_raw` stays available on the host copy; batches go to the GPU.
#[derive(Clone)]
pub struct GpuComparator {
    host: Arc<Vec<Vec<f32>>>,
    store: Arc<StoreHandle>,
}

impl GpuComparator {
    pub fn new(data: Arc<Vec<Vec<f32>>>, device: i32) -> Self {
        let dim = data[0].len();
        let mut flat: Vec<f32> = Vec::with_capacity(data.len() * dim);
        for v in data.iter() {
            assert_eq!(v.len(), dim);
            flat.extend_from_slice(v);
        }
        let mut s = std::ptr::null_mut();
        check(unsafe {
            sys::phnsw_store_create(flat.as_ptr(), data.len() as u64, dim as u32, sys::PHNSW_METRIC_COSINE_HALF,
                                    device, &mut s)
        });
        GpuComparator { host: data, store: Arc::new(StoreHandle(s)) }
    }

    /// `compare_vec(v, Stored(id))` for a whole candidate list in one launch (lib.rs:69-73)
    pub fn compare_batch(&self, v: AbstractVector<Vec<f32>>, ids: &[VectorId]) -> Vec<f32> {
        let raw: Vec<u64> = ids.iter().map(|i| i.0 as u64).collect();
        let mut out = vec![0f32; ids.len()];
        let rc = unsafe {
            match v {
                AbstractVector::Stored(q) => sys::phnsw_distance_batch(self.store.0, std::ptr::null(), q.0 as u64,
                                                                       raw.as_ptr(), raw.len() as u64, out.as_mut_ptr()),
                AbstractVector::Unstored(q) => sys::phnsw_distance_batch(self.store.0, q.as_ptr(), 0, raw.as_ptr(),
                                                                         raw.len() as u64, out.as_mut_ptr()),
            }
        };
        check(rc);
        out
    }
}

impl Comparator for GpuComparator {
    type T = Vec<f32>;
    type Borrowable<'a> = &'a Vec<f32>;
    fn lookup(&self, v: VectorId) -> &Vec<f32> {
        &self.host[v.0]
    }
    fn compare_raw(&self, v1: &Vec<f32>, v2: &Vec<f32>) -> f32 {
        // bigvec.rs:47-53
        let mut result = 0.0;
        for (&f1, &f2) in v1.iter().zip(v2.iter()) {
            result += f1 * f2
        }
        (1.0 - result) / 2.0
    }
}

unsafe extern "C" fn progress_trampoline(user: *mut c_void, phase: *const c_char, done: u64, total: u64) -> c_int {
    let monitor = &mut *(user as *mut &mut dyn ProgressMonitor);
    let phase = CStr::from_ptr(phase).to_string_lossy();
    let state = serde_json::json!({ "phase": phase, "done": done, "total": total });
    match monitor.update(ProgressUpdate { state }) {
        Ok(()) => 0,
        Err(_) => 1, // Interrupt (progress.rs:8-10)
    }
}

/// `Hnsw<GpuComparator>` whose layers live on the GPU.
/// The collective seam of the sharded build (`phnsw_comm`): the library's own RCCL transport over xGMI -- rank 0
/// makes the 128-byte id, the host hands it to the other ranks (a file, MPI, a socket), every rank creates its
/// communicator on its own GPU -- or any transport the host supplies as two callbacks.
pub struct ShardComm {
    owned: *mut sys::phnsw_comm,
    custom: Option<Box<sys::phnsw_comm>>,
}
unsafe impl Send for ShardComm {}
impl ShardComm {
    pub fn rccl_unique_id() -> [u8; 128] {
        let mut id = [0u8; 128];
        check(unsafe { sys::phnsw_comm_rccl_unique_id(id.as_mut_ptr()) });
        id
    }
    pub fn rccl(id: &[u8; 128], rank: u32, world: u32, device: i32) -> Self {
        let mut c = std::ptr::null_mut();
        check(unsafe { sys::phnsw_comm_rccl_create(id.as_ptr(), rank, world, device, &mut c) });
        ShardComm { owned: c, custom: None }
    }
    /// a transport of the host's own (MPI, ...): see `phnsw_comm` in include/phnsw.h for the two callbacks
    pub fn custom(c: sys::phnsw_comm) -> Self {
        ShardComm { owned: std::ptr::null_mut(), custom: Some(Box::new(c)) }
    }
    /// every rank calls it: a known pattern through the all-gather and the all-reduce, verified on every rank
    pub fn selftest(&self, bytes: u64) {
        check(unsafe { sys::phnsw_comm_selftest(self.as_ptr(), bytes) });
    }
    fn as_ptr(&self) -> *const sys::phnsw_comm {
        match &self.custom {
            Some(b) => &**b as *const sys::phnsw_comm,
            None => self.owned as *const sys::phnsw_comm,
        }
    }
}
impl Drop for ShardComm {
    fn drop(&mut self) {
        if !self.owned.is_null() {
            unsafe { sys::phnsw_comm_destroy(self.owned) }
        }
    }
}

pub struct GpuHnsw {
    ix: *mut sys::phnsw_index,
    comparator: GpuComparator,
    pub build_parameters: BuildParameters,
}
unsafe impl Send for GpuHnsw {}
unsafe impl Sync for GpuHnsw {} // search entry points are thread safe (phnsw.h)

impl Drop for GpuHnsw {
    fn drop(&mut self) {
        unsafe { sys::phnsw_index_destroy(self.ix) }
    }
}

impl GpuHnsw {
    /// `Hnsw::generate(c, vs, bp, progress)`  lib.rs:825-830
    pub fn generate(c: GpuComparator, vs: Vec<VectorId>, bp: BuildParameters, progress: &mut dyn ProgressMonitor) -> Self {
        let raw: Vec<u64> = vs.iter().map(|v| v.0 as u64).collect();
        let mut ix = std::ptr::null_mut();
        let mut monitor: &mut dyn ProgressMonitor = progress;
        check(unsafe {
            sys::phnsw_build(c.store.0, raw.as_ptr(), raw.len() as u64, &bp_c(bp, 0), Some(progress_trampoline),
                             &mut monitor as *mut &mut dyn ProgressMonitor as *mut c_void, &mut ix)
        });
        GpuHnsw { ix, comparator: c, build_parameters: bp }
    }

    /// `Hnsw::generate` with every per-node phase split over the GPUs of one node (BASELINE config 4): one process
    /// per GPU, every rank calls this with the same vectors, ids and parameters and ends with the same index
    /// (`phnsw_build_sharded`: node ranges per round, all-gather of the per-node results, lib.rs:1097-1153)
    pub fn generate_sharded(c: GpuComparator, vs: Vec<VectorId>, bp: BuildParameters, comm: &ShardComm,
                            progress: &mut dyn ProgressMonitor) -> (Self, sys::phnsw_sharded_stats) {
        let raw: Vec<u64> = vs.iter().map(|v| v.0 as u64).collect();
        let mut ix = std::ptr::null_mut();
        let mut stats = sys::phnsw_sharded_stats::default();
        let mut monitor: &mut dyn ProgressMonitor = progress;
        check(unsafe {
            sys::phnsw_build_sharded(c.store.0, raw.as_ptr(), raw.len() as u64, &bp_c(bp, 0), comm.as_ptr(),
                                     Some(progress_trampoline), &mut monitor as *mut &mut dyn ProgressMonitor as *mut c_void,
                                     &mut ix, &mut stats)
        });
        (GpuHnsw { ix, comparator: c, build_parameters: bp }, stats)
    }

    /// adopt the layers of an `Hnsw` the crate built or deserialised (top first, lib.rs:587)
    pub fn from_layers(c: GpuComparator, layers: &[Layer<GpuComparator>], bp: BuildParameters) -> Self {
        let counts: Vec<u64> = layers.iter().map(|l| l.nodes.len() as u64).collect();
        let widths: Vec<u64> = layers.iter().map(|l| l.neighborhood_size as u64).collect();
        let nodes: Vec<Vec<u64>> = layers.iter().map(|l| l.nodes.iter().map(|v| v.0 as u64).collect()).collect();
        let nbrs: Vec<Vec<u64>> = layers.iter().map(|l| l.neighbors.iter().map(|n| n.0 as u64).collect()).collect();
        let np: Vec<*const u64> = nodes.iter().map(|v| v.as_ptr()).collect();
        let bp_: Vec<*const u64> = nbrs.iter().map(|v| v.as_ptr()).collect();
        let mut ix = std::ptr::null_mut();
        check(unsafe {
            sys::phnsw_index_from_layers(c.store.0, layers.len() as u32, counts.as_ptr(), widths.as_ptr(), np.as_ptr(),
                                         bp_.as_ptr(), &mut ix)
        });
        GpuHnsw { ix, comparator: c, build_parameters: bp }
    }

    pub fn comparator(&self) -> &GpuComparator {
        &self.comparator // lib.rs:648-650
    }
    pub fn layer_count(&self) -> usize {
        unsafe { sys::phnsw_index_layer_count(self.ix) as usize } // lib.rs:644-646
    }
    pub fn vector_count(&self) -> usize {
        let mut n = 0u64;
        check(unsafe {
            sys::phnsw_index_layer_info(self.ix, self.layer_count() as u32 - 1, &mut n, std::ptr::null_mut())
        });
        n as usize // lib.rs:592-594
    }

    /// `Layer { comparator, neighborhood_size, nodes, neighbors }` of layer `from_top` (lib.rs:85-91):
    /// what the crate's own `serialize` (serialize.rs:33-124) and diagnostics need
    pub fn layer(&self, from_top: usize) -> Layer<GpuComparator> {
        let (mut n, mut w) = (0u64, 0u64);
        check(unsafe { sys::phnsw_index_layer_info(self.ix, from_top as u32, &mut n, &mut w) });
        let (mut nodes, mut nbrs) = (vec![0u64; n as usize], vec![0u64; (n * w) as usize]);
        check(unsafe { sys::phnsw_index_layer_read(self.ix, from_top as u32, nodes.as_mut_ptr(), nbrs.as_mut_ptr()) });
        Layer {
            comparator: self.comparator.clone(),
            neighborhood_size: w as usize,
            nodes: nodes.into_iter().map(|v| VectorId(v as usize)).collect(),
            // u64::MAX == !0usize: the empty-slot sentinel survives the cast (types.rs:8-13)
            neighbors: nbrs.into_iter().map(|x| NodeId(x as usize)).collect(),
        }
    }

    /// `Hnsw::search(v, sp)`  lib.rs:663-665
    pub fn search(&self, v: AbstractVector<Vec<f32>>, sp: SearchParameters) -> Vec<(VectorId, f32)> {
        self.search_many(&[v], sp, 0).pop().unwrap()
    }
    /// `Hnsw::search_upto(v, sp, upto_layer_from_top)`  lib.rs:654-661
    pub fn search_upto(&self, v: AbstractVector<Vec<f32>>, sp: SearchParameters, upto: usize) -> Vec<(VectorId, f32)> {
        self.search_many(&[v], sp, upto as u32).pop().unwrap()
    }
    /// a batch of queries in one launch (what the crate's callers get from `par_iter().map(search)`,
    /// lib.rs:1107-1117, 2169-2184); all Stored or all Unstored
    pub fn search_many(&self, vs: &[AbstractVector<Vec<f32>>], sp: SearchParameters, upto: u32)
                       -> Vec<Vec<(VectorId, f32)>> {
        let (nq, ef) = (vs.len(), sp.number_of_candidates);
        let psp = sp_c(sp);
        let (mut ids, mut d, mut len) = (vec![0u64; nq * ef], vec![0f32; nq * ef], vec![0u64; nq]);
        let rc = match vs.first() {
            None => return Vec::new(),
            Some(AbstractVector::Stored(_)) => {
                let q: Vec<u64> = vs.iter().map(|v| match v {
                    AbstractVector::Stored(i) => i.0 as u64,
                    _ => panic!("search_many: mixed Stored / Unstored"),
                }).collect();
                unsafe {
                    sys::phnsw_search_batch_stored(self.ix, q.as_ptr(), nq as u64, &psp, upto, std::ptr::null(),
                                                   ids.as_mut_ptr(), d.as_mut_ptr(), len.as_mut_ptr(),
                                                   std::ptr::null_mut())
                }
            }
            Some(AbstractVector::Unstored(_)) => {
                let mut q: Vec<f32> = Vec::new();
                for v in vs {
                    match v {
                        AbstractVector::Unstored(x) => q.extend_from_slice(x),
                        _ => panic!("search_many: mixed Stored / Unstored"),
                    }
                }
                unsafe {
                    sys::phnsw_search_batch(self.ix, q.as_ptr(), nq as u64, &psp, upto, std::ptr::null(),
                                            ids.as_mut_ptr(), d.as_mut_ptr(), len.as_mut_ptr(), std::ptr::null_mut())
                }
            }
        };
        check(rc);
        (0..nq).map(|q| (0..len[q] as usize).map(|k| (VectorId(ids[q * ef + k] as usize), d[q * ef + k])).collect())
               .collect()
    }

    /// `search(v, sp)` for a batch of raw queries keeping the best `k` of each: the truncation the crate's callers do
    /// themselves (lib.rs:1118) happens on the device, before the transfer (`phnsw_search_batch_topk`)
    pub fn search_many_topk(&self, queries: &[Vec<f32>], sp: SearchParameters, k: usize) -> Vec<Vec<(VectorId, f32)>> {
        let nq = queries.len();
        let psp = sp_c(sp);
        let mut q: Vec<f32> = Vec::with_capacity(nq * queries.first().map_or(0, |x| x.len()));
        for x in queries {
            q.extend_from_slice(x);
        }
        let (mut ids, mut d, mut len) = (vec![0u64; nq * k], vec![0f32; nq * k], vec![0u64; nq]);
        check(unsafe {
            sys::phnsw_search_batch_topk(self.ix, q.as_ptr(), std::ptr::null(), nq as u64, &psp, 0, std::ptr::null(), k as u64,
                                         ids.as_mut_ptr(), d.as_mut_ptr(), len.as_mut_ptr())
        });
        (0..nq).map(|i| (0..len[i] as usize).map(|j| (VectorId(ids[i * k + j] as usize), d[i * k + j])).collect()).collect()
    }

    /// `Hnsw::search_instrumented(v, sp)`  lib.rs:667-673
    pub fn search_instrumented(&self, v: AbstractVector<Vec<f32>>, sp: SearchParameters) -> (Vec<(VectorId, f32)>, usize) {
        let ef = sp.number_of_candidates;
        let psp = sp_c(sp);
        let (mut ids, mut d, mut len, mut index) = (vec![0u64; ef], vec![0f32; ef], 0u64, 0u64);
        let rc = match &v {
            AbstractVector::Stored(i) => {
                let q = i.0 as u64;
                unsafe {
                    sys::phnsw_search_instrumented(self.ix, std::ptr::null(), &q, 1, &psp, ids.as_mut_ptr(), d.as_mut_ptr(),
                                                   &mut len, &mut index)
                }
            }
            AbstractVector::Unstored(x) => unsafe {
                sys::phnsw_search_instrumented(self.ix, x.as_ptr(), std::ptr::null(), 1, &psp, ids.as_mut_ptr(),
                                               d.as_mut_ptr(), &mut len, &mut index)
            },
        };
        check(rc);
        ((0..len as usize).map(|j| (VectorId(ids[j] as usize), d[j])).collect(), index as usize) // u64::MAX == usize::MAX
    }

    /// `Hnsw::improve_index(bp, last_recall, progress)`  lib.rs:1664-1669
    pub fn improve_index(&mut self, bp: BuildParameters, last_recall: Option<f32>,
                         progress: &mut dyn ProgressMonitor) -> f32 {
        let mut out = 0f32;
        let mut monitor: &mut dyn ProgressMonitor = progress;
        check(unsafe {
            sys::phnsw_improve_index(self.ix, &bp_c(bp, 0), last_recall.unwrap_or(f32::NAN), Some(progress_trampoline),
                                     &mut monitor as *mut &mut dyn ProgressMonitor as *mut c_void, &mut out)
        });
        out
    }
    /// `Hnsw::improve_neighbors(op, last_recall)`  lib.rs:1507-1513
    pub fn improve_neighbors(&mut self, op: OptimizationParameters, last_recall: Option<f32>) -> f32 {
        let mut bp = self.build_parameters;
        bp.optimization = op;
        let mut out = 0f32;
        check(unsafe {
            sys::phnsw_improve_neighbors_upto(self.ix, self.layer_count() as u32, &bp_c(bp, 0),
                                              last_recall.unwrap_or(f32::NAN), &mut out)
        });
        out
    }
    /// `Hnsw::stochastic_recall(op)`  lib.rs:1501-1505
    pub fn stochastic_recall(&self, op: OptimizationParameters) -> f32 {
        let mut out = 0f32;
        check(unsafe { sys::phnsw_stochastic_recall_at(self.ix, self.layer_count() as u32 - 1, &op_c(op), &mut out) });
        out
    }

    /// `Hnsw::knn(k, probe_depth)`  lib.rs:905-928 (collected: one launch over the bottom layer)
    pub fn knn(&self, k: usize, probe_depth: usize) -> Vec<(VectorId, Vec<(VectorId, f32)>)> {
        let n = self.vector_count();
        let (mut ids, mut d, mut len) = (vec![0u64; n * k], vec![0f32; n * k], vec![0u64; n]);
        check(unsafe {
            sys::phnsw_knn(self.ix, k as u64, probe_depth as u64, ids.as_mut_ptr(), d.as_mut_ptr(), len.as_mut_ptr())
        });
        self.pair_up(&ids, &d, &len, k)
    }
    /// `Hnsw::threshold_nn(threshold, probe_depth, initial_search_depth)`  lib.rs:930-962
    pub fn threshold_nn(&self, threshold: f32, probe_depth: usize, initial_search_depth: usize, max_out: usize)
                        -> Vec<(VectorId, Vec<(VectorId, f32)>)> {
        let n = self.vector_count();
        let (mut ids, mut d, mut len) = (vec![0u64; n * max_out], vec![0f32; n * max_out], vec![0u64; n]);
        check(unsafe {
            sys::phnsw_threshold_nn(self.ix, threshold, probe_depth as u64, initial_search_depth as u64,
                                    max_out as u64, ids.as_mut_ptr(), d.as_mut_ptr(), len.as_mut_ptr())
        });
        self.pair_up(&ids, &d, &len, max_out)
    }
    fn pair_up(&self, ids: &[u64], d: &[f32], len: &[u64], stride: usize) -> Vec<(VectorId, Vec<(VectorId, f32)>)> {
        let bottom = self.layer(self.layer_count() - 1);
        bottom.nodes.iter().enumerate().map(|(i, v)| {
            (*v, (0..len[i] as usize).map(|j| (VectorId(ids[i * stride + j] as usize), d[i * stride + j])).collect())
        }).collect()
    }

    /// the crate's directory format (serialize.rs:33-209), written / read by libphnsw itself
    pub fn serialize<P: AsRef<Path>>(&self, path: P) {
        let p = CString::new(path.as_ref().to_str().expect("utf-8 path")).unwrap();
        check(unsafe { sys::phnsw_index_serialize(self.ix, p.as_ptr()) });
    }
    pub fn deserialize<P: AsRef<Path>>(path: P, c: GpuComparator) -> Self {
        let p = CString::new(path.as_ref().to_str().expect("utf-8 path")).unwrap();
        let mut ix = std::ptr::null_mut();
        check(unsafe { sys::phnsw_index_deserialize(c.store.0, p.as_ptr(), &mut ix) });
        let mut bp = std::mem::MaybeUninit::<sys::phnsw_build_params>::uninit();
        check(unsafe { sys::phnsw_index_build_params(ix, bp.as_mut_ptr()) });
        let b = unsafe { bp.assume_init() };
        let sp = |s: sys::phnsw_search_params| SearchParameters {
            number_of_candidates: s.number_of_candidates as usize,
            upper_layer_candidate_count: s.upper_layer_candidate_count as usize,
            probe_depth: s.probe_depth as usize,
        };
        let build_parameters = BuildParameters {
            order: b.order as usize,
            zero_layer_neighborhood_size: b.zero_layer_neighborhood_size as usize,
            neighborhood_size: b.neighborhood_size as usize,
            optimization: OptimizationParameters {
                promotion_threshold: b.optimization.promotion_threshold,
                neighborhood_threshold: b.optimization.neighborhood_threshold,
                recall_proportion: b.optimization.recall_proportion,
                promotion_proportion: b.optimization.promotion_proportion,
                search: sp(b.optimization.search),
            },
            initial_partition_search: sp(b.initial_partition_search),
        };
        GpuHnsw { ix, comparator: c, build_parameters }
    }
}
