// The reference's benches/bench.rs with its stale call fixed: `Hnsw::generate` takes
// (comparator, vs, BuildParameters, &mut dyn ProgressMonitor) since lib.rs:825-830, the bench still
// passes (comparator, vs, 24, 48, 2) (benches/bench.rs:54-63).  Same workload: 10 000 random
// 100-d vectors, cosine-style metric, default build parameters (order 12, M 24, M0 48) -- once
// through the crate's CPU path and once through libphnsw.
#![feature(test)]
extern crate test;

use std::sync::Arc;

use parallel_hnsw::parameters::BuildParameters;
use parallel_hnsw::{Comparator, Hnsw, VectorId};
use parallel_hnsw_gpu::{GpuComparator, GpuHnsw};
use rand::{thread_rng, Rng};
use test::Bencher;

const LENGTH: usize = 10000;
const DIM: usize = 100;

#[derive(Clone)]
struct SillyComparator {
    data: Arc<Vec<Vec<f32>>>,
}

impl Comparator for SillyComparator {
    type T = Vec<f32>;
    type Borrowable<'a> = &'a Vec<f32>;
    fn lookup(&self, v: VectorId) -> Self::Borrowable<'_> {
        &self.data[v.0]
    }
    fn compare_raw(&self, v1: &Self::T, v2: &Self::T) -> f32 {
        let mut result = 0.0;
        for (&f1, &f2) in v1.iter().zip(v2.iter()) {
            result += f1 * f2
        }
        (1.0 - result) / 2.0
    }
}

fn create_test_data(length: usize) -> Arc<Vec<Vec<f32>>> {
    let mut rng = thread_rng();
    Arc::new((0..length).map(|_| {
        let v: Vec<f32> = (0..DIM).map(|_| rng.gen_range(-1.0..1.0)).collect();
        let norm = v.iter().map(|f| f * f).sum::<f32>().sqrt();
        v.into_iter().map(|f| f / norm).collect() // random_normed_vec, bigvec.rs:59-65
    }).collect())
}

#[bench]
fn generate_cpu(b: &mut Bencher) {
    let comparator = SillyComparator { data: create_test_data(LENGTH) };
    let vs: Vec<VectorId> = (0..LENGTH).map(VectorId).collect();
    b.iter(|| {
        let _result: Hnsw<_> = Hnsw::generate(comparator.clone(), vs.clone(), BuildParameters::default(), &mut ());
    });
}

#[bench]
fn generate_gpu(b: &mut Bencher) {
    let comparator = GpuComparator::new(create_test_data(LENGTH), 0);
    let vs: Vec<VectorId> = (0..LENGTH).map(VectorId).collect();
    b.iter(|| {
        let _result = GpuHnsw::generate(comparator.clone(), vs.clone(), BuildParameters::default(), &mut ());
    });
}
