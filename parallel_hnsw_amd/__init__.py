"""parallel_hnsw_amd: MI355X-native HNSW build + search behind the public surface of the
Rust crate terminusdb-labs/parallel-hnsw (hot path only)."""
from ._lib import BuildParams, OptimizationParams, PhnswError, SearchParams, build_lib, lib  # noqa: F401
from .hnsw import (EMPTY, METRIC_COSINE_HALF, METRIC_L2, METRIC_ONE_MINUS_DOT, BuildParameters, Hnsw, Layer,  # noqa: F401
                   SearchParameters, Stored, Unstored, VectorStore, PqStore, SharedPqStore, QuantizedHnsw,
                   stream_create_beside)
from .sharded import (EmulatedComm, GpuEngine, PythonEngine, ShardedBuilder, TorchComm, build_sharded,  # noqa: F401,E402
                      improve_index_sharded, sharded_tuning)
