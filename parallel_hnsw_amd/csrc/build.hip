#include "phnsw_internal.h"
// placeholder until the GPU build lands (next commit)
#define STUB(name, ...) extern "C" int name(__VA_ARGS__) { ph_set_error(#name ": not implemented yet"); return PHNSW_E_UNSUPPORTED; }
STUB(phnsw_build, phnsw_store *, const uint64_t *, uint64_t, const phnsw_build_params *, phnsw_progress_cb, void *, phnsw_index **)
STUB(phnsw_generate_layer, phnsw_index *, const uint64_t *, uint64_t, uint64_t, const phnsw_build_params *)
STUB(phnsw_link_layer, phnsw_index *, uint32_t, const phnsw_search_params *, uint64_t, uint64_t *)
STUB(phnsw_improve_index, phnsw_index *, const phnsw_build_params *, phnsw_progress_cb, void *, float *)
STUB(phnsw_improve_neighbors_upto, phnsw_index *, uint32_t, const phnsw_build_params *, float, float *)
STUB(phnsw_stochastic_recall_at, phnsw_index *, uint32_t, const phnsw_optimization_params *, float *)
