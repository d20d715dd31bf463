"""CPU-only checks of the drop-in boundary: libphnsw.so loads, exports every symbol that
include/phnsw.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import parallel_hnsw_amd as ph
from parallel_hnsw_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "phnsw.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(phnsw_[a-z0-9_]+)\s*\(", src)) - {"phnsw_progress_cb"})


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 25
    L = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), "libphnsw.so does not export %s" % n
    # and the ctypes table binds exactly the header's surface
    assert sorted(_lib.SYMBOLS) == names


def test_parameter_defaults_match_reference():
    """parameters.rs:10-18, 50-64"""
    bp = ph.BuildParameters()
    assert (bp.order, bp.zero_layer_neighborhood_size, bp.neighborhood_size) == (12, 48, 24)
    s = bp.optimization.search
    assert (s.number_of_candidates, s.upper_layer_candidate_count, s.probe_depth) == (300, 300, 2)
    i = bp.initial_partition_search
    assert (i.number_of_candidates, i.upper_layer_candidate_count, i.probe_depth) == (6, 6, 2)
    o = bp.optimization
    assert (round(o.promotion_threshold, 6), round(o.neighborhood_threshold, 6), round(o.recall_proportion, 6),
            o.promotion_proportion) == (0.01, 0.01, 0.1, 1.0)
    sp = ph.SearchParams()
    ph.lib().phnsw_default_search_params(C.byref(sp))
    assert (sp.number_of_candidates, sp.upper_layer_candidate_count, sp.probe_depth) == (300, 300, 2)


def test_struct_layouts_match_the_header():
    assert C.sizeof(ph.SearchParams) == 24
    assert C.sizeof(ph.OptimizationParams) == 40
    assert C.sizeof(ph.BuildParams) == 24 + 40 + 24 + 24


def _no_gpu():
    return ph.lib().phnsw_device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="a GPU is visible")
def test_no_cpu_fallback():
    with pytest.raises(ph.PhnswError) as e:
        ph.VectorStore(np.zeros((4, 8), dtype=np.float32))
    assert e.value.code == -2  # PHNSW_E_NO_DEVICE
    with pytest.raises(ph.PhnswError):
        ph.VectorStore.synthetic(10, 8)


def test_product_does_not_depend_on_the_oracle():
    """nothing under parallel_hnsw_amd/ or include/ may reference oracle/"""
    bad = []
    for base in ("parallel_hnsw_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r'#include\s+"[^"]*orc|\borc_\w+\s*\(|liboracle|^\s*import oracle|^\s*from oracle', txt, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
