"""The C-ABI library loads here (no GPU) and exports every symbol include/codd_knn.h
declares; argument validation that needs no device is exercised too.  No compute."""

import ctypes
import os
import re

import pytest

from codd_query_engine_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "codd_knn.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(codd_knn_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    assert declared_symbols() == sorted(name for name, _, _ in native.ABI)


def test_library_loads_and_exports_every_declared_symbol():
    lib = native.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.codd_knn_version().decode().startswith("codd_knn ")
    assert "gfx950" in lib.codd_knn_version().decode()


def test_error_codes_without_a_device():
    lib = native.load()
    h = ctypes.c_void_p()
    # argument validation happens before any HIP call
    assert lib.codd_knn_create(ctypes.byref(h), 0, 0, 0, 0) == -22
    assert "dim" in native.last_error()
    assert lib.codd_knn_create(ctypes.byref(h), 0, 768, 7, 0) == -22
    assert lib.codd_knn_create(ctypes.byref(h), 0, 768, 0, 3) == -95
    assert lib.codd_knn_create(None, 0, 768, 0, 0) == -22
    assert lib.codd_knn_destroy(None) == 0
    out = ctypes.c_int64()
    assert lib.codd_knn_count(None, ctypes.byref(out)) == -22
    assert lib.codd_knn_search(None, None, 1, 1, None, None, None) == -22
    assert lib.codd_knn_merge_keys(0, None, 1, 1, 1, None, None, None, None) == -22
    assert lib.codd_knn_merge_shards(0, None, 1, 1, 1, 1, None, None, None, None) == -22
    with pytest.raises(native.NativeLibraryError):
        native.check(-22, "demo")


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from codd_query_engine_amd import KnnClient, MetricsSemanticMetadataStore

    store = MetricsSemanticMetadataStore(KnnClient())
    with pytest.raises(native.NativeLibraryError):
        store.index_metadata("ns", {"metric_name": "cpu.usage", "description": "CPU utilization"})
