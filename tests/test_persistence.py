"""On-disk index shared by the indexer job (writer process) and the service (reader process):
KnnClient(path=...) — SURVEY.md §8(f)1.  Host logic on the checker engine; the same flow on the
HIP engine is marked gpu (stored rows must come back bit-identical, search results unchanged)."""

import contextlib
import io
import json
import os

import numpy as np
import pytest

from codd_query_engine_amd import KnnClient, MetricsSemanticMetadataStore
from codd_query_engine_amd.indexer_job import MetricsSemanticIndexerJob, StaticMetadataSource
from tests._oracle_engine import OracleEngine


def cpu_client(path):
    return KnnClient(engine_factory=lambda dim: OracleEngine(dim), path=str(path))


def gpu_client(path):
    return KnnClient(device="cuda:0", path=str(path))


CLIENTS = [pytest.param(cpu_client, id="checker-engine"), pytest.param(gpu_client, id="hip-engine", marks=pytest.mark.gpu)]
DOCS = ["CPU utilization percentage", "Memory utilization in bytes", "HTTP request latency", "Disk write throughput",
        "Network packets dropped", "Garbage collection pause time", "Queue depth of pending jobs"]


@pytest.mark.parametrize("make", CLIENTS)
def test_writer_then_reader_round_trip(tmp_path, make):
    writer = make(tmp_path)
    store = MetricsSemanticMetadataStore(writer)
    for i, d in enumerate(DOCS):
        store.index_metadata("ns", {"metric_name": f"m{i}", "description": d, "category": "c"})
    before = store.search_metadata("request latency", n_results=4)
    rows_before = store.collection._engine.read_rows()
    assert writer.persist() == 1 and writer.persist() == 0  # second call: nothing changed

    reader = make(tmp_path)  # "the service process"
    rstore = MetricsSemanticMetadataStore(reader)  # get_or_create finds the loaded collection
    assert rstore.collection.count() == len(DOCS)
    assert np.array_equal(rstore.collection._engine.read_rows(), rows_before)
    assert rstore.search_metadata("request latency", n_results=4) == before
    assert rstore.metric_exists("ns", "m3") and not rstore.metric_exists("ns", "m99")

    # the writer publishes an update; the reader picks it up on reload()
    store.index_metadata("ns", {"metric_name": "m2", "description": "HTTP request latency p99 in milliseconds"})
    store.index_metadata("ns", {"metric_name": "m7", "description": "Thread pool saturation"})
    assert writer.persist() == 1
    assert reader.reload() == 1 and reader.reload() == 0
    assert rstore.collection.count() == len(DOCS) + 1
    assert rstore.search_metadata("request latency p99", n_results=1)[0]["metric_name"] == "m2"
    assert rstore.search_metadata("request latency p99", n_results=3) == store.search_metadata("request latency p99", n_results=3)


def test_layout_on_disk_and_generation_cleanup(tmp_path):
    c = cpu_client(tmp_path)
    s = MetricsSemanticMetadataStore(c, collection_name="metrics")
    for g in range(4):
        s.index_metadata("ns", {"metric_name": f"m{g}", "description": f"doc {g}"})
        c.persist()
    cdir = tmp_path / "metrics"
    assert (cdir / "CURRENT").read_text() == "gen-00000004"
    assert sorted(p.name for p in cdir.iterdir() if p.name.startswith("gen-")) == ["gen-00000003", "gen-00000004"]
    man = json.loads((cdir / "gen-00000004" / "manifest.json").read_text())
    assert man["count"] == 4 and man["dim"] == 384 and man["padded_dim"] == 384 and man["dtype"] == "f32"
    assert man["metadata"]["hnsw:space"] == "cosine"
    assert os.path.getsize(cdir / "gen-00000004" / "rows.bin") == 4 * 384 * 4
    c.delete_collection("metrics")
    assert not cdir.exists()


def test_job_persists_what_it_indexed(tmp_path):
    prom = {"http_request_duration_seconds": [{"type": "histogram", "help": "HTTP request latency"}],
            "node_memory_MemFree_bytes": [{"type": "gauge", "help": "Free memory"}]}
    job = MetricsSemanticIndexerJob(None, cpu_client(tmp_path), None, None, None, metadata_source=lambda cfg: StaticMetadataSource(prom))
    with contextlib.redirect_stdout(io.StringIO()):
        job.run("prod:api")
    service = MetricsSemanticMetadataStore(cpu_client(tmp_path))
    assert service.search_metadata("request latency", n_results=1)[0]["metric_name"] == "http_request_duration_seconds"


def test_empty_collection_and_unknown_format(tmp_path):
    c = cpu_client(tmp_path)
    MetricsSemanticMetadataStore(c, collection_name="empty")
    assert c.persist() == 1
    again = cpu_client(tmp_path)
    assert MetricsSemanticMetadataStore(again, collection_name="empty").search_metadata("anything") == []
    man = tmp_path / "empty" / "gen-00000001" / "manifest.json"
    data = json.loads(man.read_text())
    data["format_version"] = 99
    man.write_text(json.dumps(data))
    with pytest.raises(ValueError):
        cpu_client(tmp_path)
