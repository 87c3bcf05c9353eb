"""The reference's own hot-path test scenarios, re-expressed against this build's store.

Source scenarios: /root/reference/tests/unit/codd_dal/metrics/test_metrics_semantic_metadata_store.py:39-216
and tests/integration/codd_dal/metrics/test_metrics_semantic_metadata_store_integration.py:28-175.
The reference runs them on chromadb.EphemeralClient() with the MiniLM embedder (unavailable
offline); here the client is KnnClient with the deterministic lexical embedder — on the CPU
with the checker engine (host logic) and, marked gpu, on the HIP engine through the C ABI.
"""

import pytest

from codd_query_engine_amd import KnnClient, MetricsSearchClient, MetricsSemanticMetadataStore
from tests._oracle_engine import OracleEngine


def cpu_client():
    return KnnClient(engine_factory=lambda dim: OracleEngine(dim))


def gpu_client():
    return KnnClient(device="cuda:0")


CLIENTS = [pytest.param(cpu_client, id="checker-engine"), pytest.param(gpu_client, id="hip-engine", marks=pytest.mark.gpu)]


@pytest.fixture(params=CLIENTS)
def store(request):
    return MetricsSemanticMetadataStore(request.param(), collection_name=f"test_{request.node.name}")


def test_index_metadata_basic(store):
    md = {"metric_name": "cpu.usage", "type": "gauge", "description": "CPU utilization percentage", "unit": "percent",
          "category": "system", "subcategory": "cpu"}
    assert store.index_metadata("test", md) == "test#cpu.usage"


def test_index_metadata_missing_required_field(store):
    with pytest.raises(KeyError) as exc:
        store.index_metadata("test", {"type": "gauge", "description": "Some metric"})
    assert "metric_name" in str(exc.value)


def test_index_metadata_upsert(store):
    ns, name = "test_namespace", "memory.usage"
    store.index_metadata(ns, {"metric_name": name, "description": "Memory usage version 1", "category": "system"})
    out = store.index_metadata(ns, {"metric_name": name, "description": "Memory usage version 2 updated", "category": "infrastructure"})
    assert out == f"{ns}#{name}"
    assert store.collection.count() == 1
    hits = store.search_metadata("memory usage version 2", n_results=1)
    assert len(hits) == 1 and hits[0]["metric_name"] == name
    assert "version 2" in hits[0]["description"]


def test_search_metadata_basic(store):
    store.index_metadata("test", {"metric_name": "cpu.usage", "description": "CPU utilization percentage", "category": "system"})
    store.index_metadata("test", {"metric_name": "memory.usage", "description": "Memory utilization in bytes", "category": "system"})
    hits = store.search_metadata("CPU utilization")
    assert len(hits) == 2  # min(n_results, count)
    assert hits[0]["metric_name"] == "cpu.usage"
    assert 0 <= hits[0]["similarity_score"] <= 1


def test_search_metadata_no_results(store):
    assert store.search_metadata("some query") == []


def test_search_metadata_ranking(store):
    store.index_metadata("test", {"metric_name": "http.latency", "description": "HTTP request latency in milliseconds", "golden_signal_type": "latency"})
    store.index_metadata("test", {"metric_name": "db.query.time", "description": "Database query execution time", "category": "database"})
    store.index_metadata("test", {"metric_name": "network.bandwidth", "description": "Network bandwidth usage", "category": "network"})
    hits = store.search_metadata("request latency")
    assert hits[0]["metric_name"] == "http.latency"
    scores = [h["similarity_score"] for h in hits]
    assert scores == sorted(scores, reverse=True)


def test_search_metadata_returns_all_fields(store):
    md = {"metric_name": "test.metric", "type": "gauge", "description": "Test description", "unit": "bytes",
          "category": "test_category", "subcategory": "test_subcategory", "category_description": "Category desc",
          "golden_signal_type": "throughput", "golden_signal_description": "Measures throughput",
          "meter_type": "gauge", "meter_type_description": "Gauge meter"}
    store.index_metadata("test", md)
    hit = store.search_metadata("test metric")[0]
    for key, val in md.items():
        assert hit[key] == val
    assert hit["namespace"] == "test"


def test_workflow_across_namespaces(store):
    """integration test :28-175 — search is global across namespaces; upsert keeps the id."""
    order = [
        {"metric_name": "http.request.duration.p99", "description": "99th percentile HTTP request latency and response time",
         "unit": "ms", "category": "application", "golden_signal_type": "latency"},
        {"metric_name": "http.requests.total", "description": "Total number of HTTP requests received", "category": "application",
         "golden_signal_type": "traffic"},
        {"metric_name": "cpu.utilization", "description": "CPU utilization percentage of the host", "category": "system",
         "golden_signal_type": "saturation"},
    ]
    for md in order:
        store.index_metadata("test:order_service", md)
    hits = store.search_metadata("request latency and response time", n_results=5)
    assert len(hits) >= 2
    assert hits[0]["metric_name"] == "http.request.duration.p99"
    assert hits[0]["golden_signal_type"] == "latency"
    assert 0 <= hits[0]["similarity_score"] <= 1

    store.index_metadata("test:payment_service", {"metric_name": "db.query.execution.time", "description": "Database query execution latency",
                                                  "category": "database", "golden_signal_type": "latency"})
    store.index_metadata("test:payment_service", {"metric_name": "payment.failures", "description": "Count of failed payment attempts",
                                                  "golden_signal_type": "errors"})
    names = [h["metric_name"] for h in store.search_metadata("latency", n_results=5)]
    assert "db.query.execution.time" in names[:3]

    assert store.metric_exists("test:payment_service", "payment.failures") is True
    assert store.metric_exists("test:payment_service", "nope") is False
    out = store.index_metadata("test:order_service", {**order[0], "description": "P99 latency with improved accuracy"})
    assert out == "test:order_service#http.request.duration.p99"
    top = store.search_metadata("improved accuracy latency", n_results=1)[0]
    assert top["metric_name"] == "http.request.duration.p99" and "improved accuracy" in top["description"]
    assert store.collection.count() == 5


def test_search_relevant_metrics_end_to_end(store):
    store.index_metadata("prod", {"metric_name": "api.latency.p95", "description": "API latency 95th percentile", "unit": "ms",
                                  "golden_signal_type": "latency", "type": "histogram"})
    store.index_metadata("prod", {"metric_name": "disk.free", "description": "Free disk space in bytes", "unit": "bytes"})
    rows = MetricsSearchClient(store).search_relevant_metrics("API experiencing high latency", limit=1)
    assert len(rows) == 1 and rows[0]["metric_name"] == "api.latency.p95"
    assert set(rows[0]) == {"metric_name", "similarity_score", "description", "unit", "category", "subcategory",
                            "category_description", "golden_signal_type", "golden_signal_description", "meter_type",
                            "meter_type_description"}


def test_batch_search_equals_single_searches(store):
    docs = ["CPU utilization percentage", "Memory utilization in bytes", "HTTP request latency", "Disk write throughput",
            "Network packets dropped", "Garbage collection pause time", "Queue depth of pending jobs"]
    for i, d in enumerate(docs):
        store.index_metadata("ns", {"metric_name": f"m{i}", "description": d})
    queries = ["cpu", "request latency", "", "dropped network packets", "gc pause"]
    batch = store.search_metadata_batch(queries, n_results=3)
    for q, got in zip(queries, batch):
        assert got == store.search_metadata(q, n_results=3)
