"""REST / MCP / CLI shims above the store: same shapes as the reference
(metrics_controller.py:47-58,113-135; server.py:48-88; commands/metrics.py:26-67)."""

import asyncio

import pytest
from fastapi.testclient import TestClient

from codd_query_engine_amd import KnnClient, MetricsSearchClient, MetricsSemanticMetadataStore
from codd_query_engine_amd.wire import cli_main, create_app, make_search_relevant_metrics_tool
from tests._oracle_engine import OracleEngine


def _client(kind):
    if kind == "hip-engine":
        return KnnClient(device="cuda:0")
    return KnnClient(engine_factory=lambda dim: OracleEngine(dim))


@pytest.fixture(params=["checker-engine", pytest.param("hip-engine", marks=pytest.mark.gpu)])
def search_client(request):
    """Every shim test runs on the checker engine (CPU) and, on the GPU box, on the HIP engine: route -> store -> C ABI."""
    store = MetricsSemanticMetadataStore(_client(request.param))
    store.index_metadata("prod", {"metric_name": "http_request_duration_seconds", "description": "HTTP request latency in seconds",
                                  "category": "application", "golden_signal_type": "latency", "type": "histogram"})
    store.index_metadata("prod", {"metric_name": "node_memory_MemFree_bytes", "description": "Free memory in bytes", "category": "infrastructure"})
    store.index_metadata("prod", {"metric_name": "http_requests_total", "description": "Total HTTP requests", "category": "application"})
    return MetricsSearchClient(store)


def test_rest_search_shape_and_limits(search_client):
    api = TestClient(create_app(lambda: search_client))
    r = api.post("/api/metrics/search", json={"query": "API high latency", "limit": 2})
    assert r.status_code == 200
    body = r.json()
    assert set(body) == {"results", "count"} and body["count"] == len(body["results"]) <= 2
    assert body["results"][0]["metric_name"] == "http_request_duration_seconds"
    assert set(body["results"][0]) == {"metric_name", "similarity_score", "description", "unit", "category", "subcategory",
                                       "category_description", "golden_signal_type", "golden_signal_description", "meter_type",
                                       "meter_type_description"}
    assert api.post("/api/metrics/search", json={"query": "latency"}).json()["count"] == 3  # default limit 5, 3 stored
    assert api.post("/api/metrics/search", json={"limit": 3}).status_code == 422           # missing query
    assert api.post("/api/metrics/search", json={"query": "", "limit": 3}).json() == {"results": [], "count": 0}


def test_rest_maps_failures_to_500(search_client):
    api = TestClient(create_app(lambda: search_client))
    r = api.post("/api/metrics/search", json={"query": "x", "limit": 0})  # store raises ValidationError
    assert r.status_code == 500 and "n_results must be at least 1" in r.json()["detail"]


def test_mcp_tool_forwards_and_swallows(search_client, capsys):
    api = TestClient(create_app(lambda: search_client))
    tool = make_search_relevant_metrics_tool(lambda endpoint, js: api.post(endpoint, json=js).json())
    out = asyncio.run(tool("API experiencing high latency", 1))
    assert [r["metric_name"] for r in out] == ["http_request_duration_seconds"]
    assert asyncio.run(tool("memory"))[0]["metric_name"] == "node_memory_MemFree_bytes"  # default limit 5

    def broken(endpoint, js):
        raise ConnectionError("service down")

    assert asyncio.run(make_search_relevant_metrics_tool(broken)("anything")) == []
    assert "Error searching metrics: service down" in capsys.readouterr().out


def test_cli_table_and_exit_codes(search_client, capsys):
    assert cli_main(["get-semantic-metrics", "request latency", "--limit", "2"], search_client) == 0
    out = capsys.readouterr().out
    assert "Semantic Search Results (Top 2)" in out and "http_request_duration_seconds" in out and "application" in out
    assert cli_main(["get-semantic-metrics", "   "], search_client) == 0
    assert "No metrics found matching your query." in capsys.readouterr().out
    assert cli_main(["get-semantic-metrics", "x", "--limit", "0"], search_client) == 1
    assert "Error:" in capsys.readouterr().out
