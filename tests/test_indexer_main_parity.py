"""Replays tests/golden/indexer_main_golden.json — captured by running the REFERENCE entry point
(codd_jobs/metrics_semantic_indexer_main.py) with fake clients, oracle/gen_main_golden.py — against this build's
codd_query_engine_amd/indexer_main.py: same flags and defaults, same mode selection, same printed query report byte for
byte, same arguments into the job, same exit codes (0 / 1 / 130).  Then both modes end to end on a real index."""

import contextlib
import io
import json
import logging
import os
import types

import pytest

from codd_query_engine_amd import KnnClient, indexer_main
from tests._oracle_engine import OracleEngine


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "indexer_main_golden.json")) as f:
        return json.load(f)


@pytest.fixture(autouse=True)
def quiet_logging():
    """main() configures logging to stdout + a file; with a handler already on the root logger that is a no-op, so stdout
    carries print() output only — as in the golden capture."""
    root = logging.getLogger()
    h = logging.NullHandler()
    root.addHandler(h)
    yield
    root.removeHandler(h)


class FakeRedisConnectionError(Exception):
    pass


def run_scenario(golden, sc, monkeypatch):
    rec = {"query_calls": [], "heartbeats": 0, "collections": [], "redis_kwargs": None, "job_kwargs": None, "run_kwargs": None}
    canned = golden["canned_query_response"]

    class FakeCollection:
        def query(self, query_texts, n_results):
            rec["query_calls"].append({"query_texts": list(query_texts), "n_results": n_results})
            if sc.get("query_raises"):
                raise RuntimeError("index is corrupt")
            return {"ids": [[]], "metadatas": [[]], "distances": [[]]} if sc.get("empty") else canned

    class FakeClient:
        def heartbeat(self):
            rec["heartbeats"] += 1
            if sc.get("heartbeat_fails"):
                raise ConnectionError("store is down")
            return 1

        def get_or_create_collection(self, name, metadata=None):
            rec["collections"].append(name)
            return FakeCollection()

    class FakeRedis:
        def __init__(self, **kw):
            rec["redis_kwargs"] = kw

        def ping(self):
            if sc.get("redis_down"):
                raise FakeRedisConnectionError("redis is down")
            return True

    class FakeJob:
        def __init__(self, **kw):
            rec["job_kwargs"] = {"batch_size": kw.get("batch_size"), "redis_is_fake": isinstance(kw.get("redis_client"), FakeRedis),
                                 "store_is_fake": isinstance(kw.get("chromadb_client"), FakeClient)}

        def run(self, **kw):
            rec["run_kwargs"] = kw
            if sc.get("job_raises") == "KeyboardInterrupt":
                raise KeyboardInterrupt()
            if sc.get("job_raises"):
                raise RuntimeError("job blew up")

    fake_redis = types.SimpleNamespace(Redis=FakeRedis, ConnectionError=FakeRedisConnectionError)
    monkeypatch.setattr(indexer_main, "make_knn_client", lambda config, device: FakeClient())
    monkeypatch.setattr(indexer_main, "_import_redis", lambda: fake_redis)
    monkeypatch.setattr(indexer_main, "MetricsSemanticIndexerJob", FakeJob)
    buf = io.StringIO()
    code = None
    with contextlib.redirect_stdout(buf):
        try:
            indexer_main.main(sc["argv"] + ["--log-file", ""])
        except SystemExit as e:
            code = e.code
    return code, buf.getvalue(), rec


def test_flags_and_defaults_match_the_reference(golden):
    for case in golden["parse_cases"]:
        ns = vars(indexer_main.parse_args(case["argv"]))
        assert {k: ns[k] for k in golden["shared_flags"]} == case["namespace"], case["argv"]
    # the flags this build adds all default to "off"
    extra = set(vars(indexer_main.parse_args(["--namespace", "a:b"]))) - set(golden["shared_flags"])
    assert extra == {"index_path", "device", "metadata_file", "log_file"}


def test_main_matches_the_reference_in_every_scenario(golden, monkeypatch):
    assert len(golden["scenarios"]) >= 11
    for sc in golden["scenarios"]:
        code, stdout, rec = run_scenario(golden, sc, monkeypatch)
        assert code == sc["exit_code"], sc["name"]
        assert stdout == sc["stdout"], sc["name"]
        for key in ("query_calls", "collections", "redis_kwargs", "job_kwargs", "run_kwargs"):
            assert rec[key] == sc[key], (sc["name"], key)
        assert rec["heartbeats"] == sc["heartbeats"], sc["name"]


def test_exit_codes_cover_done_failed_interrupted(golden):
    assert {sc["exit_code"] for sc in golden["scenarios"]} == {0, 1, 130}


PROM = {
    "http_request_duration_seconds": [{"type": "histogram", "help": "HTTP request latency in seconds"}],
    "http_requests_total": [{"type": "counter", "help": "Total number of HTTP requests"}],
    "node_memory_MemFree_bytes": [{"type": "gauge", "help": "Free memory in bytes"}],
    "go_gc_duration_seconds": [{"type": "summary", "help": "Pause duration of garbage collection cycles"}],
    "db_query_errors_total": [{"type": "counter", "help": "Failed database queries"}],
}


def _index_then_query(tmp_path, monkeypatch, make_client):
    meta = tmp_path / "metadata.json"
    meta.write_text(json.dumps({"status": "success", "data": PROM}))
    index_dir = str(tmp_path / "index")
    monkeypatch.setattr(indexer_main, "make_knn_client", make_client)
    monkeypatch.setattr(indexer_main, "_import_redis", lambda: None)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), pytest.raises(SystemExit) as done:
        indexer_main.main(["--namespace", "prod:api", "--index-path", index_dir, "--metadata-file", str(meta), "--batch-size", "2",
                           "--exclude-pattern", "go_", "--log-file", ""])
    assert done.value.code == 0
    assert "Indexed (Semantic): 4" in buf.getvalue() and "Excluded:           1" in buf.getvalue()
    # a second process: query mode on the directory the job wrote
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), pytest.raises(SystemExit) as done:
        indexer_main.main(["--namespace", "prod:api", "--index-path", index_dir, "--query", "failed database queries", "--query-limit", "2",
                           "--log-file", ""])
    assert done.value.code == 0
    text = buf.getvalue()
    assert "Found 2 result(s):" in text and "Total Results: 2" in text
    first = json.loads(text.split("Result #1:\n" + "-" * 70 + "\n")[1].split("\n" + "-" * 70)[0])
    assert first["metric_name"] == "db_query_errors_total" and first["namespace"] == "prod:api"


def test_both_modes_end_to_end_on_the_checker_engine(tmp_path, monkeypatch):
    _index_then_query(tmp_path, monkeypatch, lambda config, device: KnnClient(path=config.chromadb_path, engine_factory=lambda dim: OracleEngine(dim)))


@pytest.mark.gpu
def test_both_modes_end_to_end_on_the_hip_engine(tmp_path, monkeypatch):
    _index_then_query(tmp_path, monkeypatch, lambda config, device: KnnClient(path=config.chromadb_path, device="cuda:0"))


def test_http_metadata_source_speaks_the_prometheus_metadata_api():
    import httpx

    def handler(request):
        if request.url.path == "/-/healthy":
            return httpx.Response(200, text="Prometheus Server is Healthy.")
        if request.url.path == "/api/v1/metadata":
            return httpx.Response(200, json={"status": "success", "data": PROM})
        return httpx.Response(404)

    with indexer_main.HttpMetadataSource("http://prom:9090/", transport=httpx.MockTransport(handler)) as src:
        assert src.health_check() is True
        assert src.get_metric_metadata() == PROM
