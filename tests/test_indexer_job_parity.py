"""Replays tests/golden/indexer_job_golden.json — captured by running the REFERENCE indexer job
(codd_jobs/metrics_semantic_indexer_job.py) with fake collaborators, oracle/gen_job_golden.py —
against this build's MetricsSemanticIndexerJob: same counters, same calls into the stores and the
agent, same printed report, same error type and message.  Then an end-to-end run on the real store."""

import contextlib
import io
import json
import os

import pytest

from codd_query_engine_amd import KnnClient
from codd_query_engine_amd.indexer_job import (
    IndexingStats,
    MetricEnrichmentError,
    MetricsSemanticIndexerJob,
    MetricsSemanticIndexerJobError,
    StaticMetadataSource,
)
from tests._oracle_engine import OracleEngine


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "indexer_job_golden.json")) as f:
        return json.load(f)


def reference_like_enrich(metric_name, metric_type=None, description=None):
    """The fake agent the golden generator used (same outputs)."""
    if metric_name == "llm_hates_this_one":
        raise MetricEnrichmentError("model refused to answer about llm_hates_this_one, and said so at length")
    return {"metric_name": metric_name, "type": metric_type or "unknown", "description": description or f"{metric_name} (no help)",
            "unit": "seconds" if metric_name.endswith("seconds") or "seconds_" in metric_name else "",
            "category": "application" if metric_name.startswith("http") else "infrastructure",
            "subcategory": metric_name.split("_")[0], "category_description": "cat desc",
            "golden_signal_type": "latency" if "duration" in metric_name else "none",
            "golden_signal_description": "gs desc", "meter_type": metric_type or "gauge", "meter_type_description": "mt desc"}


def run_scenario(golden, sc):
    prom = dict(golden["prometheus_metadata_items"])  # a list in the file: iteration order matters
    present = set(sc.get("present", []))
    rec = {"index_calls": [], "exists_calls": [], "redis_calls": [], "enrich_calls": []}

    class Store:
        def metric_exists(self, namespace, metric_name):
            rec["exists_calls"].append([namespace, metric_name])
            return metric_name in present

        def index_metadata(self, namespace, metadata):
            rec["index_calls"].append([namespace, dict(metadata)])
            if metadata["metric_name"] == "bad name!":
                raise ValueError("metric_name contains invalid characters. Only alphanumeric, dots, dashes, underscores, and slashes are allowed")
            return f"{namespace}#{metadata['metric_name']}"

    class Names:
        def set_metric_names(self, namespace, names):
            rec["redis_calls"].append([namespace, sorted(names)])
            if sc.get("redis_fails"):
                raise ConnectionError("redis is down")

    class Agent:
        def enrich_metric_to_dict(self, metric_name, metric_type=None, description=None):
            rec["enrich_calls"].append([metric_name, metric_type, description])
            return reference_like_enrich(metric_name, metric_type, description)

    class NullClient:  # the job builds its store from the client; the scenario then swaps the store
        def get_or_create_collection(self, **kw):
            return object()

    job = MetricsSemanticIndexerJob(None, NullClient(), None, None, None, batch_size=sc.get("batch_size", 10),
                                    metadata_source=lambda cfg: StaticMetadataSource(prom, healthy=sc.get("healthy", True)),
                                    enrichment_agent=Agent(), metric_names_store=Names())
    job.semantic_store = Store()
    out = io.StringIO()
    result = {"ok": None}
    with contextlib.redirect_stdout(out):
        try:
            job.run(**sc["run"])
        except MetricsSemanticIndexerJobError as e:
            result = {"raises": type(e).__name__, "message": str(e)}
    return result, vars(job.stats), out.getvalue(), rec


def test_job_matches_reference_in_every_scenario(golden):
    assert len(golden["scenarios"]) >= 11
    for sc in golden["scenarios"]:
        result, stats, stdout, rec = run_scenario(golden, sc)
        assert result == sc["result"], sc["name"]
        assert stats == sc["stats"], sc["name"]
        for key in ("index_calls", "exists_calls", "redis_calls", "enrich_calls"):
            assert rec[key] == sc[key], (sc["name"], key)
        assert stdout == sc["stdout"], sc["name"]


def test_stats_schema_matches_reference():
    assert list(vars(IndexingStats())) == ["total_metrics", "processed_metrics", "enriched_metrics", "indexed_metrics",
                                           "failed_metrics", "skipped_metrics", "excluded_metrics"]


PROM = {
    "http_request_duration_seconds": [{"type": "histogram", "help": "HTTP request latency in seconds"}],
    "http_requests_total": [{"type": "counter", "help": "Total number of HTTP requests"}],
    "node_memory_MemFree_bytes": [{"type": "gauge", "help": "Free memory in bytes"}],
    "go_gc_duration_seconds": [{"type": "summary", "help": "Pause duration of garbage collection cycles"}],
    "db_query_errors_total": [{"type": "counter", "help": "Failed database queries"}],
}


def make_job(client):
    return MetricsSemanticIndexerJob(None, client, None, None, None, batch_size=2, metadata_source=lambda cfg: StaticMetadataSource(PROM))


def test_job_end_to_end_on_the_real_store_checker_engine():
    client = KnnClient(engine_factory=lambda dim: OracleEngine(dim))
    job = make_job(client)
    with contextlib.redirect_stdout(io.StringIO()):
        job.run("prod:api")
    assert job.stats.indexed_metrics == 5 and job.stats.failed_metrics == 0
    assert job.redis_store.names["prod:api#metric_names"] == set(PROM)
    hits = job.semantic_store.search_metadata("http request latency", n_results=2)
    assert hits[0]["metric_name"] == "http_request_duration_seconds"
    assert hits[0]["golden_signal_type"] == "latency" and hits[0]["namespace"] == "prod:api"
    # second run with skip_if_present: nothing is enriched again
    job2 = MetricsSemanticIndexerJob(None, client, None, None, None, metadata_source=lambda cfg: StaticMetadataSource(PROM))
    with contextlib.redirect_stdout(io.StringIO()):
        job2.run("prod:api", skip_if_present=True)
    assert job2.stats.skipped_metrics == 5 and job2.stats.enriched_metrics == 0
    assert job.semantic_store.collection.count() == 5


@pytest.mark.gpu
def test_job_end_to_end_on_the_hip_engine():
    client = KnnClient(device="cuda:0")
    job = make_job(client)
    with contextlib.redirect_stdout(io.StringIO()):
        job.run("prod:api", exclude_pattern="go_")
    assert job.stats.indexed_metrics == 4 and job.stats.excluded_metrics == 1
    hits = job.semantic_store.search_metadata("failed database queries", n_results=1)
    assert hits[0]["metric_name"] == "db_query_errors_total"


def test_real_store_is_fed_one_upsert_per_batch():
    """the reference upserts one row per call (store.py:236); this job validates per metric and sends each
    batch to the engine in ONE upsert.  Validation failures still surface per metric, at the same place in the report."""
    upserts = []

    class CountingEngine(OracleEngine):
        def upsert(self, slots, vecs, normalize=True):
            upserts.append(len(slots))
            return super().upsert(slots, vecs, normalize)

    prom = dict(PROM)
    prom["bad name!"] = [{"type": "gauge", "help": "fails store validation"}]
    client = KnnClient(engine_factory=lambda dim: CountingEngine(dim))
    job = MetricsSemanticIndexerJob(None, client, None, None, None, batch_size=4, metadata_source=lambda cfg: StaticMetadataSource(prom))
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        job.run("prod:api")
    assert upserts == [4, 1], upserts           # 6 metrics in batches of 4 + 2; the bad name never reaches the engine
    assert job.stats.indexed_metrics == 5 and job.stats.failed_metrics == 1 and job.stats.enriched_metrics == 6
    assert "→ Enriching: bad name! ✗ (error: metric_name contains invalid characters" in out.getvalue()
    assert job.semantic_store.collection.count() == 5


def test_a_failing_commit_reports_one_outcome_per_metric():
    """with the two-phase ingest the ✓ line used to be printed (and indexed_metrics incremented) before the batch was
    stored; a metric whose upsert then failed showed ✓ and a separate ✗.  The reference prints exactly one outcome per metric
    at that position (job.py:409-459)."""
    class FlakyEngine(OracleEngine):
        def upsert(self, slots, vecs, normalize=True):
            if len(slots) > 1:
                raise RuntimeError("device lost during the batched upsert")       # the batch as a whole is rejected ...
            if FlakyEngine.calls == 1:
                FlakyEngine.calls += 1
                raise RuntimeError("HBM error on this row")                        # ... and one of its metrics again on the retry
            FlakyEngine.calls += 1
            return super().upsert(slots, vecs, normalize)
    FlakyEngine.calls = 0

    client = KnnClient(engine_factory=lambda dim: FlakyEngine(dim))
    job = MetricsSemanticIndexerJob(None, client, None, None, None, batch_size=3, metadata_source=lambda cfg: StaticMetadataSource(dict(list(PROM.items())[:3])))
    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        job.run("prod:api")
    text = out.getvalue()
    lines = [l for l in text.splitlines() if "→ Enriching:" in l]
    assert len(lines) == 3 and all((" ✓ " in l) != (" ✗ " in l) for l in lines), text     # one outcome each, on the metric's own line
    assert sum(" ✗ (error: HBM error on this row" in l for l in lines) == 1
    assert job.stats.indexed_metrics == 2 and job.stats.failed_metrics == 1 and job.stats.enriched_metrics == 3
    assert job.semantic_store.collection.count() == 2


def test_an_interrupt_between_prepare_and_commit_counts_nothing_as_indexed():
    class Interrupting(OracleEngine):
        def upsert(self, slots, vecs, normalize=True):
            raise KeyboardInterrupt()

    client = KnnClient(engine_factory=lambda dim: Interrupting(dim))
    job = MetricsSemanticIndexerJob(None, client, None, None, None, batch_size=2, metadata_source=lambda cfg: StaticMetadataSource(PROM))
    out = io.StringIO()
    with contextlib.redirect_stdout(out), pytest.raises(KeyboardInterrupt):
        job.run("prod:api")
    assert job.stats.indexed_metrics == 0 and job.stats.enriched_metrics == 2
    assert out.getvalue().count("✗ (interrupted before the batch was stored)") == 2 and " ✓ (category" not in out.getvalue()
