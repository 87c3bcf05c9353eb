"""
oracle/gen_job_golden.py — generates tests/golden/indexer_job_golden.json by running the
REFERENCE's own indexer job (imported from /root/reference, never copied) with fake
collaborators in place of Prometheus, Redis, the LLM agent and ChromaDB.

Pins the ingest harness of the path (last row):
  codd_jobs/metrics_semantic_indexer_job.py
      run (:107-214), _filter_metrics_by_pattern (:216-257), _fetch_metrics_from_prometheus
      (:259-305), _update_redis_store (:307-327), _process_batch (:359-461), _print_summary (:463-485)
— which counter moves on which branch, the first-metadata-entry rule, limit applied while
iterating, `re.match` exclusion, what reaches the stores, the printed report, the error type.

Third-party modules the job only names in imports (chromadb, redis, opus_agent_base, httpx
consumers ...) are pre-seeded as inert stubs.  TEST INFRASTRUCTURE, this container only.
Run:  python oracle/gen_job_golden.py
"""

from __future__ import annotations

import contextlib
import io
import json
import os
import sys
from unittest.mock import MagicMock

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "indexer_job_golden.json")


def _import_job_module():
    sys.dont_write_bytecode = True
    for name in ["chromadb", "opus_agent_base", "opus_agent_base.agent", "opus_agent_base.agent.agent_builder",
                 "opus_agent_base.config", "opus_agent_base.config.config_manager", "opus_agent_base.prompt",
                 "opus_agent_base.prompt.instructions_manager"]:
        sys.modules.setdefault(name, MagicMock())
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "codd_lib"))
    for _ in range(48):
        try:
            import codd_jobs.metrics_semantic_indexer_job as job  # noqa: E402
            return job
        except ModuleNotFoundError as e:
            if not e.name or e.name.startswith("codd_"):
                raise
            sys.modules[e.name] = MagicMock()
    raise RuntimeError("could not import the reference job")


PROM = {
    "http_request_duration_seconds": [{"type": "histogram", "help": "HTTP request latency"},
                                      {"type": "gauge", "help": "second entry is ignored"}],
    "go_gc_duration_seconds": [{"type": "summary", "help": "A summary of the pause duration of garbage collection cycles that is long enough to be cut"}],
    "process_cpu_seconds_total": [{"type": "counter", "help": "Total user and system CPU time spent in seconds."}],
    "empty_entries_metric": [],
    "up": [{"type": "gauge"}],
    "node_memory_MemFree_bytes": [{"help": "Memory information field MemFree_bytes."}],
    "bad name!": [{"type": "gauge", "help": "fails store validation"}],
    "llm_hates_this_one": [{"type": "gauge", "help": "enrichment raises"}],
}

SCENARIOS = [
    {"name": "plain", "run": {"namespace": "prod:api"}},
    {"name": "dry_run", "run": {"namespace": "prod:api", "dry_run": True}},
    {"name": "limit_3", "run": {"namespace": "prod:api", "limit": 3}},
    {"name": "exclude_go_and_process", "run": {"namespace": "prod:api", "exclude_pattern": "go_|process_"}},
    {"name": "exclude_is_match_not_search", "run": {"namespace": "prod:api", "exclude_pattern": "memory"}},
    {"name": "skip_if_present", "run": {"namespace": "prod:api", "skip_if_present": True},
     "present": ["up", "http_request_duration_seconds"]},
    {"name": "batch_size_3", "batch_size": 3, "run": {"namespace": "t:s", "limit": 7}},
    {"name": "bad_regex", "run": {"namespace": "prod:api", "exclude_pattern": "("}},
    {"name": "unhealthy_prometheus", "run": {"namespace": "prod:api"}, "healthy": False},
    {"name": "redis_down", "run": {"namespace": "prod:api"}, "redis_fails": True},
    {"name": "everything_excluded", "run": {"namespace": "prod:api", "exclude_pattern": ".*"}},
]


def main():
    job = _import_job_module()

    def enrich(metric_name, metric_type=None, description=None):
        if metric_name == "llm_hates_this_one":
            raise job.MetricEnrichmentError("model refused to answer about llm_hates_this_one, and said so at length")
        return {"metric_name": metric_name, "type": metric_type or "unknown", "description": description or f"{metric_name} (no help)",
                "unit": "seconds" if metric_name.endswith("seconds") or "seconds_" in metric_name else "",
                "category": "application" if metric_name.startswith("http") else "infrastructure",
                "subcategory": metric_name.split("_")[0], "category_description": "cat desc",
                "golden_signal_type": "latency" if "duration" in metric_name else "none",
                "golden_signal_description": "gs desc", "meter_type": metric_type or "gauge", "meter_type_description": "mt desc"}

    golden = {"_generated_by": "oracle/gen_job_golden.py", "prometheus_metadata_items": [[k, v] for k, v in PROM.items()], "scenarios": []}
    for sc in SCENARIOS:
        healthy = sc.get("healthy", True)
        present = set(sc.get("present", []))
        rec = {"index_calls": [], "exists_calls": [], "redis_calls": [], "enrich_calls": []}

        class FakePromQLClient:
            def __init__(self, config=None):
                self.config = config

            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

            def health_check(self):
                return healthy

            def get_metric_metadata(self):
                return PROM

        class FakeSemanticStore:
            def metric_exists(self, namespace, metric_name):
                rec["exists_calls"].append([namespace, metric_name])
                return metric_name in present

            def index_metadata(self, namespace, metadata):
                rec["index_calls"].append([namespace, dict(metadata)])
                if metadata["metric_name"] == "bad name!":
                    raise ValueError("metric_name contains invalid characters. Only alphanumeric, dots, dashes, underscores, and slashes are allowed")
                return f"{namespace}#{metadata['metric_name']}"

        class FakeRedisStore:
            def set_metric_names(self, namespace, names):
                rec["redis_calls"].append([namespace, sorted(names)])
                if sc.get("redis_fails"):
                    raise ConnectionError("redis is down")

        class FakeAgent:
            def enrich_metric_to_dict(self, metric_name, metric_type=None, description=None):
                rec["enrich_calls"].append([metric_name, metric_type, description])
                return enrich(metric_name, metric_type, description)

        job.PromQLClient = FakePromQLClient
        j = job.MetricsSemanticIndexerJob.__new__(job.MetricsSemanticIndexerJob)
        j.prometheus_config = MagicMock(base_url="http://fake:9090")
        j.batch_size = sc.get("batch_size", 10)
        j.promql_client = None
        j.redis_store = FakeRedisStore()
        j.semantic_store = FakeSemanticStore()
        j.enrichment_agent = FakeAgent()
        j.stats = job.IndexingStats()

        out = io.StringIO()
        result = {"ok": None}
        with contextlib.redirect_stdout(out):
            try:
                j.run(**sc["run"])
            except Exception as e:  # noqa: BLE001
                result = {"raises": type(e).__name__, "message": str(e)}
        golden["scenarios"].append({**sc, "result": result, "stats": dict(vars(j.stats)), "stdout": out.getvalue(), **rec})

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(golden, f, indent=1, ensure_ascii=False, sort_keys=True)
    print("wrote", os.path.normpath(OUT), len(golden["scenarios"]), "scenarios")


if __name__ == "__main__":
    main()
