"""
oracle/gen_main_golden.py — generates tests/golden/indexer_main_golden.json by running the REFERENCE's own
entry point (codd_jobs/metrics_semantic_indexer_main.py, imported from /root/reference, never copied) with fake
clients in place of ChromaDB, Redis, the agent managers and the job class.

Pins, for the build's counterpart (codd_query_engine_amd/indexer_main.py):
  parse_args (:68-169)          names, types and defaults of every flag
  initialize_clients (:172-227) connect + ping / heartbeat, exit 1 when either store is down
  run_query_mode (:250-298)     the printed report, byte for byte
  main (:301-398)               which mode runs, what reaches the job, exit codes 0 / 1 / 130

The module configures logging at import (stdout + a log file in the working directory): the root logger is given a
NullHandler first, which makes that `basicConfig` a no-op — only `print` output is captured.
TEST INFRASTRUCTURE, this container only.  Run:  python oracle/gen_main_golden.py
"""

from __future__ import annotations

import contextlib
import io
import json
import logging
import os
import sys
import types
from unittest.mock import MagicMock

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "indexer_main_golden.json")

CANNED = {
    "ids": [["prod:api#node_memory_MemFree_bytes", "prod:api#process_resident_memory_bytes"]],
    "metadatas": [[
        {"type": "gauge", "description": "Free memory in bytes", "unit": "bytes", "category": "infrastructure", "subcategory": "node",
         "category_description": "infrastructure level metrics", "golden_signal_type": "saturation",
         "golden_signal_description": "relates to saturation", "meter_type": "gauge", "meter_type_description": "gauge of free memory",
         "namespace": "prod:api"},
        {"type": "gauge", "description": "Resident memory size in bytes — “RSS”", "unit": "bytes", "category": "runtime", "subcategory": "process",
         "category_description": "runtime level metrics", "golden_signal_type": "saturation",
         "golden_signal_description": "relates to saturation", "meter_type": "gauge", "meter_type_description": "gauge of resident memory",
         "namespace": "prod:api"},
    ]],
    "distances": [[0.25, 0.40625]],
}

SCENARIOS = [
    {"name": "query_two_results", "argv": ["--namespace", "prod:api", "--query", "memory usage metrics", "--query-limit", "3"]},
    {"name": "query_default_limit", "argv": ["--namespace", "prod:api", "--query", "memory"]},
    {"name": "query_no_results", "argv": ["--namespace", "prod:api", "--query", "nothing like it"], "empty": True},
    {"name": "query_store_down", "argv": ["--namespace", "prod:api", "--query", "memory"], "heartbeat_fails": True},
    {"name": "query_search_raises", "argv": ["--namespace", "prod:api", "--query", "memory"], "query_raises": True},
    {"name": "index_defaults", "argv": ["--namespace", "prod:api"]},
    {"name": "index_all_flags", "argv": ["--namespace", "t:s", "--batch-size", "5", "--limit", "7", "--exclude-pattern", "^(go_|process_)",
                                         "--skip-if-present", "--dry-run", "--redis-host", "r", "--redis-port", "1234", "--redis-db", "2",
                                         "--promql-url", "http://p:9090", "--log-level", "DEBUG"]},
    {"name": "index_redis_down", "argv": ["--namespace", "prod:api"], "redis_down": True},
    {"name": "index_store_down", "argv": ["--namespace", "prod:api"], "heartbeat_fails": True},
    {"name": "index_job_raises", "argv": ["--namespace", "prod:api"], "job_raises": "RuntimeError"},
    {"name": "index_interrupted", "argv": ["--namespace", "prod:api"], "job_raises": "KeyboardInterrupt"},
]
PARSE_CASES = [
    ["--namespace", "a:b"],
    ["--namespace", "a:b", "--limit", "3", "--batch-size", "2", "--query", "q", "--query-limit", "50", "--skip-if-present", "--dry-run",
     "--exclude-pattern", "x", "--redis-port", "1", "--redis-db", "3", "--redis-host", "h", "--promql-url", "u", "--chromadb-host", "c",
     "--chromadb-port", "9", "--log-level", "ERROR"],
]
SHARED_FLAGS = ["namespace", "promql_url", "redis_host", "redis_port", "redis_db", "chromadb_host", "chromadb_port", "batch_size", "limit",
                "exclude_pattern", "skip_if_present", "dry_run", "query", "query_limit", "log_level"]


class FakeRedisConnectionError(Exception):
    pass


def make_fakes(sc, rec):
    class FakeCollection:
        def query(self, query_texts, n_results):
            rec["query_calls"].append({"query_texts": list(query_texts), "n_results": n_results})
            if sc.get("query_raises"):
                raise RuntimeError("index is corrupt")
            return {"ids": [[]], "metadatas": [[]], "distances": [[]]} if sc.get("empty") else CANNED

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

    return FakeClient, FakeRedis, FakeJob


def _import_main():
    sys.dont_write_bytecode = True
    logging.getLogger().addHandler(logging.NullHandler())  # the module's basicConfig becomes a no-op
    redis_stub = types.ModuleType("redis")
    redis_stub.ConnectionError = FakeRedisConnectionError
    redis_stub.Redis = MagicMock()
    sys.modules["redis"] = redis_stub
    for name in ["chromadb", "opus_agent_base", "opus_agent_base.agent", "opus_agent_base.agent.agent_builder", "opus_agent_base.config",
                 "opus_agent_base.config.config_manager", "opus_agent_base.prompt", "opus_agent_base.prompt.instructions_manager"]:
        sys.modules.setdefault(name, MagicMock())
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "codd_lib"))
    for _ in range(48):
        try:
            import codd_jobs.metrics_semantic_indexer_main as main_mod  # noqa: E402
            return main_mod
        except ModuleNotFoundError as e:
            if not e.name or e.name.startswith("codd_"):
                raise
            sys.modules[e.name] = MagicMock()
    raise RuntimeError("could not import the reference entry point")


def main():
    m = _import_main()
    out = {"generated_by": "oracle/gen_main_golden.py from codd_jobs/metrics_semantic_indexer_main.py", "canned_query_response": CANNED,
           "shared_flags": SHARED_FLAGS, "parse_cases": [], "scenarios": []}
    for argv in PARSE_CASES:
        old = sys.argv
        sys.argv = ["prog"] + argv
        try:
            ns = vars(m.parse_args())
        finally:
            sys.argv = old
        assert sorted(ns) == sorted(SHARED_FLAGS), sorted(ns)
        out["parse_cases"].append({"argv": argv, "namespace": {k: ns[k] for k in SHARED_FLAGS}})
    for sc in SCENARIOS:
        rec = {"query_calls": [], "heartbeats": 0, "collections": [], "redis_kwargs": None, "job_kwargs": None, "run_kwargs": None}
        FakeClient, FakeRedis, FakeJob = make_fakes(sc, rec)
        m.chromadb.HttpClient = lambda host, port: FakeClient()
        m.redis.Redis = FakeRedis
        m.MetricsSemanticIndexerJob = FakeJob
        old = sys.argv
        sys.argv = ["prog"] + sc["argv"]
        buf = io.StringIO()
        code = None
        try:
            with contextlib.redirect_stdout(buf):
                try:
                    m.main()
                except SystemExit as e:
                    code = e.code
        finally:
            sys.argv = old
        out["scenarios"].append({**sc, "exit_code": code, "stdout": buf.getvalue(), **rec})
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)  # (insertion order kept: the printed JSON of query mode follows the dict order of the canned metadata)
    print("wrote", os.path.normpath(OUT), len(out["scenarios"]), "scenarios")


if __name__ == "__main__":
    main()
