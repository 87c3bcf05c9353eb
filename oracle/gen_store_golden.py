"""
oracle/gen_store_golden.py — generates tests/golden/store_wrapper_golden.json by
running the REFERENCE's own Python wrapper (never copied, only imported from
/root/reference in this container) against a recording fake ChromaDB collection.

It pins everything the reference does AROUND the k-NN call on the
search_relevant_metrics path:

  * codd_dal/metrics/metrics_semantic_metadata_store.py
      __init__ (:43-75), index_metadata (:138-245), metric_exists (:247-264),
      search_metadata (:266-341)
  * codd_lib/codd_lib/client/metrics_promql_client.py  search_relevant_metrics (:71-107)

`chromadb` and `opus_agent_base` are not installed here; they are only *named* by
the wrapper's imports (type annotations / sibling agents), so they are pre-seeded
in sys.modules as inert MagicMock modules.  The k-NN arithmetic itself is NOT
exercised by this script (it lives inside chromadb): numeric parity stays
"unpinned", see oracle/knn_oracle.c.

Run:  python oracle/gen_store_golden.py      (TEST INFRASTRUCTURE, this container only;
the GPU box has no /root/reference and uses the committed JSON).
"""

from __future__ import annotations

import json
import os
import sys
from unittest.mock import MagicMock

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "store_wrapper_golden.json")


def _import_reference():
    sys.dont_write_bytecode = True
    for name in [
        "chromadb",
        "opus_agent_base",
        "opus_agent_base.agent",
        "opus_agent_base.agent.agent_builder",
        "opus_agent_base.config",
        "opus_agent_base.config.config_manager",
        "opus_agent_base.prompt",
        "opus_agent_base.prompt.instructions_manager",
    ]:
        sys.modules.setdefault(name, MagicMock())
    sys.path.insert(0, REF)
    from codd_dal.metrics.metrics_semantic_metadata_store import MetricsSemanticMetadataStore  # noqa: E402
    from codd_engine.validation_engine.metrics.validation_result import ValidationError  # noqa: E402

    return MetricsSemanticMetadataStore, ValidationError


class RecordingCollection:
    """Stands where chromadb's Collection stands; records every call verbatim."""

    def __init__(self):
        self.calls = []
        self.next_query = None
        self.next_get = None
        self.raise_on_get = False

    def upsert(self, **kw):
        self.calls.append(["upsert", kw])

    def get(self, **kw):
        self.calls.append(["get", kw])
        if self.raise_on_get:
            raise RuntimeError("boom")
        return self.next_get

    def query(self, **kw):
        self.calls.append(["query", kw])
        return self.next_query


class RecordingClient:
    def __init__(self):
        self.calls = []
        self.collection = RecordingCollection()

    def get_or_create_collection(self, **kw):
        self.calls.append(["get_or_create_collection", kw])
        return self.collection


def _exc(fn):
    try:
        return {"ok": fn()}
    except Exception as e:  # noqa: BLE001 - we record the type the reference raises
        return {"raises": type(e).__name__, "message": str(e)}


INDEX_RECORDS = [
    ["test", {"metric_name": "cpu.usage", "type": "gauge", "description": "CPU utilization percentage",
              "unit": "percent", "category": "system", "subcategory": "cpu"}],
    ["test", {"metric_name": "http.request.duration", "type": "histogram",
              "description": "HTTP request duration in milliseconds", "unit": "ms", "category": "application",
              "subcategory": "http", "category_description": "Application-level metrics",
              "golden_signal_type": "latency", "golden_signal_description": "Measures request latency",
              "meter_type": "histogram", "meter_type_description": "Distribution of values over time"}],
    ["ns", {"metric_name": "http.latency", "description": "HTTP   request\x00 latency",
            "golden_signal_type": "latency", "category": "app"}],
    ["ns", {"metric_name": "only_name"}],
    ["ns", {"metric_name": "labels.only", "category": "c", "subcategory": "s", "meter_type": "g"}],
    ["prod:order_service", {"metric_name": "db/query-time_p99", "description": "  leading and trailing  ",
                            "unit": None, "category": "", "meter_type": "\tTimer\n"}],
    ["ns", {"metric_name": "métrique.latence", "description": "latence des requêtes"}],
    ["ns", {"metric_name": 12345, "description": 678}],
    ["", {"metric_name": "empty.namespace", "description": "d"}],
    ["ns", {"metric_name": "extra.keys", "description": "d", "not_a_field": "ignored", "namespace": "spoof"}],
    ["ns", {"metric_name": "x" * 255, "description": "y" * 2000}],
]

INDEX_ERRORS = [
    ["ns", {"type": "gauge", "description": "Some metric"}],
    ["ns", {"metric_name": ""}],
    ["ns", {"metric_name": "x" * 256}],
    ["ns", {"metric_name": "has space"}],
    ["ns", {"metric_name": "semi;colon"}],
    ["ns", {"metric_name": "a#b"}],
    ["ns", {"metric_name": "ok", "description": "y" * 2001}],
    ["ns", {"metric_name": "ok", "meter_type_description": "z" * 2001}],
    ["ns", {"metric_name": "ok\n"}],
]

MD_A = {"type": "gauge", "description": "CPU utilization percentage", "unit": "percent", "category": "system",
        "subcategory": "cpu", "category_description": "", "golden_signal_type": "", "golden_signal_description": "",
        "meter_type": "", "meter_type_description": "", "namespace": "test"}
MD_B = {"type": "", "description": "Memory utilization in bytes", "unit": "", "category": "system",
        "subcategory": "", "category_description": "", "golden_signal_type": "", "golden_signal_description": "",
        "meter_type": "", "meter_type_description": "", "namespace": "test"}

SEARCH_CASES = [
    # [query, n_results or None (default), canned collection.query response]
    ["CPU utilization", None, {"ids": [["test#cpu.usage", "test#memory.usage"]], "metadatas": [[MD_A, MD_B]],
                               "distances": [[0.25, 0.5]]}],
    ["  CPU \x00  utilization\n\n now ", 3, {"ids": [["test#cpu.usage"]], "metadatas": [[MD_A]], "distances": [[0.125]]}],
    ["anything", 5, {"ids": [[]], "metadatas": [[]], "distances": [[]]}],
    ["anything", 5, None],
    ["anything", 5, {}],
    ["anything", 5, {"ids": [["no_hash_id", "a#b#c"]], "metadatas": [[{"k": "v"}, {}]], "distances": [[0.0, 1.0]]}],
    ["missing distance", 2, {"ids": [["ns#m1", "ns#m2"]], "metadatas": [[MD_A, MD_B]], "distances": [[0.5]]}],
    ["missing metadata", 2, {"ids": [["ns#m1", "ns#m2"]], "metadatas": [[MD_A]], "distances": [[0.5, 0.75]]}],
    ["no metadatas key", 2, {"ids": [["ns#m1"]]}],
    ["override", 1, {"ids": [["ns#real"]], "metadatas": [[{"metric_name": "spoofed", "similarity_score": 42.0}]],
                     "distances": [[0.25]]}],
    ["cap", 200, {"ids": [["ns#m1"]], "metadatas": [[MD_A]], "distances": [[0.5]]}],
    ["cap-exact", 100, {"ids": [["ns#m1"]], "metadatas": [[MD_A]], "distances": [[0.5]]}],
    ["neg distance", 1, {"ids": [["ns#m1"]], "metadatas": [[MD_A]], "distances": [[-1.1920929e-07]]}],
    ["q" * 1000, 1, {"ids": [["ns#m1"]], "metadatas": [[MD_A]], "distances": [[0.5]]}],
    ["  " + "q" * 1000 + "  ", 1, {"ids": [["ns#m1"]], "metadatas": [[MD_A]], "distances": [[0.5]]}],
]

SEARCH_ERRORS = [
    ["", 5], ["   \n\t", 5], [None, 5], ["q" * 1001, 5], ["ok", 0], ["ok", -3],
    ["a " * 600, 5],
]

PROJECTION_INPUTS = [
    [],
    [{"metric_name": "cpu.usage", "similarity_score": 0.75, **MD_A}],
    [{"metric_name": "m", "similarity_score": 0.5}, {"similarity_score": 0.25, "description": "d", "extra": 1}],
    [{}],
]


def main() -> None:
    Store, ValidationError = _import_reference()
    golden: dict = {"_generated_by": "oracle/gen_store_golden.py", "_reference": "sathish316/codd_query_engine @ /root/reference"}

    # --- constructor -----------------------------------------------------------------
    c = RecordingClient()
    Store(c)
    c2 = RecordingClient()
    Store(c2, collection_name="custom_name")
    golden["ctor"] = {"default": c.calls, "custom": c2.calls}

    class FailingClient:
        def get_or_create_collection(self, **kw):
            raise RuntimeError("connection refused")

    golden["ctor_failure"] = _exc(lambda: Store(FailingClient()))

    # --- index_metadata -------------------------------------------------------------
    idx = []
    for ns, md in INDEX_RECORDS:
        c = RecordingClient()
        s = Store(c)
        ret = s.index_metadata(ns, md)
        idx.append({"namespace": ns, "metadata": md, "returns": ret, "collection_calls": c.collection.calls})
    golden["index_metadata"] = idx

    errs = []
    for ns, md in INDEX_ERRORS:
        c = RecordingClient()
        s = Store(c)
        r = _exc(lambda: s.index_metadata(ns, md))
        errs.append({"namespace": ns, "metadata": md, "result": r, "collection_calls": c.collection.calls})
    golden["index_metadata_errors"] = errs

    # --- metric_exists ----------------------------------------------------------------
    ex = []
    for canned in [{"ids": ["ns#m"]}, {"ids": []}, None, {}, {"ids": None}]:
        c = RecordingClient()
        c.collection.next_get = canned
        s = Store(c)
        ex.append({"canned": canned, "returns": s.metric_exists("ns", "m"), "collection_calls": c.collection.calls})
    c = RecordingClient()
    c.collection.raise_on_get = True
    ex.append({"canned": "raises", "returns": Store(c).metric_exists("ns", "m"), "collection_calls": c.collection.calls})
    golden["metric_exists"] = ex

    # --- search_metadata --------------------------------------------------------------
    sr = []
    for query, n, canned in SEARCH_CASES:
        c = RecordingClient()
        c.collection.next_query = canned
        s = Store(c)
        out = s.search_metadata(query) if n is None else s.search_metadata(query, n_results=n)
        sr.append({"query": query, "n_results": n, "canned": canned, "returns": out,
                   "collection_calls": c.collection.calls})
    golden["search_metadata"] = sr

    se = []
    for query, n in SEARCH_ERRORS:
        c = RecordingClient()
        c.collection.next_query = {"ids": [["ns#m1"]], "metadatas": [[MD_A]], "distances": [[0.5]]}
        s = Store(c)
        r = _exc(lambda: s.search_metadata(query, n_results=n))
        se.append({"query": query, "n_results": n, "result": r, "collection_calls": c.collection.calls})
    golden["search_metadata_errors"] = se

    # --- MetricsPromQLClient.search_relevant_metrics projection ---------------------
    proj = []
    try:
        sys.path.insert(0, os.path.join(REF, "codd_lib"))
        # the client module drags in every provider (redis, lark, ...): none is on the
        # projection's code path, so each absent third-party name becomes an inert stub
        for _ in range(32):
            try:
                from codd_lib.client.metrics_promql_client import MetricsPromQLClient  # noqa: E402
                break
            except ModuleNotFoundError as e:
                if not e.name or e.name.startswith("codd_"):
                    raise
                sys.modules[e.name] = MagicMock()

        for raw in PROJECTION_INPUTS:
            fake_self = MagicMock()
            fake_self.semantic_metadata_store.search_metadata.return_value = raw
            out = MetricsPromQLClient.search_relevant_metrics(fake_self, "some query", 7)
            call = fake_self.semantic_metadata_store.search_metadata.call_args
            proj.append({"raw": raw, "returns": out, "store_call": {"args": list(call.args), "kwargs": dict(call.kwargs)}})
        golden["search_relevant_metrics"] = proj
    except Exception as e:  # ordinary import error (missing third-party deps): record and go on
        golden["search_relevant_metrics"] = {"unavailable": f"{type(e).__name__}: {e}"}

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(golden, f, indent=1, ensure_ascii=False, sort_keys=True)
    print("wrote", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
