"""Replays tests/golden/store_wrapper_golden.json — captured from the REFERENCE wrapper
(oracle/gen_store_golden.py) — against this build's MetricsSemanticMetadataStore and
MetricsSearchClient: same collection calls, same return values, same exceptions."""

import json
import os

import pytest

from codd_query_engine_amd import MetricsSearchClient, MetricsSemanticMetadataStore, ValidationError


class RecordingCollection:
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


@pytest.fixture(scope="module")
def golden(golden_dir):
    with open(os.path.join(golden_dir, "store_wrapper_golden.json")) as f:
        return json.load(f)


def outcome(fn):
    try:
        return {"ok": fn()}
    except (ValidationError, KeyError, RuntimeError) as e:
        return {"raises": type(e).__name__, "message": str(e)}


def test_constructor_calls(golden):
    c = RecordingClient()
    MetricsSemanticMetadataStore(c)
    assert c.calls == golden["ctor"]["default"]
    c2 = RecordingClient()
    MetricsSemanticMetadataStore(c2, collection_name="custom_name")
    assert c2.calls == golden["ctor"]["custom"]


def test_constructor_failure_propagates(golden):
    class Failing:
        def get_or_create_collection(self, **kw):
            raise RuntimeError("connection refused")

    assert outcome(lambda: MetricsSemanticMetadataStore(Failing())) == golden["ctor_failure"]


def test_index_metadata_matches_reference(golden):
    assert len(golden["index_metadata"]) >= 10
    for case in golden["index_metadata"]:
        c = RecordingClient()
        s = MetricsSemanticMetadataStore(c)
        assert s.index_metadata(case["namespace"], case["metadata"]) == case["returns"]
        assert c.collection.calls == case["collection_calls"], case["metadata"]


def test_index_metadata_errors_match_reference(golden):
    for case in golden["index_metadata_errors"]:
        c = RecordingClient()
        s = MetricsSemanticMetadataStore(c)
        assert outcome(lambda: s.index_metadata(case["namespace"], case["metadata"])) == case["result"], case["metadata"]
        assert c.collection.calls == case["collection_calls"]


def test_metric_exists_matches_reference(golden):
    for case in golden["metric_exists"]:
        c = RecordingClient()
        if case["canned"] == "raises":
            c.collection.raise_on_get = True
        else:
            c.collection.next_get = case["canned"]
        s = MetricsSemanticMetadataStore(c)
        assert s.metric_exists("ns", "m") is case["returns"]
        assert c.collection.calls == case["collection_calls"]


def test_search_metadata_matches_reference(golden):
    assert len(golden["search_metadata"]) >= 12
    for case in golden["search_metadata"]:
        c = RecordingClient()
        c.collection.next_query = case["canned"]
        s = MetricsSemanticMetadataStore(c)
        got = s.search_metadata(case["query"]) if case["n_results"] is None else s.search_metadata(case["query"], n_results=case["n_results"])
        assert got == case["returns"], case["query"][:40]
        assert c.collection.calls == case["collection_calls"]


def test_search_metadata_errors_match_reference(golden):
    for case in golden["search_metadata_errors"]:
        c = RecordingClient()
        c.collection.next_query = {"ids": [["ns#m1"]], "metadatas": [[{}]], "distances": [[0.5]]}
        s = MetricsSemanticMetadataStore(c)
        assert outcome(lambda: s.search_metadata(case["query"], n_results=case["n_results"])) == case["result"]
        assert c.collection.calls == case["collection_calls"]


def test_search_relevant_metrics_projection_matches_reference(golden):
    cases = golden["search_relevant_metrics"]
    assert isinstance(cases, list) and cases, "projection goldens missing"

    class FakeStore:
        def __init__(self, raw):
            self.raw, self.calls = raw, []

        def search_metadata(self, *a, **kw):
            self.calls.append({"args": list(a), "kwargs": kw})
            return self.raw

    for case in cases:
        store = FakeStore(case["raw"])
        got = MetricsSearchClient(store).search_relevant_metrics("some query", 7)
        assert got == case["returns"]
        assert store.calls == [case["store_call"]]
        for row in got:
            assert len(row) == 11 and "type" not in row and "namespace" not in row


def test_batch_extension_agrees_with_single_calls(golden):
    """search_metadata_batch is new; it must equal per-query search_metadata."""
    canned = {"ids": [["ns#a", "ns#b"], ["ns#c"]], "metadatas": [[{"description": "A"}, {"description": "B"}], [{"description": "C"}]],
              "distances": [[0.25, 0.5], [0.75]]}
    c = RecordingClient()
    c.collection.next_query = canned
    s = MetricsSemanticMetadataStore(c)
    out = s.search_metadata_batch(["first  query", "", "second\x00 query"], n_results=500)
    assert c.collection.calls == [["query", {"query_texts": ["first query", "second query"], "n_results": 100}]]
    assert out[1] == []
    assert out[0] == [{"metric_name": "a", "similarity_score": 0.75, "description": "A"}, {"metric_name": "b", "similarity_score": 0.5, "description": "B"}]
    assert out[2] == [{"metric_name": "c", "similarity_score": 0.25, "description": "C"}]
    with pytest.raises(ValidationError):
        s.search_metadata_batch(["ok", "q" * 1001])
    with pytest.raises(ValidationError):
        s.search_metadata_batch(["ok"], n_results=0)
