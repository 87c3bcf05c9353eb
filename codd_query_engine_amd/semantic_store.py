"""
MetricsSemanticMetadataStore — the host-side operator of the search_relevant_metrics path,
same class name, constructor, three public methods, return shapes and error behaviour as
the reference's codd_dal/metrics/metrics_semantic_metadata_store.py, so that
`PromQLModule.get_semantic_store` (codd_lib/.../provider/promql_module.py:49-67) can hand
out this class unchanged.  Behaviour is pinned by tests/golden/store_wrapper_golden.json,
captured from the reference wrapper itself (oracle/gen_store_golden.py).

The client object is duck-typed exactly as the reference duck-types chromadb: anything with
`get_or_create_collection(name=, metadata=)` returning an object with upsert / get / query.
In this build that object is codd_query_engine_amd.knn_client.KnnClient (HIP engine).

One extension, needed for batched configs: `search_metadata_batch`.
"""

from __future__ import annotations

import logging
import re
from typing import Any, Iterable

from .errors import ValidationError
from .models import MetricMetadata

logger = logging.getLogger(__name__)

# limits: reference store.py:20-24
MAX_METRIC_NAME_LENGTH = 255
MAX_TEXT_FIELD_LENGTH = 2000
MAX_QUERY_LENGTH = 1000
MAX_BULK_OPERATIONS = 1000
MAX_N_RESULTS = 100

# reference store.py:28 — note `.match` + `$`: a single trailing newline passes (golden pins it)
METRIC_NAME_PATTERN = re.compile(r"^[\w._\-/]+$", re.UNICODE)

# the ten stored text fields, in the reference's order (store.py:166-177)
TEXT_FIELDS = (
    "type",
    "description",
    "unit",
    "category",
    "subcategory",
    "category_description",
    "golden_signal_type",
    "golden_signal_description",
    "meter_type",
    "meter_type_description",
)

# labelled parts of the searchable document, in order (store.py:202-207)
DOCUMENT_LABELS = (
    ("category", "Category"),
    ("subcategory", "Subcategory"),
    ("golden_signal_type", "Golden Signal"),
    ("meter_type", "Meter Type"),
)

# collection settings the reference hard-codes (store.py:63-68). The exhaustive engine only
# consumes hnsw:space; the graph parameters are accepted and recorded for parity of the call.
COLLECTION_METADATA = {
    "hnsw:space": "cosine",
    "hnsw:construction_ef": 200,
    "hnsw:search_ef": 100,
    "hnsw:M": 16,
}

_WS = re.compile(r"\s+")


class MetricsSemanticMetadataStore:
    """Index and search metric metadata by semantic similarity.

    Args:
        chromadb_client: client object (KnnClient here; chromadb.Client in the reference)
        collection_name: collection to open or create
    """

    def __init__(self, chromadb_client: Any, collection_name: str = "metrics_semantic_metadata"):
        self.chromadb_client = chromadb_client
        self.collection_name = collection_name
        try:
            self.collection = chromadb_client.get_or_create_collection(
                name=collection_name, metadata=dict(COLLECTION_METADATA)
            )
        except Exception as exc:
            logger.error(f"Failed to initialize collection '{collection_name}': {exc}")
            raise
        logger.info(f"Initialized collection '{collection_name}'")

    # ------------------------------------------------------------------ validation helpers
    def _validate_metric_name(self, metric_name: str) -> None:
        if not metric_name:
            raise ValidationError("metric_name cannot be empty")
        if len(metric_name) > MAX_METRIC_NAME_LENGTH:
            raise ValidationError(f"metric_name exceeds maximum length of {MAX_METRIC_NAME_LENGTH} characters")
        if METRIC_NAME_PATTERN.match(metric_name) is None:
            raise ValidationError(
                "metric_name contains invalid characters. "
                "Only alphanumeric, dots, dashes, underscores, and slashes are allowed"
            )

    def _validate_text_field(self, field_name: str, field_value: str) -> None:
        if field_value and len(field_value) > MAX_TEXT_FIELD_LENGTH:
            raise ValidationError(f"{field_name} exceeds maximum length of {MAX_TEXT_FIELD_LENGTH} characters")

    def _sanitize_text(self, text: str) -> str:
        """Drop NULs, trim, collapse whitespace runs to one space (store.py:117-136)."""
        if not text:
            return ""
        return _WS.sub(" ", text.replace("\x00", "").strip())

    def _clean_query(self, query: str) -> str | None:
        """None for an empty query, else the sanitised text; raises when too long."""
        if not query or not query.strip():
            return None
        cleaned = self._sanitize_text(query)
        if len(cleaned) > MAX_QUERY_LENGTH:
            raise ValidationError(f"Query exceeds maximum length of {MAX_QUERY_LENGTH} characters")
        return cleaned

    @staticmethod
    def _clamp_n_results(n_results: int) -> int:
        if n_results < 1:
            raise ValidationError("n_results must be at least 1")
        if n_results > MAX_N_RESULTS:
            logger.warning(f"n_results {n_results} exceeds maximum {MAX_N_RESULTS}, capping to maximum")
            return MAX_N_RESULTS
        return n_results

    # ------------------------------------------------------------------ ingest
    def _compose(self, namespace: str, metadata: MetricMetadata) -> tuple[str, str, dict]:
        """(document id, searchable text, stored metadata) for one record."""
        if "metric_name" not in metadata:
            raise KeyError("metric_name is required in metadata")
        metric_name = str(metadata["metric_name"])
        self._validate_metric_name(metric_name)
        for field in TEXT_FIELDS:
            value = metadata.get(field, "")
            if value:
                self._validate_text_field(field, str(value))

        parts = []
        if metadata.get("description"):
            parts.append(self._sanitize_text(str(metadata["description"])))
        for field, label in DOCUMENT_LABELS:
            if metadata.get(field):
                parts.append(f"{label}: {self._sanitize_text(str(metadata[field]))}")
        text = " | ".join(parts) if parts else metric_name

        stored = {field: self._sanitize_text(str(metadata.get(field, ""))) for field in TEXT_FIELDS}
        stored["namespace"] = namespace
        return f"{namespace}#{metric_name}", text, stored

    def index_metadata(self, namespace: str, metadata: MetricMetadata) -> str:
        """Upsert one metric; the document id `namespace#metric_name` is returned.

        Raises KeyError without `metric_name`, ValidationError for a bad name or an
        over-long field (reference store.py:138-245).
        """
        document_id, text, stored = self._compose(namespace, metadata)
        try:
            self.collection.upsert(documents=[text], metadatas=[stored], ids=[document_id])
        except Exception as exc:
            logger.error(f"Failed to index metric '{document_id}': {exc}")
            raise
        logger.debug(f"Indexed metric: {document_id}")
        return document_id

    # Two-phase ingest for the indexer job (SURVEY.md §8f1 "batched upsert"; the reference upserts one row per call,
    # store.py:236): `prepare_index` does everything index_metadata does up to the upsert — same KeyError /
    # ValidationError at the same point for a bad record — and `commit_index` sends a whole batch in ONE upsert
    # (one embedding call, one ingest launch on the device).
    def prepare_index(self, namespace: str, metadata: MetricMetadata) -> tuple:
        return self._compose(namespace, metadata)

    def commit_index(self, prepared: list) -> list[str]:
        if not prepared:
            return []
        ids = [p[0] for p in prepared]
        try:
            self.collection.upsert(documents=[p[1] for p in prepared], metadatas=[p[2] for p in prepared], ids=ids)
        except Exception as exc:
            logger.error(f"Failed to index {len(ids)} metrics ('{ids[0]}' ...): {exc}")
            raise
        logger.debug(f"Indexed {len(ids)} metrics in one upsert")
        return ids

    def index_metadata_batch(self, namespace: str, records: Iterable[MetricMetadata]) -> list[str]:
        """Extension: validate every record first, then ONE upsert (one ingest launch)."""
        composed = [self._compose(namespace, r) for r in records]
        if not composed:
            return []
        if len(composed) > MAX_BULK_OPERATIONS:
            raise ValidationError(f"bulk operation exceeds maximum of {MAX_BULK_OPERATIONS} records")
        ids = [c[0] for c in composed]
        self.collection.upsert(documents=[c[1] for c in composed], metadatas=[c[2] for c in composed], ids=ids)
        return ids

    def metric_exists(self, namespace: str, metric_name: str) -> bool:
        """True when `namespace#metric_name` is stored; any failure reads as False (store.py:247-264)."""
        try:
            found = self.collection.get(ids=[f"{namespace}#{metric_name}"])
            return bool(found and found.get("ids") and len(found["ids"]) > 0)
        except Exception as exc:
            logger.warning(f"Error checking if metric exists: {exc}")
            return False

    # ------------------------------------------------------------------ search
    @staticmethod
    def _shape_hits(ids: list, metadatas: list, distances: list) -> list[dict]:
        """Chroma-shaped hit lists -> result dicts (store.py:322-341): metric_name is the text
        after the last '#', similarity = 1 - distance (missing distance counts as 1.0), and the
        stored metadata is spread LAST, so stored keys win over the two computed ones."""
        shaped = []
        for i, doc_id in enumerate(ids):
            stored = metadatas[i] if i < len(metadatas) else {}
            distance = distances[i] if i < len(distances) else 1.0
            name = doc_id.split("#")[-1] if "#" in doc_id else doc_id
            shaped.append({"metric_name": name, "similarity_score": 1.0 - distance, **stored})
        return shaped

    def search_metadata(self, query: str, n_results: int = 10) -> list[dict]:
        """Metrics most similar to `query`, best first (reference store.py:266-341).

        Empty query -> []; query longer than 1000 chars after sanitising or n_results < 1 ->
        ValidationError; n_results above 100 is capped with a warning.
        """
        cleaned = self._clean_query(query)
        if cleaned is None:
            logger.debug("Empty query received, returning empty results")
            return []
        n_results = self._clamp_n_results(n_results)
        results = self.collection.query(query_texts=[cleaned], n_results=n_results)
        if not results or not results.get("ids") or not results["ids"][0]:
            return []
        return self._shape_hits(
            results["ids"][0], results.get("metadatas", [[]])[0], results.get("distances", [[]])[0]
        )

    def search_metadata_batch(self, queries: list[str], n_results: int = 10) -> list[list[dict]]:
        """Extension (the reference only ever sends one query): many queries, ONE engine call.

        Same per-query rules as search_metadata; an empty query yields [] at its position.
        """
        cleaned = [self._clean_query(q) for q in queries]
        n_results = self._clamp_n_results(n_results)
        live = [i for i, c in enumerate(cleaned) if c is not None]
        out: list[list[dict]] = [[] for _ in queries]
        if not live:
            return out
        results = self.collection.query(query_texts=[cleaned[i] for i in live], n_results=n_results)
        if not results or not results.get("ids"):
            return out
        all_md = results.get("metadatas") or []
        all_d = results.get("distances") or []
        for j, i in enumerate(live):
            ids = results["ids"][j] if j < len(results["ids"]) else []
            if ids:
                out[i] = self._shape_hits(ids, all_md[j] if j < len(all_md) else [], all_d[j] if j < len(all_d) else [])
        return out
