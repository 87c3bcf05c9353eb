"""Record / result / config schemas on the search_relevant_metrics path.

Counterparts (field for field) of the reference's
  MetricMetadata       codd_engine/models/metrics_common.py:4-16
  SearchResult         codd_engine/validation_engine/metrics/structured_outputs.py:7-20
  SemanticStoreConfig  codd_lib/codd_lib/config/semantic_store_config.py:8-14
"""

from __future__ import annotations

from typing import Optional

from typing_extensions import TypedDict  # pydantic needs this flavour below Python 3.12

from pydantic import BaseModel


class MetricMetadata(TypedDict, total=False):
    metric_name: str  # required
    type: str | None  # read by the store (store.py:167) although the reference's TypedDict omits it
    description: str | None
    unit: str | None
    category: str | None
    subcategory: str | None
    category_description: str | None
    golden_signal_type: str | None
    golden_signal_description: str | None
    meter_type: str | None
    meter_type_description: str | None


class SearchResult(TypedDict):
    metric_name: str
    similarity_score: float
    description: str
    unit: str
    category: str
    subcategory: str
    category_description: str
    golden_signal_type: str
    golden_signal_description: str
    meter_type: str
    meter_type_description: str


SEARCH_RESULT_DEFAULTS: dict = {
    "metric_name": "",
    "similarity_score": 0.0,
    "description": "",
    "unit": "",
    "category": "",
    "subcategory": "",
    "category_description": "",
    "golden_signal_type": "",
    "golden_signal_description": "",
    "meter_type": "",
    "meter_type_description": "",
}


class SemanticStoreConfig(BaseModel):
    """Same four fields and defaults as the reference; the rest selects the in-process engine.

    `chromadb_path` exists in the reference but is never read there; here it is the
    directory of the on-disk index shared by the indexer job and the service.
    """

    chromadb_host: str = "localhost"
    chromadb_port: int = 8000
    chromadb_path: Optional[str] = None
    collection_name: str = "metrics_semantic_metadata"
    # --- build extensions (defaults keep the reference's behaviour) ---
    device: str = "cuda:0"
    dtype: str = "f32"
    embedding_dim: int = 384
