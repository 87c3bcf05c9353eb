"""
Direct callers of the store on the search_relevant_metrics path, mirrored:

  get_semantic_store(config)          <- PromQLModule.get_semantic_store
                                         (codd_lib/codd_lib/client/provider/promql_module.py:49-67)
  MetricsSearchClient.search_relevant_metrics(query, limit=5)
                                      <- MetricsPromQLClient.search_relevant_metrics
                                         (codd_lib/codd_lib/client/metrics_promql_client.py:71-107)
"""

from __future__ import annotations

from typing import Any, Optional

from .knn_client import KnnClient
from .models import SEARCH_RESULT_DEFAULTS, SearchResult, SemanticStoreConfig
from .semantic_store import MetricsSemanticMetadataStore


def get_semantic_store(config: Optional[SemanticStoreConfig] = None, client: Any = None) -> MetricsSemanticMetadataStore:
    """The DI seam: where the reference builds `chromadb.HttpClient(host, port)`, build the
    in-process HIP-backed client instead; everything above the store is unchanged."""
    config = config or SemanticStoreConfig()
    if client is None:
        from .embedding import HashingEmbeddingFunction

        client = KnnClient(device=config.device, dtype=config.dtype, path=config.chromadb_path,
                           embedding_function=HashingEmbeddingFunction(config.embedding_dim))
    return MetricsSemanticMetadataStore(client, collection_name=config.collection_name)


def project_search_results(raw_results: list[dict]) -> list[SearchResult]:
    """Exactly the 11 SearchResult keys, '' / 0.0 for missing ones; `type` and `namespace`
    (present in the store's dicts) are dropped (metrics_promql_client.py:87-107)."""
    return [{key: row.get(key, default) for key, default in SEARCH_RESULT_DEFAULTS.items()} for row in raw_results]


class MetricsSearchClient:
    """The slice of MetricsPromQLClient that serves `search_relevant_metrics`."""

    def __init__(self, semantic_metadata_store: MetricsSemanticMetadataStore):
        self.semantic_metadata_store = semantic_metadata_store

    def search_relevant_metrics(self, query: str, limit: int = 5) -> list[SearchResult]:
        return project_search_results(self.semantic_metadata_store.search_metadata(query, n_results=limit))

    def search_relevant_metrics_batch(self, queries: list[str], limit: int = 5) -> list[list[SearchResult]]:
        """Extension: one engine call for many queries (B up to 1024)."""
        return [project_search_results(r) for r in self.semantic_metadata_store.search_metadata_batch(queries, n_results=limit)]
