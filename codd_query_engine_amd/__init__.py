"""
codd_query_engine_amd — MI355X-native replacement for the ChromaDB-backed semantic
retrieval behind Codd's `search_relevant_metrics`.

Host side mirrors the reference's Python interface for the path; the cosine top-k runs in
hand-written HIP (csrc/) behind the C ABI of include/codd_knn.h.
"""

from .errors import ValidationError
from .models import MetricMetadata, SearchResult, SemanticStoreConfig
from .embedding import HashingEmbeddingFunction, LocalTransformerEmbeddingFunction
from .knn_client import Collection, KnnClient
from .semantic_store import MetricsSemanticMetadataStore
from .metrics_search import MetricsSearchClient, get_semantic_store, project_search_results

__all__ = [
    "ValidationError",
    "MetricMetadata",
    "SearchResult",
    "SemanticStoreConfig",
    "HashingEmbeddingFunction",
    "LocalTransformerEmbeddingFunction",
    "Collection",
    "KnnClient",
    "MetricsSemanticMetadataStore",
    "MetricsSearchClient",
    "get_semantic_store",
    "project_search_results",
]
__version__ = "0.1.0"
