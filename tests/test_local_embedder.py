"""The optional local-path transformer embedder, exercised offline with a tiny randomly initialised
BERT written to a temporary directory — no weights exist in this image and none are fetched."""

import numpy as np
import pytest

from codd_query_engine_amd import KnnClient, MetricsSemanticMetadataStore
from codd_query_engine_amd.embedding import LocalTransformerEmbeddingFunction
from tests._oracle_engine import OracleEngine

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "cpu", "memory", "disk", "usage", "latency", "http", "request", "requests", "errors",
         "total", "seconds", "bytes", "queue", "kafka", "consumer", "lag", "##s", "the", "of", "per", "rate", "|", ":", "category", "type"]


@pytest.fixture(scope="module")
def model_dir(tmp_path_factory):
    import torch
    from transformers import BertConfig, BertModel, BertTokenizerFast

    d = tmp_path_factory.mktemp("tiny_bert")
    (d / "vocab.txt").write_text("\n".join(VOCAB) + "\n")
    BertTokenizerFast(vocab_file=str(d / "vocab.txt"), do_lower_case=True).save_pretrained(str(d))
    torch.manual_seed(0)
    cfg = BertConfig(vocab_size=len(VOCAB), hidden_size=32, num_hidden_layers=2, num_attention_heads=4, intermediate_size=64,
                     max_position_embeddings=64)
    BertModel(cfg).save_pretrained(str(d), safe_serialization=True)
    return str(d)


def test_embeddings_are_unit_vectors_deterministic_and_padding_invariant(model_dir):
    ef = LocalTransformerEmbeddingFunction(model_dir)
    texts = ["cpu usage", "http request latency seconds total", "kafka consumer lag", ""]
    a = ef(texts)
    assert a.shape == (4, 32) and a.dtype == np.float32 and ef.dim == 32
    assert np.allclose(np.linalg.norm(a, axis=1), 1.0, atol=1e-5)
    assert np.array_equal(a, ef(texts))
    # a text embeds the same alone as inside a padded batch (the mask keeps padding out of the mean)
    alone = np.concatenate([ef([t]) for t in texts])
    assert np.abs(alone - a).max() < 1e-5
    assert ef([]).shape == (0, 32)


def test_missing_directory_is_an_error_not_a_download(tmp_path):
    with pytest.raises(FileNotFoundError):
        LocalTransformerEmbeddingFunction(str(tmp_path / "nope"))


def test_store_runs_end_to_end_on_a_local_transformer(model_dir):
    ef = LocalTransformerEmbeddingFunction(model_dir)
    store = MetricsSemanticMetadataStore(KnnClient(engine_factory=lambda dim: OracleEngine(dim), embedding_function=ef))
    for name, desc in [("cpu.usage", "cpu usage"), ("http.latency", "http request latency seconds"), ("kafka.lag", "kafka consumer lag")]:
        store.index_metadata("ns", {"metric_name": name, "description": desc})
    hits = store.search_metadata("kafka consumer lag", n_results=3)
    assert len(hits) == 3 and {h["metric_name"] for h in hits} == {"cpu.usage", "http.latency", "kafka.lag"}
    # the document text of kafka.lag is exactly the query: its embedding is identical, so it must rank first at distance ~0
    assert hits[0]["metric_name"] == "kafka.lag" and abs(hits[0]["similarity_score"] - 1.0) < 1e-5
