"""
MetricsSemanticIndexerJob — the ingest surface of the search_relevant_metrics path, counterpart
of the reference's codd_jobs/metrics_semantic_indexer_job.py (same class name, constructor and
`run` signature, the same seven counters moving on the same branches, the same printed report,
the same error type).  Behaviour is pinned by tests/golden/indexer_job_golden.json, captured by
running the reference job itself with fake collaborators (oracle/gen_job_golden.py).

What the job feeds is this build's MetricsSemanticMetadataStore on a KnnClient: every indexed
metric becomes one normalised row in HBM (csrc/, `codd_knn_upsert_host`).

The three network collaborators of the reference are OUT OF SCOPE and
enter through seams instead of being rebuilt:
  * Prometheus (`PromQLClient`, HTTP)      -> `metadata_source`: context manager with
                                               health_check() and get_metric_metadata()
  * Redis exact-match store                -> `metric_names_store`: set_metric_names(namespace, names);
                                               default keeps the SET `<namespace>#metric_names` on the
                                               given redis client, or in memory when there is none
  * the LLM enrichment agent               -> `enrichment_agent`: enrich_metric_to_dict(metric_name=,
                                               metric_type=, description=); default is a deterministic
                                               keyword-rule stand-in so the job runs offline
"""

from __future__ import annotations

import logging
import re
from dataclasses import dataclass
from typing import Any, Callable, Optional

from .semantic_store import MetricsSemanticMetadataStore

logger = logging.getLogger(__name__)

_RULE = "=" * 70


@dataclass
class IndexingStats:
    """Counters of one job run (reference job.py:38-48)."""

    total_metrics: int = 0
    processed_metrics: int = 0
    enriched_metrics: int = 0
    indexed_metrics: int = 0
    failed_metrics: int = 0
    skipped_metrics: int = 0
    excluded_metrics: int = 0


class MetricsSemanticIndexerJobError(Exception):
    """The job could not run to completion."""


class MetricEnrichmentError(Exception):
    """An enrichment agent could not describe a metric (reference metrics_enrichment_agent.py:27)."""


# ------------------------------------------------------------------------------ stand-ins
class InMemoryMetricNamesStore:
    """Stand-in for the Redis SET store (reference codd_dal/metrics/metrics_metadata_store.py:35-48)."""

    def __init__(self):
        self.names: dict[str, set[str]] = {}

    def set_metric_names(self, namespace: str, metric_names: set[str]) -> None:
        self.names[f"{namespace}#metric_names"] = set(metric_names)


class RedisMetricNamesStore:
    """Same key and replace-all semantics as the reference store, on a caller-supplied redis client."""

    def __init__(self, redis_client: Any):
        self.redis_client = redis_client

    def set_metric_names(self, namespace: str, metric_names: set[str]) -> None:
        key = f"{namespace}#metric_names"
        self.redis_client.delete(key)
        if metric_names:
            self.redis_client.sadd(key, *metric_names)


class StaticMetadataSource:
    """A fixed `{metric: [{type, help}, ...]}` mapping shaped like Prometheus' /api/v1/metadata."""

    def __init__(self, metadata: dict, healthy: bool = True):
        self.metadata = metadata
        self.healthy = healthy

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def health_check(self) -> bool:
        return self.healthy

    def get_metric_metadata(self) -> dict:
        return self.metadata


class RuleBasedEnrichmentAgent:
    """Deterministic, offline stand-in for the LLM agent: fills the eleven MetricMetadata fields
    from keywords in the metric name and HELP text.  Not a model of the LLM's prose — only of its
    output schema (reference semantic_engine/structured_outputs.py:5-35)."""

    _SIGNALS = (
        ("latency", ("latency", "duration", "seconds", "time", "delay")),
        ("errors", ("error", "fail", "exception", "5xx", "timeout", "dropped")),
        ("saturation", ("memory", "cpu", "disk", "queue", "utilization", "usage", "free", "bytes", "load")),
        ("traffic", ("request", "total", "count", "rate", "throughput", "packets", "connections")),
    )
    _CATEGORIES = (
        ("application", ("http", "grpc", "request", "api", "rpc", "handler")),
        ("database", ("db", "sql", "query", "postgres", "mysql", "redis", "mongo")),
        ("runtime", ("go_", "jvm", "gc", "python", "process", "thread")),
        ("infrastructure", ("node", "cpu", "memory", "disk", "network", "container", "kube")),
    )
    _UNITS = (("seconds", "seconds"), ("bytes", "bytes"), ("percent", "percent"), ("ratio", "ratio"), ("total", "count"))

    def enrich_metric_to_dict(self, metric_name: str, metric_type: Optional[str] = None, description: Optional[str] = None) -> dict:
        if not metric_name:
            raise MetricEnrichmentError("metric_name is empty")
        text = f"{metric_name} {description or ''}".lower()
        pick = lambda table, default: next((label for label, keys in table if any(k in text for k in keys)), default)  # noqa: E731
        signal = pick(self._SIGNALS, "none")
        category = pick(self._CATEGORIES, "application")
        unit = next((u for suffix, u in self._UNITS if suffix in metric_name.lower()), "")
        meter = metric_type if metric_type and metric_type != "unknown" else "gauge"
        words = re.sub(r"[_.:/\-]+", " ", metric_name).strip()
        return {
            "metric_name": metric_name,
            "type": meter,
            "description": description or f"{words} metric",
            "unit": unit,
            "category": category,
            "subcategory": (re.split(r"[_.:/\-]+", metric_name) + [""])[0],
            "category_description": f"{category} level metrics",
            "golden_signal_type": signal,
            "golden_signal_description": f"relates to {signal}" if signal != "none" else "not a golden signal",
            "meter_type": meter,
            "meter_type_description": f"{meter} of {words}",
        }


# ------------------------------------------------------------------------------ the job
class MetricsSemanticIndexerJob:
    """Fetch metric metadata, record the names, enrich each metric, index it for semantic search."""

    def __init__(
        self,
        redis_client: Any,
        chromadb_client: Any,
        config_manager: Any = None,
        instructions_manager: Any = None,
        prometheus_config: Any = None,
        batch_size: int = 10,
        *,
        metadata_source: Optional[Callable[[Any], Any]] = None,
        enrichment_agent: Any = None,
        metric_names_store: Any = None,
    ):
        """Positional arguments as in the reference (job.py:66-74).  `metadata_source` is a factory
        `prometheus_config -> context manager`; the LLM managers are accepted and unused unless a
        caller's own `enrichment_agent` wants them."""
        self.prometheus_config = prometheus_config
        self.batch_size = batch_size
        self.config_manager = config_manager
        self.instructions_manager = instructions_manager
        self._metadata_source = metadata_source
        self.promql_client = None
        if metric_names_store is not None:
            self.redis_store = metric_names_store
        elif redis_client is not None:
            self.redis_store = RedisMetricNamesStore(redis_client)
        else:
            self.redis_store = InMemoryMetricNamesStore()
        self.semantic_store = MetricsSemanticMetadataStore(chromadb_client)
        self.enrichment_agent = enrichment_agent or RuleBasedEnrichmentAgent()
        self.stats = IndexingStats()
        logger.info(f"Initialized MetricsSemanticIndexerJob with batch_size={batch_size}")

    # -------------------------------------------------------------------------- run
    def run(self, namespace: str, limit: Optional[int] = None, exclude_pattern: Optional[str] = None,
            skip_if_present: bool = False, dry_run: bool = False):
        """One indexing pass over `namespace` (reference job.py:107-214)."""
        logger.info(f"Starting semantic indexing job for namespace: {namespace}")
        print(f"\n{_RULE}")
        print("METRICS SEMANTIC INDEXER JOB" + (" [DRY RUN MODE]" if dry_run else ""))
        print(_RULE)
        print(f"Namespace: {namespace}")
        print(f"Batch Size: {self.batch_size}")
        print(f"Limit: {limit if limit else 'None (all metrics)'}")
        print(f"Exclude Pattern: {exclude_pattern if exclude_pattern else 'None'}")
        print(f"Skip if Present: {skip_if_present}")
        print(f"Dry Run: {dry_run}")
        print(f"{_RULE}\n")
        try:
            print("[1/4] Fetching metrics from Prometheus...")
            metrics = self._fetch_metrics_from_prometheus(limit)
            self.stats.total_metrics = len(metrics)
            print(f"      ✓ Found {self.stats.total_metrics} metrics\n")

            if exclude_pattern:
                print(f"[1.5/4] Filtering metrics with exclude pattern: {exclude_pattern}")
                metrics = self._filter_metrics_by_pattern(metrics, exclude_pattern)
                print(f"      ✓ Filtered to {len(metrics)} metrics ({self.stats.excluded_metrics} excluded)\n")

            if dry_run:
                print("[2/4] Updating Redis metadata store... [SKIPPED - DRY RUN]\n")
                print("[3/4] Displaying metrics (no enrichment or indexing in dry run mode)...")
            else:
                print("[2/4] Updating Redis metadata store...")
                self._update_redis_store(namespace, metrics)
                print(f"      ✓ Redis updated with {len(metrics)} metric names\n")
                print("[3/4] Enriching metrics using LLM and indexing to semantic store...")
            self._process_metrics_in_batches(namespace, metrics, skip_if_present, dry_run)
            print(f"      ✓ Processed {self.stats.processed_metrics} metrics\n")

            print("[4/4] Indexing complete!")
            self._print_summary()
            self._persist()
            logger.info(f"Semantic indexing job completed successfully for namespace: {namespace}")
        except Exception as exc:
            logger.error(f"Semantic indexing job failed: {exc}", exc_info=True)
            print(f"\n✗ ERROR: {exc}\n")
            raise MetricsSemanticIndexerJobError(f"Failed to execute semantic indexing job: {exc}") from exc

    def _persist(self) -> None:
        """A persistent client (KnnClient(path=...)) writes its collections to disk after a run, so the
        service process can load what the job process indexed."""
        persist = getattr(getattr(self.semantic_store, "chromadb_client", None), "persist", None)
        if callable(persist):
            persist()

    # -------------------------------------------------------------------------- steps
    def _fetch_metrics_from_prometheus(self, limit: Optional[int]) -> list[dict]:
        """`[{metric, type, help}]`: the FIRST metadata entry of every metric that has one; `limit`
        is checked after every metric name, listed or not (reference job.py:259-305)."""
        try:
            if self._metadata_source is None:
                raise MetricsSemanticIndexerJobError("no metadata source configured (Prometheus access is out of scope: pass metadata_source=)")
            with self._metadata_source(self.prometheus_config) as client:
                self.promql_client = client
                if not client.health_check():
                    raise MetricsSemanticIndexerJobError("Prometheus health check failed")
                found: list[dict] = []
                for name, entries in client.get_metric_metadata().items():
                    if entries:
                        first = entries[0]
                        found.append({"metric": name, "type": first.get("type", "unknown"), "help": first.get("help", "")})
                    if limit and len(found) >= limit:
                        break
                return found
        except Exception as exc:
            raise MetricsSemanticIndexerJobError(f"Failed to fetch metrics from Prometheus: {exc}") from exc

    def _filter_metrics_by_pattern(self, metrics: list[dict], exclude_pattern: str) -> list[dict]:
        """Drop metrics whose NAME `re.match`es the pattern (anchored at the start, job.py:235)."""
        try:
            pattern = re.compile(exclude_pattern)
        except re.error as exc:
            raise MetricsSemanticIndexerJobError(f"Invalid regex pattern '{exclude_pattern}': {exc}") from exc
        kept = []
        for metric in metrics:
            if pattern.match(metric.get("metric", "")):
                self.stats.excluded_metrics += 1
            else:
                kept.append(metric)
        return kept

    def _update_redis_store(self, namespace: str, metrics: list[dict]) -> None:
        try:
            self.redis_store.set_metric_names(namespace, {m["metric"] for m in metrics})
        except Exception as exc:
            raise MetricsSemanticIndexerJobError(f"Failed to update Redis store: {exc}") from exc

    def _process_metrics_in_batches(self, namespace: str, metrics: list[dict], skip_if_present: bool = False,
                                    dry_run: bool = False) -> None:
        total = len(metrics)
        num_batches = (total + self.batch_size - 1) // self.batch_size
        print(f"      Processing {total} metrics in {num_batches} batches...\n")
        for b in range(num_batches):
            batch = metrics[b * self.batch_size : min((b + 1) * self.batch_size, total)]
            self._process_batch(namespace, batch, b + 1, num_batches, skip_if_present, dry_run)

    def _process_batch(self, namespace: str, batch: list[dict], batch_num: int, total_batches: int,
                       skip_if_present: bool = False, dry_run: bool = False) -> None:
        """Per metric: dry-run display | skip when present | enrich + index; a failure of one metric
        never stops the batch (reference job.py:359-461)."""
        print(f"      Batch {batch_num}/{total_batches} ({len(batch)} metrics):")
        # Batched ingest: a store that offers the two-phase API (this build's MetricsSemanticMetadataStore) validates and
        # composes every metric where the reference indexes it — same exceptions, same place in the report — and the
        # batch goes to the device in ONE upsert at the end.  Any other store (the reference's interface: one
        # index_metadata call per metric) is driven exactly as the reference drives it.
        prepare = getattr(self.semantic_store, "prepare_index", None)
        commit = getattr(self.semantic_store, "commit_index", None)
        two_phase = callable(prepare) and callable(commit)
        # Two-phase: a metric's outcome is only known once the batch's upsert has run, so the batch's report lines are
        # collected and printed behind it — the same bytes in the same order as the reference prints them one by one, with
        # exactly ONE outcome (✓ or ✗) per metric and `indexed_metrics` counting what was actually stored, also when the
        # upsert fails or the job is interrupted between prepare and commit.
        report: list[str] = []                      # the batch's output, in order (two-phase only)
        pending: list[tuple[int, Any, str]] = []    # (position in `report`, prepared record, its success suffix)
        failed_at: dict[int, BaseException] = {}
        committed = False

        def emit(text: str, end: str = "\n") -> None:
            if two_phase and not dry_run:
                report.append(text + end)
            else:
                print(text, end=end, flush=(end == ""))

        try:
            for item in batch:
                name = item["metric"]
                mtype = item.get("type", "unknown")
                help_text = item.get("help", "")
                try:
                    if dry_run:
                        self.stats.processed_metrics += 1
                        preview = help_text[:60] + "..." if len(help_text) > 60 else help_text
                        emit(f"        → {name} (type: {mtype}, desc: {preview or 'N/A'})")
                        continue
                    if skip_if_present and self.semantic_store.metric_exists(namespace, name):
                        self.stats.skipped_metrics += 1
                        emit(f"        → Skipping: {name} (already present)")
                        continue
                    self.stats.processed_metrics += 1
                    emit(f"        → Enriching: {name}", end="")
                    enriched = self.enrichment_agent.enrich_metric_to_dict(
                        metric_name=name, metric_type=mtype, description=help_text if help_text else None
                    )
                    self.stats.enriched_metrics += 1
                    done = (
                        f" ✓ (category: {enriched.get('category', 'N/A')}, "
                        f"signal: {enriched.get('golden_signal_type', 'N/A')}, "
                        f"meter_type: {enriched.get('meter_type', 'N/A')})"
                    )
                    if two_phase:
                        record = prepare(namespace, enriched)   # validates and composes here: same exceptions, same place
                        report.append("")                       # the metric's outcome, filled in behind the upsert
                        pending.append((len(report) - 1, record, done))
                        continue
                    self.semantic_store.index_metadata(namespace, enriched)
                    self.stats.indexed_metrics += 1
                    print(done)
                except Exception as exc:
                    self.stats.failed_metrics += 1
                    if type(exc).__name__ == "MetricEnrichmentError":  # ours or a caller agent's own class
                        emit(f" ✗ (enrichment failed: {str(exc)[:50]}...)")
                        logger.warning(f"Failed to enrich metric: {name}")
                    else:
                        emit(f" ✗ (error: {str(exc)[:50]}...)")
                        logger.error(f"Failed to process metric: {name}", exc_info=True)
            if pending:
                failed_at = self._commit_batch([rec for _, rec, _ in pending], commit)
            committed = True
        finally:
            # (also on KeyboardInterrupt / SystemExit between prepare and commit: nothing is reported as indexed that was not)
            for j, (pos, rec, done) in enumerate(pending):
                if committed and j not in failed_at:
                    self.stats.indexed_metrics += 1
                    report[pos] = done + "\n"
                elif j in failed_at:
                    self.stats.failed_metrics += 1
                    report[pos] = f" ✗ (error: {str(failed_at[j])[:50]}...)\n"
                    logger.error(f"Failed to index metric: {rec[0]}: {failed_at[j]}")
                else:
                    report[pos] = " ✗ (interrupted before the batch was stored)\n"
            if report:
                print("".join(report), end="", flush=True)
        print()

    def _commit_batch(self, prepared: list, commit: Callable[[list], Any]) -> dict:
        """One upsert for the batch.  If the store rejects it as a whole (a device error, not a validation error: those
        were raised per metric above), the metrics are retried one by one so that the report names the ones that failed.
        Returns {position in `prepared`: exception} for the metrics that could not be stored."""
        try:
            commit(prepared)
            return {}
        except Exception as exc:
            logger.error(f"Batched upsert of {len(prepared)} metrics failed ({exc}); retrying one by one")
        failed: dict = {}
        for j, one in enumerate(prepared):
            try:
                commit([one])
            except Exception as exc:
                failed[j] = exc
        return failed

    def _print_summary(self) -> None:
        s = self.stats
        print(f"\n{_RULE}")
        print("INDEXING SUMMARY")
        print(_RULE)
        print(f"Total Metrics:      {s.total_metrics}")
        print(f"Excluded:           {s.excluded_metrics}")
        print(f"Processed:          {s.processed_metrics}")
        print(f"Enriched (LLM):     {s.enriched_metrics}")
        print(f"Indexed (Semantic): {s.indexed_metrics}")
        print(f"Failed:             {s.failed_metrics}")
        print(f"Skipped:            {s.skipped_metrics}")
        print(_RULE)
        eligible = s.total_metrics - s.excluded_metrics
        rate = (s.indexed_metrics / eligible * 100) if eligible > 0 else 0
        print(f"Success Rate:       {rate:.1f}% (of eligible metrics)")
        print(f"{_RULE}\n")
