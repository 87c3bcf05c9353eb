"""
Wire shims above the store — the three thin surfaces through which `search_relevant_metrics` is
reached in the reference, with the same request/response shapes, so that existing callers (the
MCP host, HTTP clients, the CLI user) see no difference:

  REST  POST /api/metrics/search {query, limit=5} -> {results: [SearchResult], count}
        codd_service/codd_service/api/controllers/metrics_controller.py:47-58,113-135
        (any exception -> HTTP 500 with the message; a missing `query` -> 422 from validation)
  MCP   search_relevant_metrics(problem_json: str, limit: int = 5) -> list[dict]
        codd_mcp_server/server.py:48-88  (POSTs to the service; ANY failure -> prints and returns [])
  CLI   get-semantic-metrics QUERY --limit N   codd_cli/codd_cli/commands/metrics.py:26-67

These are plumbing, not the product: no server is started here, nothing is rebuilt from the
reference's service/MCP/CLI packages beyond the one route, the one tool and the one command.
"""

from __future__ import annotations

import os
from typing import Any, Awaitable, Callable, Optional

from pydantic import BaseModel

from .metrics_search import MetricsSearchClient
from .models import SearchResult


class MetricsSearchRequest(BaseModel):
    query: str
    limit: int = 5


class MetricsSearchResponse(BaseModel):
    results: list[SearchResult]
    count: int


def create_app(get_search_client: Callable[[], MetricsSearchClient]):
    """FastAPI app exposing the search route; `get_search_client` plays the role of the reference's
    module-global `get_client(True)` singleton (metrics_controller.py:22-38)."""
    from fastapi import APIRouter, FastAPI, HTTPException

    router = APIRouter(prefix="/api/metrics")

    @router.post("/search", response_model=MetricsSearchResponse)
    async def search_metrics(request: MetricsSearchRequest):
        try:
            results = get_search_client().search_relevant_metrics(request.query, limit=request.limit)
            return MetricsSearchResponse(results=results, count=len(results))
        except Exception as exc:  # the reference maps every failure to a 500
            raise HTTPException(status_code=500, detail=str(exc))

    app = FastAPI(title="Codd metrics search (MI355X engine)")
    app.include_router(router)
    return app


def http_transport(base_url: Optional[str] = None, timeout: float = 120.0) -> Callable[[str, dict], dict]:
    """POST json to the service, as the reference's `_make_request` does (server.py:17-44)."""
    base = base_url or os.getenv("MAVERICK_SERVICE_URL", "http://localhost:2840")

    def post(endpoint: str, json_data: dict) -> dict:
        import httpx

        with httpx.Client(timeout=timeout) as client:
            response = client.post(f"{base}{endpoint}", json=json_data)
            response.raise_for_status()
            return response.json()

    return post


def make_search_relevant_metrics_tool(post: Optional[Callable[[str, dict], dict]] = None) -> Callable[..., Awaitable[list[dict[str, Any]]]]:
    """The MCP tool body (register it with `@mcp.tool()` where fastmcp is installed).  `post` is the
    transport to the service; an in-process deployment can pass `lambda ep, js: app_call(js)`."""
    send = post or http_transport()

    async def search_relevant_metrics(problem_json: str, limit: int = 5) -> list[dict[str, Any]]:
        try:
            response = send("/api/metrics/search", {"query": problem_json, "limit": limit})
            return response.get("results", [])
        except Exception as exc:
            print(f"Error searching metrics: {exc}")
            return []

    return search_relevant_metrics


def format_results_table(results: list[dict]) -> str:
    """Plain-text rendering of the CLI table (metric, score with 3 decimals, first 50 chars of the
    description + '...', category) — commands/metrics.py:48-63."""
    if not results:
        return "No metrics found matching your query."
    rows = [(r["metric_name"], f"{r['similarity_score']:.3f}", r.get("description", "")[:50] + "...", r.get("category", "")) for r in results]
    head = ("Metric Name", "Score", "Description", "Category")
    widths = [max(len(str(x[i])) for x in rows + [head]) for i in range(4)]
    line = lambda cols: "  ".join(str(c).ljust(w) for c, w in zip(cols, widths)).rstrip()  # noqa: E731
    title = f"Semantic Search Results (Top {len(results)})"
    return "\n".join([title, line(head), line(["-" * w for w in widths])] + [line(r) for r in rows])


def cli_main(argv: Optional[list[str]] = None, search_client: Optional[MetricsSearchClient] = None) -> int:
    """`python -m codd_query_engine_amd.wire get-semantic-metrics "API latency" --limit 5 [--path DIR]`"""
    import argparse

    p = argparse.ArgumentParser(prog="codd")
    sub = p.add_subparsers(dest="cmd", required=True)
    g = sub.add_parser("get-semantic-metrics", help="Search for relevant metrics using semantic search")
    g.add_argument("query")
    g.add_argument("--limit", type=int, default=5)
    g.add_argument("--path", default=None, help="on-disk index written by the indexer job")
    args = p.parse_args(argv)
    try:
        if search_client is None:
            from .metrics_search import get_semantic_store
            from .models import SemanticStoreConfig

            search_client = MetricsSearchClient(get_semantic_store(SemanticStoreConfig(chromadb_path=args.path)))
        print(format_results_table(search_client.search_relevant_metrics(args.query, limit=args.limit)))
        return 0
    except Exception as exc:
        print(f"Error: {exc}")
        return 1


if __name__ == "__main__":
    raise SystemExit(cli_main())
