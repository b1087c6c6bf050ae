"""Geometric analysis and graph-based utilities on the MI355X -- drop-in for the reference's
`src.geo` package (src/geo/__init__.py:5-8 re-exports the same two names)."""
from .knn_graph_optimized import build_knn_graph
from .geo_shortest_paths import dijkstra_multi_source

__all__ = ["build_knn_graph", "dijkstra_multi_source"]
