"""Geometric analysis and graph-based utilities on the MI355X -- drop-in for the reference's
`src.geo` package (src/geo/__init__.py:5-8 re-exports the same two names)."""
from .geo_shortest_paths import dijkstra_multi_source

__all__ = ["dijkstra_multi_source"]
