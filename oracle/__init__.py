"""oracle/ -- CPU restatement of the reference's geodesic-codebook path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product (vqvae_amd/) never does, and fails loudly when its HIP library is missing.

Restated files (reference = m4rch1n0/vqvae snapshot 2025-09-26, paths relative to its root):
  src/geo/knn_graph_optimized.py   -> oracle/knn.py      (+ geo_oracle.c: oracle_knn, oracle_cc)
  src/geo/geo_shortest_paths.py    -> oracle/sssp.py     (+ geo_oracle.c: oracle_sssp)
  src/geo/kmeans_optimized.py      -> oracle/kmedoids.py
  src/geo/riemannian_metric.py     -> oracle/metric.py   (closed-form forward-mode tangent)
  src/models/spatial_vae.py:47-81  -> oracle/metric.py   (decoder layer algebra)
  src/scripts/build_codebook.py    -> oracle/pipeline.py

The arithmetic of the reference lives in third-party packages that are not vendored in it and are
unpinned there (requirements.txt: scipy>=1.11, scikit-learn>=1.3, bare torch/numpy).  The oracle is
pinned against the versions installed in the build container:
  scipy 1.15.3 (csgraph.dijkstra, connected_components, sparse max/min), scikit-learn 1.7.2
  (NearestNeighbors / ArgKmin), torch 2.10.0 (autograd.functional.jvp, conv/BN CPU kernels),
  numpy 2.2.6 (RandomState.randint/choice, pairwise float32 sum).
Parity is pinned by (i) the reference's own known-answer tests for Dijkstra
(tests/test_geo_shortest_paths.py:37-46,56-71,82-90, re-expressed in tests/test_oracle_*.py) and
(ii) golden vectors produced by importing the reference in the build container with
oracle/gen_golden.py and committed under tests/golden/.
"""
PINNED = {"scipy": "1.15.3", "scikit-learn": "1.7.2", "torch": "2.10.0", "numpy": "2.2.6"}
