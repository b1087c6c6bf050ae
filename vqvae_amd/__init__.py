"""vqvae_amd -- MI355X (gfx950) implementation of the geodesic-codebook path of m4rch1n0/vqvae:
kNN graph -> decoder pull-back edge lengths -> shortest paths -> k-means++/k-medoids, behind the
reference's own `src/geo` API (vqvae_amd.geo) and `build_codebook.py` CLI (vqvae_amd.scripts)."""
__version__ = "0.1.0"
