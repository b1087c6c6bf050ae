"""the legacy Riemannian codebook builder for vanilla-VAE vector latents, against the artefacts of the
reference's own build_and_save on the same seeded inputs (tests/golden/legacy_riemannian.npz): Euclidean kNN graph ->
largest component (549 of 600 nodes: codes carry -1) -> full / distance-stratified subset re-weighting through the
decoder's pull-back metric (Linear-first decoder: autograd on the GPU) -> geodesic k-medoids."""
import os

import numpy as np
import pytest
import torch
from scipy import sparse

pytestmark = pytest.mark.gpu

LEGACY_VAE = dict(in_channels=1, enc_channels=[32, 64, 128], dec_channels=[128, 64, 32], latent_dim=16, recon_loss="bce",
                  output_image_size=28, norm_type="none", mse_use_sigmoid=True)


def _config(tmp, mode):
    return {"data": {"latents_path": os.path.join(tmp, "z.pt")}, "checkpoint_path": os.path.join(tmp, "best.pt"),
            "vae_config": dict(LEGACY_VAE), "graph": {"k": 10, "metric": "euclidean", "sym": "mutual", "mode": "distance"},
            "riemannian": {"mode": mode, "max_edges": 1000, "batch_size": 256},
            "quantize": {"K": 16, "init": "kpp", "seed": 42}, "out": {"dir": os.path.join(tmp, "out_" + mode)}}


@pytest.mark.parametrize("mode", ["full", "subset"])
def test_legacy_builder_artefacts_equal_reference(golden, tmp_path, mode):
    from oracle import synthetic as syn
    from vqvae_amd.training.build_riemannian_codebook_legacy import build_and_save
    from vqvae_amd.vae import Decoder
    g = golden("legacy_riemannian")
    tmp = str(tmp_path)
    # a reference checkpoint holds encoder.* and decoder.* entries; the builder reads the decoder's only.  (seeded_state_dict
    # fills the keys in sorted order, "decoder." first: the same decoder weights the fixture was generated with)
    template = {"decoder." + k: v for k, v in Decoder(1, (128, 64, 32), 16, 28, "none").state_dict().items()}
    template["encoder.fc_mu.weight"] = torch.zeros(16, 2048)
    torch.save({"model_state_dict": syn.seeded_state_dict(template, 21), "epoch": 0}, os.path.join(tmp, "best.pt"))
    torch.save(torch.from_numpy(syn.gauss_latents(600, 16, 22)), os.path.join(tmp, "z.pt"))
    np.random.seed(123)                                   # the subset is drawn with numpy's global generator
    out = build_and_save(_config(tmp, mode))
    Wr = sparse.load_npz(out / "knn_graph_riemannian.npz").tocsr()
    Wr.sort_indices()
    We = sparse.load_npz(out / "knn_graph_euclidean.npz")
    assert We.nnz == int(g[f"{mode}/eucl_nnz"])
    np.testing.assert_array_equal(Wr.indptr, g[f"{mode}/riem_indptr"])
    np.testing.assert_array_equal(Wr.indices, g[f"{mode}/riem_indices"])
    rel = np.abs(Wr.data - g[f"{mode}/riem_data"]) / g[f"{mode}/riem_data"]
    assert np.mean(rel <= 1e-5) >= 0.999 and rel.max() < 1e-3, rel.max()
    cb = torch.load(out / "codebook.pt", weights_only=False)
    assert cb["method"] == "riemannian_geodesic" and cb["medoid_indices"].dtype == np.int32
    np.testing.assert_array_equal(cb["medoid_indices"], g[f"{mode}/medoid_indices"])
    np.testing.assert_array_equal(np.load(out / "codes.npy"), g[f"{mode}/codes"])
