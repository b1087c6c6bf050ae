"""Prior training step time at the C5 shape, single process: native (HIP graph + fused attention + one-launch AdamW) vs eager."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd.prior.codes_dataset import CodeSequences, ResidentBatches
from vqvae_amd.prior.train import train_prior
from vqvae_amd.prior.transformer import Transformer
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
codes = torch.randint(0, 512, (n, 16))
labels = torch.randint(0, 10, (n,))
for native in (True, False, True):
    torch.manual_seed(0)
    data = CodeSequences(codes, labels).to(dev)
    tl, vl = ResidentBatches(data, 256, True), ResidentBatches(data, 256, False)
    model = Transformer(num_classes=10, num_tokens=512, embed_dim=256, n_layers=4, n_head=4, max_seq_len=16, dropout=0.1).to(dev)
    model.fused_attention = native
    stamps = []
    def on_step(i, loss):
        stamps.append(time.perf_counter())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist = train_prior(model, tl, vl, epochs=1, lr=3e-4, weight_decay=0.01, device=dev, native=native, on_step=on_step)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    d = np.diff(stamps) * 1e3
    print(f"native={native}: epoch {t1 - t0:.2f} s for {len(stamps)} steps; step ms median {np.median(d):.3f} first {d[:3]} last {d[-3:]}; loss {hist['train_loss'][0]:.3f} -> {hist['train_loss'][-1]:.3f}", flush=True)
