"""Two ranks sharing the box's GPU over gloo (the C5 test's situation): prior step time, native vs eager.
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 scratch/exp_prior_dp.py"""
import os, sys, time, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vqvae_amd.prior.codes_dataset import CodeSequences, ResidentBatches
from vqvae_amd.prior.train import train_prior
from vqvae_amd.prior.transformer import Transformer
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("gloo")
rank = dist.get_rank()
n = 6400
g0 = torch.Generator().manual_seed(0)
codes = torch.randint(0, 512, (n, 16), generator=g0)
labels = torch.randint(0, 10, (n,), generator=g0)
for native in (True, False):
    torch.manual_seed(0)
    data = CodeSequences(codes, labels).to(dev)
    gen = torch.Generator().manual_seed(1)
    tl, vl = ResidentBatches(data, 256, True, gen), ResidentBatches(data, 256, False, gen)
    model = Transformer(num_classes=10, num_tokens=512, embed_dim=256, n_layers=4, n_head=4, max_seq_len=16, dropout=0.1).to(dev)
    model.fused_attention = native
    stamps = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hist = train_prior(model, tl, vl, epochs=1, lr=3e-4, weight_decay=0.01, device=dev, native=native,
                       on_step=lambda i, l: stamps.append(time.perf_counter()))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    d = np.diff(stamps) * 1e3
    print(f"rank {rank} native={native}: epoch {t1 - t0:.2f} s, {len(stamps)} steps, step ms median {np.median(d):.2f} max {d.max():.1f}", flush=True)
dist.destroy_process_group()
