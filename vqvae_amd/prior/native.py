"""MI355X-native pieces of the prior's training step (csrc/prior.hip through the C ABI) and the HIP-graph step.

At the reference's sizes (4 layers x 256 dims, 15 tokens, batch 256: src/models/transformer.py, configs/*/transformer.yaml) a
training step is ~300 launches over tensors of a few hundred kilobytes -- bound by launches and intermediate round trips.
What this module does about it:
  * `causal_attention`   one fused kernel for  q k^T * scale -> causal mask -> softmax -> dropout -> @ v  and one for its
                         whole backward (6 + 11 launches and 5 intermediates in torch), as a torch.autograd.Function;
  * `ArenaAdamW`         torch.optim.AdamW's update over the model's single flat arena in ONE launch (lr and step on the
                         device, so the launch never changes);
  * `GraphedStep`        forward + backward of a fixed-shape batch captured once in a HIP graph and replayed: no Python,
                         no per-kernel launch cost ("HIP graphs instead of a tracing compiler").
Dropout masks are drawn by torch (its Philox stream is graph-safe) and handed to the kernel as bytes.
"""
from typing import Optional, Tuple

import torch

from .. import _lib
from .._device import ptr, stream_ptr


def attention_kernel_covers(T: int, head_dim: int) -> bool:
    return T <= 16 and head_dim in (16, 32, 64)


class _CausalAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: torch.Tensor, n_head: int, p_drop: float, keep: Optional[torch.Tensor] = None):
        B, T, C3 = qkv.shape
        C = C3 // 3
        qkv = qkv.contiguous()
        out = torch.empty((B, T, C), dtype=torch.float32, device=qkv.device)
        probs = torch.empty((B, n_head, T, T), dtype=torch.float32, device=qkv.device)
        if keep is None and p_drop > 0.0:
            keep = torch.rand((B, n_head, T, T), device=qkv.device) >= p_drop
        if keep is not None:
            keep = keep.to(torch.bool).contiguous()
        scale = 1.0 / (1.0 - p_drop) if p_drop > 0.0 else 1.0
        with torch.cuda.device(qkv.device):
            _lib.check(_lib.load().geo_prior_attention_fwd(ptr(qkv), ptr(keep), scale, B, T, n_head, C // n_head, ptr(out),
                                                           ptr(probs), stream_ptr()), "geo_prior_attention_fwd")
        ctx.save_for_backward(qkv, probs, keep)
        ctx.meta = (n_head, scale)
        return out

    @staticmethod
    def backward(ctx, dout: torch.Tensor):
        qkv, probs, keep = ctx.saved_tensors
        n_head, scale = ctx.meta
        B, T, C3 = qkv.shape
        dout = dout.contiguous()
        dqkv = torch.empty_like(qkv)
        with torch.cuda.device(qkv.device):
            _lib.check(_lib.load().geo_prior_attention_bwd(ptr(qkv), ptr(probs), ptr(keep), scale, ptr(dout), B, T, n_head,
                                                           C3 // 3 // n_head, ptr(dqkv), stream_ptr()), "geo_prior_attention_bwd")
        return dqkv, None, None, None


def causal_attention(qkv: torch.Tensor, n_head: int, p_drop: float = 0.0, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """qkv f32 [B, T, 3C] on the GPU (the c_attn projection) -> context f32 [B, T, C]; standard causal mask.
    `keep` (bool [B, H, T, T]) fixes the dropout mask instead of drawing one (tests)."""
    return _CausalAttention.apply(qkv, n_head, float(p_drop), keep)


class ArenaAdamW:
    """AdamW over ONE flat parameter (the Transformer's arena): same update as torch.optim.AdamW(params, lr, betas=(0.9, 0.999),
    eps=1e-8, weight_decay), one kernel launch.  `param_groups` mimics the optimiser interface the lr scheduler needs."""

    def __init__(self, arena: torch.nn.Parameter, lr: float, weight_decay: float, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-8):
        assert arena.is_cuda and arena.dtype == torch.float32 and arena.is_contiguous()
        self.arena = arena
        self.exp_avg = torch.zeros_like(arena)
        self.exp_avg_sq = torch.zeros_like(arena)
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=arena.device)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=arena.device)
        self.betas, self.eps, self.weight_decay = betas, float(eps), float(weight_decay)
        self.param_groups = [{"lr": float(lr), "initial_lr": float(lr), "params": [arena]}]
        self._lr_on_device = float(lr)

    def step(self) -> None:
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_on_device:                   # the scheduler moved it (once per epoch)
            self.lr_dev.fill_(lr)
            self._lr_on_device = lr
        self.step_dev += 1
        a = self.arena
        with torch.cuda.device(a.device):
            _lib.check(_lib.load().geo_prior_adamw(ptr(a.data), ptr(a.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), a.numel(),
                                                   ptr(self.lr_dev), ptr(self.step_dev), self.betas[0], self.betas[1], self.eps,
                                                   self.weight_decay, stream_ptr()), "geo_prior_adamw")


class GraphedStep:
    """forward + backward of `loss_fn(x, y, labels) -> scalar tensor` for ONE batch shape, captured in a HIP graph.
    The gradient lands in model.arena.grad (a static buffer); `run` copies the batch into static inputs and replays."""

    def __init__(self, model, loss_fn, x: torch.Tensor, y: torch.Tensor, labels: Optional[torch.Tensor]):
        self.model, self.loss_fn = model, loss_fn
        self.x, self.y = x.clone(), y.clone()
        self.labels = labels.clone() if labels is not None else None
        arena = model.arena
        if arena.grad is None:
            arena.grad = torch.zeros_like(arena)
        side = torch.cuda.Stream(device=arena.device)
        side.wait_stream(torch.cuda.current_stream(arena.device))
        with torch.cuda.stream(side):                  # warm-up outside the capture (lazy initialisations, autotuning)
            for _ in range(2):
                arena.grad.zero_()
                loss_fn(self.x, self.y, self.labels).backward()
        torch.cuda.current_stream(arena.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            arena.grad.zero_()
            self.loss = loss_fn(self.x, self.y, self.labels)
            self.loss.backward()

    def run(self, x: torch.Tensor, y: torch.Tensor, labels: Optional[torch.Tensor]) -> torch.Tensor:
        self.x.copy_(x)
        self.y.copy_(y)
        if self.labels is not None:
            self.labels.copy_(labels)
        self.graph.replay()
        return self.loss
