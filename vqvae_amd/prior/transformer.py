"""Autoregressive prior over code sequences, built around ONE flat parameter arena.

Contract kept from the reference (src/models/transformer.py:10-133): the constructor arguments, `forward(idx, y=None)`
-> logits (B, T, num_tokens), the attributes the training loop reads (`num_classes`, ...), the state-dict entry names and
shapes (a reference `best.pt` loads with strict=True and ours loads there), and the initial weights a given torch seed
produces.  Everything else is this repo's own design:

* all 3.3 M weights of the model live in a single contiguous float32 buffer (`self.arena`, the only nn.Parameter).
  The optimiser therefore runs one fused element-wise update, the data-parallel step reduces ONE buffer with ONE
  collective (vqvae_amd/prior/train.py), and the fused HIP block kernels take one base pointer plus the offsets of
  `self.layout` instead of forty tensor arguments;
* the network is evaluated functionally from named views into that arena -- there are no sub-modules;
* `state_dict` / `load_state_dict` translate between the arena and the reference's per-tensor names.

Architecture (fixed by the checkpoint contract): token + learned position (+ class) embedding, `n_layers` pre-LayerNorm
blocks of causal multi-head attention (QKV in one projection) and a 4x GELU MLP, final LayerNorm, untied output head
without bias.  Sequences are H*W - 1 = 15 tokens.
"""
import math
from collections import OrderedDict
from typing import Dict, List, NamedTuple, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import native


class Slot(NamedTuple):
    """One named tensor of the arena: where it lives and how a fresh model fills it."""
    name: str
    shape: Tuple[int, ...]
    offset: int
    fill: str            # "normal" (N(0, 0.02)), "zeros", "ones"
    born: str            # how torch creates it before the re-initialisation: "linear_w", "linear_b", "embedding", "none"


def _plan(num_classes: int, num_tokens: int, C: int, n_layers: int, T: int) -> List[Slot]:
    """Arena layout in the reference's construction order (which is also its state-dict order)."""
    spec: List[Tuple[str, Tuple[int, ...], str, str]] = [
        ("pos_emb", (1, T, C), "normal", "none"),
        ("token_emb.weight", (num_tokens, C), "normal", "embedding"),
    ]
    if num_classes > 0:
        spec.append(("class_emb.weight", (num_classes, C), "normal", "embedding"))
    for i in range(n_layers):
        b = f"blocks.{i}."
        spec += [(b + "ln1.weight", (C,), "ones", "none"), (b + "ln1.bias", (C,), "zeros", "none"),
                 (b + "ln2.weight", (C,), "ones", "none"), (b + "ln2.bias", (C,), "zeros", "none"),
                 (b + "attn.c_attn.weight", (3 * C, C), "normal", "linear_w"), (b + "attn.c_attn.bias", (3 * C,), "zeros", "linear_b"),
                 (b + "attn.c_proj.weight", (C, C), "normal", "linear_w"), (b + "attn.c_proj.bias", (C,), "zeros", "linear_b"),
                 (b + "mlp.0.weight", (4 * C, C), "normal", "linear_w"), (b + "mlp.0.bias", (4 * C,), "zeros", "linear_b"),
                 (b + "mlp.2.weight", (C, 4 * C), "normal", "linear_w"), (b + "mlp.2.bias", (C,), "zeros", "linear_b")]
    spec += [("ln_f.weight", (C,), "ones", "none"), ("ln_f.bias", (C,), "zeros", "none"),
             ("head.weight", (num_tokens, C), "normal", "linear_w")]
    slots, off = [], 0
    for name, shape, fill, born in spec:
        slots.append(Slot(name, shape, off, fill, born))
        off += (math.prod(shape) + 63) // 64 * 64          # every tensor starts on a 256-byte boundary
    return slots


class Transformer(nn.Module):
    def __init__(self, num_classes: int, num_tokens: int, embed_dim: int, n_layers: int, n_head: int, max_seq_len: int,
                 dropout: float = 0.1):
        super().__init__()
        if embed_dim % n_head:
            raise AssertionError(f"embed_dim {embed_dim} is not a multiple of n_head {n_head}")
        self.num_classes, self.num_tokens, self.embed_dim = num_classes, num_tokens, embed_dim
        self.n_layers, self.n_head, self.max_seq_len, self.dropout = n_layers, n_head, max_seq_len, float(dropout)
        self.layout: List[Slot] = _plan(num_classes, num_tokens, embed_dim, n_layers, max_seq_len)
        last = self.layout[-1]
        self.arena = nn.Parameter(torch.zeros(last.offset + math.prod(last.shape)))
        # the reference keeps one (1, 1, T, T) lower-triangular buffer per attention layer IN its state dict and masks
        # where that buffer is zero; the buffers are carried (and honoured, should a checkpoint hold other values)
        self.register_buffer("mask_bias", torch.tril(torch.ones(max_seq_len, max_seq_len)).repeat(n_layers, 1, 1),
                             persistent=False)
        self._standard_mask = True      # mask buffers are the lower triangle (checked whenever a checkpoint is loaded)
        self.fused_attention = True     # GPU: csrc/prior.hip's attention kernels when they cover the shape
        self._fresh_weights()

    # ------------------------------------------------------------------------------------------ parameters
    def views(self) -> Dict[str, torch.Tensor]:
        """name -> view into the arena (autograd flows back into `self.arena.grad`, one buffer)."""
        a = self.arena
        return {s.name: a[s.offset:s.offset + math.prod(s.shape)].view(s.shape) for s in self.layout}

    @torch.no_grad()
    def _fresh_weights(self) -> None:
        """The weights the reference holds after construction under the same torch seed (transformer.py:43-53 re-draws
        every Linear / Embedding weight and the positions from N(0, 0.02), zeroes biases, unit LayerNorm).  torch first
        creates each Linear / Embedding with its default initialiser, which advances the global CPU generator; those
        draws are replayed into scratch so that the N(0, 0.02) draws that follow see the generator where the reference's
        do.  The re-draw visits modules children-first: embeddings, blocks, head -- and the positions last."""
        scratch = torch.empty(max(math.prod(s.shape) for s in self.layout))
        for s in self.layout:                                  # construction: default initialisers
            n = math.prod(s.shape)
            if s.born == "embedding":
                scratch[:n].view(s.shape).normal_()
            elif s.born in ("linear_w", "linear_b"):
                scratch[:n].view(s.shape).uniform_()
        v = self.views()
        order = [s for s in self.layout if s.name != "pos_emb"] + [self.layout[0]]
        for s in order:                                        # re-initialisation
            if s.fill == "normal":
                v[s.name].copy_(torch.empty(s.shape).normal_(mean=0.0, std=0.02))
            elif s.fill == "ones":
                v[s.name].fill_(1.0)
            else:
                v[s.name].zero_()

    # ------------------------------------------------------------------------------------------ checkpoint contract
    def _mask_entries(self) -> Dict[str, torch.Tensor]:
        T = self.max_seq_len
        return {f"blocks.{i}.attn.bias": self.mask_bias[i].view(1, 1, T, T) for i in range(self.n_layers)}

    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False):
        """The reference's entries, in its order: one tensor per weight plus the (1, 1, T, T) lower-triangular mask of
        every attention layer."""
        out = OrderedDict() if destination is None else destination
        masks = self._mask_entries()
        with torch.set_grad_enabled(keep_vars):
            for name, t in self.views().items():
                out[prefix + name] = t if keep_vars else t.detach().clone()
            for name, t in masks.items():
                out[prefix + name] = t.clone()
        return out

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        views, masks = self.views(), self._mask_entries()
        missing = [k for k in list(views) + list(masks) if k not in state_dict]
        unexpected = [k for k in state_dict if k not in views and k not in masks]
        if "arena" in state_dict and len(state_dict) == 1:      # our own flat form
            missing, unexpected = [], []
            with torch.no_grad():
                self.arena.copy_(state_dict["arena"])
            return nn.modules.module._IncompatibleKeys(missing, unexpected)
        errors = [f"size mismatch for {k}: {tuple(state_dict[k].shape)} vs {tuple(views[k].shape)}"
                  for k in views if k in state_dict and tuple(state_dict[k].shape) != tuple(views[k].shape)]
        if strict and (missing or unexpected):
            errors.append(f"missing keys {missing}, unexpected keys {unexpected}")
        if errors:
            raise RuntimeError("Error(s) in loading state_dict for Transformer: " + "; ".join(errors))
        with torch.no_grad():
            for k, t in list(views.items()) + list(masks.items()):
                if k in state_dict:
                    t.copy_(state_dict[k])
            T = self.max_seq_len
            self._standard_mask = bool(((self.mask_bias == 0) == (torch.tril(torch.ones(T, T, device=self.mask_bias.device)) == 0)).all())
        return nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, idx: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, T = idx.shape
        assert T <= self.max_seq_len, f"Sequence length {T} exceeds model max length {self.max_seq_len}"
        p, C, H = self.views(), self.embed_dim, self.n_head
        drop = self.dropout if self.training else 0.0
        x = F.embedding(idx, p["token_emb.weight"]) + p["pos_emb"][:, :T]
        x = F.dropout(x, drop, self.training)
        if y is not None:
            x = x + F.embedding(y, p["class_emb.weight"])[:, None, :]
        scale = 1.0 / math.sqrt(C // H)
        hidden = self.mask_bias[:, :T, :T] == 0                                   # (layers, T, T): True = not attended
        fused = (self.fused_attention and idx.is_cuda and self._standard_mask and native.attention_kernel_covers(T, C // H))
        for i in range(self.n_layers):
            b = f"blocks.{i}."
            h = F.layer_norm(x, (C,), p[b + "ln1.weight"], p[b + "ln1.bias"])
            qkv = F.linear(h, p[b + "attn.c_attn.weight"], p[b + "attn.c_attn.bias"])
            if fused:
                a = native.causal_attention(qkv, H, drop)                         # one kernel (and one for its backward)
            else:
                q, k, v = qkv.view(B, T, 3, H, C // H).permute(2, 0, 3, 1, 4)     # each (B, H, T, C/H)
                w = (q @ k.transpose(-2, -1) * scale).masked_fill(hidden[i], float("-inf"))
                w = F.dropout(torch.softmax(w, dim=-1), drop, self.training)
                a = (w @ v).transpose(1, 2).reshape(B, T, C)
            x = x + F.dropout(F.linear(a, p[b + "attn.c_proj.weight"], p[b + "attn.c_proj.bias"]), drop, self.training)
            h = F.layer_norm(x, (C,), p[b + "ln2.weight"], p[b + "ln2.bias"])
            h = F.linear(F.gelu(F.linear(h, p[b + "mlp.0.weight"], p[b + "mlp.0.bias"])), p[b + "mlp.2.weight"], p[b + "mlp.2.bias"])
            x = x + F.dropout(h, drop, self.training)
        return F.linear(F.layer_norm(x, (C,), p["ln_f.weight"], p["ln_f.bias"]), p["head.weight"])
