"""Decoder-only Transformer over code sequences -- the reference's prior (src/models/transformer.py:10-133) with the
same constructor, the same parameter / buffer names (a reference `best.pt` state dict loads with strict=True) and the
same arithmetic: pre-LayerNorm blocks, causal self-attention as explicit QK^T / softmax / V products (sequences are
H*W - 1 = 15 tokens: nothing to tile), GELU MLP of width 4x, learned positions, optional class embedding added to every
position, untied output head.  PyTorch-ROCm modules; the data-parallel training loop is vqvae_amd/prior/train.py."""
import math
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


class CausalSelfAttention(nn.Module):
    def __init__(self, embed_dim: int, n_head: int, max_seq_len: int, dropout: float):
        super().__init__()
        assert embed_dim % n_head == 0
        self.c_attn = nn.Linear(embed_dim, 3 * embed_dim)
        self.c_proj = nn.Linear(embed_dim, embed_dim)
        self.attn_dropout = nn.Dropout(dropout)
        self.resid_dropout = nn.Dropout(dropout)
        self.n_head, self.embed_dim = n_head, embed_dim
        # lower-triangular mask, kept in the state dict under the reference's name
        self.register_buffer("bias", torch.tril(torch.ones(max_seq_len, max_seq_len)).view(1, 1, max_seq_len, max_seq_len))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, T, C = x.shape
        hd = C // self.n_head
        q, k, v = (t.view(B, T, self.n_head, hd).transpose(1, 2) for t in self.c_attn(x).split(self.embed_dim, dim=2))
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd))
        att = att.masked_fill(self.bias[:, :, :T, :T] == 0, float("-inf"))
        att = self.attn_dropout(F.softmax(att, dim=-1))
        y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
        return self.resid_dropout(self.c_proj(y))


class Block(nn.Module):
    def __init__(self, embed_dim: int, n_head: int, max_seq_len: int, dropout: float):
        super().__init__()
        self.ln1 = nn.LayerNorm(embed_dim)
        self.ln2 = nn.LayerNorm(embed_dim)
        self.attn = CausalSelfAttention(embed_dim, n_head, max_seq_len, dropout)
        self.mlp = nn.Sequential(nn.Linear(embed_dim, 4 * embed_dim), nn.GELU(), nn.Linear(4 * embed_dim, embed_dim),
                                 nn.Dropout(dropout))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x + self.attn(self.ln1(x))
        return x + self.mlp(self.ln2(x))


class Transformer(nn.Module):
    def __init__(self, num_classes: int, num_tokens: int, embed_dim: int, n_layers: int, n_head: int, max_seq_len: int,
                 dropout: float = 0.1):
        super().__init__()
        self.num_classes, self.num_tokens, self.embed_dim = num_classes, num_tokens, embed_dim
        self.n_layers, self.n_head, self.max_seq_len = n_layers, n_head, max_seq_len
        self.token_emb = nn.Embedding(num_tokens, embed_dim)
        self.pos_emb = nn.Parameter(torch.zeros(1, max_seq_len, embed_dim))
        self.drop = nn.Dropout(dropout)
        if num_classes > 0:
            self.class_emb = nn.Embedding(num_classes, embed_dim)
        self.blocks = nn.ModuleList([Block(embed_dim, n_head, max_seq_len, dropout) for _ in range(n_layers)])
        self.ln_f = nn.LayerNorm(embed_dim)
        self.head = nn.Linear(embed_dim, num_tokens, bias=False)
        self.apply(self._init_weights)

    def _init_weights(self, module):
        """Reference initialisation (transformer.py:43-53): N(0, 0.02) for Linear / Embedding weights and the positions,
        zero biases, unit LayerNorm."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            torch.nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if isinstance(module, nn.Linear) and module.bias is not None:
                torch.nn.init.zeros_(module.bias)
        elif isinstance(module, nn.LayerNorm):
            torch.nn.init.zeros_(module.bias)
            torch.nn.init.ones_(module.weight)
        elif isinstance(module, Transformer):
            torch.nn.init.normal_(module.pos_emb, mean=0.0, std=0.02)

    def forward(self, idx: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, T = idx.shape
        assert T <= self.max_seq_len, f"Sequence length {T} exceeds model max length {self.max_seq_len}"
        x = self.drop(self.token_emb(idx) + self.pos_emb[:, :T, :])
        if y is not None:
            x = x + self.class_emb(y).unsqueeze(1)
        for block in self.blocks:
            x = block(x)
        return self.head(self.ln_f(x))
