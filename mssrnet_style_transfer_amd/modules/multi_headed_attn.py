"""MultiHeadedAttention on the gfx950 kernels.

Drop-in for onmt/modules/multi_headed_attn.py:11-230 of the reference: same constructor,
attribute names (state-dict keys ``linear_{keys,values,query}.{weight,bias}``,
``final_linear.*``), forward signature, return values and ``layer_cache`` protocol.
"""
import torch
import torch.nn as nn

from .. import ops


class MultiHeadedAttention(nn.Module):
    def __init__(self, head_count, model_dim, dropout=0.1, max_relative_positions=0):
        assert model_dim % head_count == 0
        if max_relative_positions != 0:
            raise NotImplementedError("relative positions are unused by MSSRNet (max_relative_positions=0)")
        self.dim_per_head = model_dim // head_count
        self.model_dim = model_dim
        super().__init__()
        self.head_count = head_count
        self.linear_keys = nn.Linear(model_dim, head_count * self.dim_per_head)
        self.linear_values = nn.Linear(model_dim, head_count * self.dim_per_head)
        self.linear_query = nn.Linear(model_dim, head_count * self.dim_per_head)
        self.softmax = nn.Softmax(dim=-1)       # attribute kept for pickle/ABI parity; not called
        self.dropout = nn.Dropout(dropout)      # holds p; the kernel path reads .p and .training
        self.final_linear = nn.Linear(model_dim, model_dim)
        self.max_relative_positions = max_relative_positions

    def _proj(self, lin, x):
        return ops.linear(x, lin.weight, lin.bias)

    def forward(self, key, value, query, mask=None, layer_cache=None, attn_type=None):
        """key/value/query: [B, len, d] batch-first.  mask: Bool [B,1,Lk] or [B,Lq,Lk], True =
        masked (filled with -1e18 before the fp32 softmax).  Returns (output [B,Lq,d],
        attention [B,H,Lq,Lk] = probabilities BEFORE dropout)."""
        B = key.size(0)
        H, dh = self.head_count, self.dim_per_head

        def shape(x):       # [B, len, d] -> [B, H, len, dh] view, no copy
            return x.view(B, -1, H, dh).transpose(1, 2)

        if layer_cache is not None:
            if attn_type == "self":
                q = self._proj(self.linear_query, query)
                k = shape(self._proj(self.linear_keys, query))
                v = shape(self._proj(self.linear_values, query))
                if layer_cache["self_keys"] is not None:
                    k = torch.cat((layer_cache["self_keys"], k), dim=2)
                if layer_cache["self_values"] is not None:
                    v = torch.cat((layer_cache["self_values"], v), dim=2)
                layer_cache["self_keys"] = k
                layer_cache["self_values"] = v
            elif attn_type == "context":
                q = self._proj(self.linear_query, query)
                if layer_cache["memory_keys"] is None:
                    k = shape(self._proj(self.linear_keys, key))
                    v = shape(self._proj(self.linear_values, value))
                else:
                    k, v = layer_cache["memory_keys"], layer_cache["memory_values"]
                layer_cache["memory_keys"] = k
                layer_cache["memory_values"] = v
            else:
                raise ValueError("layer_cache needs attn_type 'self' or 'context'")
        else:
            k = shape(self._proj(self.linear_keys, key))
            v = shape(self._proj(self.linear_values, value))
            q = self._proj(self.linear_query, query)
        q = shape(q)
        ctx, attn = ops.attention(q, k, v, mask, self.dropout.p, self.training)
        Lq = q.size(2)
        context = ctx.permute(0, 2, 1, 3).reshape(B, Lq, H * dh)     # a view: ctx is stored [B,Lq,H,dh]
        output = self._proj(self.final_linear, context)
        return output, attn

    def update_dropout(self, dropout):
        self.dropout.p = dropout
