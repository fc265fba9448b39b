"""Hot-path modules with the reference's names (onmt/modules/__init__.py re-exports)."""
from .embeddings import Embeddings, PositionalEncoding
from .multi_headed_attn import MultiHeadedAttention
from .position_ffn import PositionwiseFeedForward
from .util_class import Cast, Elementwise

__all__ = ["Embeddings", "PositionalEncoding", "MultiHeadedAttention", "PositionwiseFeedForward",
           "Cast", "Elementwise"]
