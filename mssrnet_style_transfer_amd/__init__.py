"""MI355X-native hot path of MSSRNet: the OpenNMT-derived transformer encoder/decoder
training step (MultiHeadedAttention, PositionwiseFeedForward, LayerNorm, positional
embeddings; forward, backward and the WGAN-GP double-backward) on hand-written gfx950 HIP
kernels behind a C-ABI library (include/mssr_hip.h).  See DESIGN.md.
"""
__version__ = "0.1.0"
