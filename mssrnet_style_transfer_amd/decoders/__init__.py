from .transformer import TransformerDecoder, TransformerDecoderLayer

str2dec = {"transformer": TransformerDecoder}
__all__ = ["TransformerDecoder", "TransformerDecoderLayer", "str2dec"]
