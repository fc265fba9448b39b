from .transformer import TransformerEncoder, TransformerEncoderLayer

str2enc = {"transformer": TransformerEncoder}
__all__ = ["TransformerEncoder", "TransformerEncoderLayer", "str2enc"]
