"""Token-id conventions and tokenizers of ``deephumor.data`` that the caption path depends on
(vocab.py:5-42, tokenizers.py:14-29).  Dataset / crawler / langdetect utilities are out of scope."""
from .vocab import SPECIAL_TOKENS, Vocab, build_vocab
from .tokenizers import Tokenizer, WordPunctTokenizer, CharTokenizer

__all__ = ["SPECIAL_TOKENS", "Vocab", "build_vocab", "Tokenizer", "WordPunctTokenizer", "CharTokenizer"]
