"""Regex tokenizers of the reference (tokenizers.py:14-29): special tokens in ``<>`` stay whole."""
import re


class Tokenizer:
    def tokenize(self, text):
        raise NotImplementedError


class WordPunctTokenizer(Tokenizer):
    """Words (with ``'`` and ``<>``) or runs of punctuation."""
    token_pattern = re.compile(r"[<\w'>]+|[^\w\s]+")

    def tokenize(self, text):
        return self.token_pattern.findall(text)


class CharTokenizer(Tokenizer):
    """Single characters, except ``<special>`` tokens."""
    token_pattern = re.compile(r"<\w+>|.")

    def tokenize(self, text):
        return self.token_pattern.findall(text)
