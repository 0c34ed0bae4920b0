"""Token vocabulary with the reference's id layout: the six special tokens first
(``<pad>``=0 ``<unk>``=1 ``<bos>``=2 ``<eos>``=3 ``<sep>``=4 ``<emp>``=5, vocab.py:5-12), then the sorted tokens.
The beam kernels rely on pad 0 / unk 1 / eos 3."""
from collections import Counter

SPECIAL_TOKENS = {'PAD': '<pad>', 'UNK': '<unk>', 'BOS': '<bos>', 'EOS': '<eos>', 'SEP': '<sep>', 'EMPTY': '<emp>'}


class Vocab:
    def __init__(self, tokens, special_tokens=tuple(SPECIAL_TOKENS.values())):
        rest = sorted(t for t in tokens if t not in special_tokens)
        self.tokens = list(special_tokens) + rest
        self.stoi = {tok: i for i, tok in enumerate(self.tokens)}
        self.itos = dict(enumerate(self.tokens))

    def __iter__(self):
        return iter(self.tokens)

    def __len__(self):
        return len(self.tokens)

    def save(self, filepath):
        with open(filepath, 'w') as f:
            f.writelines(tok + '\n' for tok in self.tokens)

    @staticmethod
    def load(filepath):
        with open(filepath) as f:
            return Vocab([line.rstrip('\n') for line in f])


def build_vocab(documents, tokenizer, min_df=7):
    """Vocabulary of the tokens that occur in at least ``min_df`` documents (vocab.py:45-70)."""
    df = Counter()
    for text in documents:
        df.update(set(tokenizer.tokenize(text.lower())))
    return Vocab([tok for tok, n in df.items() if n >= min_df])
