"""Text <-> token helpers either side of the caption path.

Host-side string work with the behaviour of ``deephumor/experiments/inference.py:11-89``: the prompt prefix goes in
as token ids (``text_to_seq``), the generated ids come out as text (``seq_to_text``) and are cut into the meme's
top/bottom blocks (``split_caption``)."""
import re

import torch

from ..data import SPECIAL_TOKENS

# a space followed by a run of punctuation: the space is dropped when a block is cleaned (inference.py:8)
_SPACE_BEFORE_PUNCT = re.compile(r"( )([!#$%&\()*+,\-.\/:;<=>?@\\^{|}~]+)")
_SPECIAL = re.compile(r"<\w+>")


def text_to_seq(text, vocab, tokenizer):
    """``str`` -> int64 ``[1, seq_len]``: lower-case, tokenize, out-of-vocabulary tokens become ``<unk>``."""
    unk = vocab.stoi[SPECIAL_TOKENS['UNK']]
    ids = [vocab.stoi.get(tok, unk) for tok in tokenizer.tokenize(text.lower())]
    return torch.tensor(ids).unsqueeze(0)


def seq_to_text(seq, vocab, delimiter=' '):
    """1-D token tensor -> text, cut before the first ``<eos>``."""
    ids = seq.detach().cpu().reshape(-1).tolist()
    eos = vocab.stoi[SPECIAL_TOKENS['EOS']]
    if eos in ids:
        ids = ids[:ids.index(eos)]
    return delimiter.join(vocab.itos[i] for i in ids)


def _clean_block(block):
    block = _SPECIAL.sub('', block).strip(' \t\n\r\f\v')
    return _SPACE_BEFORE_PUNCT.sub(r'\2', block)


def split_caption(text, num_blocks=None):
    """Splits a caption at ``<sep>`` into cleaned blocks; pads with '' / truncates to ``num_blocks``."""
    blocks = [_clean_block(b) for b in text.split(SPECIAL_TOKENS['SEP'])]
    if num_blocks is None:
        return blocks
    return (blocks + [''] * max(0, num_blocks - len(blocks)))[:num_blocks]


IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def images_to_tensor(images_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """Decoded, already resized RGB images uint8 ``[N, H, W, 3]`` on the GPU -> the normalised fp32 ``[N, 3, H, W]``
    batch the encoders take: the notebook's ``ToTensor`` + ``Normalize`` (deephumor_demo.ipynb:565-567) as one
    device kernel."""
    from .. import hip
    dev = images_u8.device
    return hip.normalize_u8_hwc(images_u8.contiguous(), torch.tensor(mean, dtype=torch.float32, device=dev),
                                torch.tensor(std, dtype=torch.float32, device=dev))
