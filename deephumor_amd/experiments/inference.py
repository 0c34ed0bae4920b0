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


_COEFF_CACHE = {}


def resize_coefficients(in_size, out_size):
    """Filter table of one axis of Pillow's antialiased BILINEAR resample (``precompute_coeffs`` + ``normalize_coeffs_8bpc``
    of Pillow's Resample.c): ``(bounds int32 [out, 2] = (first source index, count), weights int32 [out, ksize])`` with
    22-bit fixed-point weights.  Host-side double precision arithmetic, like Pillow's; a few KB per (in, out) pair."""
    import numpy as np
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale                                   # triangle filter of half-width 1, stretched when shrinking
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    weights = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        n = min(int(center + support + 0.5), in_size) - lo
        arg = (np.arange(n, dtype=np.float64) + lo - center + 0.5) / filterscale
        w = np.where(np.abs(arg) < 1.0, 1.0 - np.abs(arg), 0.0)
        if w.sum() != 0.0:
            w = w / w.sum()
        weights[xx, :n] = np.where(w < 0, -0.5 + w * (1 << 22), 0.5 + w * (1 << 22)).astype(np.int64)
        bounds[xx] = (lo, n)
    return bounds, weights


def _coeffs(in_size, out_size, dev):
    if in_size == out_size:
        return None
    key = (in_size, out_size, str(dev))
    if key not in _COEFF_CACHE:
        b, w = resize_coefficients(in_size, out_size)
        _COEFF_CACHE[key] = (torch.from_numpy(b).to(dev), torch.from_numpy(w).to(dev))
    return _COEFF_CACHE[key]


def resize_images(images_u8, size=(224, 224)):
    """Decoded RGB images uint8 ``[N, H, W, 3]`` on the GPU -> uint8 ``[N, size[0], size[1], 3]``: the notebook's
    ``transforms.Resize((224, 224))`` (deephumor_demo.ipynb:565; Pillow's antialiased bilinear resample) as device kernels,
    bit-identical to Pillow (``tests/golden/g9_resize.npz``)."""
    from .. import hip
    n, h, w, c = images_u8.shape
    return hip.resize_u8_hwc(images_u8.contiguous(), size[0], size[1], _coeffs(w, size[1], images_u8.device),
                             _coeffs(h, size[0], images_u8.device))


def preprocess_images(images_u8, size=(224, 224), mean=IMAGENET_MEAN, std=IMAGENET_STD, dtype=torch.float32):
    """The whole notebook transform (Resize -> ToTensor -> Normalize, deephumor_demo.ipynb:565-567) on device for a batch
    of equally sized decoded images uint8 ``[N, H, W, 3]``.  ``dtype=torch.float32``: the fp32 ``[N, 3, 224, 224]`` batch
    the reference's encoders take; ``torch.bfloat16`` / ``torch.float16``: the normalised batch already in the 16-bit
    channels-last ``[N, 224, 224, 8]`` layout the matrix-core stem convolution reads (normalise + pack fused, no fp32 tensor
    in front of conv1) -- ``ImageEncoder`` accepts it in place of the NCHW batch and gives identical features."""
    from .. import hip
    dev = images_u8.device
    x = resize_images(images_u8, size) if tuple(images_u8.shape[1:3]) != tuple(size) else images_u8.contiguous()
    m, s = torch.tensor(mean, dtype=torch.float32, device=dev), torch.tensor(std, dtype=torch.float32, device=dev)
    if dtype == torch.float32:
        return hip.normalize_u8_hwc(x, m, s)
    return hip.normalize_pack_u8(x, m, s, out_dtype=dtype)


def images_to_tensor(images_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """Decoded, already resized RGB images uint8 ``[N, H, W, 3]`` on the GPU -> the normalised fp32 ``[N, 3, H, W]``
    batch the encoders take: the notebook's ``ToTensor`` + ``Normalize`` (deephumor_demo.ipynb:565-567) as one
    device kernel."""
    from .. import hip
    dev = images_u8.device
    return hip.normalize_u8_hwc(images_u8.contiguous(), torch.tensor(mean, dtype=torch.float32, device=dev),
                                torch.tensor(std, dtype=torch.float32, device=dev))
