"""Name-keyed deterministic synthetic weights for the captioning models.

There is no network on the build or GPU boxes, so neither the torchvision ResNet-50
checkpoint the reference downloads (encoders.py:34) nor the released caption checkpoints
(deephumor_demo.ipynb:620-629) exist here.  Parity fixtures and the benchmark therefore use
synthetic weights that are

* a pure function of ``(seed, state-dict key, shape)`` -- every tensor is drawn from its own
  numpy Philox stream keyed by the CRC32 of its name, so the reference model (in the build
  container), the oracle and this package all receive bit-identical tensors regardless of
  construction order, device or torch version;
* scaled so activations stay O(1) through the 53-conv trunk with *eval-mode* BatchNorm
  (running stats near identity), and so greedy top-1/top-2 logit margins are far above fp32
  noise ("Hard parts").

The generator only needs ``{key: shape}``; it works on ``model.state_dict()`` of either this
package's models or the reference's (identical key layout).
"""
import re
import zlib

import numpy as np
import torch

__all__ = ["synth_state_dict", "load_synthetic", "synth_images"]


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def _normal(seed, name, shape, std):
    x = _rng(seed, name).standard_normal(size=tuple(shape))
    return torch.from_numpy((x * std).astype(np.float32))


def _uniform(seed, name, shape, lo, hi):
    x = _rng(seed, name).random(size=tuple(shape))
    return torch.from_numpy((lo + (hi - lo) * x).astype(np.float32))


_BN_LEAF = re.compile(r"(^|\.)(bn\d?|downsample\.1|resnet\.1|bn)\.(weight|bias|running_mean|running_var)$")


def _is_batchnorm(key, keys):
    base = key.rsplit(".", 1)[0]
    return (base + ".running_mean") in keys


def synth_state_dict(reference_sd, seed=1234, logit_std=2.5):
    """Returns a new state dict with the same keys/shapes/dtypes as ``reference_sd``.

    ``reference_sd`` values are only consulted for shape, dtype and for the fixed
    ``*.scale`` parameters (sqrt(head_dim) / sqrt(hid_dim), transformers.py:77-80,424-427),
    which are copied through unchanged.
    """
    keys = set(reference_sd.keys())
    is_transformer = any(k.endswith("tok_embedding.weight") for k in keys)
    out = {}
    for key, ref in reference_sd.items():
        shape = tuple(ref.shape)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros(shape, dtype=ref.dtype)
            continue
        if leaf == "scale" and ref.numel() == 1:
            out[key] = ref.detach().clone().cpu()
            continue
        if _is_batchnorm(key, keys):
            is_bn1d = key.startswith("encoder.bn.") or ".image_encoder.bn." in key
            if leaf == "weight":
                if is_bn1d and is_transformer:
                    # the image slot is divided by sqrt(hid_dim) with the token embeddings
                    # (transformers.py:462); keep it visible next to them
                    t = _uniform(seed, key, shape, 20.0, 30.0)
                elif key.endswith("bn3.weight"):
                    t = _uniform(seed, key, shape, 0.25, 0.35)   # damp the residual branch
                else:
                    t = _uniform(seed, key, shape, 0.9, 1.1)
            elif leaf == "bias":
                t = _uniform(seed, key, shape, -0.1, 0.1)
            elif leaf == "running_mean":
                t = _uniform(seed, key, shape, -0.1, 0.1)
            else:  # running_var
                t = _uniform(seed, key, shape, 0.9, 1.1)
            out[key] = t.to(ref.dtype)
            continue
        if ref.dim() == 4:  # conv weight, He init
            fan_in = shape[1] * shape[2] * shape[3]
            out[key] = _normal(seed, key, shape, (2.0 / fan_in) ** 0.5)
            continue
        if "embedding" in key:
            if key.endswith("pos_embedding.weight"):
                std = 0.5
            elif key.endswith("tok_embedding.weight"):
                std = 8.0    # divided by sqrt(hid_dim)=22.6 in forward
            else:
                std = 1.0
            out[key] = _normal(seed, key, shape, std)
            continue
        if ".lstm." in key:
            # strong recurrent weights: keeps greedy sequences varied instead of collapsing
            # onto one repeated token
            k = 6.0 / (shape[-1] ** 0.5) if ref.dim() == 2 else 0.1
            out[key] = _uniform(seed, key, shape, -k, k)
            continue
        if "_ln." in key:  # LayerNorm affine
            if leaf == "weight":
                out[key] = _uniform(seed, key, shape, 0.9, 1.1)
            else:
                out[key] = _uniform(seed, key, shape, -0.1, 0.1)
            continue
        if key.endswith("classifier.weight"):
            # LSTM hidden states have rms ~0.25, LayerNorm outputs rms ~1
            gain = 1.0 if is_transformer else 4.0
            out[key] = _normal(seed, key, shape, gain * logit_std / (shape[1] ** 0.5))
            continue
        if key.endswith("classifier.bias"):
            t = _normal(seed, key, shape, 0.5)
            # greedy decode (top_k=1) raises in the reference when <unk> (index 1) is the
            # arg-max, because every logit is then filtered (beam.py:35-36); keep it unlikely
            t[1] = -30.0
            out[key] = t
            continue
        if ref.dim() == 2:  # generic Linear
            # trunk features have rms ~5: bring the image embeddings back to O(1)
            gain = 0.2 if key.endswith("encoder.linear.weight") and shape[1] == 2048 else 1.0
            # post-LN residual stack: damp the sub-layer outputs so token/position/image
            # information survives 6 layers instead of collapsing onto one repeated token
            if key.endswith("fc_o.weight") or key.endswith("fc_2.weight"):
                gain = 0.3
            out[key] = _normal(seed, key, shape, gain / (shape[1] ** 0.5))
            continue
        if ref.dim() == 1:  # generic bias
            out[key] = _uniform(seed, key, shape, -0.05, 0.05)
            continue
        raise KeyError(f"synth_state_dict: no rule for {key} {shape}")
    # the reference shares ONE Embedding between the label encoder and the LSTM decoder
    # (caption_models.py:125): both keys must hold the same tensor
    a, b = "encoder.label_encoder.embedding.weight", "decoder.embedding.weight"
    if a in out and b in out:
        out[b] = out[a]
    return out


def load_synthetic(model, seed=1234, logit_std=2.5):
    """Loads synthetic weights into ``model`` (any nn.Module with the reference key layout)."""
    sd = synth_state_dict(model.state_dict(), seed=seed, logit_std=logit_std)
    model.load_state_dict(sd)
    return model


def synth_images(n, seed=0, size=224, first=0):
    """Deterministic synthetic ``[n, 3, size, size]`` fp32 images (ImageNet-normalised range).

    Pure white noise looks the same to a conv trunk after global average pooling, so every
    image would get the same caption.  Each image therefore gets its own contrast, per-channel
    brightness and a smooth 2-D gradient on top of the noise.  Image ``i`` depends only on
    ``(seed, first + i)`` -- a rank that owns images ``[first, first+n)`` of a larger batch
    generates exactly the rows the single-GPU run would.
    """
    out = np.empty((n, 3, size, size), dtype=np.float32)
    yy, xx = np.meshgrid(np.linspace(-1, 1, size), np.linspace(-1, 1, size), indexing="ij")
    for i in range(n):
        g = np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFF, 0x1A6E0000 + first + i]))
        contrast = g.uniform(0.3, 1.5)
        bright = g.uniform(-1.0, 1.0, size=(3, 1, 1))
        gx, gy = g.uniform(-1.0, 1.0, size=2)
        noise = g.standard_normal(size=(3, size, size))
        out[i] = (contrast * noise + bright + gx * xx + gy * yy).astype(np.float32)
    return torch.from_numpy(out)
