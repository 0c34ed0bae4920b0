"""Image captioning models: encoder + decoder compositions on the gfx950 kernels.

Drop-in for ``deephumor.models.caption_models`` (reference caption_models.py:9-461): same class
names, constructor arguments, ``forward`` / ``generate`` / ``save`` / ``from_pretrained`` and
checkpoint format ``{'model': state_dict, 'hp': dict}``.  New: ``generate_batch`` (N images at a
time; the reference handles exactly one) and ``CaptioningTransformerWithLabels`` (BASELINE
config 5).
"""
import torch
from torch import nn

from .encoders import ImageEncoder, ImageLabelEncoder, SpatialImageLabelEncoder
from .rnn_models import LSTMDecoder
from .transformers import SelfAttentionTransformerDecoder, TransformerDecoder


class _CaptioningBase(nn.Module):
    _GEN_KEYS = ("caption", "max_len", "temperature", "beam_size", "top_k", "eos_index")

    def save(self, ckpt_path):
        """Saves the model's state and hyperparameters (reference caption_models.py:76-81)."""
        torch.save({'model': self.state_dict(), 'hp': self._hp}, ckpt_path)

    @classmethod
    def from_pretrained(cls, ckpt_path):
        """Loads and builds the model from a checkpoint file (reference caption_models.py:83-98)."""
        ckpt = torch.load(ckpt_path, map_location='cpu')
        model = cls(**ckpt['hp'])
        model.load_state_dict(ckpt['model'])
        return model

    @staticmethod
    def _one(toks, lens):
        return toks[0, :int(lens[0])].squeeze()


class CaptioningLSTM(_CaptioningBase):
    """LSTM-based image captioning model (reference caption_models.py:9-98)."""

    def __init__(self, num_tokens, emb_dim=256, hidden_size=512, num_layers=2,
                 enc_dropout=0.3, dec_dropout=0.1):
        super().__init__()
        self.encoder = ImageEncoder(emb_dim=emb_dim, dropout=enc_dropout)
        self.decoder = LSTMDecoder(num_tokens=num_tokens, emb_dim=emb_dim, hidden_size=hidden_size,
                                   num_layers=num_layers, dropout=dec_dropout)
        self._hp = {'num_tokens': num_tokens, 'emb_dim': emb_dim, 'hidden_size': hidden_size,
                    'num_layers': num_layers, 'enc_dropout': enc_dropout, 'dec_dropout': dec_dropout}

    def forward(self, images, captions, lengths=None):
        return self.decoder(self.encoder(images), captions, lengths)

    def generate_batch(self, images, caption=None, max_len=25, temperature=1.0, beam_size=10, top_k=50,
                       eos_index=3, **kw):
        return self.decoder.generate_batch(self.encoder(images), caption=caption, max_len=max_len,
                                           temperature=temperature, beam_size=beam_size, top_k=top_k,
                                           eos_index=eos_index, **kw)

    def generate(self, image, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        return self._one(*self.generate_batch(image, caption, max_len, temperature, beam_size, top_k, eos_index, **kw))


class CaptioningLSTMWithLabels(_CaptioningBase):
    """LSTM captioning model conditioned on image + text label (reference caption_models.py:101-195).
    The label encoder and the decoder share ONE embedding (caption_models.py:125)."""

    def __init__(self, num_tokens, emb_dim=256, hidden_size=512, num_layers=2,
                 enc_dropout=0.3, dec_dropout=0.1):
        super().__init__()
        self.encoder = ImageLabelEncoder(num_tokens=num_tokens, emb_dim=emb_dim, dropout=enc_dropout)
        self.decoder = LSTMDecoder(num_tokens=num_tokens, emb_dim=emb_dim, hidden_size=hidden_size,
                                   num_layers=num_layers, dropout=dec_dropout,
                                   embedding=self.encoder.label_encoder.embedding)
        self._hp = {'num_tokens': num_tokens, 'emb_dim': emb_dim, 'hidden_size': hidden_size,
                    'num_layers': num_layers, 'enc_dropout': enc_dropout, 'dec_dropout': dec_dropout}

    def forward(self, images, captions, lengths, labels):
        return self.decoder(self.encoder(images=images, labels=labels), captions, lengths)

    def generate_batch(self, images, labels, caption=None, max_len=25, temperature=1.0, beam_size=10, top_k=50,
                       eos_index=3, **kw):
        return self.decoder.generate_batch(self.encoder(images, labels), caption=caption, max_len=max_len,
                                           temperature=temperature, beam_size=beam_size, top_k=top_k,
                                           eos_index=eos_index, **kw)

    def generate(self, image, label, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        return self._one(*self.generate_batch(image, label, caption, max_len, temperature, beam_size, top_k,
                                              eos_index, **kw))


class _TransformerHP:
    def _set_hp(self, num_tokens, hid_dim, n_layers, n_heads, pf_dim, enc_dropout, dec_dropout, pad_index, max_len):
        self._hp = {'num_tokens': num_tokens, 'hid_dim': hid_dim, 'n_layers': n_layers, 'n_heads': n_heads,
                    'pf_dim': pf_dim, 'enc_dropout': enc_dropout, 'dec_dropout': dec_dropout,
                    'pad_index': pad_index, 'max_len': max_len}


class CaptioningTransformerBase(_CaptioningBase, _TransformerHP):
    """Transformer captioning model without encoder attention (reference caption_models.py:198-327)."""

    def __init__(self, num_tokens, hid_dim=512, n_layers=6, n_heads=8, pf_dim=2048,
                 enc_dropout=0.3, dec_dropout=0.1, pad_index=0, max_len=128):
        super().__init__()
        self.encoder = ImageEncoder(emb_dim=hid_dim, dropout=enc_dropout, spatial_features=False)
        self.decoder = SelfAttentionTransformerDecoder(num_tokens=num_tokens, hid_dim=hid_dim, n_layers=n_layers,
                                                       n_heads=n_heads, pf_dim=pf_dim, dropout=dec_dropout,
                                                       pad_index=pad_index, max_len=max_len)
        self._set_hp(num_tokens, hid_dim, n_layers, n_heads, pf_dim, enc_dropout, dec_dropout, pad_index, max_len)

    def forward(self, images, captions, lengths=None):
        return self.decoder(captions, start_emb=self.encoder(images))

    def generate_batch(self, images, caption=None, max_len=25, temperature=1.0, beam_size=10, top_k=50,
                       eos_index=3, **kw):
        return self.decoder.generate_batch(self.encoder(images), caption=caption, max_len=max_len,
                                           temperature=temperature, beam_size=beam_size, top_k=top_k,
                                           eos_index=eos_index, **kw)

    def generate(self, image, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        return self._one(*self.generate_batch(image, caption, max_len, temperature, beam_size, top_k, eos_index, **kw))


class CaptioningTransformer(_CaptioningBase, _TransformerHP):
    """Transformer captioning model attending over the 7x7 spatial image features
    (reference caption_models.py:330-461)."""

    def __init__(self, num_tokens, hid_dim=512, n_layers=6, n_heads=8, pf_dim=2048,
                 enc_dropout=0.3, dec_dropout=0.1, pad_index=0, max_len=128):
        super().__init__()
        self.encoder = ImageEncoder(emb_dim=hid_dim, dropout=enc_dropout, spatial_features=True)
        self.decoder = TransformerDecoder(num_tokens=num_tokens, hid_dim=hid_dim, n_layers=n_layers,
                                          n_heads=n_heads, pf_dim=pf_dim, dropout=dec_dropout,
                                          pad_index=pad_index, max_len=max_len)
        self._set_hp(num_tokens, hid_dim, n_layers, n_heads, pf_dim, enc_dropout, dec_dropout, pad_index, max_len)

    def forward(self, images, captions, lengths=None):
        image_emb, image_spatial_emb = self.encoder(images)
        return self.decoder(captions, enc_out=image_spatial_emb, start_emb=image_emb)

    def generate_batch(self, images, caption=None, max_len=25, temperature=1.0, beam_size=10, top_k=50,
                       eos_index=3, **kw):
        image_emb, image_spatial_emb = self.encoder(images)
        return self.decoder.generate_batch(image_emb, image_spatial_emb, caption=caption, max_len=max_len,
                                           temperature=temperature, beam_size=beam_size, top_k=top_k,
                                           eos_index=eos_index, **kw)

    def generate(self, image, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        return self._one(*self.generate_batch(image, caption, max_len, temperature, beam_size, top_k, eos_index, **kw))


class CaptioningTransformerWithLabels(_CaptioningBase, _TransformerHP):
    """BASELINE config 5: ImageLabelEncoder (with spatial features) + CaptioningTransformer decoder.
    No reference class exists; composition defined in SURVEY.md 8(a) row A4 from reference parts."""

    def __init__(self, num_tokens, hid_dim=512, n_layers=6, n_heads=8, pf_dim=2048,
                 enc_dropout=0.3, dec_dropout=0.1, pad_index=0, max_len=128):
        super().__init__()
        self.encoder = SpatialImageLabelEncoder(num_tokens=num_tokens, emb_dim=hid_dim, dropout=enc_dropout)
        self.decoder = TransformerDecoder(num_tokens=num_tokens, hid_dim=hid_dim, n_layers=n_layers,
                                          n_heads=n_heads, pf_dim=pf_dim, dropout=dec_dropout,
                                          pad_index=pad_index, max_len=max_len)
        self._set_hp(num_tokens, hid_dim, n_layers, n_heads, pf_dim, enc_dropout, dec_dropout, pad_index, max_len)

    def forward(self, images, captions, lengths, labels):
        start, spatial = self.encoder(images, labels)
        return self.decoder(captions, enc_out=spatial, start_emb=start)

    def generate_batch(self, images, labels, caption=None, max_len=25, temperature=1.0, beam_size=10, top_k=50,
                       eos_index=3, **kw):
        start, spatial = self.encoder(images, labels)
        return self.decoder.generate_batch(start, spatial, caption=caption, max_len=max_len,
                                           temperature=temperature, beam_size=beam_size, top_k=top_k,
                                           eos_index=eos_index, **kw)

    def generate(self, image, label, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        return self._one(*self.generate_batch(image, label, caption, max_len, temperature, beam_size, top_k,
                                              eos_index, **kw))
