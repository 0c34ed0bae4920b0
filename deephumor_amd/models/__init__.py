"""Mirror of ``deephumor.models`` (reference models/__init__.py:1-25) on the gfx950 kernels."""
from .encoders import ImageEncoder, ImageLabelEncoder, LabelEncoder, SpatialImageLabelEncoder
from .beam import BeamSearchHelper
from .rnn_models import LSTMDecoder
from .transformers import (
    TransformerEncoder,
    TransformerDecoder,
    SelfAttentionTransformerDecoder,
    MultiHeadAttentionLayer,
)
from .caption_models import (
    CaptioningLSTM,
    CaptioningLSTMWithLabels,
    CaptioningTransformerBase,
    CaptioningTransformer,
    CaptioningTransformerWithLabels,
)

__all__ = [
    'ImageEncoder', 'ImageLabelEncoder', 'LabelEncoder', 'SpatialImageLabelEncoder', 'BeamSearchHelper',
    'LSTMDecoder', 'TransformerEncoder', 'TransformerDecoder', 'SelfAttentionTransformerDecoder',
    'MultiHeadAttentionLayer', 'CaptioningLSTM', 'CaptioningLSTMWithLabels', 'CaptioningTransformerBase',
    'CaptioningTransformer', 'CaptioningTransformerWithLabels',
]
