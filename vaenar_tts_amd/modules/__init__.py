"""Mirror of the reference's ``modules`` exports (/root/reference/modules/__init__.py:1-5)."""
from .encoder import TransformerEncoder
from .decoder import TransformerDecoder
from .prior import TransformerPrior
from .posterior import TransformerPosterior
from .length_predictor import DenseLengthPredictor

__all__ = ["TransformerEncoder", "TransformerDecoder", "TransformerPrior", "TransformerPosterior",
           "DenseLengthPredictor"]
