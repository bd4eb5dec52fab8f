"""unitspeech_amd: MI355X-native (gfx950) implementation of the UnitSpeech diffusion-decoder hot path.

Public surface mirrors `unitspeech/unitspeech.py` of the reference: `UnitSpeech`, `GradLogPEstimator2d`."""
from .params import DecoderConfig, synthetic_inputs, synthetic_state_dict  # noqa: F401
from .optim import FusedAdam  # noqa: F401
from .unitspeech import GradLogPEstimator2d, UnitSpeech  # noqa: F401

__all__ = ["UnitSpeech", "GradLogPEstimator2d", "FusedAdam", "DecoderConfig", "synthetic_state_dict", "synthetic_inputs"]
