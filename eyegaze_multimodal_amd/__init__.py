"""eyegaze_multimodal_amd — MI355X-native (gfx950) implementation of the dual-stream window classifier hot path
of roseDwayane/EyeGaze-Multimodal (train_art.py loop over DualEEGTransformer).  See DESIGN.md / INTEGRATION.md."""
from ._lib import EG_BF16, EG_F16, EG_F32, EgError, LIB_PATH  # noqa: F401
from .dual_eeg_transformer import DualEEGTransformer  # noqa: F401
from .optim import HipAdamW  # noqa: F401

__all__ = ["DualEEGTransformer", "HipAdamW", "EgError", "EG_BF16", "EG_F16", "EG_F32", "LIB_PATH"]
from . import ops  # noqa: E402,F401  (registers the eyegaze::* operators with torch.library)
