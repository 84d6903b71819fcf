"""MI355X-native semi-Markov decode path of dpfried/action-segmentation.

Host side (Python / PyTorch-ROCm) mirrors the reference's operator interface for this path only:
``SemiMarkovModule`` (score_features / viterbi / log_likelihood / fit_supervised) and ``SemiMarkovModel``
(--classifier semimarkov).  The arithmetic runs in ``libsmmdp.so`` (hand-written HIP for gfx950, C ABI in
``include/smmdp.h``); there is no CPU fallback: without the library or without a GPU the ops raise.
"""
from . import _build, _lib  # noqa: F401

__all__ = ["_build", "_lib"]
