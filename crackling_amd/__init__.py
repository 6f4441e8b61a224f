"""crackling_amd -- MI355X-native drop-in for Crackling's ISSL off-target scoring step.

Only the hot path of bmds-lab/Crackling is here: `isslScoreOfftargets`
(reference: src/ISSL/isslScoreOfftargets.cpp, called from src/crackling/Crackling.py:727-837) and the
producer of its input format, `isslCreateIndex` (src/ISSL/isslCreateIndex.cpp).  The compute lives in
libissl_hip.so (hand-written HIP for gfx950 behind the C ABI of include/issl_hip.h); this package is
the thin host-side mirror used by tests, bench.py and Python callers.  Importing it without the built
library raises immediately -- there is no CPU fallback.
"""
from ._lib import lib, IsslError, LIB_PATH  # noqa: F401
from .scorer import (  # noqa: F401
    IsslIndex,
    IsslNode,
    METHODS,
    encode_guides,
    extract_offtargets,
    decode_guides,
    format_scores,
    format_scores_native,
    run_scorer_binary,
    parse_scorer_output,
    verdicts,
)

__all__ = [
    "IsslIndex", "IsslNode", "IsslError", "METHODS", "encode_guides", "extract_offtargets", "decode_guides", "format_scores", "format_scores_native",
    "run_scorer_binary", "parse_scorer_output", "verdicts", "lib", "LIB_PATH",
]
