"""pgrc_amd -- MI355X-native implementation of PgRC's read-to-pseudogenome matching path.

The product is `libpgrc_match.so` (hand-written HIP kernels for gfx950 behind the C ABI of
`include/pgrc_match.h`).  This package is the thin host-side mirror of the reference's matcher
interface (matching/ReadsMatchers.h) on top of that ABI; it contains no compute of its own.
"""
from ._lib import LIB_PATH, NOT_MATCHED_CNT, NOT_MATCHED_POS, PgrcMatchError  # noqa: F401
from .matchers import (  # noqa: F401
    AbstractReadsApproxMatcher,
    CopMEMReadsApproxMatcher,
    DefaultReadsApproxMatcher,
    DefaultReadsExactMatcher,
    DefaultReadsMatcher,
    InterleavedReadsApproxMatcher,
    MatchContext,
    copmem_params,
    mapReadsIntoPg,
)
from .textmatch import CopMEMMatcher  # noqa: F401
from .readsets import DividedPCLReadsSets  # noqa: F401
from . import synth  # noqa: F401
