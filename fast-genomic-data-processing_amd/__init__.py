"""MI355X-native PairHMM likelihood engine, sort/mark-duplicate core and Smith-Waterman aligner.

Python is only a thin ctypes veneer over the C ABI of ``libmgx.so`` (include/mgx_pairhmm.h,
include/mgx_sortdedup.h, include/mgx_smithwaterman.h, include/mgx_bgzf.h); there is no Python or CPU compute path.  If the HIP library has not
been built, or no HIP device is visible, every entry point raises.
"""
from . import native  # noqa: F401
from .native import MgxError, lib_path, load  # noqa: F401
from .pairhmm import PairHMMEngine, PairHMMBatch, PairHMMQueue  # noqa: F401
from .sortdedup import SortDedupEngine, Routed  # noqa: F401
from .smithwaterman import SmithWatermanEngine  # noqa: F401
from .bgzf import BgzfCompressor, BgzfBatch, BgzfStore  # noqa: F401
from . import pairhmm, sortdedup, smithwaterman, bgzf, synth  # noqa: F401
