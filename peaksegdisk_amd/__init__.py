"""peaksegdisk_amd -- MI355X-native drop-in for PeakSegDisk's PeakSegFPOP hot path.

The constrained functional-pruning dynamic program of tdhock/PeakSegDisk
(src/PeakSegFPOPLog.cpp + src/funPieceListLog.cpp) as hand-written HIP kernels for gfx950
behind the reference's own entry points.  See DESIGN.md and INTEGRATION.md.
"""
from . import _native  # noqa: F401  (fails loudly when the HIP library is not built)
from .api import (PeakSegError, PeakSegFPOP_dir, PeakSegFPOP_df, PeakSegFPOP_file,  # noqa: F401
                  PeakSegFPOP_vec, PeakSegFPOP_dir_batch, col_name_list, paste,
                  sequentialSearch_dir, sequentialSearch_dir_batch, writeBedGraph)
from .grid import ProblemSet  # noqa: F401

__all__ = ["PeakSegFPOP_file", "PeakSegFPOP_dir", "PeakSegFPOP_df", "PeakSegFPOP_vec",
           "PeakSegFPOP_dir_batch", "sequentialSearch_dir", "sequentialSearch_dir_batch",
           "writeBedGraph", "col_name_list", "paste",
           "ProblemSet",
           "PeakSegError"]
