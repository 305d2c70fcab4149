"""agcn_amd: MI355X-native implementation of the AGCN adaptive graph-convolution hot path
(SGC_LL / SGC_LL_Reslap forward + backward), behind the reference's layer call contract.

Python here is host code only: hand-written HIP kernels (csrc/) are reached through the C ABI
of include/agcn.h via ctypes.  There is no CPU or eager fallback."""
from . import _build, _ffi  # noqa: F401
from ._ffi import AgcnError  # noqa: F401
from .batch import GraphBatch, Plan, RowViews, TileViews  # noqa: F401
from .graph_structure import Graph, MolGraph, laplacian_csr  # noqa: F401
from .layers import (SGC_LL, SGC_LL_Reslap, GraphGatherMol, DenseMol, BlockEnd, DenseBlockEnd, GraphPoolMol,  # noqa: F401
                     MultitaskHead, Layer, set_debug_checks)
from .graphs import (SequentialGraphMol, ResidualGraphMol, ResidualGraphMolResLap, DenseConnectedGraph,  # noqa: F401
                     DenseConnectedGraphResLap, SegmentationGraph)
from .store import GraphStore, ResidentStore  # noqa: F401
from .engine import GraphedStep, SgcLLEpochLanes, SgcLLEpochPipeline, SgcLLEpochStep, SgcLLStep  # noqa: F401
from .functional import (check_device_status, capture_scope, release_reserve_pool, reserve_pool_bytes,  # noqa: F401
                         dispatch, run_checked, device_timeouts, recovered_calls, param_grad_fork, no_member_flags)
from . import dist  # noqa: F401

__version__ = "0.1.0"
