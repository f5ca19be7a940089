"""dg-sct_amd: MI355X-native DG-SCT cross-modal adapter path (hand-written gfx950 kernels behind a C ABI).

The directory name follows the reference repo name and is not a valid Python identifier; import it as
``import dgsct_amd`` (repo-root shim) or ``importlib.import_module("dg-sct_amd")``.
"""
from ._lib import Lib, default_lib, LIB_PATH, PARAM_NAMES          # noqa: F401
from .ops import AdapterSpec, map_pool                             # noqa: F401
from .adapter import VisualAdapter, bicubic_matrix                 # noqa: F401
from .stack import AdapterStack, ave_stage_shapes                  # noqa: F401
from .dp import GradAllReducer, init_process_group                 # noqa: F401
from .temporal import TemporalAttention, TemporalAttentionAVS, TemporalAttentionAVVP, frame_scale   # noqa: F401
from .backbone import FrozenBlocks, HTSATBlock, SwinV2Block         # noqa: F401
