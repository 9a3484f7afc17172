from . import depth_aa, mlbw, row_flow_v3  # noqa: F401  (registers iw3.depth_aa, sbs.row_flow_v3, sbs.mlbw + factories)
from .depth_aa import DepthAA  # noqa: F401
