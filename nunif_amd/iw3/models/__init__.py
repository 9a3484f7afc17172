from . import depth_aa, light_inpaint_v1, light_video_inpaint_v1, mlbw, row_flow_v3  # noqa: F401  (registers iw3.depth_aa, sbs.row_flow_v3, sbs.mlbw, inpaint.light_inpaint_v1)
from .depth_aa import DepthAA  # noqa: F401
