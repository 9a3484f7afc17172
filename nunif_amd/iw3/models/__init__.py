from . import mlbw, row_flow_v3  # noqa: F401  (registers sbs.row_flow_v3, sbs.mlbw and the mlbw_l* factories)
