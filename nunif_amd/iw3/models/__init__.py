from . import row_flow_v3  # noqa: F401  (registers sbs.row_flow_v3)
