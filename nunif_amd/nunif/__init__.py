"""Mirror of the reference's ``nunif`` package, hot-path subset (see nunif_amd/__init__.py)."""
