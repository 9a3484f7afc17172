"""The temporal window of the video inpaint path (the reference's ``FrameQueue``, ``iw3/inpaint_utils.py:98-188``) lives in
``side_model.py`` as :class:`~nunif_amd.iw3.side_model.StereoWindow`; this module keeps the reference's module path."""
from .side_model import StereoWindow  # noqa: F401
