"""nunif_amd — MI355X-native engine behind nunif's tiled-inference hot path.

Sub-packages mirror the reference's module tree for the hot path only:

* ``nunif_amd.nunif``   — model contract / registry / ``.pth`` format, ``tiled_render`` + ``SeamBlending``
* ``nunif_amd.waifu2x`` — swin_unet / cunet models on HIP kernels, ``Waifu2x`` context, hub API
* ``nunif_amd.iw3``     — depth post-processing and stereo warps

All compute goes through ``libnunif_hip.so`` (``nunif_amd/csrc``, C ABI in ``include/nunif_hip.h``).
"""
__version__ = "0.1.0"
