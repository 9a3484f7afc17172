"""``waifu2x.upconv_7`` (reference ``waifu2x/models/upconv_7.py``): implemented next to VGG7 in ``vgg_7.py``."""
from .vgg_7 import UpConv7  # noqa: F401
