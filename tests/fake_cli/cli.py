"""A stand-in for the reference's ``waifu2x/cli.py`` in ``tests/test_launch.py`` (``NUNIF_AMD_LAUNCH_CLI_MODULE``): same argument
surface (``-i``, ``-o``, ``--gpu``, ``-r``) and the same LISTING calls as ``waifu2x/ui_utils.py:385-409`` — through the live
reference's ``ImageLoader.listdir`` — but instead of rendering it records which files this process was given."""
import argparse
import json
import os

from oracle import refstub

refstub.install()           # the build container has no torchvision / PyAV: inert stubs, as in tests/test_install.py

from nunif.utils.image_loader import ImageLoader
from nunif.utils.ui import is_image, is_text, list_subdir


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--input", "-i", required=True)
    ap.add_argument("--output", "-o", required=True)
    ap.add_argument("--gpu", "-g", type=int, nargs="+", default=[0])
    ap.add_argument("--recursive", "-r", action="store_true")
    ap.add_argument("--method", "-m", default="scale")
    args = ap.parse_args()
    files = []
    if os.path.isdir(args.input):
        dirs = list_subdir(args.input, include_root=True, excludes=args.output) if args.recursive else [args.input]
        for d in dirs:
            files += ImageLoader.listdir(d)
    elif is_text(args.input):
        with open(args.input, encoding="utf-8") as f:
            files = [ln.strip() for ln in f if ln.strip() and not ln.startswith("#")]
    elif is_image(args.input):
        files = [args.input]
    import nunif.utils.render as R
    os.makedirs(args.output, exist_ok=True)
    rec = {"rank": int(os.environ.get("LOCAL_RANK", "0")), "world": int(os.environ.get("WORLD_SIZE", "1")), "gpu": args.gpu,
           "files": files, "method": args.method, "tiled_render_module": R.tiled_render.__module__}
    with open(os.path.join(args.output, f"rank{rec['rank']}.json"), "w") as f:
        json.dump(rec, f)
