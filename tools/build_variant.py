"""A/B builds: ``python tools/build_variant.py <name> [-D FLAG ...] [--files a.hip b.hip ...]`` links
``nunif_amd/libnunif_hip_<name>.so`` = the product library with the named sources recompiled with the extra ``-D`` flags (all
sources when no file is named).  Pick it on the GPU box with ``NUNIF_HIP_LIB=$PWD/nunif_amd/libnunif_hip_<name>.so``
(nunif_amd/_hip.py).  ``--snapshot`` just copies the current product library under that name (the "base" of an A/B pair)."""
import argparse
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nunif_amd import build as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("-D", dest="defs", action="append", default=[])
    ap.add_argument("--files", nargs="*", default=[])
    ap.add_argument("--snapshot", action="store_true")
    a = ap.parse_args()
    out = os.path.join(B.HERE, f"libnunif_hip_{a.name}.so")
    B.build(verbose=False)
    if a.snapshot:
        shutil.copyfile(B.LIB, out)
        print("snapshot", out)
        return
    vdir = os.path.join("/tmp", f"nunif_variant_{a.name}")
    os.makedirs(vdir, exist_ok=True)
    objs = []
    for src in B.sources():
        base = os.path.basename(src)
        if a.files and base not in a.files:
            objs.append(os.path.join(B.OBJ, base + ".o"))
            continue
        obj = os.path.join(vdir, base + ".o")
        cmd = [B.hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(base, []) + [f"-D{d}" for d in a.defs] + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"hipcc failed for {src}:\n{r.stderr}")
        objs.append(obj)
    r = subprocess.run([B.hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(f"link failed:\n{r.stderr}")
    print("built", out)


if __name__ == "__main__":
    main()
