"""Build ``libnunif_hip.so`` (gfx950) in-tree with hipcc.  ``python -m nunif_amd.build [--force]``.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is git-ignored but
travels to the GPU box with the snapshot.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libnunif_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-x", "hip", "-Wall", "-Wno-unused-function",
         "-fno-gpu-rdc"]


# the stitcher replays the reference's fp32 recurrence bit-exactly: no FMA contraction in that file
# the swin kernels are VALU-issue bound: hipcc's SLP vectoriser packs adjacent fp32 ops into v_pk_*_f32, which issue at
# half rate and drag an s_nop behind each dependent use on gfx950 (tools/ubench_valu.hip) — keep them scalar
NO_SLP = ["-fno-slp-vectorize"]
EXTRA_FLAGS = {"stitch.hip": ["-ffp-contract=off"], "iw3_warp.hip": ["-ffp-contract=off"],
               "iw3_depth.hip": ["-ffp-contract=off"], "image_ops.hip": ["-ffp-contract=off"], "swin_block_tail.hip": NO_SLP, "swin_block_tail_ws.hip": NO_SLP, "swin_qkv_attn_r.hip": NO_SLP + ["-fno-honor-nans"], "swin_block96.hip": NO_SLP + ["-fno-honor-nans"],
               "swin_kernels.hip": NO_SLP, "swin_patchup.hip": NO_SLP, "swin_patchdown.hip": NO_SLP, "cunet_up.hip": NO_SLP, "cunet_head.hip": NO_SLP,
               "cunet_kernels.hip": NO_SLP, "rowflow.hip": NO_SLP, "depth_aa.hip": NO_SLP, "depth_anything.hip": NO_SLP + ["-fno-honor-nans"], "depth_temporal.hip": NO_SLP, "conv3_lds.hip": NO_SLP}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "nunif_hip.h"))
    return hs


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm >= 7.0)")
    return exe


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    newest = max(os.path.getmtime(p) for p in [src] + headers())
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [hipcc()] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        results = list(ex.map(lambda s: _compile(s, force), sources()))
    objs = [o for o, _ in results]
    if any(changed for _, changed in results) or not os.path.exists(LIB):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[nunif_amd.build] linked {LIB}")
    elif verbose:
        print(f"[nunif_amd.build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
