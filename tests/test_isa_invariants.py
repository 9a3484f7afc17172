"""Static checks on the gfx950 ISA of the hot kernels (hipcc cross-compiles without a GPU): the properties below were each lost at
least once while a kernel was being changed, silently, and each costs tens of per cent — a register count that halves the resident
workgroups, a spill, a `flat_load` whose `s_waitcnt vmcnt(0)` drains a prefetch, a hand-counted wait the compiler no longer emits
(profiles/r04_isa_notes.md).  The numbers are occupancy limits of the launch shape, not tuning targets."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nunif_amd import build  # noqa: E402

FILES = ["swin_qkv_attn_r.hip", "swin_block_tail.hip", "swin_block_tail_ws.hip", "conv3_dma.hip", "depth_mlp.hip", "iw3_warp.hip",
         "cunet_head.hip"]


def _asm(fname):
    src = os.path.join(build.CSRC, fname)
    out = os.path.join("/tmp", f"nunif_isa_{fname}.s")
    newest = max(os.path.getmtime(p) for p in [src] + build.headers())
    if not (os.path.exists(out) and os.path.getmtime(out) >= newest):
        flags = [x for x in build.FLAGS if x != "-fPIC"] + build.EXTRA_FLAGS.get(fname, [])
        subprocess.run([build.hipcc()] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"), "-o", out, src],
                       check=True, capture_output=True)
    return open(out).read()


@pytest.fixture(scope="module")
def kernels():
    """{mangled kernel name: (body, sgprs, vgprs, spills)} over FILES"""
    try:
        build.hipcc()
    except RuntimeError:
        pytest.skip("hipcc not available")
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        texts = list(ex.map(_asm, FILES))
    found = {}
    for text in texts:
        meta = {m.group(1): (int(m.group(2)), int(m.group(3)), int(m.group(4))) for m in re.finditer(
            r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)}
        for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
            if m.group(1) in meta:
                found[m.group(1)] = (m.group(2),) + meta[m.group(1)]
    return found


def _pick(kernels, *needles):
    hits = [(k, v) for k, v in kernels.items() if all(n in k for n in needles)]
    assert hits, (needles, sorted(kernels)[:5])
    return hits


def _is_level1_attention_wm(name):
    return "qkv_attn_r_kernel" in name and "ILi96ELi16ELi6ELi16ELb1E" in name


def test_no_spills_no_scratch_in_the_dispatched_kernels(kernels):
    for name, (body, sg, vg, sp) in kernels.items():
        assert vg <= 256, (name, vg)
        if _is_level1_attention_wm(name):
            continue                                        # its own test below
        if "qkv_attn_r_kernel" in name and "ILi96ELi16E" in name:
            # the pixel-major instance (launches beyond 2^31 att bytes, the debug taps): a handful of per-lane constants in scratch
            assert sp <= 16, (name, sp)
            continue
        assert sp == 0 and "scratch_" not in body, (name, sp)


def test_level1_attention_keeps_scratch_out_of_its_head_loop(kernels):
    """Round 6: with the next window's x requested inside the last head (DIET & 32) the window-major level-1 attention runs at exactly
    128 registers and hipcc parks some per-lane constants in scratch — in the launch prologue and on the path of the windows of the last
    row / column (5 % of the windows).  That measured FASTER than the spill-free form (338.5 -> 333.4 us, profiles/r06h_attn_early_x_ab.txt);
    what must not happen is a scratch access in the common path of the window loop: the six unrolled heads (the code between the
    `sched_barrier` markers that separate them, and the first head in front of the first marker up to the loop header) are scratch-free,
    and so is the stretch that issues the early x loads on the common path."""
    (name, (body, sg, vg, sp)), = [kv for kv in kernels.items() if _is_level1_attention_wm(kv[0])]
    lines = body.splitlines()
    marks = [i for i, ln in enumerate(lines) if "sched_barrier" in ln]
    assert len(marks) == 6, len(marks)                       # six unrolled heads, one marker behind each
    loop_head = max(i for i, ln in enumerate(lines) if "Loop Header" in ln or "Inner Loop" in ln)       # the window loop is the last loop
    # (hipcc rotates the loop: the stretch behind the LAST marker is the last head's softmax with the early x request in it)
    hot = lines[loop_head:marks[-1]]
    assert sum(1 for i in marks if i > loop_head) >= 5
    assert not any("scratch_" in ln for ln in hot), [ln for ln in hot if "scratch_" in ln][:3]
    # behind the last marker: reloads on the rare branch only, never a spill STORE inside the window loop
    tail = lines[marks[-1]:]
    assert not any("scratch_store" in ln for ln in lines[loop_head:]), "a spill store inside the window loop"
    assert sum("scratch_load" in ln for ln in tail) <= 12 and sp <= 16, (sp, sum("scratch_load" in ln for ln in tail))


def test_forward_warp_keeps_two_rows_per_cu(kernels):
    # 1024 threads x 2 workgroups per CU = 8 waves per SIMD: <= 64 VGPRs, and <= 80 SGPRs (beyond that the hardware keeps fewer waves
    # resident whatever the compiler's "Occupancy: 8" says: 129 vs 87 us at 102 SGPRs in round 4, profiles/r04_fw_trace.txt; 103-109 vs
    # 70 us at 86 in round 5, profiles/r05ag_fw_pairs.txt) — both instantiations (the diet form and round 4's, kept for A/B runs)
    hits = _pick(kernels, "forward_warp_kernel")
    assert len(hits) == 3                   # <DIET, 1>, <DIET, 2> (rows beyond 2 048 pixels), round 4's form
    for name, (body, sg, vg, sp) in hits:
        assert vg <= 64 and sg <= 80, (name, vg, sg)


def test_cunet_head_keeps_four_workgroups_per_cu(kernels):
    # 16 x 16 tiles: 37 KiB of LDS = four workgroups of four waves per CU = four waves per SIMD: <= 128 VGPRs
    for name, (body, sg, vg, sp) in _pick(kernels, "cunet_head_kernel", "ILi16E"):
        assert vg <= 128 and body.count("v_mfma") == 24, (name, vg)           # 6 pixel groups x 2 n-tiles x 2 k-steps


def test_level1_swin_kernels_hold_their_occupancy_and_address_spaces(kernels):
    for name, (body, sg, vg, sp) in _pick(kernels, "qkv_attn_r_kernel", "ILi96ELi16E"):
        assert vg <= 128, (name, vg)                          # 16 waves per workgroup: four per SIMD
    for name, (body, sg, vg, sp) in _pick(kernels, "proj_mlp_r_kernel"):
        assert "flat_load" not in body and "flat_store" not in body, name      # a laundered LDS pointer drains the weight prefetch
    for name, (body, sg, vg, sp) in _pick(kernels, "proj_mlp_ws_kernel"):
        assert "flat_load" not in body, name


def test_hand_counted_waits_are_what_was_written(kernels):
    # depth MLP: the weight ring is inline asm; every fc1 k-step waits vmcnt(22), every fc2 k-step vmcnt(21) (tail: 18 .. 0)
    (name, (body, sg, vg, sp)), = _pick(kernels, "da_mlp_split_kernel")
    assert body.count("s_waitcnt vmcnt(22)") == 36 and body.count("s_waitcnt vmcnt(21)") >= 16, name
    assert body.count("v_mfma_f32_16x16x32") == 576
    # conv3_dma: 2 DMA instructions per wave and chunk, two chunks ahead -> vmcnt(2) at every chunk boundary but the first
    for name, (body, sg, vg, sp) in _pick(kernels, "conv3_dma_kernel", "ILi4ELi64ELb0ELi1ELi2ELi1E"):
        assert body.count("s_waitcnt vmcnt(2)") >= 8 and "global_load_lds_dwordx4" in body, name
