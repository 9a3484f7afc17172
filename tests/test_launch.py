"""``python -m nunif_amd.launch``: the reference's ``--gpu 0 1 ...`` list becomes one process per GPU, each
running the reference's OWN ``cli.main()`` on the installed engine over its share of the input files.  CPU only: the ranks are
real processes started through ``torch.distributed.run`` on 127.0.0.1, the CLI is a stand-in that makes the reference's listing
calls and records what it was given (``tests/fake_cli/cli.py``) — nothing is rendered.

Reference: ``waifu2x/ui_utils.py:231,385-409`` / ``iw3/utils.py:1957,2254-2437`` (``--gpu``, input kinds),
``nunif/utils/image_loader.py:40-51`` (listing), ``nunif/utils/video.py:1622-1757`` (the in-process multi-device pool this replaces).
"""
import json
import os
import subprocess
import sys

import pytest

from oracle import refstub
from nunif_amd import launch as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not refstub.reference_available(), reason="/root/reference is not mounted here")


def test_gpu_list_parsing_and_input_kinds(tmp_path):
    assert L.split_gpu_args(["-i", "a", "--gpu", "0", "1", "3", "-o", "b"]) == ([0, 1, 3], ["-i", "a", "-o", "b"])
    assert L.split_gpu_args(["-g", "2", "-m", "scale"]) == ([2], ["-m", "scale"])
    assert L.split_gpu_args(["--gpu", "1", "1", "0", "--tta"]) == ([1, 0], ["--tta"])
    assert L.split_gpu_args(["-i", "x"]) == ([], ["-i", "x"])
    (tmp_path / "d").mkdir()
    (tmp_path / "l.txt").write_text("a.png\n")
    assert [L.classify_input(str(p)) for p in (tmp_path / "d", tmp_path / "l.txt", "x.png", "y.mp4", "z.yml")] == \
        ["dir", "list", "image", "video", "config"]
    assert L.shard(range(7), 1, 3) == [1, 4]
    # what one process per GPU cannot shard is refused with the reason, not silently run on one GPU: an export config, and a
    # waifu2x video (iw3's single video IS sharded, by frame: test_one_video_is_frame_sharded_over_two_ranks)
    assert L.main(["iw3", "-i", "export.yml", "-o", "out", "--gpu", "0", "1"]) == 2
    assert L.main(["waifu2x", "-i", "movie.mp4", "-o", "out", "--gpu", "0", "1"]) == 2


def _run(args, tmp_path, cli="fake_cli.cli"):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests"), refstub.REFERENCE_ROOT,
                                                       os.environ.get("PYTHONPATH", "")]),
               NUNIF_AMD_LAUNCH_TEST_HOOKS="1", NUNIF_AMD_LAUNCH_CLI_MODULE=cli)
    r = subprocess.run([sys.executable, "-m", "nunif_amd.launch"] + args, env=env, cwd=str(tmp_path), capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return r


def _make_images(d, names):
    d.mkdir(parents=True, exist_ok=True)
    for n in names:
        (d / n).write_bytes(b"")
    return sorted(str(d / n) for n in names if n.endswith(".png"))


def test_two_ranks_split_a_directory_and_a_list_between_them(tmp_path):
    imgs = _make_images(tmp_path / "in", [f"f{i:02d}.png" for i in range(7)] + ["notes.md"])
    sub = _make_images(tmp_path / "in" / "sub", ["a.png", "b.png", "c.png"])
    out = tmp_path / "out"
    _run(["waifu2x", "-m", "noise", "-i", str(tmp_path / "in"), "-o", str(out), "--gpu", "2", "5", "-r"], tmp_path)
    recs = [json.load(open(out / f"rank{r}.json")) for r in range(2)]
    assert [r["gpu"] for r in recs] == [[2], [5]] and all(r["world"] == 2 and r["method"] == "noise" for r in recs)
    # every listing is cut r, r + 2, ...: together the ranks cover each directory exactly once, in the CLI's own order
    assert recs[0]["files"] == imgs[0::2] + sub[0::2] and recs[1]["files"] == imgs[1::2] + sub[1::2]
    # the ranks run on the installed engine: the reference's name now resolves to ours
    assert all(r["tiled_render_module"].startswith("nunif_amd.") for r in recs)
    lst = tmp_path / "files.txt"
    lst.write_text("# picked by hand\n" + "\n".join(imgs[:5]) + "\n")
    out2 = tmp_path / "out2"
    _run(["waifu2x", "-i", str(lst), "-o", str(out2), "--gpu", "0", "1"], tmp_path)
    recs = [json.load(open(out2 / f"rank{r}.json")) for r in range(2)]
    assert recs[0]["files"] == imgs[0:5:2] and recs[1]["files"] == imgs[1:5:2]


def test_one_gpu_runs_in_process_on_the_whole_input(tmp_path):
    imgs = _make_images(tmp_path / "in", ["a.png", "b.png", "c.png"])
    out = tmp_path / "out"
    _run(["waifu2x", "-i", str(tmp_path / "in"), "-o", str(out), "--gpu", "3"], tmp_path)
    rec = json.load(open(out / "rank0.json"))
    assert rec["files"] == imgs and rec["gpu"] == [3] and rec["world"] == 1 and rec["tiled_render_module"].startswith("nunif_amd.")


@pytest.mark.parametrize("case", ["ema", "cut_last_of_batch"])
def test_one_video_is_frame_sharded_over_two_ranks(tmp_path, case):
    """BASELINE configs[3] through the CLI line: ``iw3 -i movie.mp4 --gpu 0 1`` = two ranks, every rank runs the decode loop, batch
    b is processed on rank b mod 2 (EMA look-ahead replayed across the ranks), rank 0 alone hands frames to its encoder — and
    they are the frames of the reference's own single-process scheduler (``tests/golden/frame_pool.npz``), bit for bit."""
    import numpy as np
    import torch
    out = tmp_path / "out"
    _run(["iw3", "-i", "movie.mp4", "-o", str(out), "--gpu", "0", "1", "--case", case], tmp_path, cli="fake_cli.iw3_cli")
    rec = json.load(open(out / "rank0.json"))
    golden = np.load(os.path.join(ROOT, "tests", "golden", "frame_pool.npz"))[case + "_frames"]
    assert rec["world"] == 2 and rec["pool"].endswith("ShardedFrameCallbackPool") and rec["av_frames"]
    assert rec["frames_encoded"] == golden.shape[0] and sum(rec["calls"]) == golden.shape[0]
    assert torch.equal(torch.load(out / "frames.pt"), torch.from_numpy(golden))
    assert not (out / "rank1.json").exists()             # rank 1 encoded nothing and its scratch output is gone


def _trace(tmp_path, rank):
    return json.load(open(tmp_path / f"trace{rank}.json"))


def test_ranks_take_the_same_exit_through_process_video(tmp_path, monkeypatch):
    """with one video on N ranks the early exits of ``process_video_full`` (iw3/utils.py:1000-1011) fired on rank 0
    only — ranks > 0 write to a scratch ``--output`` — and rank 0 then waited in a barrier against the others' collectives.  The
    decision is now rank 0's, broadcast: (a) a fresh output runs sharded, through the reference's own ``process_video``; (b) the
    same line again without ``--yes`` and (c) with ``--resume`` leaves on BOTH ranks (no prompt, no hang: the 300 s timeout of
    ``_run`` is the assertion); (d) ``--low-vram`` = the per-frame route runs on rank 0 alone."""
    import numpy as np
    import torch
    monkeypatch.setenv("NUNIF_AMD_FAKE_CLI_TRACE", str(tmp_path))
    (tmp_path / "out").mkdir()
    out = tmp_path / "out" / "movie_sbs.mp4"            # a FILE name, as the early exits test it; the stand-in's frames go beside it
    base = ["iw3", "-i", "movie.mp4", "-o", str(out), "--gpu", "0", "1", "--case", "ema", "--via-process-video"]
    _run(base + ["--yes"], tmp_path, cli="fake_cli.iw3_cli")
    golden = np.load(os.path.join(ROOT, "tests", "golden", "frame_pool.npz"))["ema_frames"]
    assert torch.equal(torch.load(str(out) + ".frames/frames.pt"), torch.from_numpy(golden)) and out.is_file()
    assert _trace(tmp_path, 0)["ran"] == [str(out)] and len(_trace(tmp_path, 1)["ran"]) == 1
    for extra in ([], ["--resume"]):
        r = _run(base + extra, tmp_path, cli="fake_cli.iw3_cli")
        assert _trace(tmp_path, 0)["ran"] == [] and _trace(tmp_path, 1)["ran"] == []
        if not extra:
            assert "pass --yes" in r.stderr
    r = _run(base + ["--yes", "--low-vram"], tmp_path, cli="fake_cli.iw3_cli")
    assert _trace(tmp_path, 0)["ran"] == [str(out)] and _trace(tmp_path, 1)["ran"] == []
    assert "first GPU only" in r.stderr


def test_export_of_one_video_with_several_gpus_is_refused():
    assert L.main(["iw3", "-i", "movie.mp4", "-o", "out", "--gpu", "0", "1", "--export"]) == 2
