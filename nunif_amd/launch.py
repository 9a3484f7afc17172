"""``python -m nunif_amd.launch {waifu2x|iw3} <the reference CLI's own arguments>`` — the reference command line, unchanged, on
the HIP engine, and its ``--gpu 0 1 2 3`` list turned into ONE PROCESS PER GPU.

The reference handles ``--gpu 0 1 ...`` inside one process: ``nn.DataParallel`` over the tile minibatch
(``nunif/models/register.py:44-61``, ``nunif/models/data_parallel.py:8-68``) and one worker thread per device in
``FrameCallbackPool`` (``nunif/utils/video.py:1622-1757``, ``iw3/utils.py:709-831``).  On MI355X the unit is one process per GPU
(DESIGN.md §7), so this launcher

* with one GPU (or none named): calls ``nunif_amd.install()`` and the reference's ``<tool>.cli.main()`` in this process;
* with N GPUs: re-launches itself as N ranks through ``torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
  127.0.0.1`` (the same launcher ``bench.py --gpus N`` uses); rank r binds ``--gpu <r-th id of the list>``, installs the engine
  and runs the SAME reference ``main()`` over its share of the input FILES: every directory listing the CLI makes under ``-i``
  (``ImageLoader.listdir`` ``nunif/utils/image_loader.py:40-51``, ``VU.list_videos`` ``nunif/utils/video.py:44-49``) returns
  files ``r, r + N, r + 2N, ...`` of the sorted listing, a text-list input (``waifu2x/ui_utils.py:406-409``,
  ``iw3/utils.py:2425-2437``) is cut the same way.  Output names and directory layout are the CLI's own, so the N ranks together
  write exactly what one process would have; there is no collective on this path (frames / files are independent units —
  north_star) and no rank waits for another.

ONE video on N GPUs (``iw3 -i movie.mp4 --gpu 0 1 ...``, BASELINE configs[3]) is FRAME-sharded: every rank runs the reference's own
decode loop (``VU.process_video``, ``nunif/utils/video.py:956-1170``) over the whole file, and the two names that loop's batch
route resolves — ``iw3.utils.bind_batch_frame_callback`` and ``VU.FrameCallbackPool`` (``iw3/utils.py:1137-1153``) — are bound to
``nunif_amd.iw3.frame_pipeline``'s: batch ``b`` is uploaded and processed by rank ``b mod N`` only, the EMA depth normalisation is
replayed across the ranks from an all-gather of two scalars per frame (``ShardedStereoStream``), finished frames reach rank 0 by a
gather and leave through ITS encoder; the other ranks' ``process_video`` sees an empty stream and writes a scratch file that is
removed.  Decoding N times is the price of not touching the reference's loop (a 1080p software decode runs at several hundred
frames per second per process; the GPUs' share of a frame is ~0.5 ms).

The ranks decide TOGETHER what happens to the file (``install_video_guards``): rank 0 evaluates the reference's early exits
(``--resume`` / ``--skip-error`` / an existing output without ``--yes``, ``iw3/utils.py:1000-1011``) against the real ``--output``
and broadcasts the verdict; scene detection and its cache run on rank 0 and ``segment_pts`` is broadcast.  Option sets that take
the reference's PER-FRAME routes (``--low-vram``, ``--debug-depth``, ``--keyframe``, VideoDepthAnything, the inpaint side
models: temporal state, sharded by scene segment — not built) run on the first GPU of the list only, the other ranks leave.

What it refuses, with the reason: a ``.yml`` export config or ``--export`` of one video with several GPUs, and a waifu2x video
with several GPUs (its frames are independent, but its CLI route does not pass through the two names above).  A single image
runs on the first GPU of the list.

The reference checkout must be importable (``PYTHONPATH=/path/to/nunif``): this is a launcher FOR it, it carries no CLI of its own.
"""
import mimetypes
import os
import shutil
import socket
import subprocess
import sys

TOOLS = ("waifu2x", "iw3")
_RANK_FLAG = "--nunif-amd-rank-entry"


def split_gpu_args(argv):
    """-> (gpu id list, argv without the ``--gpu`` / ``-g`` option).  The reference declares ``--gpu/-g type=int nargs="+"``
    (``waifu2x/ui_utils.py:231``, ``iw3/utils.py:1957``)."""
    gpus, rest, i = [], [], 0
    while i < len(argv):
        a = argv[i]
        if a in ("--gpu", "-g"):
            i += 1
            while i < len(argv) and argv[i].lstrip("-").isdigit():
                gpus.append(int(argv[i]))
                i += 1
            continue
        if a.startswith("--gpu="):
            gpus += [int(v) for v in a.split("=", 1)[1].split()]
            i += 1
            continue
        rest.append(a)
        i += 1
    seen, uniq = set(), []
    for g in gpus:
        if g not in seen:
            seen.add(g)
            uniq.append(g)
    return uniq, rest


def option_value(argv, *names):
    for i, a in enumerate(argv):
        if a in names and i + 1 < len(argv):
            return argv[i + 1]
        for n in names:
            if n.startswith("--") and a.startswith(n + "="):
                return a.split("=", 1)[1]
    return None


def _mime(path, kind):
    m = mimetypes.guess_type(path)[0]               # the reference's own test: nunif/utils/ui.py:46-58
    return bool(m and m.startswith(kind))


def classify_input(path):
    if path is None:
        return "none"
    if os.path.isdir(path):
        return "dir"
    if _mime(path, "text"):
        return "list"
    if _mime(path, "image"):
        return "image"
    if _mime(path, "video"):
        return "video"
    if path.endswith((".yml", ".yaml")):
        return "config"
    return "other"


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def shard(seq, rank, world):
    return list(seq)[rank::world]


def install_listing_shards(rank, world, input_root):
    """Every directory listing under ``input_root`` that the reference CLI makes returns this rank's share.  Listings elsewhere
    (model directories, an ``--import`` rgb / depth pair outside the input) are left alone."""
    import importlib
    root = os.path.realpath(input_root)

    def under_root(directory):
        d = os.path.realpath(directory)
        return d == root or d.startswith(root + os.sep)

    il = importlib.import_module("nunif.utils.image_loader")
    orig_list_images = il.list_images

    def list_images(directory, *a, **kw):
        files = orig_list_images(directory, *a, **kw)
        return shard(files, rank, world) if under_root(directory) else files

    il.list_images = list_images                     # ImageLoader.listdir resolves the module global at call time
    try:
        vu = importlib.import_module("nunif.utils.video")
    except Exception:                                # PyAV missing: the CLI cannot list videos either
        vu = None
    if vu is not None and hasattr(vu, "list_videos"):
        orig_list_videos = vu.list_videos

        def list_videos(directory, *a, **kw):
            files = orig_list_videos(directory, *a, **kw)
            return shard(files, rank, world) if under_root(directory) else files

        vu.list_videos = list_videos


def shard_text_list(path, rank, world, out):
    """The reference reads one path per line, ``#`` starts a comment (``iw3/utils.py:2431-2436``, ``waifu2x/ui_utils.py:208``).
    ``out``: a file of this run's own (``tempfile.mkstemp``: two launcher runs with the same world size must not share it)."""
    with open(path, mode="r", encoding="utf-8") as f:
        files = [ln.strip() for ln in f.readlines()]
    files = [ln for ln in files if ln and not ln.startswith("#")]
    with open(out, mode="w", encoding="utf-8") as f:
        f.write("\n".join(shard(files, rank, world)) + "\n")
    return out


def replace_option(argv, names, value):
    out, i, done = [], 0, False
    while i < len(argv):
        a = argv[i]
        if a in names and i + 1 < len(argv):
            out += [a, value]
            i += 2
            done = True
            continue
        out.append(a)
        i += 1
    if not done:
        out += [names[0], value]
    return out


def tool_main(tool):
    import importlib
    mod = f"{tool}.cli"
    if os.environ.get("NUNIF_AMD_LAUNCH_TEST_HOOKS") == "1":                   # the launcher's own tests run a stand-in CLI
        mod = os.environ.get("NUNIF_AMD_LAUNCH_CLI_MODULE") or mod
    return importlib.import_module(mod).main


def _to_av_frame(use_16bit):
    """Finished HWC tensor -> what the reference's encode loop takes (``VU.to_frame``, nunif/utils/video.py:236-245)."""
    def convert(frame):
        import av
        arr = frame.cpu().numpy()
        if use_16bit:
            return av.VideoFrame.from_ndarray(arr.view("uint16") if arr.dtype != "uint16" else arr, format="rgb48le")
        return av.VideoFrame.from_ndarray(arr, format="rgb24")
    return convert


def install_frame_sharding(rank, world):
    """One video, ``world`` ranks: bind the batch route of ``iw3.utils.process_video_full`` (iw3/utils.py:1137-1153) to the
    frame-sharded scheduler.  Needs an initialised process group."""
    import importlib
    from nunif_amd.iw3 import frame_pipeline as FP
    iu = importlib.import_module("iw3.utils")
    vu = importlib.import_module("nunif.utils.video")

    def sharded_pool(frame_callback, batch_size, device, use_16bit=False, **kw):
        kw.pop("max_workers", None), kw.pop("max_batch_queue", None)
        return FP.ShardedFrameCallbackPool(frame_callback, batch_size, device, use_16bit=use_16bit,
                                           to_output=_to_av_frame(use_16bit), **kw)

    iu.bind_batch_frame_callback = FP.bind_batch_frame_callback
    vu.FrameCallbackPool = sharded_pool


_SINGLE_FRAME_DEPTH = {"VideoDepthAnything", "VideoDepthAnythingStreaming"}
_INPAINT_METHODS = {"forward_inpaint", "mlbw_l2_inpaint"}


def video_decision(iu, vu, input_filename, output_path, args, depth_model):
    """What ``iw3.utils.process_video_full`` (iw3/utils.py:974-1170) will do with this file, decided ONCE (on rank 0, with the real
    ``--output``) so that N ranks act alike: ``"skip"`` = one of its early exits fires (``--resume`` with the output present,
    ``--skip-error`` with an error file, an existing output without ``--yes`` — the reference would prompt on stdin, which N ranks
    under torchrun cannot answer); ``"rank0"`` = the reference takes a per-frame route that does not pass through the two names the
    launcher rebinds (``bind_single_frame_callback`` / ``bind_vda_frame_callback``, iw3/utils.py:1098-1135: ``--low-vram``,
    ``--debug-depth``, a depth or side model with temporal state; also ``--keyframe``), so one rank runs it and the others leave;
    ``"go"`` = the frame-sharded batch route."""
    import os.path as path
    output_parent_dir = path.basename(output_path)
    input_parent_dir = path.basename(path.dirname(input_filename))
    if iu.is_output_dir(output_path) or (output_parent_dir != "" and output_parent_dir == input_parent_dir):
        output_filename = path.join(output_path, iu.make_output_filename(path.basename(input_filename), args, video=True))
    else:
        output_filename = output_path
    if (getattr(args, "resume", False) and path.exists(output_filename)) or \
            (getattr(args, "skip_error", False) and path.exists(vu.make_error_file_path(output_filename))):
        return "skip"
    if not getattr(args, "yes", False) and path.exists(output_filename):
        sys.stderr.write(f"nunif_amd.launch: '{output_filename}' already exists; pass --yes to overwrite it (no prompt under "
                         "one process per GPU).  Skipped.\n")
        return "skip"
    name = depth_model.get_name() if hasattr(depth_model, "get_name") else ""
    if (getattr(args, "keyframe", False) or getattr(args, "low_vram", False) or getattr(args, "debug_depth", False)
            or name in _SINGLE_FRAME_DEPTH or getattr(args, "method", None) in _INPAINT_METHODS
            or getattr(depth_model, "has_temporal_state", False)):
        sys.stderr.write("nunif_amd.launch: this option set takes the reference's per-frame route (temporal state or --low-vram / "
                         "--debug-depth / --keyframe), which shards by scene segment — not built.  Running on the first GPU only.\n")
        return "rank0"
    return "go"


def install_video_guards(rank, world, real_output, iu=None, vu=None, bcast=None):
    """One video on N ranks: every rank must take the SAME way through ``iw3.utils.process_video`` (the early exits of
    ``process_video_full`` fired on rank 0 only — ranks > 0 have a scratch ``--output`` — and rank 0 then sat in a barrier against
    the others' all-gather).  Rank 0 decides (``video_decision``), the decision is broadcast, and scene detection + its cache
    (iw3/utils.py:1015-1036) run on rank 0 only with ``segment_pts`` broadcast.  Returns the state dict (``mode`` = last decision)."""
    import importlib
    iu = iu or importlib.import_module("iw3.utils")
    vu = vu or importlib.import_module("nunif.utils.video")
    if bcast is None:
        import torch.distributed as dist

        def bcast(obj):
            box = [obj]
            dist.broadcast_object_list(box, src=0)
            return box[0]
    state = {"mode": None}
    orig_process_video = iu.process_video

    def process_video(input_filename, output_path, args, depth_model, side_model):
        mode = video_decision(iu, vu, input_filename, real_output, args, depth_model) if rank == 0 else None
        mode = state["mode"] = bcast(mode)
        if mode == "skip" or (mode == "rank0" and rank > 0):
            return None
        return orig_process_video(input_filename, output_path, args, depth_model, side_model)

    iu.process_video = process_video

    def on_rank0(fn):
        def wrapper(*a, **kw):
            if state["mode"] != "go":                     # rank 0 alone is in here: nobody to talk to
                return fn(*a, **kw)
            return bcast(fn(*a, **kw) if rank == 0 else None)
        return wrapper

    if hasattr(iu, "try_load_scene_cache"):
        iu.try_load_scene_cache = on_rank0(iu.try_load_scene_cache)
    if hasattr(iu, "SBD") and hasattr(iu.SBD, "detect_boundary"):
        iu.SBD.detect_boundary = on_rank0(iu.SBD.detect_boundary)
    if hasattr(iu, "save_scene_cache"):
        orig_save = iu.save_scene_cache

        def save_scene_cache(*a, **kw):
            return orig_save(*a, **kw) if rank == 0 else None

        iu.save_scene_cache = save_scene_cache
    return state


def run_in_process(tool, argv, gpu, rank=0, world=1):
    """install() + the reference's ``<tool>.cli.main()`` with ``--gpu <gpu>`` and, for world > 1, this rank's share of the files."""
    import tempfile
    argv = list(argv)
    if gpu is not None:
        argv += ["--gpu", str(gpu)]
    src = option_value(argv, "--input", "-i")
    kind = classify_input(src)
    from nunif_amd import install as engine_install
    # load the CLI layer BEFORE install() so that its ``from x import y`` copies exist and get rebound (nunif_amd/install.py)
    main = tool_main(tool)
    if not engine_install.is_installed():
        engine_install.install(strict=False)
    scratch = []
    guard = None
    if world > 1:
        if kind == "dir":
            install_listing_shards(rank, world, src)
        elif kind == "list":
            fd, lst = tempfile.mkstemp(prefix=f"nunif_amd_shard_{rank}_of_{world}_", suffix=".txt")
            os.close(fd)
            scratch.append(lst)
            argv = replace_option(argv, ("--input", "-i"), shard_text_list(src, rank, world, lst))
        elif kind == "video" and tool == "iw3":
            import torch
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                backend = "nccl" if (gpu is not None and gpu >= 0 and torch.cuda.is_available()) else "gloo"
                if backend == "nccl":
                    torch.cuda.set_device(gpu)
                dist.init_process_group(backend)
            install_frame_sharding(rank, world)
            guard = install_video_guards(rank, world, option_value(argv, "--output", "-o"))
            if rank > 0:
                # this rank's encoder sees an empty stream: its file goes to a scratch directory
                tmp = tempfile.mkdtemp(prefix=f"nunif_amd_rank{rank}_")
                scratch.append(tmp)
                argv = replace_option(argv, ("--output", "-o"), tmp)
        elif rank > 0:
            return 0                                  # a single image: the first GPU of the list renders it
    old = sys.argv
    sys.argv = [f"{tool}.cli"] + argv
    try:
        main()
    finally:
        sys.argv = old
        for p in scratch:
            if os.path.isdir(p):
                shutil.rmtree(p, ignore_errors=True)
            elif os.path.exists(p):
                os.remove(p)
        if world > 1 and kind == "video":
            import torch.distributed as dist
            if dist.is_initialized():
                # "rank0": the other ranks left at once and rank 0 may run for hours — a barrier would only meet the watchdog
                if guard is None or guard["mode"] != "rank0":
                    dist.barrier()
                dist.destroy_process_group()
    return 0


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    as_rank = False
    if argv and argv[0] == _RANK_FLAG:
        as_rank, argv = True, argv[1:]
    if not argv or argv[0] not in TOOLS:
        sys.stderr.write(f"usage: python -m nunif_amd.launch {{{'|'.join(TOOLS)}}} <arguments of the reference's <tool>.cli>\n")
        return 2
    tool, argv = argv[0], argv[1:]
    gpus, rest = split_gpu_args(argv)
    if as_rank:
        rank, world = int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
        gpus = gpus or list(range(world))
        assert len(gpus) == world, (gpus, world)
        return run_in_process(tool, rest, gpus[rank], rank, world)
    if len(gpus) <= 1:
        return run_in_process(tool, rest, gpus[0] if gpus else None)
    kind = classify_input(option_value(rest, "--input", "-i"))
    exporting = any(a in ("--export", "--export-disparity") for a in rest)
    if kind == "video" and tool == "iw3" and exporting:
        sys.stderr.write("nunif_amd.launch: --export of ONE video with several GPUs: the export route (iw3/utils.py export_video) "
                         "does not pass through the frame scheduler the launcher shards.  Run it with one GPU.\n")
        return 2
    if kind in ("config", "other", "none") or (kind == "video" and tool != "iw3"):
        sys.stderr.write(
            f"nunif_amd.launch: --gpu {' '.join(map(str, gpus))} with a single {kind} input for {tool}.  One process per GPU "
            "shards FILES (a directory or a text list) and, for iw3, the FRAMES of one video (batch b on rank b mod N); this "
            "input is neither.  Run it with one GPU, or pass a directory / list of files.\n")
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={len(gpus)}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), "-m", "nunif_amd.launch", _RANK_FLAG, tool,
           "--gpu", *map(str, gpus), *rest]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


if __name__ == "__main__":
    sys.exit(main())
