"""``python bench.py --gpus 2`` END TO END on two CPU ranks: ``relaunch_as_ranks`` -> ``torch.distributed.run`` ->
rendezvous on 127.0.0.1 -> timed loop (barrier, MAX over ranks) -> delivery leg -> iw3 / cunet / config-5 legs -> ONE JSON line on
rank 0 -> process group down, exit code 0.  ``NUNIF_BENCH_BACKEND=gloo`` swaps every device function for a torch-CPU stand-in
(``bench.dry_standins``): nothing in the line is a measurement, the control flow is the driver's N > 1 run.  The other gloo tests
(tests/test_parallel_gloo.py) call the legs; this one goes through the launcher.

Reference for what the ranks replace: ``nunif/utils/video.py:1622-1757`` (FrameCallbackPool, device round-robin),
``iw3/utils.py:709-831``, ``nunif/models/data_parallel.py:41-68``."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, extra_env=None, timeout=420):
    env = dict(os.environ, NUNIF_BENCH_BACKEND="gloo", PYTHONPATH=os.pathsep.join([ROOT, os.environ.get("PYTHONPATH", "")]))
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_two_ranks_through_the_launcher_print_one_line():
    r, lines = _bench(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]                  # rank 0 alone speaks
    rec = json.loads(lines[0])
    assert rec["ok"] and rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["data"].startswith("dry") and rec["unit"] == "MPix/s" and rec["value"] > 0
    mg = rec["multi_gpu"]
    assert mg["ranks_seen"] == 2 and mg["world_size"] == 2 and mg["distinct_pci_bus_ids"] == 2 and mg["backend"] == "gloo"
    # the delivery leg: every frame of the sharded render reached rank 0, and its rate sits next to `value` in the top-level keys
    g = rec["gathered"]
    assert g["frames_delivered"] == g["frames"] > 0 and rec["gathered_value"] == g["value"]
    # BASELINE's other metrics at N > 1
    assert rec["iw3"]["world"] == 2 and rec["iw3"]["frames_delivered"] == rec["iw3"]["frames"]
    assert rec["cunet"]["world"] == 2 and rec["config5"]["world"] == 2 and rec["config5"]["scaling"].startswith("replicas")
    assert rec["config5"]["ms_per_frame_per_gpu"] == 21.0      # priced at the slowest rank (stand-in: 20 + rank ms)


def test_leg_switches_and_world_size_mismatch():
    r, lines = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-iw3", "--no-cunet", "--no-config5"])
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(lines[0])
    assert rec["ok"] and "iw3" not in rec and "cunet" not in rec and "config5" not in rec and rec["gathered"]["frames_delivered"] > 0
    # launched by hand with the wrong world size: refused before any collective
    r, lines = _bench(["--gpus", "2", "--steps", "1"], extra_env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode != 0 and not lines and "WORLD_SIZE=3" in (r.stderr + r.stdout)


def test_four_ranks_deliver_every_frame():
    """The driver's scaling run goes 1 / 2 / 4 / 8: the same line at four ranks (odd frame counts per rank, three peers sending to
    rank 0 per round)."""
    r, lines = _bench(["--gpus", "4", "--steps", "2", "--warmup", "1", "--no-cunet", "--no-config5"])
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads(lines[0])
    assert rec["ok"] and rec["n_gpus"] == 4 and rec["multi_gpu"]["ranks_seen"] == 4 and rec["multi_gpu"]["distinct_pci_bus_ids"] == 4
    assert rec["gathered"]["frames_delivered"] == rec["gathered"]["frames"] == 16
    assert rec["iw3"]["world"] == 4 and rec["iw3"]["frames_delivered"] == rec["iw3"]["frames"] == 32
