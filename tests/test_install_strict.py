"""``nunif_amd.install(strict=True)`` WITHOUT the reference on ``sys.path``: the documented ImportError (not an
UnboundLocalError out of the roll-back handler), nothing left bound, and ``strict=False`` reports instead of raising.
Runs in a child interpreter whose path holds the repo only, so it does not depend on /root/reference being mounted."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path[:] = [p for p in sys.path if "reference" not in p]
for m in ("nunif", "waifu2x", "iw3"):
    assert m not in sys.modules
import nunif_amd.install as inst
try:
    inst.install()
except ImportError as e:
    assert "is the nunif checkout on sys.path?" in str(e), str(e)
else:
    raise SystemExit("install() did not raise")
assert not inst.is_installed()
rep = inst.install(strict=False, registry=False)
assert rep["patched"] == {} and rep["skipped"], rep
inst.uninstall()
print("ok")
"""


def test_strict_install_without_the_reference_raises_the_documented_import_error():
    env = dict(os.environ, PYTHONPATH=REPO)
    r = subprocess.run([sys.executable, "-c", CHILD], cwd=REPO, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
