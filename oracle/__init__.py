"""CPU oracle for the nunif hot path (waifu2x tiled SR; iw3 depth post-processing + stereo warps).

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker / the reported CPU baseline — never as the thing measured or shipped.  The product
package (``nunif_amd``) never imports ``oracle`` and fails loudly when ``libnunif_hip.so`` is missing.

Every function is a torch-CPU fp32 (or pure-integer) restatement of a reference function and cites the
reference file:line it follows.  Pinning: the reference ships **no** golden vectors for this path
 except ``nunif/modules/replication_pad2d.py:109-126``; therefore the oracle is pinned
against *outputs of the reference itself run in the build container* — ``tests/golden/make_golden.py``
imports ``/root/reference`` through ``oracle/refstub.py`` and writes the fixtures under ``tests/golden/``
that the ``-m "not gpu"`` tests replay (on the GPU box ``/root/reference`` does not exist).

Not pinned (stated also in DESIGN.md): torchvision's ``SwinTransformerBlock`` itself — see
``oracle/tv_swin_block.py``.
"""
