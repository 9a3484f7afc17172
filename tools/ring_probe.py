import sys, time, torch, numpy as np
sys.path.insert(0, ".")
from nunif_amd.iw3 import _ops
from nunif_amd.nunif.utils.render import tiled_render
from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x
from nunif_amd.synthetic import swin_unet_state_dict
torch.set_grad_enabled(False)
m = SwinUNet2x().eval(); m.load_state_dict(swin_unet_state_dict(102, 2)); m = m.to("cuda:0")
H, W = 1080, 1920
host = [np.random.randint(0, 256, (H, W, 3), dtype=np.uint8) for _ in range(4)]
dev = torch.device("cuda:0")
import ctypes, os
def host_alloc(shape, flags):
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    n = int(np.prod(shape)); p = ctypes.c_void_p()
    rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), ctypes.c_uint(flags)); assert rc == 0, rc
    arr = np.ctypeslib.as_array((ctypes.c_uint8 * n).from_address(p.value)).reshape(shape)
    return torch.from_numpy(arr)
FL = int(os.environ.get("FLAGS", "-1"), 0)
if FL >= 0:
    h_in, h_out = host_alloc((H, W, 3), FL), host_alloc((2 * H, 2 * W, 3), FL)
    torch.Tensor.is_pinned = lambda self, *a, **k: True      # probe only: let the wrappers accept the raw host buffers
else:
    h_in = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()
    h_out = torch.empty((2 * H, 2 * W, 3), dtype=torch.uint8).pin_memory()
st = torch.cuda.Stream(dev)
ev = torch.cuda.Event()
import os
V = os.environ.get("V", "host")
x8 = torch.from_numpy(host[0]).to(dev)
d_out = torch.empty((2 * H, 2 * W, 3), dtype=torch.uint8, device=dev)
names = ["memcpy_in", "to_tensor", "render", "to_frame", "record", "sync", "copy_out"]
rows = []
pre = np.empty((2 * H, 2 * W, 3), np.uint8)
for i in range(14):
    t = [time.perf_counter()]
    h_in.copy_(torch.from_numpy(host[i % 4])); t.append(time.perf_counter())
    with torch.cuda.stream(st):
        x = _ops.frame_to_tensor(h_in, device=dev) if V in ("host", "in") else _ops.frame_to_tensor(x8); t.append(time.perf_counter())
        y = tiled_render(x, m, tile_size=256, batch_size=45); t.append(time.perf_counter())
        _ops.to_frame(y, 8, out=h_out if V in ("host", "out") else d_out); t.append(time.perf_counter())
        ev.record(st); t.append(time.perf_counter())
    (st.synchronize() if V == "dev_streamsync" else ev.synchronize()); t.append(time.perf_counter())
    if os.environ.get("PRE"):
        np.copyto(pre, h_out.numpy())
    else:
        o = h_out.numpy().copy()
    t.append(time.perf_counter())
    rows.append([round((t[k + 1] - t[k]) * 1e3, 2) for k in range(len(names))])
print(names)
print(V, "sync column:", [r[5] for r in rows])
