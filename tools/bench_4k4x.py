import sys, time, torch
sys.path.insert(0, ".")
from nunif_amd.nunif.utils.render import tiled_render
from nunif_amd.waifu2x.models.swin_unet import SwinUNet4x
from nunif_amd.synthetic import swin_unet_state_dict
torch.set_grad_enabled(False)
m = SwinUNet4x().eval(); m.load_state_dict(swin_unet_state_dict(104, 4)); m = m.to("cuda:0")
x = torch.rand(3, 2160, 3840, device="cuda")
for bs in (34, 85):
    for _ in range(2): tiled_render(x, m, tile_size=256, batch_size=bs)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(4): tiled_render(x, m, tile_size=256, batch_size=bs)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 4
    print("4K 4x batch", bs, "ms/frame", round(dt * 1e3, 2), "input MPix/s", round(2160 * 3840 / dt / 1e6, 1), "output MPix/s", round(16 * 2160 * 3840 / dt / 1e6, 1))
