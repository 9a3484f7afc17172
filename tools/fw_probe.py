import sys, time, torch
sys.path.insert(0, ".")
from nunif_amd.iw3.forward_warp import apply_divergence_forward_warp
from nunif_amd.iw3.dilation import dilate_edge
from nunif_amd.synthetic import synth_depth
dev = "cuda:0"
def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters
for (B, H, W) in ((2, 1080, 1920), (1, 1080, 1920), (1, 1440, 2560), (1, 2160, 3840), (1, 1080, 3840), (1, 2160, 3000)):
    c = torch.rand(B, 3, H, W, device=dev); d = synth_depth(1, B, H, W, "smooth_edges").to(dev)
    for kind in ("smooth_edges",):
        t = timeit(lambda: apply_divergence_forward_warp(c, d, 2.0, 0.5, method="forward_fill", width_base=False))
        t2 = timeit(lambda: apply_divergence_forward_warp(c, d, 2.0, 0.5, method="forward", width_base=False))
        print(B, H, W, "fill %.1f us  noFill %.1f us  %.2f TB/s" % (t * 1e6, t2 * 1e6, B * H * W * 40 / t / 1e12))
ds = synth_depth(2, 2, 392, 686, "smooth_edges").to(dev) * 5
print("dilate_edge(2) 2x392x686: %.1f us" % (timeit(lambda: dilate_edge(ds, 2)) * 1e6))
