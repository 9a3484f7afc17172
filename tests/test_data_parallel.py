"""In-process multi-device wrappers (nunif_amd/nunif/models/data_parallel.py) on a fake device table: two host "devices".

The reference's ``Waifu2x(gpus=[0, 1, ...])`` / ``BaseDepthModel.load(gpu=[0, 1, ...])`` go through ``DataParallelInference``
(tile minibatch split over the devices, ``nunif/models/data_parallel.py:8-38``) and ``DeviceSwitchInference`` (:53-68).  The
device-side behaviour (streams, events, peer copies) needs two GPUs; what is checked here is the host logic: chunk sizes ==
torch.nn.parallel.scatter's, order of the gathered results, attribute fall-through, the generic tiled_render route.
"""
import pytest
import torch

from nunif_amd.nunif.models.data_parallel import DataParallelInference, DeviceSwitchInference, chunk_sizes
from nunif_amd.nunif.models.register import create_model, data_parallel_model


class _Net(torch.nn.Module):
    i2i_scale, i2i_offset, i2i_blend_size, i2i_default_batch_size, name = 1, 2, 0, 4, "fake"

    def __init__(self):
        super().__init__()
        self.calls = []
        self.w = torch.nn.Parameter(torch.tensor(2.0))

    def find_valid_tile_size(self, t):
        return t

    def forward(self, x):
        self.calls.append(x.shape[0])
        return x[:, :, 2:-2, 2:-2] * self.w + 1.0

    def infer_delta(self, x, flip=False):
        self.calls.append(("delta", x.shape[0], flip))
        return x * 3


@pytest.mark.parametrize("n,parts", [(1, 2), (4, 2), (5, 2), (5, 4), (8, 3), (45, 8), (3, 8)])
def test_chunk_sizes_equal_tensor_chunk(n, parts):
    assert chunk_sizes(n, parts) == [c.shape[0] for c in torch.arange(n).chunk(parts)]


def test_data_parallel_inference_splits_and_gathers_in_order():
    net = _Net().eval()
    dp = DataParallelInference(net, device_ids=[-1, -1, -1])          # three host "devices"
    assert len(dp.replicas) == 3 and dp.replicas[0] is net and dp.replicas[1] is not net
    x = torch.rand(7, 3, 12, 12)
    with torch.inference_mode():
        y = dp(x)
        ref = net.forward(x)
    assert torch.equal(y, ref)
    assert [r.calls[0] for r in dp.replicas] == [3, 3, 1]
    assert dp.i2i_offset == 2 and dp.find_valid_tile_size(64) == 64 and not hasattr(dp, "render_frame")
    assert data_parallel_model(dp, [-1, -1]) is dp                       # wrapped once
    assert data_parallel_model(net, [-1]) is net


def test_device_switch_inference_dispatches_calls_and_methods():
    net = _Net().eval()
    ds = DeviceSwitchInference(net, device_ids=[-1])
    x = torch.rand(2, 3, 8, 8)
    with torch.inference_mode():
        assert torch.equal(ds(x), net.forward(x))
        assert torch.equal(ds.infer_delta(x, flip=True), x * 3)
    assert ("delta", 2, True) in net.calls and ds.i2i_scale == 1


def test_create_model_with_several_device_ids_wraps_like_the_reference():
    import nunif_amd.waifu2x.models.swin_unet  # noqa: F401  (registers the names)
    m = create_model("waifu2x.swin_unet_2x", device_ids=[-1, -1])
    assert isinstance(m, DataParallelInference) and m.name == "waifu2x.swin_unet_2x" and len(m.replicas) == 2
    assert m.replicas[1] is not m.replicas[0]
    a, b = m.replicas[0].state_dict(), m.replicas[1].state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a)
    one = create_model("waifu2x.swin_unet_2x", device_ids=[-1])
    assert not isinstance(one, DataParallelInference)


def test_wrappers_survive_deepcopy_and_report_unknown_devices():
    import copy
    net = _Net().eval()
    dp = DataParallelInference(net, device_ids=[-1, -1])
    dp2 = copy.deepcopy(dp)                       # __getattr__ must raise AttributeError while __dict__ is still empty
    assert isinstance(dp2, DataParallelInference) and len(dp2.replicas) == 2
    ds = DeviceSwitchInference(net, device_ids=[-1])
    assert isinstance(copy.deepcopy(ds), DeviceSwitchInference)
    with pytest.raises(ValueError, match="replicas exist on"):
        ds(torch.zeros(1, 3, 8, 8, device="meta"))


@pytest.mark.gpu
def test_data_parallel_on_two_real_gpus_is_bit_identical_to_one():
    """Needs two devices (the driver's multi-GPU node): ``dp(x)`` over [0, 1] == the single-device forward, bit for bit."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    from nunif_amd.synthetic import swin_unet_state_dict
    from nunif_amd.waifu2x.models.swin_unet import SwinUNet2x
    m = SwinUNet2x().eval()
    m.load_state_dict(swin_unet_state_dict(102, 2))
    one = m.to("cuda:0")
    x = torch.rand(5, 3, 64, 64, device="cuda:0")
    with torch.inference_mode():
        ref = one(x)
        dp = DataParallelInference(one, device_ids=[0, 1])
        got = dp(x)
    torch.cuda.synchronize()
    assert got.device == ref.device and torch.equal(got, ref)
