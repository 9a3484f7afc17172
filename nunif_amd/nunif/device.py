"""Device naming and autocast policy.  Mirrors ``nunif/device.py`` (create_device :12-31, autocast :58-71).

ROCm shows up as ``cuda`` in PyTorch; there is no mps/xpu branch here.  The HIP engine computes in fp16 storage /
fp32 accumulate regardless of autocast, so ``autocast`` only matters for torch modules mixed into the path.
"""
import torch


def create_device_name(device_id):
    if isinstance(device_id, (list, tuple)):
        assert len(device_id) > 0
        device_id = device_id[0]
    if device_id < 0:
        return "cpu"
    if not torch.cuda.is_available():
        raise ValueError("No GPU available. Use `--gpu -1` for CPU.")
    return f"cuda:{device_id}"


def create_device(device_id):
    return torch.device(create_device_name(device_id))


def device_is(device, name):
    return device.type == name if isinstance(device, torch.device) else name in str(device)


def device_is_cpu(device):
    return device_is(device, "cpu")


def device_is_cuda(device):
    return device_is(device, "cuda")


def autocast(device, dtype=None, enabled=True):
    if device_is_cpu(device):
        return torch.autocast(device_type="cpu", dtype=torch.bfloat16, enabled=False)  # reference disables it
    device_type = device.split(":")[0] if isinstance(device, str) else device.type
    return torch.autocast(device_type=device_type, dtype=dtype, enabled=enabled)
