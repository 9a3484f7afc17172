"""In-process multi-device inference behind the reference's own entry points.

Mirrors ``nunif/models/data_parallel.py`` (reference): ``DataParallelInference`` :8-38 / ``DataParallelWrapper`` :41-50 — the
tile MINIBATCH of ``tiled_render`` is split over the listed devices (what ``Waifu2x(gpus=[0, 1, ...])`` and
``waifu2x.cli --gpu 0 1 ...`` get through ``create_model(device_ids=...)`` / ``data_parallel_model``, register.py:44-61) —
and ``DeviceSwitchInference`` :53-68 — one replica per device, the call goes to the replica of ``x.device`` (what
``BaseDepthModel.load(gpu=[...])`` and iw3's side models use, base_depth_model.py:129-133, utils.py:2361).

The reference builds on ``torch.nn.parallel`` (replicate / scatter / parallel_apply on Python threads / gather).  Engine
models are not ``nn.Module`` parameter holders — a replica is a deep copy with its own engine handle and workspace on its
device — and nothing here needs threads: every chunk's kernels are queued asynchronously on ITS device's stream, so the
devices run concurrently from one Python thread; the gather waits on per-device HIP events, never on the host.
One process per GPU with frame sharding (``nunif_amd.parallel``) remains the faster layout (no per-minibatch copies
over xGMI); this module is what makes the reference's ``--gpu 0 1 ...`` call sites work unchanged.
"""
import copy

import torch

from ..device import create_device


def _as_devices(device_ids):
    if device_ids is None:
        device_ids = list(range(torch.cuda.device_count()))
    return [d if isinstance(d, torch.device) else (torch.device(d) if isinstance(d, str) else create_device(d))
            for d in device_ids]


def replicate(module, devices):
    """One copy of ``module`` per device (the first device keeps the module itself, like torch's replicate keeps cuda:0's
    parameters).  Engine models copy their host weights and rebuild the device engine lazily (Model.__deepcopy__)."""
    replicas = []
    for k, dev in enumerate(devices):
        if hasattr(module, "replica"):               # device-bound engines (HipDepthAnythingV2): rebuilt from their host weights
            m = module.replica(dev)
        else:
            m = module if k == 0 else copy.deepcopy(module)
            m = m.to(dev)
        replicas.append(m.eval() if hasattr(m, "eval") else m)
    return replicas


def chunk_sizes(n, parts):
    """torch.nn.parallel.scatter's split of a batch of ``n`` over ``parts`` devices: ceil-sized chunks, trailing devices
    may get nothing (``Tensor.chunk`` semantics)."""
    step = (n + parts - 1) // parts if n else 0
    sizes, left = [], n
    for _ in range(parts):
        s = min(step, left)
        if s > 0:
            sizes.append(s)
        left -= s
    return sizes


class DataParallelInference:
    """``model(minibatch)`` with the minibatch split over ``device_ids`` along ``dim`` and the results concatenated on
    ``output_device`` (default: the first id).  Attribute access falls through to the wrapped module (``i2i_scale``,
    ``find_valid_tile_size``, ...), except ``render_frame``: the whole-frame fast path of ``tiled_render`` is a
    single-device launch sequence, so a multi-device model takes the generic gather -> model -> stitch route."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0):
        self.dim = dim
        self.devices = _as_devices(device_ids)
        assert len(self.devices) > 0
        self.device_ids = list(device_ids) if device_ids is not None else list(range(len(self.devices)))
        self.output_device = self.devices[0] if output_device is None else _as_devices([output_device])[0]
        self.module = module.eval() if hasattr(module, "eval") else module
        self.replicas = replicate(self.module, self.devices)
        self._streams = {}

    def _stream(self, dev):
        if dev.type != "cuda":
            return None
        if dev not in self._streams:
            self._streams[dev] = torch.cuda.Stream(dev)
        return self._streams[dev]

    def __call__(self, x, **kwargs):
        sizes = chunk_sizes(x.shape[self.dim], len(self.devices))
        chunks = torch.split(x, sizes, dim=self.dim) if sizes else []
        src_dev = x.device
        src_event = None
        if src_dev.type == "cuda":
            src_event = torch.cuda.Event()
            src_event.record(torch.cuda.current_stream(src_dev))
        outs, events = [], []
        for chunk, dev, replica in zip(chunks, self.devices, self.replicas):
            st = self._stream(dev)
            if st is None:                                            # host "devices" (the CPU tests): sequential
                outs.append(replica(chunk.to(dev), **kwargs))
                events.append(None)
                continue
            with torch.cuda.device(dev), torch.cuda.stream(st):
                if src_event is not None:
                    st.wait_event(src_event)                          # the minibatch was produced on the caller's stream
                xi = chunk.to(dev, non_blocking=True)
                if xi.data_ptr() == chunk.data_ptr():
                    chunk.record_stream(st)
                zi = replica(xi, **kwargs)
                ev = torch.cuda.Event()
                ev.record(st)
            outs.append(zi)
            events.append(ev)
        out_dev = self.output_device
        if out_dev.type != "cuda":
            return torch.cat([o.to(out_dev) for o in outs], dim=self.dim)
        with torch.cuda.device(out_dev):
            cur = torch.cuda.current_stream(out_dev)
            parts = []
            for o, ev in zip(outs, events):
                if ev is not None:
                    cur.wait_event(ev)
                parts.append(o.to(out_dev, non_blocking=True))
                o.record_stream(cur)
            return torch.cat(parts, dim=self.dim) if len(parts) > 1 else parts[0]

    forward = __call__

    def __getattr__(self, name):
        # (copy.deepcopy / pickle probe ``__deepcopy__`` / ``__getstate__`` on an instance whose __init__ has not run:
        #  that must be an AttributeError, not a KeyError)
        if name == "render_frame" or "module" not in self.__dict__ or (name.startswith("__") and name.endswith("__")):
            raise AttributeError(name)      # dunder probes (__deepcopy__, __getstate__) are about the WRAPPER, never delegated
        return getattr(self.__dict__["module"], name)

    def get_device(self):
        return self.output_device

    def to(self, device):                # the wrapper owns its placement (register.py: create_model keeps a wrapped model as is)
        return self

    def eval(self):
        return self


# ``nn.DataParallel`` subclass in the reference (register.py:44-49); the same object here
DataParallelWrapper = DataParallelInference


class DeviceSwitchInference:
    """One replica per device; ``model(x)`` runs on the replica that lives on ``x.device`` (reference :53-68)."""

    def __init__(self, module, device_ids=None):
        self._devices = _as_devices(device_ids)
        self._module = module.eval() if hasattr(module, "eval") else module
        self._replicas = replicate(self._module, self._devices)

    def _replica(self, device):
        device = torch.device(device)
        if device not in self._devices:
            raise ValueError(f"DeviceSwitchInference: tensor on {device}, replicas exist on {[str(d) for d in self._devices]}")
        return self._replicas[self._devices.index(device)]

    def __call__(self, x, *args, **kwargs):
        return self._replica(x.device)(x, *args, **kwargs)

    def __getattr__(self, name):
        if "_module" not in self.__dict__ or (name.startswith("__") and name.endswith("__")):
            raise AttributeError(name)
        attr = getattr(self.__dict__["_module"], name)
        if callable(attr) and hasattr(attr, "__self__") and attr.__self__ is self.__dict__["_module"]:
            # a method of the wrapped model that takes tensors (``infer_delta``, ``infer``, ...): same dispatch as __call__
            def on_device(*args, **kwargs):
                dev = next((a.device for a in list(args) + list(kwargs.values()) if torch.is_tensor(a)), None)
                target = self._module if dev is None else self._replica(dev)
                return getattr(target, name)(*args, **kwargs)
            return on_device
        return attr
