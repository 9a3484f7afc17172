"""``--depth-model`` -> depth model object.  Mirrors ``iw3/depth_model_factory.py`` :10-33 (first class whose ``supported``
accepts the name) for the model families the HIP engine carries; ZoeDepth / DepthPro / DA3 are external ``torch.hub`` nets
that are not restated and raise ``ValueError`` like an unknown name does in the reference."""
from .named_depth_models import DepthAnythingModel, NullDepthModel
from .video_depth_anything_model import VideoDepthAnythingModel
from .video_depth_anything_streaming_model import VideoDepthAnythingStreamingModel


def create_depth_model(model_type):
    for cls in (DepthAnythingModel, VideoDepthAnythingModel, VideoDepthAnythingStreamingModel, NullDepthModel):
        if cls.supported(model_type):
            return cls(model_type)
    raise ValueError(f"{model_type} is not supported")
