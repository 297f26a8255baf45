"""MI355X-native video-ASL clip -> logits path (gfx950 HIP kernels behind a forward()/classify_clip() surface)."""
from .spec import SPECS, ModelSpec, param_shapes

__all__ = ["SPECS", "ModelSpec", "param_shapes", "AslClipModel", "init_state_dict", "gather_logits", "shard_batch"]


def __getattr__(name):
    # model/_lib load libaslhip.so; keep `import video_asl_recognition_amd` itself light so spec-only
    # users (and the oracle-side tests) do not need the shared object.
    if name in ("AslClipModel", "init_state_dict"):
        from . import model
        return getattr(model, name)
    if name in ("gather_logits", "shard_batch"):
        from . import dist
        return getattr(dist, name)
    raise AttributeError(name)
