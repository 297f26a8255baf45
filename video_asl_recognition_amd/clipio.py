"""Host-side steps either side of the clip -> logits path: checkpoint import, frame sampling,
keypoint wire format, label map / top-k decoding and a double-buffered clip stream.

The reference snapshot holds no code for any of these (/root/reference/README.md:1 is a bare
title; /root/reference/.gitignore:1-5 only shows that the project produced ``*.pt`` / ``*.pth``
checkpoints and a ``pose_estimation`` stage), so every contract below is stated here in full.
Nothing in this module computes on the CPU what the GPU path computes: it prepares inputs
(indices, normalised keypoints, state_dict keys) and decodes outputs.
"""
from __future__ import annotations

import json
from typing import Dict, Iterable, Iterator, List, Optional, Sequence, Tuple

import torch

from .spec import ModelSpec, SPECS, param_shapes

# ----------------------------------------------------------------------------- checkpoints

_WRAPPER_KEYS = ("state_dict", "model_state_dict", "model", "net", "weights")
_PREFIXES = ("module.", "model.", "net.")


def _strip(k: str) -> str:
    changed = True
    while changed:
        changed = False
        for p in _PREFIXES:
            if k.startswith(p):
                k, changed = k[len(p):], True
    return k


def map_state_dict(sd: Dict[str, torch.Tensor], spec: ModelSpec) -> Dict[str, torch.Tensor]:
    """Brings a checkpoint's parameter names to the ones ``AslClipModel.load_state_dict`` takes.

    Accepted spellings: the model's own names (``backbone.layer1.0.conv1.weight``, ``fc.weight`` ...),
    the same under ``module.`` / ``model.`` / ``net.`` prefixes (DataParallel and wrapper modules), and
    a bare torchvision ResNet (``conv1.weight``, ``layer1.0...``, ``fc.*``), whose trunk names gain the
    ``backbone.`` prefix while ``fc.*`` stays the classifier.  ``num_batches_tracked`` entries are dropped.
    A classifier with a different class count is an error (shape check in ``load_state_dict``).
    """
    want = param_shapes(spec)
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        k = _strip(k)
        if k not in want and ("backbone." + k) in want and not k.startswith("fc."):
            k = "backbone." + k
        out[k] = v
    return out


def load_checkpoint(path: str, spec) -> Dict[str, torch.Tensor]:
    """Reads a ``*.pt`` / ``*.pth`` file with ``torch.load(weights_only=True)`` (nothing in the file is
    executed), unwraps ``{"state_dict": ...}``-style containers and maps the names (``map_state_dict``)."""
    spec = SPECS[spec] if isinstance(spec, str) else spec
    obj = torch.load(path, map_location="cpu", weights_only=True)
    for _ in range(3):
        if isinstance(obj, dict) and not any(isinstance(v, torch.Tensor) for v in obj.values()):
            for wk in _WRAPPER_KEYS:
                if wk in obj and isinstance(obj[wk], dict):
                    obj = obj[wk]
                    break
            else:
                break
    if not isinstance(obj, dict) or not any(isinstance(v, torch.Tensor) for v in obj.values()):
        raise ValueError(f"{path}: no state_dict found")
    return map_state_dict({k: v for k, v in obj.items() if isinstance(v, torch.Tensor)}, spec)


# ----------------------------------------------------------------------------- frame sampling

def sample_frame_indices(n_frames: int, T: int, mode: str = "uniform") -> List[int]:
    """Indices of the T frames a clip model sees out of a video of ``n_frames``.

    ``uniform``: the video is cut into T equal segments and the centre frame of each is taken
    (index ``floor((i + 0.5) * n / T)``); a video shorter than T keeps all its frames in order and
    repeats the last one (``min(i, n - 1)``).  ``head``: the first T frames, padded the same way.
    """
    if n_frames <= 0 or T <= 0:
        raise ValueError("n_frames and T must be positive")
    if mode == "head" or n_frames <= T:
        return [min(i, n_frames - 1) for i in range(T)]
    if mode != "uniform":
        raise ValueError(f"unknown sampling mode {mode!r}")
    return [min(((2 * i + 1) * n_frames) // (2 * T), n_frames - 1) for i in range(T)]


def center_crop_box(h: int, w: int, aspect: float = 1.0) -> Tuple[int, int, int, int]:
    """(top, left, height, width) of the largest centred window of the given width/height ratio.
    ``classify_clip`` resizes whatever it is given to S x S; cropping first keeps the aspect ratio."""
    if h <= 0 or w <= 0 or aspect <= 0:
        raise ValueError("bad geometry")
    ch, cw = h, int(round(h * aspect))
    if cw > w:
        cw, ch = w, int(round(w / aspect))
    return (h - ch) // 2, (w - cw) // 2, ch, cw


def sample_clip(video: torch.Tensor, T: int, mode: str = "uniform", crop: bool = True) -> torch.Tensor:
    """uint8 video [N,H,W,3] -> uint8 clip [T,h,w,3] (frame sampling, then the centred square crop)."""
    if video.dim() != 4 or video.shape[-1] != 3 or video.dtype != torch.uint8:
        raise ValueError("video must be uint8 [N,H,W,3]")
    idx = torch.tensor(sample_frame_indices(video.shape[0], T, mode), device=video.device)
    clip = video.index_select(0, idx)
    if crop:
        t, l, ch, cw = center_crop_box(video.shape[1], video.shape[2])
        clip = clip[:, t:t + ch, l:l + cw]
    return clip.contiguous()


# ----------------------------------------------------------------------------- keypoints

def normalize_keypoints(kp: torch.Tensor, height: int, width: int, mode: str = "image") -> torch.Tensor:
    """Pixel-space hand landmarks [..., 21, 2] (x, y) -> the float32 tensor the two-stream model takes.

    ``image``: x / width, y / height (range [0, 1)).  ``wrist``: landmark 0 (the wrist) is subtracted
    and the hand is scaled by its largest wrist distance, so the result is translation- and
    scale-free.  A missing hand -- any non-finite coordinate, or all 21 landmarks at (0, 0) -- becomes
    all zeros in either mode.
    """
    if kp.shape[-1] != 2 or kp.shape[-2] != 21:
        raise ValueError(f"keypoints must end in [21, 2], got {tuple(kp.shape)}")
    kp = kp.to(torch.float32)
    finite = torch.isfinite(kp).all(dim=-1).all(dim=-1)
    present = finite & (kp.abs().sum(dim=(-1, -2)) > 0)
    kp = torch.where(torch.isfinite(kp), kp, torch.zeros_like(kp))
    if mode == "image":
        out = kp / torch.tensor([float(width), float(height)], device=kp.device)
    elif mode == "wrist":
        rel = kp - kp[..., :1, :]
        scale = rel.norm(dim=-1).amax(dim=-1, keepdim=True).clamp_min(1e-6)[..., None]
        out = rel / scale
    else:
        raise ValueError(f"unknown keypoint mode {mode!r}")
    return out * present[..., None, None].to(out.dtype)


# ----------------------------------------------------------------------------- labels

class LabelMap:
    """Class index -> gloss.  Reads the WLASL release's list form (``[{"gloss": "book", ...}, ...]``,
    index = position), a plain list of strings, or a ``{"0": "book", ...}`` / ``{"book": 0, ...}`` mapping."""

    def __init__(self, glosses: Sequence[str]):
        self.glosses = list(glosses)

    @classmethod
    def from_json(cls, path: str, num_classes: Optional[int] = None) -> "LabelMap":
        with open(path) as f:
            obj = json.load(f)
        if isinstance(obj, list):
            g = [e["gloss"] if isinstance(e, dict) else str(e) for e in obj]
        elif isinstance(obj, dict):
            if all(isinstance(v, int) for v in obj.values()):
                obj = {v: k for k, v in obj.items()}
            pairs = sorted((int(k), str(v)) for k, v in obj.items())
            if [k for k, _ in pairs] != list(range(len(pairs))):
                raise ValueError(f"{path}: class indices must be 0..n-1")
            g = [v for _, v in pairs]
        else:
            raise ValueError(f"{path}: unsupported label file")
        if num_classes is not None:
            if len(g) < num_classes:
                raise ValueError(f"{path}: {len(g)} labels for {num_classes} classes")
            g = g[:num_classes]
        return cls(g)

    def __len__(self):
        return len(self.glosses)

    def decode(self, idx: torch.Tensor, prob: torch.Tensor) -> List[List[Tuple[str, float]]]:
        """(idx [B,k], prob [B,k]) from ``classify_clip`` -> per clip a list of (gloss, probability)."""
        idx, prob = idx.cpu().tolist(), prob.cpu().tolist()
        return [[(self.glosses[i], float(p)) for i, p in zip(ri, rp)] for ri, rp in zip(idx, prob)]


# ----------------------------------------------------------------------------- streaming

class ClipStream:
    """Classifies a sequence of host-resident uint8 clip batches with the host-to-device copy of
    batch i+1 overlapping the compute of batch i: two pinned staging buffers, two device buffers,
    one copy stream.  Yields ``(idx, prob)`` per batch exactly as ``model.classify_clip`` would."""

    def __init__(self, model, topk: int = 5):
        self.model, self.topk = model, int(topk)
        self._copy = torch.cuda.Stream(device=model.device)
        # flat byte buffers, grown to the largest batch seen and viewed at each batch's shape
        self._pin: List[Optional[torch.Tensor]] = [None, None]
        self._dev: List[Optional[torch.Tensor]] = [None, None]
        self._free: List[Optional[torch.cuda.Event]] = [None, None]   # device buffer no longer read

    def _stage(self, slot: int, batch: torch.Tensor):
        """Starts the transfer of `batch` into device buffer `slot`; returns (event, device view)."""
        if batch.dtype != torch.uint8 or batch.dim() != 5:
            raise ValueError("batches must be uint8 [B,T,H0,W0,3]")
        nbytes = batch.numel()
        # The previous user of this slot must have finished reading before the buffer is overwritten -- or
        # released: a dropped buffer goes back to the allocator and may be handed out again at once.
        if self._free[slot] is not None:
            self._free[slot].synchronize()
            self._free[slot] = None
        if self._dev[slot] is None or self._dev[slot].numel() < nbytes:
            self._dev[slot] = None
            # Allocated with the copy stream current: the caching allocator keeps a pool per stream, so this
            # block cannot be one that classify_clip's kernels, still queued on the compute streams, have
            # released on the host but not finished using, and the copy below needs no ordering against them.
            with torch.cuda.stream(self._copy):
                self._dev[slot] = torch.empty(nbytes, dtype=torch.uint8, device=self.model.device)
        src = batch.contiguous()
        if not src.is_pinned():                   # pageable memory: stage it (a pinned batch is transferred as it is
            if self._pin[slot] is None or self._pin[slot].numel() < nbytes:   # and must stay untouched until its
                self._pin[slot] = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)   # result has been yielded)
            pin = self._pin[slot][:nbytes].view(batch.shape)
            pin.copy_(src)
            src = pin
        dev = self._dev[slot][:nbytes].view(batch.shape)
        with torch.cuda.stream(self._copy):
            dev.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy)
        return ev, dev

    def __call__(self, batches: Iterable[torch.Tensor]) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        it = iter(batches)
        try:
            ready, dev = self._stage(0, next(it))
        except StopIteration:
            return
        slot = 0
        nxt = next(it, None)
        while True:
            torch.cuda.current_stream().wait_event(ready)
            out = self.model.classify_clip(dev, topk=self.topk)      # asynchronous launches
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream())
            self._free[slot] = done
            if nxt is None:
                yield out
                return
            # host copy into pinned memory and the H2D transfer run while the GPU works on this batch
            ready, dev = self._stage(slot ^ 1, nxt)
            slot ^= 1
            nxt = next(it, None)
            yield out
