"""Device-side frame preprocessing: the reference's host transform
(/root/reference/src/utils/dataloader.py:18-32, src/real_time_inference.py:16-28) as one HIP kernel."""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def preprocess_frames(frames_u8: torch.Tensor, crop: int = 224, device: str | torch.device = "cuda:0") -> torch.Tensor:
    """uint8 BGR frames [..., H, W, 3] (OpenCV layout) -> CLIP-normalised fp32 [..., 3, crop, crop] on `device`:
    ToTensor -> Resize(crop, bicubic) -> CenterCrop(crop) -> BGR->RGB -> Normalize."""
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() < 3 or frames_u8.shape[-1] != 3:
        raise ValueError("expected a uint8 tensor [..., H, W, 3]")
    lib = _lib.load()
    dev = torch.device(device)
    x = frames_u8.to(dev).contiguous()
    lead, (H, W) = x.shape[:-3], x.shape[-3:-1]
    nf = 1
    for d in lead:
        nf *= d
    out = torch.empty((*lead, 3, crop, crop), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gitcap_preprocess(ctypes.c_void_p(x.data_ptr()), nf, H, W, ctypes.c_void_p(out.data_ptr()), crop,
                                   ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise _lib.GitcapError(f"gitcap_preprocess failed (status {rc}): frames {H}x{W} -> {crop}")
    return out
