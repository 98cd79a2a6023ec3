"""CPU oracle of the frame transform.  TEST INFRASTRUCTURE ONLY (see oracle/git_oracle.py).

Restates ``image_transform()`` of the reference (/root/reference/src/utils/dataloader.py:18-32;
same code at src/real_time_inference.py:16-28) with plain torch, following torchvision 0.16.0
(requirements.txt:2) for tensors: ToTensor = uint8 HWC -> fp32 CHW / 255; Resize(224, BICUBIC) on a
tensor = F.interpolate(mode='bicubic', align_corners=False, antialias=False) with the shorter side
mapped to 224 and the longer to int(224 * long / short), no clamping for float images;
CenterCrop offsets int(round((size - 224) / 2.0)); BGR->RGB (dataloader.py:14-16); Normalize with
the CLIP mean/std (dataloader.py:26-29).  torchvision itself is absent offline: PARITY UNPINNED by
the reference; the known-answer cases in tests/test_preprocess.py pin the constants.
"""
import torch
import torch.nn.functional as F

MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def image_transform(frame_hwc_bgr_u8: torch.Tensor, crop: int = 224) -> torch.Tensor:
    x = frame_hwc_bgr_u8.permute(2, 0, 1).float() / 255.0                      # ToTensor
    _, h, w = x.shape
    if h <= w:
        nh, nw = crop, int(crop * w / h)
    else:
        nh, nw = int(crop * h / w), crop
    if (nh, nw) != (h, w):
        x = F.interpolate(x[None], size=(nh, nw), mode="bicubic", align_corners=False, antialias=False)[0]
    top, left = int(round((nh - crop) / 2.0)), int(round((nw - crop) / 2.0))  # CenterCrop
    x = x[:, top:top + crop, left:left + crop]
    x = x[[2, 1, 0], ...]                                                      # BGR -> RGB
    mean = torch.tensor(MEAN)[:, None, None]
    std = torch.tensor(STD)[:, None, None]
    return (x - mean) / std
