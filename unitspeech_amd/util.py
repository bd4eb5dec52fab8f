"""Pure-torch helpers the decoder's callers use (host-side, device-agnostic).

Mirrors `unitspeech/util.py:20-24` (sequence_mask), `:27-40` (generate_path) and `:55-59`
(fix_len_compatibility) of the reference."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def sequence_mask(length: torch.Tensor, max_length=None) -> torch.Tensor:
    if max_length is None:
        max_length = length.max()
    ar = torch.arange(int(max_length), dtype=length.dtype, device=length.device)
    return ar.unsqueeze(0) < length.unsqueeze(1)


def fix_len_compatibility(length, num_downsamplings_in_unet: int = 3) -> int:
    step = 2 ** num_downsamplings_in_unet
    length = int(length)
    return length if length % step == 0 else (length // step + 1) * step


def generate_path(duration: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    b, t_x, t_y = mask.shape
    ends = torch.cumsum(duration, 1).view(b * t_x)
    filled = sequence_mask(ends, t_y).to(mask.dtype).view(b, t_x, t_y)
    # row i covers the frames in [ends[i-1], ends[i]): subtract the previous row's prefix
    path = filled - F.pad(filled, (0, 0, 1, 0))[:, :-1]
    return path * mask
