"""Synthetic clips honouring the reference's sample-dict contract.

Contract (`/root/reference/data/Adobe_test_3.py:196`, `Adobe_arbitrary_test.py:161-168`):
  LQs  float32 [B,n,3,h,w] RGB in [0,1];  GT [B,T+2,3,H,W] (first/last duplicated);
  time list of T tensors [B,1] with values i/(T-1);  optional `scale`.
The frames are a low-pass random texture translated by a known sub-pixel shift per frame so that the
flow extractor sees trackable structure (SURVEY.md §8(d) "Synthetic inputs").
"""
import torch
import torch.nn.functional as F


def smooth_video(n_frames, h, w, seed=0, batch=1, shift=(1.3, -0.7), blur=5):
    g = torch.Generator().manual_seed(seed)
    pad = 16
    hh, ww = h + 2 * pad, w + 2 * pad
    base = torch.rand(batch, 3, hh, ww, generator=g, dtype=torch.float32)
    k = torch.ones(1, 1, blur, blur) / float(blur * blur)
    for _ in range(2):
        base = F.conv2d(F.pad(base.reshape(batch * 3, 1, hh, ww), (blur // 2,) * 4, mode="reflect"), k)
        base = base.reshape(batch, 3, hh, ww)
    lo = base.amin(dim=(1, 2, 3), keepdim=True)
    hi = base.amax(dim=(1, 2, 3), keepdim=True)
    base = (base - lo) / (hi - lo + 1e-12)
    ys = torch.arange(h, dtype=torch.float32) + pad
    xs = torch.arange(w, dtype=torch.float32) + pad
    frames = []
    for i in range(n_frames):
        dx, dy = shift[0] * i, shift[1] * i
        gx = ((xs + dx) / (ww - 1)) * 2 - 1
        gy = ((ys + dy) / (hh - 1)) * 2 - 1
        grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], dim=-1)
        grid = grid.unsqueeze(0).expand(batch, -1, -1, -1)
        frames.append(F.grid_sample(base, grid, mode="bilinear", padding_mode="border", align_corners=True))
    return torch.stack(frames, dim=1).clamp_(0, 1).contiguous()


def synthetic_sample(h, w, scale, n_times, n_frames=4, batch=1, seed=0, with_scale_key=True):
    """One sample dict in collated form, as `test.py:162-185` consumes it."""
    lqs = smooth_video(n_frames, h, w, seed=seed, batch=batch)
    H, W = int(round(h * scale)), int(round(w * scale))
    g = torch.Generator().manual_seed(seed + 1)
    gt = torch.rand(batch, n_times + 2, 3, H, W, generator=g, dtype=torch.float32)
    times = [torch.full((batch, 1), i / max(n_times - 1, 1), dtype=torch.float32) for i in range(n_times)]
    sample = {"LQs": lqs, "GT": gt, "time": times}
    if with_scale_key:
        sample["scale"] = [[H], [W]]
    return sample
