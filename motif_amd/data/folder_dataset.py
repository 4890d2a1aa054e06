"""Folder datasets: the sample-dict producers of the reference's evaluation (`/root/reference/data/Adobe_test_3.py:32-196`,
`data/Adobe_arbitrary_test.py:161-168`) for PNG frame folders, with the pixel conversion on the GPU.

Layout as the reference reads it: `<dataroot_GT>/<video>/<NNN>.png` (HR frames) and `<dataroot_LQ>/<video>/<NNN>.png` (LR frames
of the same indices).  A clip = `ref_num` input frames taken every (1 + interval)-th frame, sliding by (1 + interval)
(`Adobe_test_3.py:92-107`); the ground truth is the run of HR frames between the two centre inputs.  `mode`:
  "mid"        `Adobe_test_3.AdobeDataset`: GT indices [0,0,1,2,2], times 0, 1/2, 1                      (Adobe_test_3.py:158-166)
  "arbitrary"  `Adobe_arbitrary_test`: every frame between the centre inputs, times i/(len-1)            (Adobe_arbitrary_test.py:161-168)
What differs from the reference, on purpose: the hard-coded list / pickle paths (`Adobe_test_3.py:71,84`) are replaced by a directory
listing, files are decoded with PIL (RGB) instead of cv2 (BGR), and `__getitem__` returns the frames as uint8 -- the
`astype(float32) / 255`, channel reorder and HWC -> CHW of `Adobe_test_3.py:171-195` run in `motif_frames_u8_to_f32` on the device
(`decode_batch`), bit-identical to the reference's arithmetic (tests/test_kernels_gpu.py::test_frame_decode_encode_bit_exact).
"""
import os

import numpy as np
import torch
import torch.utils.data as data


def _list_frames(folder):
    names = [f for f in os.listdir(folder) if f.lower().endswith(".png")]
    return sorted(names, key=lambda f: int(os.path.splitext(f)[0]))


def read_png_u8(path):
    """uint8 [H,W,3] RGB (PIL); grey images are replicated, alpha dropped (`Adobe_test_3.py:176-182`)."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


class FolderClipDataset(data.Dataset):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.GT_root, self.LQ_root = opt["dataroot_GT"], opt.get("dataroot_LQ") or opt["dataroot_GT"]
        self.mode = opt.get("mode", "mid")
        if self.mode not in ("mid", "arbitrary"):
            raise ValueError("mode must be 'mid' or 'arbitrary'")
        interval = int(opt.get("interval", 1))
        ref_num = int(opt.get("ref_num", 4))
        videos = opt.get("videos") or sorted(d for d in os.listdir(self.GT_root) if os.path.isdir(os.path.join(self.GT_root, d)))
        self.file_list, self.gt_list = [], []
        interval_num = ref_num - 1
        step = 1 + interval
        for video in videos:
            frames = _list_frames(os.path.join(self.GT_root, video))
            index = 0
            while index + step * interval_num < len(frames):                           # Adobe_test_3.py:98-107
                inputs = [frames[i] for i in range(index, index + step * interval_num + 1, step)]
                gts = [frames[i] for i in range(index + step * (interval_num // 2), index + step * (interval_num // 2 + 1) + 1)]
                self.file_list.append([os.path.join(video, f) for f in inputs])
                self.gt_list.append([os.path.join(video, f) for f in gts])
                index += step

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, index):
        gts = self.gt_list[index]
        if self.mode == "mid":
            idx = [0, 0, len(gts) // 2, len(gts) - 1, len(gts) - 1]                    # [0,0,1,2,2] for interval 1
            times = [torch.tensor([i / 2.0]) for i in (0, 1, 2)]
        else:
            idx = [0] + list(range(len(gts))) + [len(gts) - 1]
            times = [torch.tensor([i / (len(idx) - 3)]) for i in idx[1:-1]]
        lq = np.stack([read_png_u8(os.path.join(self.LQ_root, f)) for f in self.file_list[index]], 0)
        gt = np.stack([read_png_u8(os.path.join(self.GT_root, gts[i])) for i in idx], 0)
        return {"LQs_u8": torch.from_numpy(lq), "GT_u8": torch.from_numpy(gt), "key": times, "time": times}


def collate_u8(samples):
    """DataLoader collate: stack the uint8 frames, collate `time` into T tensors [B,1] (`data/__init__.py:129`)."""
    out = {"LQs_u8": torch.stack([s["LQs_u8"] for s in samples], 0), "GT_u8": torch.stack([s["GT_u8"] for s in samples], 0)}
    T = len(samples[0]["time"])
    out["time"] = [torch.cat([s["time"][i][None] for s in samples], 0) for i in range(T)]
    return out


def decode_batch(batch, device="cuda", scale=None):
    """uint8 batch -> the sample dict `VideoSRBaseModel.feed_data` takes: LQs [B,n,3,h,w], GT [B,T+2,3,H,W] fp32 in [0,1] on the
    device, decoded by `motif_frames_u8_to_f32` (frames are RGB already: swap_rb off); `scale` adds the `scale` key as
    `test.py:180-182` does."""
    from .. import ops
    out = {"time": [t.to(device) for t in batch["time"]]}
    for k_in, k_out in (("LQs_u8", "LQs"), ("GT_u8", "GT")):
        u8 = batch[k_in].to(device)
        B, n, h, w, _ = u8.shape
        out[k_out] = ops.frames_u8_to_f32(u8.reshape(B * n, h, w, 3), swap_rb=False).view(B, n, 3, h, w)
    if scale is not None:
        h, w = out["LQs"].shape[-2:]
        out["scale"] = [[int(4 * np.ceil(h / 4)) * scale], [int(4 * np.ceil(w / 4)) * scale]]
    return out
