"""Test infrastructure only: the reference's frame decode / encode arithmetic, restated literally in numpy.

decode  -- data/Adobe_test_3.py:171-195: cv2.imread (uint8 HWC, BGR) -> `astype(np.float32) / 255.` -> `[:, :, :, [2, 1, 0]]`
           -> transpose to NCHW.
encode  -- utils/util.py:105-129 `tensor2img`: clamp(0,1), RGB->BGR, CHW->HWC, `(img * 255.0).round()`, uint8;
           demo.py:94-99: clamp, permute, `* 255`, `astype(np.uint8)` (truncation), RGB kept.
Pinned: these are numpy one-liners of the cited lines; tests/test_oracle.py checks them against hand-computed vectors.
"""
import numpy as np


def decode(frames_u8_bgr):
    x = frames_u8_bgr.astype(np.float32) / 255.
    x = x[:, :, :, [2, 1, 0]]
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def encode_tensor2img(frames):
    x = np.clip(frames.astype(np.float32), 0.0, 1.0)
    x = np.transpose(x[:, [2, 1, 0], :, :], (0, 2, 3, 1))
    return (x * 255.0).round().astype(np.uint8)


def encode_demo(frames):
    x = np.clip(frames.astype(np.float32), 0.0, 1.0)
    x = np.transpose(x, (0, 2, 3, 1))
    return (x * 255).astype(np.uint8)
