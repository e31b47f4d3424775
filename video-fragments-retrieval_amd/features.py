"""The extractor front-end of ``get_rgb_features.py`` around the HIP VGG19-fc7 stack.

* ``sample_frames(num_frames, fps, num_segments)`` -- the frame-index selection of ``DiDeMoDataset.__getitem__``
  (``get_rgb_features.py:45-60``): 25 frames per 5-second segment (5 fps), the last segment possibly short.  Host integer /
  float64 work, vectorised: the reference's running ``curr_step += step`` is a float64 cumulative sum and its ``round`` is
  round-half-even, which is exactly ``np.rint(np.cumsum(...))``.
* ``extract_video(frames_u8, fps, num_segments, weights)`` -- index-select + ImageNet normalisation + VGG19 up to fc7 in ONE
  pass over the selected frames on the device (the reference moves 16 frames at a time to the GPU and back, ``:145-148``).
* ``extract_dataset(dataset_info, decoder, features_dir, weights)`` -- the main loop (``:105-156``): resume by skipping
  ``{model}_ft_{video}.npy`` files that exist and the videos listed in ``missed_videos_features.json``, one ``.npy`` per
  video, unreadable videos appended to the missed list.

Decoding mp4 (``torchvision.io.read_video``, ``:40-44``) is codec I/O and stays the caller's: ``decoder(video, num_segments)``
returns ``(uint8 frames [T, H, W, 3], fps)`` -- or ``(None, 0)`` for an unreadable file.  Both extractor variants of the
reference are built: ``model_type="vgg19"`` (fc7, 4096-d, ``:122-126``; ``weights`` = (conv_w, conv_b, fc6, fc7)) and
``model_type="resnet152"`` (global average pool, 2048-d, ``:127-131``; ``weights`` = ``_vfr.resnet_pack(state_dict)``).  The
pretrained weights themselves are a network fetch and the caller's to provide.
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

SELECT_FPS = 25          # frames kept per segment (get_rgb_features.py:17-19: 5 fps x 5 s)
SEC_PER_SEGMENT = 5
VGG19_CFG = [64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"]


def sample_frames(num_frames: int, fps: float, num_segments: int) -> np.ndarray:
    """Indices (int64, ascending) of the frames ``get_rgb_features.py:45-60`` keeps from a decoded clip of ``num_frames``."""
    if num_frames <= 0:
        return np.zeros(0, np.int64)
    video_length = num_frames / fps
    if round(video_length) == num_segments * SEC_PER_SEGMENT:                 # all segments are full (:48-49)
        step = (num_frames - 1) / (SELECT_FPS * num_segments + 1)
    else:                                                                     # last segment shorter than 5 s (:50-51)
        step = (round((num_segments - 1) * SEC_PER_SEGMENT * fps) - 1) / (SELECT_FPS * (num_segments - 1) + 1)
    stop = min(num_frames, SELECT_FPS * num_segments * step) - 1
    if not step > 0:
        return np.zeros(1, np.int64)
    # the loop appends while curr_step <= stop - step: at most this many steps (a few spare ones, cut by the test below)
    n = int(max(stop / step, 0)) + 3
    cur = np.cumsum(np.full(n, step, np.float64))                             # curr_step after 1, 2, ... additions
    prev = np.concatenate([[0.0], cur[:-1]])
    keep = prev <= stop - step                                                # the addition happened
    keep = np.logical_and.accumulate(keep)
    return np.concatenate([[0], np.rint(cur[keep]).astype(np.int64)])


def extract_video(frames_u8, fps: float, num_segments: int, weights, cfg=None, model_type: str = "vgg19") -> torch.Tensor:
    """uint8 frames [T, H, W, 3] (device tensor or numpy) -> features [T_sel, FEATURE_DIM[model_type]] on the device.
    ``cfg``: the VGG layer list, or ``(blocks, width)`` of the ResNet (defaults: VGG-19 "E" / ResNet-152)."""
    from . import _vfr
    if model_type == "resnet152":
        dev = weights[0][0].device
    elif model_type == "vgg19":
        dev = weights[0][0].device
    else:
        raise ValueError(f"unknown extractor {model_type!r} (get_rgb_features.py:122-131 has vgg19 and resnet152)")
    fr = torch.as_tensor(frames_u8).to(dev)
    idx = torch.from_numpy(sample_frames(int(fr.shape[0]), fps, num_segments)).to(dev)
    sel = fr.index_select(0, idx).contiguous()                                # (:59) -- a gather, then everything in HIP
    if model_type == "resnet152":
        blocks, width = cfg if cfg is not None else (_vfr.RESNET152_BLOCKS, 64)
        return _vfr.resnet_pool(sel, weights, blocks, width)
    conv_w, conv_b, fc6, fc7 = weights
    return _vfr.vgg_fc7(sel, VGG19_CFG if cfg is None else cfg, conv_w, conv_b, fc6, fc7)


def extract_dataset(dataset_info, decoder, features_dir, weights, model_type: str = "vgg19", cfg=None,
                    missed_path="missed_videos_features.json"):
    """The extraction loop with the reference's resume / skip-and-record behaviour.  Returns (written, missed) video lists."""
    if model_type not in ("vgg19", "resnet152"):
        raise ValueError(f"unknown extractor {model_type!r} (get_rgb_features.py:122-131 has vgg19 and resnet152)")
    ft = Path(features_dir)
    ft.mkdir(exist_ok=True)
    prefix = f"{model_type}_ft_"
    done = [f.stem[len(prefix):] for f in sorted(ft.glob(f"{prefix}*.npy"))]                  # (:105-106)
    missed_path = Path(missed_path)
    missed = json.loads(missed_path.read_text()) if missed_path.exists() else []            # (:107-111)
    skip = set(done) | set(missed)
    written = []
    for item in dataset_info:                                                                 # (:115-116)
        video, nseg = item["video"], item["num_segments"]
        if video in skip:
            continue
        frames, fps = decoder(video, nseg)
        if frames is None or len(frames) == 0:                                                # (:75-78,152-153)
            missed.append(video)
            continue
        feats = extract_video(frames, fps, nseg, weights, cfg, model_type)
        np.save(ft / f"{prefix}{video}", feats.cpu().numpy())                                # (:150-151)
        written.append(video)
    missed_path.write_text(json.dumps(missed))                                               # (:155-156)
    return written, missed
