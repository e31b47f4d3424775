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


def select_frames(frames_u8, fps: float, num_segments: int, pin: bool = False) -> torch.Tensor:
    """The kept frames of a decoded clip (``:45-59``), selected WHERE THE FRAMES ARE: a host array is gathered on the host (one
    frame in six of a 30 fps clip survives: the H2D copy carries 23 MB instead of 135), a device tensor on the device.
    ``pin``: gather into page-locked memory (for an asynchronous copy).  A clip whose frame count is already the selection's
    (``fps`` <= 0: pre-selected) passes through."""
    fr = torch.as_tensor(frames_u8)
    if fps is not None and fps > 0:
        idx = torch.from_numpy(sample_frames(int(fr.shape[0]), fps, num_segments))
        if fr.device.type == "cpu":
            out = torch.empty((len(idx),) + tuple(fr.shape[1:]), dtype=fr.dtype, pin_memory=pin)
            torch.index_select(fr, 0, idx, out=out)
            return out
        return fr.index_select(0, idx.to(fr.device)).contiguous()
    return fr.contiguous()


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
    sel = select_frames(frames_u8, fps, num_segments).to(dev)                 # (:59) -- a gather, then everything in HIP
    if model_type == "resnet152":
        blocks, width = cfg if cfg is not None else (_vfr.RESNET152_BLOCKS, 64)
        return _vfr.resnet_pool(sel, weights, blocks, width)
    conv_w, conv_b, fc6, fc7 = weights
    return _vfr.vgg_fc7(sel, VGG19_CFG if cfg is None else cfg, conv_w, conv_b, fc6, fc7)


def extract_dataset(dataset_info, decoder, features_dir, weights, model_type: str = "vgg19", cfg=None,
                    missed_path="missed_videos_features.json", pipeline: bool = True, progress=None, stats=None):
    """The extraction loop with the reference's resume / skip-and-record behaviour.  Returns (written, missed) video lists.

    ``pipeline=True`` (ROCm device only): the loop of ``get_rgb_features.py:134-153`` as a three-stage pipeline -- a reader thread
    decodes video i + 1 into page-locked memory while the device runs the stack on video i (frames H2D, index-select,
    normalise, the whole network: all queued asynchronously on the compute stream), and a writer thread waits for video
    i - 1's D2H copy (into a page-locked slot, behind an event) and ``np.save``s it.  The device never waits for the decoder, the
    copy back or the file system; files and lists are those of the serial loop (``written`` lists a video once its file is
    saved).  ``progress(video)``: called by the device loop after a video is queued (measurement hook); ``stats`` (a dict, if
    given) receives ``max_write_queue`` = the deepest the writer's queue got.
    Files have the reference's shapes: ``[T, 4096]`` for vgg19, ``[T, 2048, 1, 1]`` for resnet152 (``children()[:-1]`` ends in the
    average pool, ``get_rgb_features.py:129-131``; ``model/data.py:166`` reshapes either)."""
    if model_type not in ("vgg19", "resnet152"):
        raise ValueError(f"unknown extractor {model_type!r} (get_rgb_features.py:122-131 has vgg19 and resnet152)")
    ft = Path(features_dir)
    ft.mkdir(exist_ok=True)
    prefix = f"{model_type}_ft_"
    done = [f.stem[len(prefix):] for f in sorted(ft.glob(f"{prefix}*.npy"))]                  # (:105-106)
    missed_path = Path(missed_path)
    missed = json.loads(missed_path.read_text()) if missed_path.exists() else []            # (:107-111)
    skip = set(done) | set(missed)
    todo = [item for item in dataset_info if item["video"] not in skip]                       # (:115-116)
    dev = _weights_device(weights, model_type)
    written = []
    stats = stats if stats is not None else {}
    stats["max_write_queue"] = 0
    if not (pipeline and dev is not None and dev.type == "cuda"):
        for item in todo:
            video, nseg = item["video"], item["num_segments"]
            frames, fps = decoder(video, nseg)
            if frames is None or len(frames) == 0:                                            # (:75-78,152-153)
                missed.append(video)
                continue
            feats = extract_video(frames, fps, nseg, weights, cfg, model_type)
            np.save(ft / f"{prefix}{video}", _file_shape(feats.cpu().numpy(), model_type))  # (:150-151)
            written.append(video)
        missed_path.write_text(json.dumps(missed))                                           # (:155-156)
        return written, missed

    import queue
    import threading
    decoded = queue.Queue(maxsize=2)            # reader -> main: (video, nseg, pinned uint8 frames | None, fps)
    to_write = queue.Queue(maxsize=3)           # main -> writer: (video, event, pinned float32 features)
    errors = []

    stop = threading.Event()

    def reader():
        try:
            for item in todo:
                if stop.is_set():
                    break
                video, nseg = item["video"], item["num_segments"]
                frames, fps = decoder(video, nseg)
                if frames is None or len(frames) == 0:
                    decoded.put((video, nseg, None, 0.0))
                    continue
                decoded.put((video, nseg, select_frames(frames, fps, nseg, pin=True), 0.0))    # kept frames only, page-locked
        except BaseException as e:              # noqa: BLE001 -- handed to the main thread
            errors.append(e)
        finally:
            decoded.put(None)

    def writer():
        try:
            while True:
                job = to_write.get()
                if job is None:
                    return
                video, event, host = job
                event.synchronize()
                np.save(ft / f"{prefix}{video}", _file_shape(host.numpy(), model_type))     # (:150-151)
                written.append(video)                                                        # only once the file exists
                stats["max_write_queue"] = max(stats["max_write_queue"], to_write.qsize())
        except BaseException as e:              # noqa: BLE001
            errors.append(e)
            while to_write.get() is not None:   # keep draining so the main thread never blocks on a full queue
                pass

    threads = [threading.Thread(target=reader, daemon=True), threading.Thread(target=writer, daemon=True)]
    for t in threads:
        t.start()
    try:
        # everything below is queued on the current stream OF THE WEIGHTS' DEVICE: the event that releases a slot to the writer
        # must be recorded on the stream that carries the D2H copy, whatever the caller's current device is
        with torch.cuda.device(dev):
            while True:
                job = decoded.get()
                if job is None:
                    break
                video, nseg, frames, fps = job
                if frames is None:                                                            # (:75-78,152-153)
                    missed.append(video)
                    continue
                feats = extract_video(frames.to(dev, non_blocking=True), fps, nseg, weights, cfg, model_type)
                host = torch.empty(feats.shape, dtype=feats.dtype, pin_memory=True)
                host.copy_(feats, non_blocking=True)
                event = torch.cuda.Event()
                event.record(torch.cuda.current_stream(dev))
                to_write.put((video, event, host))
                if progress is not None:
                    progress(video)
    finally:
        stop.set()
        while threads[0].is_alive():            # an exception above: let the reader finish its put, then stop
            try:
                decoded.get(timeout=0.05)
            except queue.Empty:
                pass
        to_write.put(None)
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    missed_path.write_text(json.dumps(missed))                                               # (:155-156)
    return written, missed


def _file_shape(feats: np.ndarray, model_type: str) -> np.ndarray:
    """The array as the reference writes it (get_rgb_features.py:147-151): resnet152 keeps the pooled map's two unit axes."""
    return feats.reshape(feats.shape[0], -1, 1, 1) if model_type == "resnet152" else feats


def _weights_device(weights, model_type):
    try:
        return weights[0][0].device
    except (TypeError, IndexError, AttributeError):
        return None
