"""Host helpers with the reference's ``utils`` surface (``model/utils.py``).

On the hot path: ``generate_moments`` (defines the moment index space the kernels emit, ``:71-75``) and
``get_iou`` (ground-truth labelling, ``:78-82``).  The filesystem helpers keep ``main.py`` working as a
caller; they are plain Python and not accelerated.
"""
from __future__ import annotations

import itertools
import json
import re
from pathlib import Path

import numpy as np


def generate_moments(num_segments: int):
    """All contiguous clip spans of a video with ``num_segments`` clips, in the reference's order:
    the ``n`` single clips first, then every pair ``(s, e)``, ``s < e`` lexicographically.  The position
    in this list is the local moment id used by every kernel (``csrc/vfr_math.h: moment_index``)."""
    singles = [(t, t) for t in range(num_segments)]
    return singles + list(itertools.combinations(range(num_segments), 2))


def moment_index(num_segments: int, start_t: int, end_t: int) -> int:
    """Closed form of ``generate_moments(n).index((s, e))``."""
    if start_t == end_t:
        return start_t
    n, s, e = num_segments, start_t, end_t
    return n + (s * (2 * n - s - 1)) // 2 + (e - s - 1)


def get_iou(times, start_t, end_t):
    """IoU of the inclusive clip span ``[start_t, end_t]`` with each annotated span in ``times``."""
    spans = np.array(times)
    lo, hi = spans[:, 0], spans[:, 1]
    overlap = np.maximum(np.minimum(hi, end_t) + 1 - np.maximum(lo, start_t), 0)
    hull = np.maximum(hi, end_t) + 1 - np.minimum(lo, start_t)
    return overlap / hull


def read_json(json_file):
    with open(json_file) as fh:
        return json.load(fh)


def load_missed_videos(missed_videos_path):
    path = Path(missed_videos_path) / "missed_videos_features.json"
    return read_json(path) if path.exists() else []


def load_dataset_info(dataset_type, dataset_directory, missed_videos_path):
    """-> (annotations dict keyed by annotation_id, list of unique videos).

    The reference returns ``list(set(videos))`` whose order depends on the hash seed (Q11); global moment
    ids follow the video order, so this keeps first-appearance order to make runs reproducible."""
    skip = set(load_missed_videos(missed_videos_path))
    annotations, videos = {}, {}
    for row in read_json(Path(dataset_directory) / f"{dataset_type}_data.json"):
        if row["video"] in skip:
            continue
        annotations[row["annotation_id"]] = dict(video=row["video"], description=row["description"], times=row["times"])
        videos.setdefault(row["video"], None)
    return annotations, list(videos)


def get_existing_experiments(exper_dir):
    return sorted(int(d.name) for d in Path(exper_dir).iterdir() if d.is_dir() and re.fullmatch(r"\d+", d.name))


def start_new_experiment(exper_dir):
    root = Path(exper_dir)
    if root.is_dir():
        name = str(get_existing_experiments(root)[-1] + 1)
    else:
        root.mkdir(exist_ok=False)
        name = "0"
    new_dir = root / name
    new_dir.mkdir(exist_ok=False)
    return new_dir


def str2bool(param):
    text = str(param).lower()
    return True if text == "true" else False if text == "false" else None


def grad_norm(model):
    norms = [p.grad.data.norm().cpu().item() for _, p in model.named_parameters() if p.grad is not None]
    return sum(norms) / len(norms)
