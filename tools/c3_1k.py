#!/usr/bin/env python3
"""BASELINE.json configs[2] literally: the extractor loop of get_rgb_features.py:134-151 over 1 000 videos.

    python tools/c3_1k.py [videos=1000] [model=vgg19|resnet152|both] > profiles/r4_c3_1k.txt

`features.extract_dataset` (reader thread -> pinned H2D -> frame selection + the whole network -> D2H -> np.save, pipelined)
over `videos` synthetic videos of 900 decoded 224x224x3 frames at 30 fps (-> 150 kept frames each).  The decoder is a seeded
frame store in host memory (decode speed is the codec's, not ours): 8 distinct clips, handed out round-robin.  Reports
frames/s over the WHOLE run and per 100-video window (sustained clocks: the last window must look like the first), the
deepest the writer's queue got (3 = its capacity: the file system is the bottleneck), and the kernel-only rate measured in
the same process for comparison.  Full-width random weights (the pretrained ones are a network fetch)."""
import sys
import tempfile
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from vfr_amd import _vfr, features, synth  # noqa: E402

DEV = "cuda:0"


def vgg_weights():
    g = torch.Generator(device=DEV); g.manual_seed(7)
    cw, cb, cin = [], [], 3
    for c in features.VGG19_CFG:
        if c == "M":
            continue
        cw.append(torch.randn((c, cin, 3, 3), device=DEV, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        cb.append(torch.zeros(c, device=DEV))
        cin = c
    fc6 = (torch.randn((4096, 512 * 49), device=DEV, generator=g) * 0.01, torch.zeros(4096, device=DEV))
    fc7 = (torch.randn((4096, 4096), device=DEV, generator=g) * 0.01, torch.zeros(4096, device=DEV))
    return cw, cb, fc6, fc7


def run(model_type, nvid, clips):
    weights = vgg_weights() if model_type == "vgg19" else _vfr.resnet_pack(synth.resnet_weights(seed=3), device=DEV)
    gflop_frame = 39.26 if model_type == "vgg19" else 23.1
    # kernel-only rate (one resident 150-frame video, 3 passes)
    sel = torch.from_numpy(clips[0][::6].copy()).to(DEV)
    for _ in range(2):
        features.extract_video(sel, 0.0, 6, weights, model_type=model_type)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        features.extract_video(sel, 0.0, 6, weights, model_type=model_type)
    torch.cuda.synchronize()
    k_only = 150 * 3 / (time.perf_counter() - t0)
    info = [dict(video=f"v{i:05d}", num_segments=6) for i in range(nvid)]
    decoder = lambda video, nseg: (clips[int(video[1:]) % len(clips)], 30.0)
    stamps, stats = [], {}
    with tempfile.TemporaryDirectory() as td:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        written, missed = features.extract_dataset(info, decoder, Path(td) / f"features_{model_type}", weights, model_type=model_type,
                                                   missed_path=Path(td) / "missed.json", progress=lambda v: stamps.append(time.perf_counter()),
                                                   stats=stats)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        files = sorted((Path(td) / f"features_{model_type}").glob("*.npy"))
        shape = np.load(files[0], mmap_mode="r").shape
        nbytes = sum(f.stat().st_size for f in files)
    frames = 150 * len(written)
    print(f"== {model_type}: {len(written)} videos written ({len(missed)} missed), {frames} frames, files {shape} x {len(files)} = {nbytes / 1e9:.2f} GB")
    print(f"whole run: {dt:.2f} s -> {frames / dt:.0f} frames/s = {len(written) / dt:.2f} videos/s = {frames / dt * gflop_frame / 1e3:.1f} TFLOP/s "
          f"({frames / dt / k_only * 100:.1f} % of the kernel-only rate {k_only:.0f} frames/s measured in this process)")
    print(f"writer queue: deepest {stats.get('max_write_queue')} of 3 slots")
    print("per 100-video window (device loop's queue times; the queues hold at most 2 + 3 videos):")
    prev = t0
    for w in range(0, len(stamps), 100):
        end = stamps[min(w + 100, len(stamps)) - 1]
        n = min(w + 100, len(stamps)) - w
        print(f"  videos {w:4d}-{w + n - 1:4d}: {end - prev:6.2f} s  {150 * n / (end - prev):7.0f} frames/s")
        prev = end


def main():
    nvid = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    base = synth.frames_u8(150, 224, 224, seed=11)
    clips = []
    for i in range(8):                                       # 8 distinct 900-frame clips (a permutation of the frames, each shown 6 times)
        perm = np.random.RandomState(100 + i).permutation(150)
        clips.append(np.ascontiguousarray(base[perm][np.repeat(np.arange(150), 6)]))
    print(f"# tools/c3_1k.py {nvid} {which}: BASELINE.json configs[2] (get_rgb_features.py:134-151), {torch.cuda.get_device_name(0)}, "
          f"8 distinct synthetic clips of 900 x 224 x 224 x 3 uint8 frames at 30 fps")
    for m in (("vgg19", "resnet152") if which == "both" else (which,)):
        run(m, nvid, clips)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
