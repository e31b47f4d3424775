"""MI355X-native cross-modal scoring hot path of video-fragments-retrieval.

Host side (Python on PyTorch-ROCm) mirrors the reference's operator surface -- ``models.CALModel``,
``evaluate.evaluate``, ``evaluate_single.evaluate``, ``data`` / ``utils`` helpers -- and calls the
hand-written gfx950 kernels in ``lib/libvfr.so`` through a flat C ABI (``include/vfr.h``).
Import as ``vfr_amd`` (see ``vfr_amd.py`` at the repo root).
"""
__version__ = "0.1.0"
