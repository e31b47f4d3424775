"""The sibling-import shims (``video-fragments-retrieval_amd/dropin``): with that directory first on ``sys.path`` the
reference's ``import data, models, utils, evaluate, evaluate_single`` (``model/main.py:11-14``, ``model/evaluate.py:8-10``)
resolve to this package.  Checked in a fresh interpreter: every name the reference's evaluation / validation path touches
(``model/main.py:121-212,268-356``, ``model/evaluate.py:28-90,173-185``) exists and works; the training samplers
(``data.CustomBatchSampler`` / ``data.custom_collate``, ``model/main.py:313-318``) are out of scope and absent -- INTEGRATION.md
says so."""
import subprocess
import sys
import textwrap
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
DROPIN = ROOT / "video-fragments-retrieval_amd" / "dropin"

SCRIPT = textwrap.dedent("""
    import sys
    sys.path.insert(0, sys.argv[1])
    sys.path.insert(1, sys.argv[2])                     # tests/ (helpers)
    import data, models, utils, evaluate, evaluate_single
    import torch
    from torch.utils.data import DataLoader

    # module-level names of the evaluate / validate path
    for mod, names in ((data, "FEATURE_DIM EMBEDDING_DIM SELECT_FPS FRAMES_PER_SEC SEC_PER_SEGMENT POOLING WordIndexer "
                              "CustomDataset VideoBatchSampler LanguageBatchSampler validate_collate"),
                       (models, "CALModel init_weights"),
                       (utils, "generate_moments get_iou load_dataset_info load_missed_videos start_new_experiment str2bool grad_norm"),
                       (evaluate, "evaluate get_metrics validate_epoch"),
                       (evaluate_single, "evaluate get_metrics")):
        for n in names.split():
            assert hasattr(mod, n), (mod.__name__, n)
    assert data.FEATURE_DIM["vgg19"] == 4096 and data.EMBEDDING_DIM == 100
    assert utils.str2bool("True") is True and utils.str2bool("junk") is None
    # out of scope (training-only sampling, SURVEY 2): absent, not half-implemented
    assert not hasattr(data, "CustomBatchSampler") and not hasattr(data, "custom_collate")

    # the calls main.py:345-356 / evaluate.py:173-185 make, through the shim modules, on the CPU device
    from helpers import MemoryDataset, problem
    p = problem(12, 9, "didemo", feat_dim=64, hidden=16, seed=3)
    emb = torch.from_numpy(p["sd"]["word_embedding.weight"])
    model = models.CALModel(visual_input_dim=64 * 2 + 2, pretrained_emb=emb, emb_dim=data.EMBEDDING_DIM, hidden_size=16,
                            dropout_rate=0.3, normalize_lang=False)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in p["sd"].items()})
    model.to("cpu").eval()
    ds = MemoryDataset(p["seg"], p["ctx"], p["counts"], p["tokens"], p["own"], p["times"])
    vi = DataLoader(ds, shuffle=False, collate_fn=data.validate_collate,
                    batch_sampler=data.VideoBatchSampler(ds.videos, ds.num_segments_info))
    li = DataLoader(ds, shuffle=False, collate_fn=data.validate_collate,
                    batch_sampler=data.LanguageBatchSampler(ds.annotations, ds.num_segments_info))
    assert set(li.batch_sampler.moments) >= set(range(7))
    out = evaluate.evaluate(model, vi, li, ds.annotations, "cpu")
    assert set(out) == {"model, IoU=0.5", "model, IoU=0.7"} and set(out["model, IoU=0.5"]) == {"R@1", "R@10", "R@100", "MR"}
    val = evaluate.validate_epoch(model, vi, li, ds.annotations, "cpu", size=5)
    assert set(val) == {"CustomRecall", "MedianRank", "MeanReciprocalRank", "pr_curve"}
    prior = {n: utils.generate_moments(n) for n in (5, 6)}
    single = evaluate_single.evaluate(model, vi, li, ds.annotations, "cpu", model_types=["model", "prior"], prior=prior)
    assert set(single["model"]) == {"Rank@1", "Rank@5", "Rank@10", "mIoU"}
    # the training-mode forward main.py:58-61 makes stays differentiable
    model.train()
    x = torch.randn(3, 130, requires_grad=True)
    model(x).sum().backward()
    assert x.grad is not None and utils.grad_norm(model) > 0
    print("DROPIN-OK")
""")


def test_dropin_modules_resolve_and_drive_the_evaluators():
    r = subprocess.run([sys.executable, "-c", SCRIPT, str(DROPIN), str(ROOT / "tests")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
