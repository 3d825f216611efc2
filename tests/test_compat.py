"""CPU: configuration / checkpoint compatibility layer (closerlook3d_amd/compat.py) against golden data produced
by the reference's own `utils/config.py` and `models/build.py` (tests/golden/make_compat_golden.py)."""
import glob
import json
import os

import sys
import types

import pytest
import torch

from closerlook3d_amd import compat

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REF_CFGS = "/root/reference/pytorch/cfgs"


def _golden(name):
    with open(os.path.join(GOLDEN, name)) as fh:
        return json.load(fh)


def test_default_config_equals_reference():
    assert json.loads(json.dumps(compat.default_config())) == _golden("reference_configs.json")["defaults"]


@pytest.mark.skipif(not os.path.isdir(REF_CFGS), reason="the reference's YAML files are only in the build container")
def test_all_reference_yamls_load_like_the_reference():
    want = _golden("reference_configs.json")["merged"]
    paths = sorted(glob.glob(os.path.join(REF_CFGS, "*", "*.yaml")))
    assert len(paths) == len(want) == 20
    for p in paths:
        got = json.loads(json.dumps(compat.load_config(p)))
        assert got == want[os.path.relpath(p, REF_CFGS)], p


def test_unknown_key_is_rejected(tmp_path):
    f = tmp_path / "bad.yaml"
    f.write_text("not_an_option: 1\n")
    with pytest.raises(ValueError, match="must exist"):
        compat.load_config(str(f))
    f.write_text("pospool:\n  reduction: 'avg'\nwidth: 72\n")
    cfg = compat.load_config(str(f))
    assert cfg.pospool.reduction == "avg" and cfg.pospool.position_embedding == "xyz" and cfg.width == 72


def _cfg_from_golden(rel):
    cfg = compat.Config(_golden("reference_configs.json")["merged"][rel])
    if rel.startswith("partnet"):
        cfg.num_parts, cfg.num_classes = [4, 2, 6], 3
    return cfg


@pytest.mark.parametrize("rel", ["modelnet/pospool_xyz_avg.yaml", "partnet/pseudo_grid.yaml",
                                 "s3dis/pointwisemlp_dp_fi_df_fc1.yaml"])
def test_model_state_dict_matches_reference(rel):
    want = _golden("state_dict_models.json")[rel]
    model = compat.build_model(_cfg_from_golden(rel))
    got = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert got == want


def test_reference_style_checkpoint_round_trip(tmp_path):
    cfg = _cfg_from_golden("modelnet/pospool_xyz_avg.yaml")
    cfg.width, cfg.nsamples, cfg.npoints = 12, [8] * 5, [64, 32, 16, 8]
    torch.manual_seed(0)
    src = compat.build_model(cfg)
    src.init_weights()
    ckpt = {"config": dict(cfg), "model": {"module." + k: v for k, v in src.state_dict().items()},  # as saved under DDP
            "optimizer": {}, "scheduler": {}, "epoch": 17, "best_acc": 0.5}
    path = str(tmp_path / "current.pth")
    torch.save(ckpt, path)
    dst = compat.build_model(cfg)
    meta = compat.load_reference_checkpoint(dst, path)
    assert meta["epoch"] == 17 and "model" not in meta
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
    bad = dict(ckpt, model={k: v for k, v in list(ckpt["model"].items())[:-1]})
    torch.save(bad, path)
    with pytest.raises(RuntimeError, match="Missing key"):
        compat.load_reference_checkpoint(compat.build_model(cfg), path)


def test_checkpoint_with_pickled_easydict_config_loads_without_easydict(tmp_path):
    """The reference pickles its `easydict.EasyDict` config into every checkpoint; loading must not need the package."""
    if "easydict" in sys.modules:
        pytest.skip("easydict is installed here")
    cfg = _cfg_from_golden("modelnet/pospool_xyz_avg.yaml")
    cfg.width, cfg.nsamples, cfg.npoints = 12, [8] * 5, [64, 32, 16, 8]
    src = compat.build_model(cfg)
    fake = types.ModuleType("easydict")
    fake.EasyDict = type("EasyDict", (dict,), {"__module__": "easydict"})
    sys.modules["easydict"] = fake
    try:
        path = str(tmp_path / "best.pth")
        torch.save({"config": fake.EasyDict(width=12), "model": src.state_dict(), "epoch": 3, "best_acc": 0.1}, path)
    finally:
        del sys.modules["easydict"]
    meta = compat.load_reference_checkpoint(compat.build_model(cfg), path)
    assert meta["epoch"] == 3 and dict(meta["config"]) == {"width": 12}
    assert "easydict" not in sys.modules


def test_untrusted_checkpoint_cannot_run_code(tmp_path):
    """a checkpoint is read through a restricted unpickler; a pickled callable is refused unless the caller
    says the file is trusted."""
    import os

    class Evil:
        def __reduce__(self):
            return (os.system, ("echo pwned > " + str(tmp_path / "pwned"),))

    cfg = _cfg_from_golden("modelnet/pospool_xyz_avg.yaml")
    cfg.width, cfg.nsamples, cfg.npoints = 12, [8] * 5, [64, 32, 16, 8]
    model = compat.build_model(cfg)
    path = str(tmp_path / "evil.pth")
    torch.save({"model": model.state_dict(), "scheduler": Evil()}, path)
    with pytest.raises(Exception, match="trusted=True"):
        compat.load_reference_checkpoint(model, path)
    assert not (tmp_path / "pwned").exists()


def test_checkpoint_with_real_optimizer_and_step_scheduler_state_loads(tmp_path):
    """the reference saves optimizer.state_dict() and scheduler.state_dict() into every checkpoint
    (function/train_modelnet_dist.py:157-164).  Its default schedule is MultiStepLR behind a GradualWarmupScheduler
    whose state nests the inner scheduler's under 'after_scheduler' (utils/lr_scheduler.py:41-50); MultiStepLR keeps its
    milestones in a collections.Counter.  Such a file must load through the restricted unpickler."""
    cfg = _cfg_from_golden("modelnet/pospool_xyz_avg.yaml")
    cfg.width, cfg.nsamples, cfg.npoints = 12, [8] * 5, [64, 32, 16, 8]
    torch.manual_seed(0)
    src = compat.build_model(cfg)
    opt = torch.optim.SGD(src.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    for p in src.parameters():  # one step so that the momentum buffers exist
        p.grad = torch.zeros_like(p)
    opt.step()
    inner = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[30, 60, 60], gamma=0.1)
    # the warm-up wrapper's state_dict, shaped as the reference builds it: its own fields + the inner scheduler's state
    sched_state = {"multiplier": 100.0, "warmup_epoch": 5, "base_lrs": [0.01], "last_epoch": 3, "_step_count": 4,
                   "after_scheduler": inner.state_dict()}
    assert type(sched_state["after_scheduler"]["milestones"]).__name__ == "Counter"
    path = str(tmp_path / "current.pth")
    torch.save({"config": dict(cfg), "model": {"module." + k: v for k, v in src.state_dict().items()},
                "optimizer": opt.state_dict(), "scheduler": sched_state, "epoch": 4, "best_acc": 0.25}, path)
    dst = compat.build_model(cfg)
    meta = compat.load_reference_checkpoint(dst, path)  # trusted=False: the restricted unpickler
    assert meta["scheduler"]["after_scheduler"]["milestones"] == inner.state_dict()["milestones"]
    assert meta["epoch"] == 4 and len(meta["optimizer"]["state"]) == len(list(src.parameters()))
    for (ka, a), (kb, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert ka == kb and torch.equal(a, b)
