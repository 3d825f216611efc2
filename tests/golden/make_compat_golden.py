"""Golden data for closerlook3d_amd/compat.py, produced by the REFERENCE's own code (build container only):

  reference_configs.json   utils/config.py's defaults and, for each of the 20 cfgs/**/*.yaml, the merged
                           configuration `update_config` produces (keys and values, as JSON)
  state_dict_models.json   parameter/buffer names and shapes of models/build.py's three model wrappers built
                           from cfgs/modelnet/pospool_xyz_avg.yaml, cfgs/partnet/pseudo_grid.yaml and
                           cfgs/s3dis/pointwisemlp_dp_fi_df_fc1.yaml (what a released checkpoint's 'model'
                           entry contains)

    python tests/golden/make_compat_golden.py
"""
import copy
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/pytorch"
OUT = os.path.join(ROOT, "tests", "golden")

import make_operator_golden as mog  # noqa: E402  (stubs for easydict and pt_custom_ops._ext)


def main():
    mog._install_stubs()
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "ops", "pt_custom_ops"))
    import yaml
    _load = yaml.load
    yaml.load = lambda f, Loader=None: _load(f, Loader=yaml.SafeLoader)  # the reference calls yaml.load(f) (PyYAML < 6 signature)
    import utils.config as rc
    defaults = copy.deepcopy(rc.config)
    merged = {}
    for path in sorted(glob.glob(os.path.join(REF, "cfgs", "*", "*.yaml"))):
        rc.config.clear()
        rc.config.update(copy.deepcopy(defaults))
        rc.update_config(path)
        merged[os.path.relpath(path, os.path.join(REF, "cfgs"))] = json.loads(json.dumps(rc.config))
    with open(os.path.join(OUT, "reference_configs.json"), "w") as fh:
        json.dump({"defaults": json.loads(json.dumps(defaults)), "merged": merged}, fh, indent=0, sort_keys=True)

    # PseudoGrid's kernel-point generator asks torch.distributed for the rank and caches under JOB_LOG_DIR
    import tempfile
    import torch.distributed as dist
    os.environ["JOB_LOG_DIR"] = tempfile.mkdtemp(prefix="cl3d_golden_")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    from models import build as rb
    shapes = {}
    for rel, fn in (("modelnet/pospool_xyz_avg.yaml", rb.build_classification),
                    ("partnet/pseudo_grid.yaml", rb.build_multi_part_segmentation),
                    ("s3dis/pointwisemlp_dp_fi_df_fc1.yaml", rb.build_scene_segmentation)):
        rc.config.clear()
        rc.config.update(copy.deepcopy(defaults))
        rc.update_config(os.path.join(REF, "cfgs", rel))
        if rel.startswith("partnet"):
            rc.config.num_parts = [4, 2, 6]   # set by the dataset at run time (train_partnet_dist.py)
            rc.config.num_classes = 3
        model, _ = fn(rc.config)
        shapes[rel] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(OUT, "state_dict_models.json"), "w") as fh:
        json.dump(shapes, fh, indent=0, sort_keys=True)
    print({k: len(v) for k, v in shapes.items()}, len(merged))


if __name__ == "__main__":
    main()
