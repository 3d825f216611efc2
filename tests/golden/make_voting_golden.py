"""Golden values for closerlook3d_amd/voting.py's metric functions, produced by the REFERENCE's own
utils/util.py (build container only):  IoU_from_confusions, s3dis_metrics, sub_s3dis_metrics on seeded logits.

    python tests/golden/make_voting_golden.py        ->  tests/golden/voting_metrics.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/pytorch"
OUT = os.path.join(ROOT, "tests", "golden", "voting_metrics.npz")


def main():
    import sklearn.metrics as skm
    _cm = skm.confusion_matrix
    # the reference passes `labels` positionally (scikit-learn < 1.0 signature)
    skm.confusion_matrix = lambda y_true, y_pred, labels=None, **kw: _cm(y_true, y_pred, labels=labels, **kw)
    sys.path.insert(0, REF)
    import utils.util as ru
    rng = np.random.default_rng(77)
    C = 13
    sizes, full = [700, 1200, 310], [2500, 4100, 900]
    logits = [rng.normal(size=(C, n)).astype(np.float32) for n in sizes]
    logits[2][9:] -= 50.0          # scene 2 never predicts classes 9..12
    sub_labels = [rng.integers(0, C, size=n).astype(np.int32) for n in sizes]
    proj = [rng.integers(0, n, size=m).astype(np.int32) for n, m in zip(sizes, full)]
    labels = [rng.integers(0, 11, size=m).astype(np.int32) for m in full]   # classes 11, 12 absent from the truth
    prop = np.array([np.sum([np.sum(l == v) for l in labels]) for v in range(C)], dtype=np.float32)
    iou, miou = ru.s3dis_metrics(C, logits, proj, labels)
    siou, smiou = ru.sub_s3dis_metrics(C, logits, sub_labels, prop)
    conf = rng.integers(0, 50, size=(4, C, C)).astype(np.int32)
    conf[:, 5, :] = 0               # an absent class
    out = {"num_classes": C, "prop": prop, "iou": iou, "miou": miou, "sub_iou": siou, "sub_miou": smiou,
           "conf": conf, "conf_iou": ru.IoU_from_confusions(conf)}
    for i in range(3):
        out[f"logits{i}"], out[f"sub_labels{i}"], out[f"proj{i}"], out[f"labels{i}"] = logits[i], sub_labels[i], proj[i], labels[i]
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "mIoU", miou, "sub mIoU", smiou)


if __name__ == "__main__":
    main()
