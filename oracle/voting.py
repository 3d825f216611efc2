"""CPU restatement (numpy) of the reference's vote bookkeeping and scene metrics -- TEST INFRASTRUCTURE ONLY
(imported by tests/ only; the product is closerlook3d_amd/voting.py).

  collect()             pytorch/function/train_s3dis_dist.py:295-300 (array set-up) and :357-369 (the per-element
                        update inside `validate`), statement by statement
  iou_from_confusions   pytorch/utils/util.py:111-137
  s3dis_metrics         pytorch/utils/util.py:140-150
  sub_s3dis_metrics     pytorch/utils/util.py:153-165

Pinned: the three metric functions against values the reference's own utils/util.py produced
(tests/golden/voting_metrics.npz, tests/golden/make_voting_golden.py); collect() is inline code of a training
script and cannot be called, so it is restated here line for line.
"""
import numpy as np


def new_arrays(num_classes, cloud_sizes):
    sums = [np.zeros((num_classes, n), dtype=np.float32) for n in cloud_sizes]
    counts = [np.zeros((1, n), dtype=np.float32) + 1e-6 for n in cloud_sizes]
    votes = [np.zeros((num_classes, n), dtype=np.float32) for n in cloud_sizes]
    running = [np.zeros((num_classes, n), dtype=np.float32) for n in cloud_sizes]
    return sums, counts, votes, running


def collect(arrays, pred, mask, input_inds, cloud_label, test_smooth=0.95):
    sums, counts, votes, running = arrays
    for ib in range(pred.shape[0]):
        mask_i = mask[ib].astype(bool)
        logits = pred[ib][:, mask_i]
        inds = input_inds[ib][mask_i]
        c_i = int(cloud_label[ib])
        sums[c_i][:, inds] = sums[c_i][:, inds] + logits
        counts[c_i][:, inds] += 1
        votes[c_i] = sums[c_i] / counts[c_i]
        running[c_i][:, inds] = test_smooth * running[c_i][:, inds] + (1 - test_smooth) * logits


def confusion(targets, preds, num_classes):
    c = np.zeros((num_classes, num_classes), dtype=np.int64)
    for t, p in zip(np.asarray(targets).reshape(-1), np.asarray(preds).reshape(-1)):
        if 0 <= t < num_classes and 0 <= p < num_classes:
            c[t, p] += 1
    return c


def iou_from_confusions(confusions):
    tp = np.diagonal(confusions, axis1=-2, axis2=-1)
    tp_plus_fn = np.sum(confusions, axis=-1)
    tp_plus_fp = np.sum(confusions, axis=-2)
    iou = tp / (tp_plus_fp + tp_plus_fn - tp + 1e-6)
    mask = tp_plus_fn < 1e-3
    counts = np.sum(1 - mask, axis=-1, keepdims=True)
    miou = np.sum(iou, axis=-1, keepdims=True) / (counts + 1e-6)
    iou += mask * miou
    return iou


def s3dis_metrics(num_classes, vote_logits, validation_proj, validation_labels):
    confs = [confusion(t, np.argmax(l[:, p], axis=0), num_classes)
             for l, p, t in zip(vote_logits, validation_proj, validation_labels)]
    iou = iou_from_confusions(np.sum(np.stack(confs), axis=0))
    return iou, np.mean(iou)


def sub_s3dis_metrics(num_classes, validation_logits, validation_labels, val_proportions):
    confs = [confusion(t, np.argmax(l, axis=0), num_classes) for l, t in zip(validation_logits, validation_labels)]
    c = np.sum(np.stack(confs), axis=0).astype(np.float32)
    c *= np.expand_dims(val_proportions / (np.sum(c, axis=1) + 1e-6), 1)
    iou = iou_from_confusions(c)
    return iou, np.mean(iou)
