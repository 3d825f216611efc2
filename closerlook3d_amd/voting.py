"""Vote bookkeeping of scene-segmentation evaluation, kept on the device.

The reference's `validate` (`pytorch/function/train_s3dis_dist.py:295-300,357-369`) copies every batch element's
logits, mask and scene indices to the host and updates three numpy arrays per scene there -- and re-divides the
WHOLE scene's sums by its counts for every element.  Here the three arrays live in HBM, an element's update is
three indexed device operations with no host round trip, and the quotient is formed when it is asked for.

    votes = VoteAccumulator(num_classes, [len(l) for l in sub_clouds_points_labels], device="cuda")
    for points, mask, features, labels, cloud_label, input_inds in loader:        # any number of voting passes
        votes.update(model(points, mask, features), mask, input_inds, cloud_label)
    iou, miou = scene_metrics(votes, projections, clouds_points_labels)            # == s3dis_metrics(...)

Same arithmetic as the reference, element by element (float32 sums, counts starting at 1e-6, running mean with
`test_smooth`), so the voted logits are bit-identical to its numpy arrays; `input_inds` of one element are
distinct scene points where the mask is set (a sphere crop), which is what makes `a[:, inds] = a[:, inds] + x`
and an indexed add the same thing.
"""
import torch


class VoteAccumulator:
    def __init__(self, num_classes, cloud_sizes, test_smooth=0.95, device="cuda", running=None):
        """`running`: optional list of [num_classes, n_i] tensors carried over from earlier epochs (the
        reference keeps `runing_vote_logits` across validation calls, train_s3dis_dist.py:229-231)."""
        self.num_classes = int(num_classes)
        self.test_smooth = float(test_smooth)
        self.device = torch.device(device)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.sums = [torch.zeros((self.num_classes, int(n)), **f32) for n in cloud_sizes]
        self.counts = [torch.full((1, int(n)), 1e-6, **f32) for n in cloud_sizes]          # :297-298
        if running is None:
            self.running = [torch.zeros((self.num_classes, int(n)), **f32) for n in cloud_sizes]
        else:
            self.running = [r.to(**f32).clone() for r in running]

    def update(self, pred, mask, input_inds, cloud_label):
        """pred [B, num_classes, N] logits, mask [B, N] (0/1), input_inds [B, N] scene indices,
        cloud_label: B scene ids (host sequence or tensor).  Reference: train_s3dis_dist.py:357-369."""
        labels = cloud_label.tolist() if torch.is_tensor(cloud_label) else list(cloud_label)
        pred = pred.to(self.device, torch.float32)
        mask = mask.to(self.device)
        input_inds = input_inds.to(self.device, torch.long)
        s = self.test_smooth
        for ib, c in enumerate(labels):  # in order: two elements of a batch may overlap in the same scene
            c = int(c)
            keep = mask[ib].bool()
            inds = input_inds[ib][keep]
            logits = pred[ib][:, keep]
            self.sums[c].index_add_(1, inds, logits)
            self.counts[c].index_add_(1, inds, torch.ones((1, inds.numel()), dtype=torch.float32, device=self.device))
            run = self.running[c]
            run[:, inds] = s * run[:, inds] + (1 - s) * logits

    def vote_logits(self, cloud):
        """[num_classes, n] mean logits of a scene (points never voted on: 0 / 1e-6 = 0, as in the reference)."""
        return self.sums[cloud] / self.counts[cloud]

    def predictions(self, cloud, proj=None, running=False):
        """arg-max class per scene point; with `proj` (indices of the nearest sub-sampled point of every original
        point, datasets/S3DIS.py:262-270) per ORIGINAL point, as s3dis_metrics does (utils/util.py:143)."""
        logits = self.running[cloud] if running else self.vote_logits(cloud)
        if proj is not None:
            logits = logits[:, torch.as_tensor(proj, device=self.device).long()]
        return torch.argmax(logits, dim=0)


def confusion_matrix(targets, preds, num_classes):
    """[num_classes, num_classes] int64, rows = truth, columns = prediction (sklearn's convention, which the
    reference uses with labels = arange(num_classes): entries outside the label set are dropped)."""
    targets = targets.long().reshape(-1)
    preds = preds.long().reshape(-1)
    ok = (targets >= 0) & (targets < num_classes) & (preds >= 0) & (preds < num_classes)
    flat = targets[ok] * num_classes + preds[ok]
    return torch.bincount(flat, minlength=num_classes * num_classes).reshape(num_classes, num_classes)


def iou_from_confusions(conf):
    """Reference: utils/util.py:111-137 (absent classes take the mean IoU of the present ones)."""
    if not conf.is_floating_point():   # integer counts divide in double; a float32 (rescaled) matrix stays float32
        conf = conf.to(torch.float64)
    tp = torch.diagonal(conf, dim1=-2, dim2=-1)
    tp_fn = conf.sum(-1)
    tp_fp = conf.sum(-2)
    iou = tp / (tp_fp + tp_fn - tp + 1e-6)
    absent = tp_fn < 1e-3
    present = (~absent).sum(-1, keepdim=True)
    miou = iou.sum(-1, keepdim=True).to(torch.float64) / (present.to(torch.float64) + 1e-6)
    return iou + (absent * miou).to(iou.dtype)


def scene_metrics(votes, projections, clouds_points_labels, running=False):
    """(IoU per class, mIoU) over the original points of all scenes.  Reference: s3dis_metrics, utils/util.py:140-150."""
    total = torch.zeros((votes.num_classes, votes.num_classes), dtype=torch.int64, device=votes.device)
    for c, (proj, labels) in enumerate(zip(projections, clouds_points_labels)):
        preds = votes.predictions(c, proj, running=running)
        total += confusion_matrix(torch.as_tensor(labels, device=votes.device), preds, votes.num_classes)
    iou = iou_from_confusions(total)
    return iou, iou.mean()


def sub_scene_metrics(votes, sub_clouds_points_labels, val_proportions, running=False):
    """The same on the sub-sampled scenes, confusion rows rescaled to the class proportions of the full
    validation set.  Reference: sub_s3dis_metrics, utils/util.py:153-165."""
    total = torch.zeros((votes.num_classes, votes.num_classes), dtype=torch.int64, device=votes.device)
    for c, labels in enumerate(sub_clouds_points_labels):
        preds = votes.predictions(c, running=running)
        total += confusion_matrix(torch.as_tensor(labels, device=votes.device), preds, votes.num_classes)
    conf = total.to(torch.float32)
    prop = torch.as_tensor(val_proportions, dtype=torch.float32, device=votes.device)
    conf = conf * (prop / (conf.sum(1) + 1e-6)).unsqueeze(1)
    iou = iou_from_confusions(conf)
    return iou, iou.mean()
