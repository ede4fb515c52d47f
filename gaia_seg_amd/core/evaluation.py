"""Cross-architecture evaluation — host-side mirror of
gaiaseg/core/evaluation/cross_arch_eval_hooks.py:24-167.

For every anchor of ``val_sampler.traverse()``: broadcast the meta, ``manipulate_arch``, run the
model over the validation loader in test mode and report mIoU / mAcc / aAcc.  Differences by
design: predictions never leave the device — each batch is folded into a CxC confusion matrix by a
HIP kernel (``gs_confusion_matrix``) and the matrices (not pickled per-image results as in
gaiaseg/apis/test.py:119-173) are summed across ranks with one all-reduce.
"""
import torch

from ..hip import lib as _lib
from ..hip.runtime import current_stream_ptr
from . import dist as gdist
from .dynamic import fold_dict
from .runner import Hook


def confusion_matrix(pred, label, num_classes, ignore_index=255, out=None):
    """out[l, p] += #{pixels with label l predicted p}; int64 device tensors of equal numel."""
    L = _lib.load()
    pred = pred.contiguous().long()
    label = label.contiguous().long()
    assert pred.numel() == label.numel()
    if out is None:
        out = torch.zeros((num_classes, num_classes), dtype=torch.int64, device=pred.device)
    _lib.check(L.gs_confusion_matrix(pred.data_ptr(), label.data_ptr(), pred.numel(), num_classes,
                                     -1 if ignore_index is None else int(ignore_index),
                                     out.data_ptr(), current_stream_ptr()), "gs_confusion_matrix")
    return out


def metrics_from_confusion(conf):
    """mmseg's mIoU metric family from a confusion matrix (rows = label, cols = prediction)."""
    conf = conf.double()
    tp = conf.diag()
    per_label = conf.sum(1)
    per_pred = conf.sum(0)
    union = per_label + per_pred - tp
    iou = tp / union
    acc = tp / per_label
    return dict(aAcc=float(tp.sum() / conf.sum().clamp_min(1)),
                mIoU=float(iou[~torch.isnan(iou)].mean()) if bool((~torch.isnan(iou)).any()) else 0.0,
                mAcc=float(acc[~torch.isnan(acc)].mean()) if bool((~torch.isnan(acc)).any()) else 0.0,
                IoU=iou.tolist(), Acc=acc.tolist())


def evaluate_model(model, loader, num_batches, num_classes, ignore_index=255):
    """Test-mode pass over ``num_batches`` batches of dict(img, img_metas, gt_semantic_seg)."""
    was_training = model.training
    model.eval()
    conf = None
    it = iter(loader)
    with torch.no_grad():
        for _ in range(num_batches):
            batch = next(it)
            img, metas, gt = batch["img"], batch["img_metas"], batch["gt_semantic_seg"]
            preds = model.simple_test_device(img, metas)
            conf = confusion_matrix(preds, gt, num_classes, ignore_index, conf)
    if gdist.is_dist():
        import torch.distributed as dist
        dist.all_reduce(conf)
    model.train(was_training)
    return metrics_from_confusion(conf)


class CrossArchEvalHook(Hook):
    """Every ``interval`` iterations evaluate every val anchor (cross_arch_eval_hooks.py:59-92)."""

    def __init__(self, dataloader, model_sampler, interval=1, num_batches=4, num_classes=19,
                 ignore_index=255, logger=None):
        self.dataloader, self.sampler = dataloader, model_sampler
        self.interval, self.num_batches = interval, num_batches
        self.num_classes, self.ignore_index = num_classes, ignore_index
        self.logger = logger
        self.results = []

    def after_train_iter(self, runner):
        if not self.every_n_iters(runner, self.interval):
            return
        self.evaluate(runner)

    def evaluate(self, runner):
        self.sampler.set_mode("traverse")
        metas = self.sampler.traverse()
        out = {}
        # the training subnet must survive the evaluation: with manipulate_arch=False (or a fixed
        # arch) nobody re-applies it before the next iteration
        backbone = runner.model.backbone
        saved = dict(key=runner.arch_key, name=runner.arch_name, meta=runner.arch_meta,
                     arch={"backbone": {k: v for k, v in backbone.state_dict_of_arch().items()
                                        if v is not None}})
        for i, meta in enumerate(metas):
            meta = gdist.broadcast_object(meta, src=0)   # :59 broadcast_object(fold_dict(meta))
            runner.model.manipulate_arch(fold_dict(meta)["arch"])
            res = evaluate_model(runner.model, self.dataloader, self.num_batches,
                                 self.num_classes, self.ignore_index)
            name = meta.get("name", str(i))
            out[name] = res
            msg = "eval %s: mIoU %.4f mAcc %.4f aAcc %.4f" % (name, res["mIoU"], res["mAcc"], res["aAcc"])
            if gdist.rank() == 0:
                (self.logger.info if self.logger else print)(msg)
        self.sampler.set_mode("sample")
        self.results.append((runner.iter + 1, out))
        runner.model.manipulate_arch(saved["arch"])
        runner.arch_key, runner.arch_name, runner.arch_meta = saved["key"], saved["name"], saved["meta"]
        runner.refresh_active()
        return out


DistCrossArchEvalHook = CrossArchEvalHook  # the matrices are all-reduced; one class serves both
