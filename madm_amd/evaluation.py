"""Semantic-segmentation evaluator on the device (eval part):
``DSECSemSegEvaluator.process / evaluate`` of /root/reference/evaluation/d2_evaluator.py:99-127,240-275.

``process`` keeps the prediction on the GPU: argmax (first maximal class) -> confusion matrix
``conf[(K+1) * pred + gt]`` with 64-bit integer atomics -- bit-exact with the reference's
``np.bincount`` -- and ``evaluate`` derives mIoU / fwIoU / mACC / pACC from the (K+1)^2 matrix in float64 on the
host exactly as the reference does (which uses the removed ``np.float`` alias; float64 here).  Unlike the
reference, whose cross-rank gather is commented out (:228-238), ``evaluate(dist=...)`` can sum the matrices of
all ranks with one all-reduce of (K+1)^2 int64 values."""
import numpy as np
import torch

from . import ops


class SemSegEvaluator:
    def __init__(self, num_classes, class_names=None, ignore_label=255, convert_pred_list=None):
        self._num_classes = num_classes
        self._class_names = list(class_names) if class_names is not None else [str(i) for i in range(num_classes)]
        self._ignore_label = ignore_label
        self.convert_pred_list = convert_pred_list
        self._conf = None

    def reset(self):
        self._conf = None

    def process(self, inputs, outputs):
        for data, output in zip(inputs, outputs):
            sem = output["sem_seg"]
            pred = ops.argmax_nchw(sem.contiguous())[0]                     # output["sem_seg"][0].argmax(dim=0)
            if self.convert_pred_list is not None:
                raise NotImplementedError("convert_pred_list (d2_evaluator.py:108-112) is not used by the shipped configs")
            gt = data["target_label"]
            if gt.dim() == 3 and gt.shape[0] == 1:
                gt = gt[0]
            gt = gt.to(pred.device).long()
            if self._conf is None:
                self._conf = torch.zeros((self._num_classes + 1, self._num_classes + 1), dtype=torch.int64, device=pred.device)
                torch.cuda.current_stream(pred.device).synchronize()   # later calls may come on other streams (once per reset)
            ops.confusion_matrix(pred.reshape(-1), gt.reshape(-1), self._num_classes, self._ignore_label, self._conf)

    def confusion(self, dist=None):
        conf = self._conf.clone()
        if dist is not None:
            dist.all_reduce(conf)     # SUM over ranks: every rank scored its shard of the dataset
        return conf.cpu().numpy()

    def evaluate(self, dist=None):
        conf = self.confusion(dist)
        K = self._num_classes
        acc = np.full(K, np.nan, dtype=np.float64)
        iou = np.full(K, np.nan, dtype=np.float64)
        tp = conf.diagonal()[:-1].astype(np.float64)
        pos_gt = np.sum(conf[:-1, :-1], axis=0).astype(np.float64)
        class_weights = pos_gt / np.sum(pos_gt)
        pos_pred = np.sum(conf[:-1, :-1], axis=1).astype(np.float64)
        acc_valid = pos_gt > 0
        acc[acc_valid] = tp[acc_valid] / pos_gt[acc_valid]
        iou_valid = (pos_gt + pos_pred) > 0
        union = pos_gt + pos_pred - tp
        iou[acc_valid] = tp[acc_valid] / union[acc_valid]
        res = {"mIoU": 100 * np.sum(iou[acc_valid]) / np.sum(iou_valid),
               "fwIoU": 100 * np.sum(iou[acc_valid] * class_weights[acc_valid])}
        for i, name in enumerate(self._class_names):
            res[f"IoU-{name}"] = 100 * iou[i]
        res["mACC"] = 100 * np.sum(acc[acc_valid]) / np.sum(acc_valid)
        res["pACC"] = 100 * np.sum(tp) / np.sum(pos_gt)
        for i, name in enumerate(self._class_names):
            res[f"ACC-{name}"] = 100 * acc[i]
        return {"sem_seg": res}


def inference_on_dataset(model, data_loader, evaluator, streams=4, range_check=None, runner="graphed"):
    """``inference_on_dataset`` of /root/reference/evaluation/evaluator.py:30-139 (the loop at :75-93) on the throughput
    launch path: every ``inputs`` of the loader goes through ``pipeline.GraphedInference.submit`` (whole-forward hipGraphs,
    ``streams`` images in flight) and ``evaluator.process`` is enqueued on the slot's stream right behind the forward --
    argmax and confusion-matrix kernels, no host sync per image where the reference calls ``torch.cuda.synchronize()`` (:87).
    One runner per image shape (the graphs are captured for a shape; DSEC / DELIVER / FMB test images have one size each).
    The extractor's input-range assert is deferred (pipeline.DeferredRangeCheck): it raises from a later iteration or at the
    end.  Returns ``evaluator.evaluate()`` ({} when it returns None, as the reference does)."""
    from .pipeline import GraphedInference, StagedInference
    assert runner in ("graphed", "staged")   # whole-forward graphs on `streams` streams | encoder / UNet / decoder + head stage graphs
    evaluator.reset()
    runners = {}
    was_training = bool(getattr(model, "training", False))
    if hasattr(model, "eval"):
        model.eval()        # inference_context(model) of the reference (evaluator.py:142-155): eval mode, restored afterwards
    try:
        with torch.no_grad():
            for idx, inputs in enumerate(data_loader):
                shape = tuple(inputs[0]['target_second_modality'].shape)
                rn = runners.get(shape)
                if rn is None:
                    rn = runners[shape] = (StagedInference(model, inputs, range_check=range_check) if runner == "staged" else
                                           GraphedInference(model, inputs, streams=streams, range_check=range_check))
                outputs, done, slot = rn.submit(inputs)
                with torch.cuda.stream(rn.stream_of(slot)):
                    evaluator.process(inputs, outputs)
            for rn in runners.values():
                rn.drain()
    finally:
        # an exception (the deferred range assert, a loader error) must not leave work in flight on the runners' streams
        for rn in runners.values():
            rn.quiesce()
        if was_training and hasattr(model, "train"):
            model.train()
    results = evaluator.evaluate()
    return {} if results is None else results
