"""``Evaler``: batched model + NMS with the reference's three timers (host-side mirror of reference
yolov6/core/evaler.py:67-151, 507-513, 578-608).

``predict`` is a caller of the hot path and keeps the reference's protocol: pre-process (to device, cast,
/255), inference (``outputs, _ = model(imgs)``), NMS (``multi_label=True``, max_det 300), each bracketed by
``time_sync``.  ``eval`` (reference :153-283) reports the speed figures and the LP accuracy metric; its matching
loops run as one kernel launch per batch (``yolov6.utils.lp_metric``).
"""
import os

import torch

from yolov6.utils.events import LOGGER
from yolov6.utils.checkpoint import load_checkpoint
from yolov6.utils.nms import non_max_suppression, xywh2xyxy
from yolov6.utils import lp_metric
from yolov6.utils.torch_utils import time_sync, get_model_info


class Evaler:
    def __init__(self, data, batch_size=32, img_size=640, conf_thres=0.03, iou_thres=0.65, device='', half=True,
                 save_dir='', **unused):
        self.data = data
        self.batch_size = batch_size
        self.img_size = img_size
        self.conf_thres = conf_thres
        self.iou_thres = iou_thres
        self.device = device
        self.half = half
        self.save_dir = save_dir

    def init_model(self, model, weights, task):
        if task != 'train':
            model = load_checkpoint(weights, map_location=self.device)
            self.stride = int(model.stride.max())
            from yolov6.layers.common import RepVGGBlock
            for layer in model.modules():
                if isinstance(layer, RepVGGBlock):
                    layer.switch_to_deploy()
            LOGGER.info("Switch model to deploy modality.")
            LOGGER.info("Model Summary: {}".format(get_model_info(model, self.img_size)))
        model.half() if self.half else model.float()
        if task != 'train' and self.device.type != 'cpu':   # warm-up on the final dtype: builds + tunes the engine
            model(torch.zeros(1, 3, self.img_size, self.img_size).to(self.device).type_as(next(model.parameters())))
        return model

    def predict(self, model, dataloader, task):
        """dataloader yields (imgs uint8 [B,3,H,W], targets [T,21] or None, paths, shapes).  Returns, like the reference
        (:103-151), (pred_results, total_targets, vis_outputs, vis_imgs): per batch the per-image detections [n,28] and
        labels [m,20] (8 ids, xyxy box and 8 corner coordinates in pixels of the network input)."""
        self.speed_result = torch.zeros(4, device=self.device)
        pred_results, total_targets = [], []
        vis_outputs, vis_imgs = [], None
        on_gpu = self.device.type == 'cuda'
        marks = []      # GPU: per batch four events on the current stream (pre | inference | NMS), read after the loop
        for i, (imgs, targets, paths, shapes) in enumerate(dataloader):
            c, h, w = imgs.shape[1:]
            if on_gpu:   # a new batch shape binds + tunes the engine once: outside every timer
                from yolov6.hip import runtime
                runtime.prepare_for(model, imgs.shape, torch.float16 if self.half else torch.float32)
            t1 = self._mark(on_gpu)
            imgs = imgs.to(self.device, non_blocking=True)
            imgs = imgs.half() if self.half else imgs.float()
            imgs /= 255
            batch_targets = self.split_targets(targets, imgs.shape[0], h, w)
            t2 = self._mark(on_gpu)
            outputs, _ = model(imgs)
            t3 = self._mark(on_gpu)
            outputs = non_max_suppression(outputs, self.conf_thres, self.iou_thres, multi_label=True)
            t4 = self._mark(on_gpu)
            if on_gpu:
                marks.append((t1, t2, t3, t4))
            else:
                self.speed_result[1] += t2 - t1
                self.speed_result[2] += t3 - t2
                self.speed_result[3] += t4 - t3
            self.speed_result[0] += len(outputs)
            pred_results.append(outputs)
            total_targets.append(batch_targets)
            if i == 0:
                vis_num = min(len(imgs), 8)
                vis_outputs, vis_imgs = outputs[:vis_num], imgs[:vis_num]
        if marks:       # the reference's three buckets (evaler.py:104-140) as device time between HIP events: one sync in all
            torch.cuda.synchronize(self.device)
            for t1, t2, t3, t4 in marks:
                self.speed_result[1] += t1.elapsed_time(t2) * 1e-3
                self.speed_result[2] += t2.elapsed_time(t3) * 1e-3
                self.speed_result[3] += t3.elapsed_time(t4) * 1e-3
        return pred_results, total_targets, vis_outputs, vis_imgs

    def split_targets(self, targets, batch, h, w):
        """Reference :120-128: labels [T,21] = (image index, 8 ids, xywh box, 8 corner coords), all normalised ->
        per image [m,20] with the box as xyxy and every coordinate in pixels (x * w, y * h)."""
        out = [torch.zeros((0, 20), device=self.device) for _ in range(batch)]
        if targets is None or len(targets) == 0:
            return out
        targets = targets.to(self.device).float().clone()
        targets[:, 9:13] = xywh2xyxy(targets[:, 9:13])
        targets[:, 9:21:2] *= w
        targets[:, 10:21:2] *= h
        idx = targets[:, 0].long()
        for b in range(batch):
            out[b] = targets[idx == b, 1:]
        return out

    def eval(self, preds, targets, model=None, task='val'):
        """Reference :153-283: speed report, then [mAP, mAP_50, mAP_75, mAP_50_95, recall, mAP_list, recall_list]."""
        if hasattr(self, 'speed_result'):
            self.eval_speed(task)
        assert len(preds) == len(targets), 'predict imgs count is not match with targets!'
        c = lp_metric.counts(preds, targets)
        if int(c[lp_metric.UNBINNED]):
            LOGGER.warning('%d matched labels have IoU >= 1.0 and fit no IoU bin: skipped (the reference re-uses a stale bin '
                           'index for them)' % int(c[lp_metric.UNBINNED]))
        return lp_metric.finish(c)

    def _mark(self, on_gpu):
        """Bucket boundary: a HIP event recorded on the current stream (GPU) or the synchronised wall clock (CPU)."""
        if not on_gpu:
            return time_sync()
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.device))
        return ev

    def eval_speed(self, task):
        """ms per image for pre-process / inference / NMS, like the reference's --task speed report."""
        n_samples = max(self.speed_result[0].item(), 1)
        pre_time, inf_time, nms_time = (1000 * self.speed_result[1:].cpu().numpy() / n_samples).tolist()
        for n, v in zip(["pre-process", "inference", "NMS"], [pre_time, inf_time, nms_time]):
            LOGGER.info("Average {} time: {:.2f} ms".format(n, v))
        return pre_time, inf_time, nms_time

    @staticmethod
    def check_task(task):
        if task not in ['train', 'val', 'test', 'speed']:
            raise Exception("task argument error: only support 'train' / 'val' / 'test' / 'speed' task.")

    @staticmethod
    def check_thres(conf_thres, iou_thres, task):
        if task in ('val', 'test'):
            if conf_thres > 0.03:
                LOGGER.warning(f'The best conf_thresh when evaluate the model is less than 0.03, while you set it to: {conf_thres}')
            if iou_thres != 0.65:
                LOGGER.warning(f'The best iou_thresh when evaluate the model is 0.65, while you set it to: {iou_thres}')
        if task == 'speed' and conf_thres < 0.4:
            LOGGER.warning(f'The best conf_thresh when test the speed of the model is larger than 0.4, while you set it to: {conf_thres}')

    @staticmethod
    def reload_device(device, model, task):
        if task == 'train':
            return next(model.parameters()).device
        if device == 'cpu':
            os.environ['CUDA_VISIBLE_DEVICES'] = '-1'
        elif device:
            os.environ['CUDA_VISIBLE_DEVICES'] = device
            assert torch.cuda.is_available()
        cuda = device != 'cpu' and torch.cuda.is_available()
        return torch.device('cuda:0' if cuda else 'cpu')
