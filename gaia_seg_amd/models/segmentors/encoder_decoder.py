"""EncoderDecoder segmentor with mmseg's contract.

mmseg's EncoderDecoder is an absent dependency; its train / inference flow is restated in the
reference tree at gaiaseg/models/segmentors/dynamic_encoder_decoder-distill-backup (1).py:85-143
(forward_train, loss aggregation) and gaiaseg/models/segmentors/dynamic_distiller.py:245-307,416-521
(extract_feat, encode_decode, slide / whole inference, simple_test, aug_test), which this class
follows.  ``train_step`` / ``_parse_losses`` follow SURVEY.md Appendix A12.
"""
from collections import OrderedDict

import torch
import torch.distributed as dist
import torch.nn as nn

from ...hip import ops
from ...hip.runtime import Act, tape_function
from .. import builder


def add_prefix(inputs, prefix):
    return {"%s.%s" % (prefix, name): value for name, value in inputs.items()}


class EncoderDecoder(nn.Module):
    def __init__(self, backbone, decode_head, neck=None, auxiliary_head=None, train_cfg=None,
                 test_cfg=None, pretrained=None):
        super().__init__()
        self.fp16_enabled = False
        self.backbone = builder.build_backbone(backbone)
        if neck is not None:
            self.neck = builder.build_neck(neck)
        self._init_decode_head(decode_head)
        self._init_auxiliary_head(auxiliary_head)
        from ...core.config import ConfigDict
        self.train_cfg = ConfigDict(train_cfg) if isinstance(train_cfg, dict) else train_cfg
        self.test_cfg = ConfigDict(test_cfg) if isinstance(test_cfg, dict) else test_cfg
        self.init_weights(pretrained=pretrained)
        assert self.with_decode_head

    # ---- structure ----
    @property
    def with_neck(self):
        return hasattr(self, "neck") and self.neck is not None

    @property
    def with_auxiliary_head(self):
        return hasattr(self, "auxiliary_head") and self.auxiliary_head is not None

    @property
    def with_decode_head(self):
        return hasattr(self, "decode_head") and self.decode_head is not None

    def _init_decode_head(self, decode_head):
        self.decode_head = builder.build_head(decode_head)
        self.align_corners = self.decode_head.align_corners
        self.num_classes = self.decode_head.num_classes

    def _init_auxiliary_head(self, auxiliary_head):
        if auxiliary_head is not None:
            if isinstance(auxiliary_head, list):
                self.auxiliary_head = nn.ModuleList(
                    [builder.build_head(h) for h in auxiliary_head])
            else:
                self.auxiliary_head = builder.build_head(auxiliary_head)

    def init_weights(self, pretrained=None):
        self.backbone.init_weights(pretrained=pretrained)
        self.decode_head.init_weights()
        if self.with_auxiliary_head:
            if isinstance(self.auxiliary_head, nn.ModuleList):
                for aux_head in self.auxiliary_head:
                    aux_head.init_weights()
            else:
                self.auxiliary_head.init_weights()

    # ---- forward pieces ----
    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def _resize_logits(self, logits, size):
        """mmseg.ops.resize(bilinear) of a logits tensor through the HIP kernel (no grad path).
        The class dimension is padded to a float4 multiple (the kernels move whole float4s)."""
        from ...hip.runtime import Tape, round_up
        n, c, h, w = logits.shape
        size = (int(size[0]), int(size[1]))
        if size == (h, w):
            return logits  # bilinear resize to the same size is the identity
        ld = round_up(c, 4)
        nhwc = logits.detach().permute(0, 2, 3, 1)
        in_place = (nhwc.stride() == (h * w * ld, w * ld, ld, 1) and nhwc.data_ptr() % 16 == 0
                    and nhwc.untyped_storage().nbytes() // 4 - nhwc.storage_offset() >= n * h * w * ld)
        if in_place:
            full = nhwc.as_strided((n, h, w, ld), nhwc.stride())
        else:
            full = torch.zeros((n, h, w, ld), dtype=torch.float32, device=logits.device)
            full[..., :c].copy_(nhwc)
        out = ops.bilinear(Tape(enabled=False), Act(full, False), size, self.align_corners)
        return out.t[..., :c].permute(0, 3, 1, 2)

    def encode_decode(self, img, img_metas):
        x = self.extract_feat(img)
        out = self._decode_head_forward_test(x, img_metas)
        return self._resize_logits(out, img.shape[2:])

    def _decode_head_forward_train(self, x, img_metas, gt_semantic_seg):
        loss_decode = self.decode_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg)
        return add_prefix(loss_decode, "decode")

    def _decode_head_forward_test(self, x, img_metas):
        return self.decode_head.forward_test(x, img_metas, self.test_cfg)

    def _auxiliary_head_forward_train(self, x, img_metas, gt_semantic_seg):
        losses = dict()
        if isinstance(self.auxiliary_head, nn.ModuleList):
            for idx, aux_head in enumerate(self.auxiliary_head):
                loss_aux = aux_head.forward_train(x, img_metas, gt_semantic_seg, self.train_cfg)
                losses.update(add_prefix(loss_aux, "aux_%d" % idx))
        else:
            loss_aux = self.auxiliary_head.forward_train(x, img_metas, gt_semantic_seg,
                                                         self.train_cfg)
            losses.update(add_prefix(loss_aux, "aux"))
        return losses

    def forward_dummy(self, img):
        return self.encode_decode(img, None)

    def forward_train(self, img, img_metas, gt_semantic_seg):
        x = self.extract_feat(img)
        losses = dict()
        losses.update(self._decode_head_forward_train(x, img_metas, gt_semantic_seg))
        if self.with_auxiliary_head:
            losses.update(self._auxiliary_head_forward_train(x, img_metas, gt_semantic_seg))
        return losses

    # ---- inference (dynamic_distiller.py:416-521) ----
    def slide_inference(self, img, img_meta, rescale):
        h_stride, w_stride = self.test_cfg.stride
        h_crop, w_crop = self.test_cfg.crop_size
        batch_size, _, h_img, w_img = img.size()
        num_classes = self.num_classes
        h_grids = max(h_img - h_crop + h_stride - 1, 0) // h_stride + 1
        w_grids = max(w_img - w_crop + w_stride - 1, 0) // w_stride + 1
        preds = img.new_zeros((batch_size, num_classes, h_img, w_img))
        count_mat = img.new_zeros((batch_size, 1, h_img, w_img))
        for h_idx in range(h_grids):
            for w_idx in range(w_grids):
                y1 = h_idx * h_stride
                x1 = w_idx * w_stride
                y2 = min(y1 + h_crop, h_img)
                x2 = min(x1 + w_crop, w_img)
                y1 = max(y2 - h_crop, 0)
                x1 = max(x2 - w_crop, 0)
                crop_img = img[:, :, y1:y2, x1:x2].contiguous()
                crop_seg_logit = self.encode_decode(crop_img, img_meta)
                preds[:, :, y1:y2, x1:x2] += crop_seg_logit
                count_mat[:, :, y1:y2, x1:x2] += 1
        assert (count_mat == 0).sum() == 0
        preds = preds / count_mat
        if rescale:
            preds = self._resize_logits(preds, img_meta[0]["ori_shape"][:2])
        return preds

    def whole_inference(self, img, img_meta, rescale):
        seg_logit = self.encode_decode(img, img_meta)
        if rescale and tuple(seg_logit.shape[2:]) != tuple(img_meta[0]["ori_shape"][:2]):
            seg_logit = self._resize_logits(seg_logit, img_meta[0]["ori_shape"][:2])
        return seg_logit

    def inference(self, img, img_meta, rescale):
        assert self.test_cfg.mode in ["slide", "whole"]
        ori_shape = img_meta[0]["ori_shape"]
        assert all(_["ori_shape"] == ori_shape for _ in img_meta)
        if self.test_cfg.mode == "slide":
            seg_logit = self.slide_inference(img, img_meta, rescale)
        else:
            seg_logit = self.whole_inference(img, img_meta, rescale)
        output = torch.softmax(seg_logit, dim=1)
        flip = img_meta[0].get("flip", False)
        if flip:
            flip_direction = img_meta[0]["flip_direction"]
            assert flip_direction in ["horizontal", "vertical"]
            output = output.flip(dims=(3,)) if flip_direction == "horizontal" else output.flip(dims=(2,))
        return output

    def _whole_argmax(self, img, img_meta, rescale):
        """whole_inference + softmax + argmax in one fused kernel: argmax of the bilinearly resized
        logits straight from the low-resolution head output (softmax is monotone; the 19xHxW
        tensor never exists).  Valid when the two resizes of the reference (to the input size, then
        to ori_shape, dynamic_distiller.py:252-262,461-473) collapse into one, i.e. the image was
        not rescaled; otherwise the generic path below is taken."""
        import ctypes
        from ...hip import lib as _lib
        from ...hip.runtime import current_stream_ptr
        from ..losses.cross_entropy_loss import _ce_desc
        x = self.extract_feat(img)
        logits = self._decode_head_forward_test(x, img_meta)
        size = tuple(img.shape[2:])
        d = _ce_desc(logits, size, None, self.align_corners)
        seg = torch.empty((img.shape[0],) + size, dtype=torch.int64, device=img.device)
        L = _lib.load()
        _lib.check(L.gs_resize_argmax(ctypes.byref(d), logits.data_ptr(), seg.data_ptr(), None,
                                      current_stream_ptr()), "gs_resize_argmax")
        return seg

    def simple_test_device(self, img, img_meta, rescale=True):
        """simple_test that keeps the label map on the device: int64 [N, H, W]."""
        ori = tuple(img_meta[0]["ori_shape"][:2])
        fused = (self.test_cfg.mode == "whole" and not img_meta[0].get("flip", False)
                 and (not rescale or ori == tuple(img.shape[2:])))
        if fused:
            return self._whole_argmax(img, img_meta, rescale)
        return self.inference(img, img_meta, rescale).argmax(dim=1)

    def simple_test(self, img, img_meta, rescale=True):
        ori = tuple(img_meta[0]["ori_shape"][:2])
        fused = (self.test_cfg.mode == "whole" and not img_meta[0].get("flip", False)
                 and (not rescale or ori == tuple(img.shape[2:])))
        if fused:
            seg_pred = self._whole_argmax(img, img_meta, rescale)
        else:
            seg_logit = self.inference(img, img_meta, rescale)
            seg_pred = seg_logit.argmax(dim=1)
        return list(seg_pred.cpu().numpy())

    def aug_test(self, imgs, img_metas, rescale=True):
        assert rescale
        seg_logit = self.inference(imgs[0], img_metas[0], rescale)
        for i in range(1, len(imgs)):
            seg_logit += self.inference(imgs[i], img_metas[i], rescale)
        seg_logit /= len(imgs)
        seg_pred = seg_logit.argmax(dim=1)
        return list(seg_pred.cpu().numpy())

    def forward_test(self, imgs, img_metas, **kwargs):
        for var, name in [(imgs, "imgs"), (img_metas, "img_metas")]:
            if not isinstance(var, list):
                raise TypeError("%s must be a list, but got %s" % (name, type(var)))
        if len(imgs) != len(img_metas):
            raise ValueError("num of augmentations (%d) != num of image meta (%d)"
                             % (len(imgs), len(img_metas)))
        if len(imgs) == 1:
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        return self.aug_test(imgs, img_metas, **kwargs)

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    # ---- runner interface ----
    def train_step(self, data_batch, optimizer=None, **kwargs):
        losses = self(**data_batch)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data_batch["img_metas"]))

    def val_step(self, data_batch, **kwargs):
        return self(**data_batch, **kwargs)

    @staticmethod
    def _parse_losses(losses):
        log_vars = OrderedDict()
        for loss_name, loss_value in losses.items():
            if isinstance(loss_value, torch.Tensor):
                log_vars[loss_name] = loss_value.mean()
            elif isinstance(loss_value, list):
                log_vars[loss_name] = sum(_loss.mean() for _loss in loss_value)
            else:
                raise TypeError("%s is not a tensor or list of tensors" % loss_name)
        loss = sum(_value for _key, _value in log_vars.items() if "loss" in _key)
        log_vars["loss"] = loss
        # one batched all-reduce for all log scalars instead of one per entry; values are kept on
        # the device (no host sync inside the step) — loggers call .item() when they print
        names = list(log_vars.keys())
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            stacked = torch.stack([log_vars[n].detach().float() for n in names])
            dist.all_reduce(stacked)
            stacked /= dist.get_world_size()
            for i, n in enumerate(names):
                log_vars[n] = stacked[i]
        else:
            for n in names:
                log_vars[n] = log_vars[n].detach()
        return loss, log_vars

