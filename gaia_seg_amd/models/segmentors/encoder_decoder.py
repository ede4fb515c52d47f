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
            self._plan_aux_fork()

    def _plan_aux_fork(self):
        """Tell the backbone behind which stage the auxiliary heads have all their inputs: the
        branch stream they run on forks from the training stream there (ops.prefork_branch), so
        they overlap the later stages as well as the decode head."""
        heads = (list(self.auxiliary_head) if isinstance(self.auxiliary_head, nn.ModuleList)
                 else [self.auxiliary_head])
        out_indices = getattr(self.backbone, "out_indices", None)
        if out_indices is None or self.with_neck:
            return
        used = set()
        for h in heads:
            idx = h.in_index if isinstance(h.in_index, (list, tuple)) else [h.in_index]
            used.update(i % len(out_indices) for i in idx)
        self.backbone.aux_fork_stage = out_indices[max(used)]

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
        if not self.with_auxiliary_head:
            losses.update(self._decode_head_forward_train(x, img_metas, gt_semantic_seg))
            return losses
        # The decode head and the auxiliary head(s) are independent consumers of x
        # ("dynamic_encoder_decoder-distill-backup (1).py":85-143; same call order here): the
        # auxiliary heads and their losses are queued on a branch stream (forward here, backward by
        # autograd's stream semantics) that forked from the training stream where their inputs were
        # complete — behind the backbone stage they read, or here at the latest — so they run beside
        # the rest of the backbone and the decode head.
        dev = img.device
        branch = ops.BRANCH_AUX and dev.type == "cuda"
        if branch and not getattr(getattr(self, "backbone", None), "_aux_forked", False):
            ops.prefork_branch(dev, ops.SLOT_AUX)
        losses.update(self._decode_head_forward_train(x, img_metas, gt_semantic_seg))
        with ops.branch_scope(dev, branch, forked=True):
            loss_aux = self._auxiliary_head_forward_train(x, img_metas, gt_semantic_seg)
        if branch:
            ops.join_branch(dev, ops.SLOT_AUX)     # whoever sums the losses reads both
        losses.update(loss_aux)
        return losses

    # ---- test time: one fused epilogue kernel per view (core/inference.py, csrc/inference.hip) ----
    def _fused(self):
        from ...core.inference import FusedInference
        eng = getattr(self, "_fused_engine", None)
        if eng is None or eng.num_classes != self.num_classes or eng.align_corners != bool(self.align_corners):
            eng = FusedInference(self.num_classes, self.align_corners)
            self._fused_engine = eng
        return eng

    def _view(self, img, img_meta, rescale, probs_in=None, want_probs=False, want_labels=True):
        """Epilogue of ONE (possibly flipped) view: window logits -> accumulate / normalise / rescale
        -> softmax -> flip back -> (+ probs_in) -> argmax.  Semantics of the reference's `inference`
        (dynamic_distiller.py:475-508) with slide / whole mode from ``test_cfg``."""
        meta0 = img_meta[0]
        ori = meta0["ori_shape"]
        if any(m["ori_shape"] != ori for m in img_meta):
            raise ValueError("all images of a test batch must share ori_shape")
        mode = self.test_cfg.mode
        kw = {}
        if mode == "slide":
            kw = dict(crop_size=self.test_cfg.crop_size, stride=self.test_cfg.stride)
        flip = meta0.get("flip_direction", "horizontal") if meta0.get("flip", False) else None
        if flip not in (None, "horizontal", "vertical"):
            raise ValueError("flip_direction must be 'horizontal' or 'vertical', got %r" % (flip,))

        def logits_fn(batch):
            return self._decode_head_forward_test(self.extract_feat(batch), img_meta)
        return self._fused()(logits_fn, img, mode=mode, out_size=tuple(ori[:2]) if rescale else None,
                             flip=flip, probs_in=probs_in, want_probs=want_probs,
                             want_labels=want_labels, **kw)

    def inference(self, img, img_meta, rescale):
        """Class probabilities [N, C, H, W] of one view (the tensor the reference's API returns)."""
        return self._view(img, img_meta, rescale, want_probs=True, want_labels=False)[1]

    def slide_inference(self, img, img_meta, rescale):
        raise NotImplementedError("slide_inference has no stand-alone form here: the window loop, "
                                  "normalisation and rescale run inside the fused epilogue "
                                  "(use inference / simple_test with test_cfg.mode='slide')")

    def whole_inference(self, img, img_meta, rescale):
        """Logits of the whole image at the input size, rescaled to ori_shape if asked."""
        logit = self.encode_decode(img, img_meta)
        if rescale:
            logit = self._resize_logits(logit, img_meta[0]["ori_shape"][:2])
        return logit

    def simple_test_device(self, img, img_meta, rescale=True):
        """simple_test that keeps the label map on the device: int64 [N, H, W]."""
        return self._view(img, img_meta, rescale)[0]

    def simple_test(self, img, img_meta, rescale=True):
        return list(self.simple_test_device(img, img_meta, rescale).cpu().numpy())

    def aug_test(self, imgs, img_metas, rescale=True):
        """Mean of the per-view probabilities, then argmax (dynamic_distiller.py:523-540): the running
        sum is carried through the fused kernel; the last view writes only the label map (the
        division by the number of views does not change the argmax)."""
        if not rescale:
            raise ValueError("aug_test needs rescale=True (views of different sizes)")
        acc = None
        for k, (img, meta) in enumerate(zip(imgs, img_metas)):
            last = k == len(imgs) - 1
            labels, acc = self._view(img, meta, rescale, probs_in=acc, want_probs=not last,
                                     want_labels=last)
        return list(labels.cpu().numpy())

    def forward_test(self, imgs, img_metas, **kwargs):
        if not isinstance(imgs, list) or not isinstance(img_metas, list):
            raise TypeError("forward_test takes lists (one entry per augmentation), got %s / %s"
                            % (type(imgs).__name__, type(img_metas).__name__))
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
                # (the heads return scalars: their mean is themselves, without a reduce launch)
                log_vars[loss_name] = loss_value if loss_value.dim() == 0 else loss_value.mean()
            elif isinstance(loss_value, list):
                log_vars[loss_name] = sum(_loss.mean() for _loss in loss_value)
            else:
                raise TypeError("%s is not a tensor or list of tensors" % loss_name)
        parts = [_value for _key, _value in log_vars.items() if "loss" in _key]
        loss = parts[0]
        for _value in parts[1:]:   # (sum() would start from the integer 0: one more launch)
            loss = loss + _value
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

