"""Common part of the dynamic decode heads (mmseg BaseDecodeHead contract).

Mirrors the shared code of the reference heads: ``_init_inputs`` / ``_transform_inputs`` /
``cls_seg`` / ``losses`` / ``forward_train`` / ``forward_test``
(gaiaseg/models/decode_heads/fcn_head.py:139-275, dynamic_fcn_head.py:36-88,137-159).
A head forward is ONE autograd node over HIP kernels; ``losses`` feeds the low-resolution logits
to the fused resize+CE kernel.
"""
from abc import ABCMeta, abstractmethod

import torch
import torch.nn as nn

from ...core.bricks import DynamicConv2d, normal_init
from ...core.dynamic import DynamicMixin
from ...hip import ops
from ...hip.runtime import Act, tape_function
from ..builder import build_loss, build_pixel_sampler
from ..losses import seg_loss_and_accuracy


class DynamicBaseDecodeHead(nn.Module, DynamicMixin, metaclass=ABCMeta):
    search_space = set()

    def __init__(self, in_channels, channels, *, num_classes, dropout_ratio=0.1, conv_cfg=None,
                 norm_cfg=None, act_cfg=dict(type="ReLU"), in_index=-1, input_transform=None,
                 loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0),
                 ignore_index=255, sampler=None, align_corners=False, dynamic_conv_seg=True):
        super().__init__()
        self._init_inputs(in_channels, in_index, input_transform)
        self.channels = channels
        self.num_classes = num_classes
        self.dropout_ratio = dropout_ratio
        self.conv_cfg, self.norm_cfg, self.act_cfg = conv_cfg, norm_cfg, act_cfg
        self.in_index = in_index
        self.loss_decode = build_loss(loss_decode)
        self.ignore_index = ignore_index
        self.align_corners = align_corners
        self.sampler = build_pixel_sampler(sampler, context=self) if sampler is not None else None
        # reference: DynamicConv2d(channels, num_classes, kernel_size=1) (dynamic_fcn_head.py:76);
        # the UPer head inherits a plain nn.Conv2d — same kernel, fixed width
        self.conv_seg = DynamicConv2d(channels, num_classes, kernel_size=1, padding=0)
        self.dropout = nn.Dropout2d(dropout_ratio) if dropout_ratio > 0 else None
        self.fp16_enabled = False

    def extra_repr(self):
        return "input_transform=%s, ignore_index=%s, align_corners=%s" % (
            self.input_transform, self.ignore_index, self.align_corners)

    def _init_inputs(self, in_channels, in_index, input_transform):
        # fcn_head.py:139-173
        if input_transform is not None:
            assert input_transform in ["resize_concat", "multiple_select"]
        self.input_transform = input_transform
        self.in_index = in_index
        if input_transform is not None:
            assert isinstance(in_channels, (list, tuple))
            assert isinstance(in_index, (list, tuple))
            assert len(in_channels) == len(in_index)
            if input_transform == "resize_concat":
                self.in_channels = sum(in_channels)
            else:
                self.in_channels = in_channels
        else:
            assert isinstance(in_channels, int)
            assert isinstance(in_index, int)
            self.in_channels = in_channels

    def init_weights(self):
        normal_init(self.conv_seg, mean=0, std=0.01)

    # ---- input selection on Acts (fcn_head.py:179-202) ----
    def _select_index(self, n_inputs):
        if self.input_transform in ("resize_concat", "multiple_select"):
            return [i % n_inputs for i in self.in_index]
        return [self.in_index % n_inputs]

    def _transform_acts(self, tape, acts):
        """acts: the selected inputs, in in_index order."""
        if self.input_transform == "resize_concat":
            n, h, w = acts[0].N, acts[0].H, acts[0].W
            ctot = sum(a.C for a in acts)
            cat = Act.empty(n, h, w, ctot, acts[0].t.device)
            c0 = 0
            for a in acts:
                sl = cat.slice(c0, c0 + a.C)
                if (a.H, a.W) == (h, w):
                    ops.copy_into(tape, a, sl)
                else:
                    ops.bilinear(tape, a, (h, w), self.align_corners, out=sl)
                c0 += a.C
            return cat
        if self.input_transform == "multiple_select":
            return list(acts)
        return acts[0]

    def cls_seg_act(self, tape, feat):
        # Dropout2d -> 1x1 conv (fcn_head.py:248-253)
        if self.dropout is not None:
            feat = ops.dropout2d(tape, feat, self.dropout.p, self.dropout.training)
        return self.conv_seg.forward_act(tape, feat)

    @abstractmethod
    def forward_acts(self, tape, x):
        """x: the transformed input(s) as Act / list[Act]; returns the logits Act."""

    def forward(self, inputs):
        inputs = list(inputs)
        idx = self._select_index(len(inputs))
        selected = [inputs[i] for i in idx]
        needs = any(p.requires_grad for p in self.parameters())

        def runner(tape, acts):
            return [self.forward_acts(tape, self._transform_acts(tape, acts))]
        return tape_function(runner, selected, needs)[0]

    def forward_train(self, inputs, img_metas, gt_semantic_seg, train_cfg, **kwargs):
        if kwargs.get("teacher_logits") is not None or kwargs.get("aux_teacher_logits") is not None:
            raise NotImplementedError(
                "the in-place distillation branch (dynamic_fcn_head.py:178-227) is outside the "
                "supernet-training path: EncoderDecoder.forward_train never passes teacher logits")
        seg_logits = self.forward(inputs)
        return self.losses(seg_logits, gt_semantic_seg)

    def forward_test(self, inputs, img_metas, test_cfg):
        return self.forward(inputs)

    def losses(self, seg_logit, seg_label):
        """dynamic_fcn_head.py:137-159: resize -> (sampler) -> CE -> accuracy; the resize is fused
        into the CE kernel (the full-resolution logits are never materialised)."""
        loss = dict()
        seg_weight = None
        if self.sampler is not None:
            seg_weight = self.sampler.sample(seg_logit, seg_label)
        cw = getattr(self.loss_decode, "class_weight", None)
        if cw is not None:
            cw = seg_logit.new_tensor(cw)
        loss_seg, acc_seg = seg_loss_and_accuracy(
            seg_logit, seg_label, seg_weight, cw, self.ignore_index, self.align_corners,
            getattr(self.loss_decode, "loss_weight", 1.0))
        loss["loss_seg"] = loss_seg
        loss["acc_seg"] = acc_seg
        return loss
