"""Oracle vs the reference's own outputs and analytic known answers (CPU only).

* PINNED: oracle.ops.accuracy / weight_reduce_loss against tests/golden/ref_loss_utils.npz, produced
  by RUNNING gaiaseg/models/losses/{accuracy,utils}.py of the reference
  (tests/golden/make_ref_loss_fixtures.py).
* Known-answer tests for the contracts that have no runnable reference:
  leading-slice conv, CE mean over ALL pixels, bilinear align_corners=False, adaptive-pool bins,
  BN biased/unbiased variance, OHEM threshold rule.
* Metamorphic: R50 anchor == textbook bottleneck ResNet-50 parameter count; depth prefix.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ops as O
from oracle.model import ODynamicResNet, OEncoderDecoder

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_loss_utils.npz")


def test_accuracy_matches_reference_run():
    g = np.load(GOLD)
    for i in range(3):
        pred = torch.from_numpy(g["acc%d_pred" % i])
        target = torch.from_numpy(g["acc%d_target" % i])
        want = torch.from_numpy(g["acc%d_top1" % i])
        got = O.accuracy(pred, target)
        assert torch.equal(got, want), (i, got, want)


def test_weight_reduce_loss_matches_reference_run():
    g = np.load(GOLD)
    for i in range(2):
        loss = torch.from_numpy(g["wrl%d_loss" % i])
        w = torch.from_numpy(g["wrl%d_weight" % i])
        assert torch.equal(O.weight_reduce_loss(loss), torch.from_numpy(g["wrl%d_mean" % i]))
        assert torch.equal(O.weight_reduce_loss(loss, w), torch.from_numpy(g["wrl%d_wmean" % i]))
        assert torch.equal(O.weight_reduce_loss(loss, w, reduction="sum"),
                           torch.from_numpy(g["wrl%d_sum" % i]))
        assert torch.equal(O.weight_reduce_loss(loss, w, reduction="none"),
                           torch.from_numpy(g["wrl%d_none" % i]))
        assert torch.equal(O.weight_reduce_loss(loss, w, avg_factor=17.0),
                           torch.from_numpy(g["wrl%d_avg" % i]))


def test_leading_slice_conv_known_answer():
    # weight[o, i] = 10*o + i ; x channel i = i+1  ->  y[o] = sum_{i<ci} (10 o + i)(i + 1)
    w = torch.zeros(4, 5, 1, 1)
    for o in range(4):
        for i in range(5):
            w[o, i, 0, 0] = 10 * o + i
    x = torch.arange(1, 4).float().view(1, 3, 1, 1)          # 3 active input channels
    y = O.dyn_conv2d(x, w, None, width=2)                     # 2 active output channels
    want = [sum((10 * o + i) * (i + 1) for i in range(3)) for o in range(2)]
    assert y.flatten().tolist() == want


def test_ce_mean_counts_ignored_pixels():
    torch.manual_seed(0)
    logit = torch.randn(1, 19, 8, 8)
    label = torch.randint(0, 19, (1, 8, 8))
    label[0, :2] = 255
    per = F.cross_entropy(logit, label, reduction="none", ignore_index=255)
    over_all = O.cross_entropy(logit, label)
    assert torch.allclose(over_all, per.sum() / 64)
    assert not torch.allclose(over_all, per.sum() / 48)       # NOT the mean over valid pixels


def test_bilinear_align_false_known_answer():
    x = torch.tensor([[0., 1.], [0., 1.]]).view(1, 1, 2, 2)
    y = O.resize(x, size=(4, 4), mode="bilinear", align_corners=False)
    assert torch.allclose(y[0, 0, 0], torch.tensor([0., .25, .75, 1.]))


def test_adaptive_pool_bin_edges():
    # H = 16, s = 3: bins [0,6) [5,11) [10,16)  (floor(i*H/s), ceil((i+1)*H/s))
    x = torch.arange(16.).view(1, 1, 16, 1)
    y = F.adaptive_avg_pool2d(x, (3, 1)).flatten()
    assert torch.allclose(y, torch.tensor([2.5, 7.5, 12.5]))


def test_dyn_bn_slice_and_running_stats():
    torch.manual_seed(0)
    rm, rv = torch.zeros(8), torch.ones(8)
    w, b = torch.ones(8) * 2, torch.ones(8)
    x = torch.randn(4, 5, 3, 3)
    y = O.dyn_batch_norm(x, rm, rv, w, b, training=True)
    mu = x.mean((0, 2, 3))
    var_b = x.var((0, 2, 3), unbiased=False)
    var_u = x.var((0, 2, 3), unbiased=True)
    assert torch.allclose(y, (x - mu.view(1, -1, 1, 1)) / (var_b.view(1, -1, 1, 1) + 1e-5).sqrt() * 2 + 1, atol=1e-5)
    assert torch.allclose(rm[:5], 0.1 * mu, atol=1e-6) and torch.all(rm[5:] == 0)
    assert torch.allclose(rv[:5], 0.9 + 0.1 * var_u, atol=1e-6) and torch.all(rv[5:] == 1)


def test_ohem_threshold_rule():
    torch.manual_seed(0)
    logit = torch.randn(2, 5, 6, 6)
    label = torch.randint(0, 5, (2, 1, 6, 6))
    label[0, 0, 0] = 255
    w = O.ohem_pixel_weights(logit, label, thresh=0.7, min_kept=10)
    prob = F.softmax(logit, 1).gather(1, label.clamp(max=4)).squeeze(1)
    valid = label.squeeze(1) != 255
    srt = prob[valid].sort()[0]
    thr = max(float(srt[min(20, srt.numel() - 1)]), 0.7)
    assert torch.equal(w.bool(), (prob < thr) & valid)


R50 = dict(stem={"width": 64}, body={"width": [64, 128, 256, 512], "depth": [3, 4, 6, 3]})


def _supernet():
    return ODynamicResNet(3, 64, [80, 160, 320, 640], [4, 6, 29, 4])


def test_r50_anchor_is_textbook_resnet50_size():
    # the R50 anchor of the supernet has the 23.51 M parameters of a bottleneck ResNet-50 body
    net = ODynamicResNet(3, 64, [64, 128, 256, 512], [3, 4, 6, 3])
    n = sum(p.numel() for p in net.parameters())
    assert abs(n - 23.51e6) < 0.02e6
    assert abs(sum(p.numel() for p in _supernet().parameters()) - 84.78e6) < 0.02e6


def test_slice_equals_standalone_on_cpu_oracle():
    torch.manual_seed(0)
    sup = ODynamicResNet(3, 32, [32, 64, 96, 128], [2, 2, 3, 2]).eval()
    sub = ODynamicResNet(3, 16, [16, 48, 64, 96], [1, 2, 2, 1]).eval()
    for m in sup.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
    sd = sup.state_dict()
    sub.load_state_dict({k: sd[k][tuple(slice(0, s) for s in v.shape)] if v.dim() else sd[k]
                         for k, v in sub.state_dict().items()})
    sup.manipulate_arch(dict(stem={"width": 16}, body={"width": [16, 48, 64, 96], "depth": [1, 2, 2, 1]}))
    x = torch.randn(1, 3, 64, 64)
    for a, b in zip(sup(x), sub(x)):
        assert torch.allclose(a, b, atol=1e-5)


def test_encoder_decoder_loss_aggregation():
    torch.manual_seed(0)
    cfg = dict(backbone=dict(in_channels=3, stem_width=16, body_width=[16, 32, 48, 64], body_depth=[1, 1, 1, 1]),
               decode_head=dict(type="DynamicFCNHead", in_channels=256, in_index=3, channels=32, num_classes=19,
                                dropout_ratio=0.0, loss_decode=dict(loss_weight=1.0)),
               auxiliary_head=dict(type="DynamicFCNHead", in_channels=192, in_index=2, channels=16, num_convs=1,
                                   concat_input=False, num_classes=19, dropout_ratio=0.0,
                                   loss_decode=dict(loss_weight=0.4)))
    m = OEncoderDecoder(**cfg).train()
    img, gt = torch.randn(2, 3, 64, 64), torch.randint(0, 19, (2, 1, 64, 64))
    losses = m.forward_train(img, gt)
    assert set(losses) == {"decode.loss_seg", "decode.acc_seg", "aux.loss_seg", "aux.acc_seg"}
    total, log_vars = m.parse_losses(losses)
    assert torch.allclose(total, losses["decode.loss_seg"] + losses["aux.loss_seg"])
