"""The reference's own forward WIRING, pinned (CPU): tests/golden/ref_wiring.{npz,json} hold outputs and
call traces of the reference's forward methods -- cut out of /root/reference by AST and run on
plain-torch stand-in children by tests/golden/make_ref_wiring_fixtures.py (the reference is absent at
test time).  oracle/model.py, given the same weights and inputs, must reproduce

  * every output tensor (the stand-ins and the oracle's bricks are the same torch ops, so equality is
    to rounding: a different loop bound, concat order or resize argument shows as an O(1) error), and
  * the call trace: which child runs in which order on which shape, every resize's
    (size, mode, align_corners).

  DynamicResLayer.forward             gaiaseg/models/utils/dynamic_res_layer.py:166-172
  DynamicResNet.forward               gaiaseg/models/backbones/dynamic_resnet.py:405-421
  DynamicPPM.forward / PSPHead.forward  dynamic_psp_head.py:62-73, psp_head.py:228-241
  DynamicUPerHead.forward             dynamic_uper_head.py:81-131
  DynamicFCNHead.forward              dynamic_fcn_head.py:128-135 (+ fcn_head.py:179-202, 248-253)

The product modules replay the same fixtures on the MI355X in tests/test_ref_wiring_gpu.py."""
import json
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 2e-6


@pytest.fixture(scope="module")
def wiring():
    with open(os.path.join(GOLD, "ref_wiring.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLD, "ref_wiring.npz"))


def state_dict_of(npz, prefix):
    return {k[len(prefix):]: torch.from_numpy(npz[k]) for k in npz.files if k.startswith(prefix)}


def close(a, b, tol=TOL):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) < tol


class OracleTrace:
    """forward pre-hooks on the oracle's bricks + a recording oracle.ops.resize: the same
    [(name, input shape, ...)] records as the generator's stand-ins write."""

    def __init__(self, root, kinds, rename=lambda n: n):
        from oracle import ops as O
        self.records, self.handles, self.O = [], [], O
        for name, m in root.named_modules():
            if isinstance(m, kinds):
                self.handles.append(m.register_forward_pre_hook(
                    lambda mod, args, _n=rename(name): self.records.append([_n, list(args[0].shape)])))

    def __enter__(self):
        self._resize = self.O.resize

        def resize(input, size=None, scale_factor=None, mode="nearest", align_corners=None):
            self.records.append(["resize", list(input.shape),
                                 [[int(s) for s in size], mode, bool(align_corners)]])
            return self._resize(input, size, scale_factor, mode, align_corners)
        self.O.resize = resize
        return self

    def __exit__(self, *exc):
        self.O.resize = self._resize
        for h in self.handles:
            h.remove()
        return False


def ref_trace(trace, drop=("dropout",)):
    """the reference trace in the oracle's vocabulary: (name, shape) for children, (+ args) for
    resize; the p = 0 dropout call of the reference has no counterpart module in the oracle."""
    out = []
    for name, shape, extra in trace:
        if name in drop:
            continue
        out.append([name, shape, extra] if name == "resize" else [name, shape])
    return out


def test_res_layer_runs_the_first_depth_state_blocks(wiring):
    from oracle.model import OBottleneck, OResLayer
    meta, npz = wiring
    m = meta["res_layer"]
    layer = OResLayer(m["inplanes"], m["planes"], m["depth_max"], stride=m["stride"])
    layer.load_state_dict(state_dict_of(npz, "reslayer_sd/"), strict=False)
    layer.train()
    x = torch.from_numpy(npz["reslayer_x"])
    for run in m["runs"]:
        layer.depth_state = run["depth_state"]
        with OracleTrace(layer, OBottleneck) as tr:
            y = layer(x)
        key = "reslayer_d%d%s_y" % (run["depth_state"], "_deploy" if run["deploying"] else "")
        assert close(y, npz[key]), key
        assert tr.records == ref_trace(run["trace"])
        assert len(run["trace"]) == run["depth_state"]
        if run["deploying"]:   # deploy_forward deletes the unused blocks, then runs the same loop
            assert run["blocks_left"] == run["depth_state"]


def test_backbone_forward_stem_stages_and_out_indices(wiring):
    from oracle.model import OBottleneck, ODynamicResNet
    meta, npz = wiring
    for c in meta["resnet"]:
        net = ODynamicResNet(3, c["stem_width"], c["width"], c["depth_max"], strides=tuple(c["strides"]),
                             out_indices=tuple(c["out_indices"]), deep_stem=c["deep_stem"])
        missing = net.load_state_dict(state_dict_of(npz, "resnet_%s_sd/" % c["tag"]), strict=False)
        assert not [k for k in missing.missing_keys if "num_batches_tracked" not in k]
        assert not missing.unexpected_keys
        net.manipulate_arch({"body": {"depth": c["depth"]}})
        net.train()
        with OracleTrace(net, OBottleneck) as tr:
            outs = net(torch.from_numpy(npz["resnet_%s_x" % c["tag"]]))
        assert isinstance(outs, tuple) and len(outs) == c["n_outs"] == len(c["out_indices"])
        for i, o in enumerate(outs):
            assert close(o, npz["resnet_%s_out%d" % (c["tag"], i)], 1e-5), (c["tag"], i)
        assert tr.records == ref_trace(c["trace"])
        assert len(c["trace"]) == sum(c["depth"])


def _head_inputs(npz, kind, tag):
    feats, i = [], 0
    while "%s_%s_in%d" % (kind, tag, i) in npz.files:
        feats.append(torch.from_numpy(npz["%s_%s_in%d" % (kind, tag, i)]))
        i += 1
    return feats


def _run_head(head, npz, kind, c):
    from oracle.model import OConv, OConvModule
    sd = state_dict_of(npz, "%s_%s_sd/" % (kind, c["tag"]))
    res = head.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "num_batches" not in k]
    head.train()
    with OracleTrace(head, (OConvModule,), ) as tr:
        h = head.conv_seg.register_forward_pre_hook(
            lambda mod, args: tr.records.append(["conv_seg", list(args[0].shape)]))
        y = head(_head_inputs(npz, kind, c["tag"]))
        h.remove()
    assert close(y, npz["%s_%s_logits" % (kind, c["tag"])], 1e-5), (kind, c["tag"])
    assert tr.records == ref_trace(c["trace"]), (kind, c["tag"])


def test_psp_head_wiring(wiring):
    from oracle.model import OPSPHead
    meta, npz = wiring
    for c in meta["psp"]:
        head = OPSPHead(c["in_channels"], c["channels"], c["num_classes"], tuple(c["pool_scales"]),
                        dropout_ratio=0.0, in_index=c["in_index"], align_corners=c["align_corners"])
        _run_head(head, npz, "psp", c)
        # psp_head.py:239 hands the channel record of the concat to the bottleneck as 2nd positional
        # argument: [C4, 512 x len(pool_scales)] -- x first, then the pyramid levels in scale order
        bott = [t for t in c["trace"] if t[0] == "bottleneck"][0]
        assert bott[2] == [[c["in_channels"]] + [c["channels"]] * len(c["pool_scales"])]
        assert bott[1][1] == c["in_channels"] + c["channels"] * len(c["pool_scales"])


def test_uper_head_wiring(wiring):
    from oracle.model import OUPerHead
    meta, npz = wiring
    for c in meta["uper"]:
        head = OUPerHead(c["in_channels"], c["channels"], c["num_classes"],
                         list(range(len(c["in_channels"]))), tuple(c["pool_scales"]),
                         dropout_ratio=0.0, align_corners=c["align_corners"])
        _run_head(head, npz, "uper", c)


def test_fcn_head_wiring(wiring):
    from oracle.model import OFCNHead
    meta, npz = wiring
    for c in meta["fcn"]:
        head = OFCNHead(c["in_channels"], c["channels"], c["num_classes"], num_convs=c["num_convs"],
                        kernel_size=c["kernel_size"], concat_input=c["concat_input"],
                        dropout_ratio=0.0, in_index=c["in_index"])
        _run_head(head, npz, "fcn", c)


def test_losses_wiring_and_values(wiring):
    """`losses` of the FCN / PSP heads run from the reference's own source with the reference's own
    cross_entropy, weight_reduce_loss and accuracy: oracle.ops.seg_losses reproduces loss_seg and
    acc_seg, and the recorded call order is resize(size = label size, bilinear, align_corners) ->
    [sampler.sample(resized logits, label [N,1,H,W])] -> loss_decode(resized logits, label [N,H,W],
    weight, ignore_index = 255) -> accuracy."""
    from oracle import ops as O
    meta, npz = wiring
    for c in meta["losses"]:
        t = c["tag"]
        logits = torch.from_numpy(npz["losses_%s_logits" % t])
        label = torch.from_numpy(npz["losses_%s_label" % t])
        pw = torch.from_numpy(npz["losses_%s_pixel_weight" % t])
        cw = torch.from_numpy(npz["losses_%s_class_weight" % t]) if c["class_weight"] else None
        sampler = (lambda lg, lb: pw) if c["sampler"] else None
        got = O.seg_losses(logits, label, c["loss_weight"], 255, c["align_corners"], sampler, cw)
        assert close(got["loss_seg"], npz["losses_%s_loss_seg" % t], 1e-6), t
        assert close(got["acc_seg"], npz["losses_%s_acc_seg" % t], 1e-6), t
        names = [r[0] for r in c["trace"]]
        assert names == ["resize"] + (["sampler.sample"] if c["sampler"] else []) + ["loss_decode"]
        rs = c["trace"][0]
        assert rs[2] == [list(label.shape[2:]), "bilinear", c["align_corners"]]
        ld = c["trace"][-1]
        assert ld[1] == [logits.shape[0], logits.shape[1]] + list(label.shape[2:])     # resized logits
        assert ld[2] == [[label.shape[0]] + list(label.shape[2:]), c["sampler"], 255]  # squeezed label
        assert c["keys"] == ["acc_seg", "loss_seg"] + (["resize_logit"] if c["head"] == "psp" else [])


def standin_low_logits(img, w1x1):
    """the fixture's stand-in decode head (tests/golden/make_ref_wiring_fixtures.py)"""
    import torch.nn.functional as F
    return F.conv2d(F.avg_pool2d(img, 8, ceil_mode=True), w1x1)


class _FakeSegmentor:
    """what oracle/inference.py needs of a model: encode_decode + decode_head.align_corners"""

    def __init__(self, w1x1, align):
        import types
        self.w, self.decode_head = w1x1, types.SimpleNamespace(align_corners=align)

    def encode_decode(self, img):
        from oracle import ops as O
        return O.resize(standin_low_logits(img, self.w), size=img.shape[2:], mode="bilinear",
                        align_corners=self.decode_head.align_corners)


def _labels_agree(got, want, prob=None):
    got, want = torch.as_tensor(got), torch.as_tensor(want)
    assert got.shape == want.shape
    return int((got != want).sum())


def test_inference_epilogue_against_the_reference_methods(wiring):
    """encode_decode -> slide / whole inference -> rescale -> softmax -> flip back -> argmax, and
    aug_test's mean over views, as the reference's own methods computed them
    (dynamic_distiller.py:252-262, 416-540): oracle/inference.py reproduces the probabilities (1e-6)
    and every label."""
    from oracle import inference as OI
    meta, npz = wiring
    inf = meta["inference"]
    w = torch.from_numpy(npz["inf_w1x1"])
    for c in inf["cases"]:
        model = _FakeSegmentor(w, c["align_corners"])
        img = torch.from_numpy(npz["inf_%s_img" % c["tag"]])
        m = dict(ori_shape=tuple(c["ori_shape"]) + (3,), flip=c["flip"], flip_direction=c["flip_direction"])
        cfg = dict(mode=c["mode"], crop_size=c["crop_size"], stride=c["stride"])
        prob = OI.inference(model, img, m, cfg, rescale=True)
        key = "inf_%s_prob" % c["tag"]
        if key in npz.files:
            assert close(prob, npz[key], 1e-6), c["tag"]
        assert _labels_agree(prob.argmax(1), npz["inf_%s_seg" % c["tag"]]) == 0, c["tag"]
    a = inf["aug"]
    model = _FakeSegmentor(w, False)
    imgs = [torch.from_numpy(npz["inf_aug_img%d" % i]) for i in range(len(a["views"]))]
    metas = [dict(ori_shape=tuple(a["ori_shape"]) + (3,), flip=v["flip"], flip_direction=v["flip_direction"])
             for v in a["views"]]
    seg = OI.aug_test(model, imgs, metas, dict(mode="whole"), rescale=True)
    assert _labels_agree(seg, npz["inf_aug_seg"]) == 0


def test_segmentor_forward_train_prefixes_and_call_order(wiring):
    """EncoderDecoder.forward_train as the reference tree restates it ("dynamic_encoder_decoder-
    distill-backup (1).py":85-143): decode head first, then the auxiliary head(s); every head gets
    (features, img_metas, gt, train_cfg); keys 'decode.*', 'aux.*' or 'aux_<i>.*' in that order.  The
    product segmentor's forward_train (host logic, no GPU needed) reproduces calls, keys, values."""
    import torch.nn as nn
    from gaia_seg_amd.models.segmentors import DynamicEncoderDecoder
    meta, _ = wiring
    for c in meta["forward_train"]:
        calls = []

        class Head(nn.Module):
            def __init__(self, name, base):
                super().__init__()
                self.name, self.base = name, base

            def forward_train(self, x, img_metas, gt, train_cfg):
                calls.append([self.name, [list(t.shape) for t in x], len(img_metas), list(gt.shape), train_cfg])
                return {"loss_seg": torch.tensor(self.base + 0.25), "acc_seg": torch.tensor(self.base * 10)}
        seg = DynamicEncoderDecoder.__new__(DynamicEncoderDecoder)
        nn.Module.__init__(seg)
        seg.train_cfg = "TRAIN_CFG"
        seg.decode_head = Head("decode_head", 1.0)
        if c["aux"] == "one":
            seg.auxiliary_head = Head("auxiliary_head", 2.0)
        elif c["aux"] == "list":
            seg.auxiliary_head = nn.ModuleList([Head("auxiliary_head.0", 2.0), Head("auxiliary_head.1", 3.0)])
        seg.extract_feat = lambda img: (img[:, :1] * 2, img[:, 1:] * 3)
        losses = seg.forward_train(torch.zeros(2, 3, 8, 8), [{}, {}], torch.zeros(2, 1, 8, 8, dtype=torch.long))
        assert calls == c["calls"], c["tag"]
        assert list(losses.keys()) == c["order"], c["tag"]
        assert {k: float(v) for k, v in losses.items()} == c["losses"], c["tag"]
        # total loss = the sum over the entries whose key contains 'loss' (mmseg _parse_losses, [3P])
        loss, log_vars = seg._parse_losses(losses)
        assert abs(float(loss) - sum(v for k, v in c["losses"].items() if "loss" in k)) < 1e-6
        assert set(log_vars) == set(c["losses"]) | {"loss"}
