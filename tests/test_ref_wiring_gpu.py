"""The product modules (HIP kernels) replay the reference-wiring fixtures of
tests/golden/make_ref_wiring_fixtures.py on the MI355X: given the fixture's weights and inputs,
DynamicResLayer / DynamicResNet / DynamicPSPHead / DynamicUPerHead / DynamicFCNHead of this package must
produce the outputs the REFERENCE's own forward methods produced on plain-torch children
(gaiaseg/models/utils/dynamic_res_layer.py:159-172, backbones/dynamic_resnet.py:405-421,
decode_heads/dynamic_psp_head.py:62-73, psp_head.py:228-241, dynamic_uper_head.py:81-131,
dynamic_fcn_head.py:128-135).  All convolution / BatchNorm weights are distinct random tensors, so a
wrong concat-slice offset, block order, resize argument or top-down add order changes the logits by
O(1); the tolerance only covers fp32 summation order."""
import json
import os

import numpy as np
import pytest
import torch

from test_ref_wiring import GOLD, _head_inputs, close, state_dict_of

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 2e-4
CONV, DBN, SBN = dict(type="DynConv2d"), dict(type="DynBN", requires_grad=True), dict(type="SyncBN", requires_grad=True)


@pytest.fixture(scope="module")
def wiring():
    with open(os.path.join(GOLD, "ref_wiring.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLD, "ref_wiring.npz"))


def _load(mod, sd):
    res = mod.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not [k for k in res.missing_keys if "num_batches_tracked" not in k], res.missing_keys
    return mod.to(DEV).train()


def _cl(x):
    return torch.as_tensor(x).to(DEV).contiguous(memory_format=torch.channels_last)


def test_res_layer_forward_and_deploy(hip_lib, wiring):
    from gaia_seg_amd.core.bricks import DynamicBottleneck
    from gaia_seg_amd.models.utils import DynamicResLayer
    meta, npz = wiring
    m = meta["res_layer"]
    x = _cl(npz["reslayer_x"])
    for run in m["runs"]:
        layer = DynamicResLayer(DynamicBottleneck, m["inplanes"], m["planes"], depth=m["depth_max"],
                                stride=m["stride"], conv_cfg=CONV, norm_cfg=DBN)
        layer = _load(layer, state_dict_of(npz, "reslayer_sd/"))
        layer.manipulate_depth(run["depth_state"])
        if run["deploying"]:
            layer._deploying = True
        with torch.no_grad():
            y = layer(x)
        key = "reslayer_d%d%s_y" % (run["depth_state"], "_deploy" if run["deploying"] else "")
        assert close(y.cpu(), npz[key], TOL), key
        assert len(layer) == run["blocks_left"]     # deploy_forward drops the unused blocks


def test_backbone_forward(hip_lib, wiring):
    from gaia_seg_amd.models import build_backbone
    meta, npz = wiring
    for c in meta["resnet"]:
        net = build_backbone(dict(type="DynamicResNet", in_channels=3, stem_width=c["stem_width"],
                                  body_width=c["width"], body_depth=c["depth_max"], num_stages=4,
                                  strides=tuple(c["strides"]), out_indices=tuple(c["out_indices"]),
                                  deep_stem=c["deep_stem"], conv_cfg=CONV, norm_cfg=DBN, style="pytorch"))
        net = _load(net, state_dict_of(npz, "resnet_%s_sd/" % c["tag"]))
        net.manipulate_arch({"body": {"depth": c["depth"], "width": c["width"]}})
        with torch.no_grad():
            outs = net(torch.as_tensor(npz["resnet_%s_x" % c["tag"]]).to(DEV))
        assert isinstance(outs, tuple) and len(outs) == c["n_outs"]
        for i, o in enumerate(outs):
            assert close(o.cpu(), npz["resnet_%s_out%d" % (c["tag"], i)], TOL), (c["tag"], i)


def _head_case(kind, c, npz, cfg):
    from gaia_seg_amd.models import build_head
    cfg = dict(cfg, conv_cfg=CONV, norm_cfg=SBN, dropout_ratio=0.0, num_classes=c["num_classes"],
               channels=c["channels"], align_corners=c.get("align_corners", False),
               loss_decode=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=1.0))
    head = _load(build_head(cfg), state_dict_of(npz, "%s_%s_sd/" % (kind, c["tag"])))
    feats = [_cl(f) for f in _head_inputs(npz, kind, c["tag"])]
    with torch.no_grad():
        y = head(feats)
    want = npz["%s_%s_logits" % (kind, c["tag"])]
    assert tuple(y.shape) == want.shape
    assert close(y.cpu(), want, TOL), (kind, c["tag"])


def test_psp_head(hip_lib, wiring):
    meta, npz = wiring
    for c in meta["psp"]:
        _head_case("psp", c, npz, dict(type="DynamicPSPHead", in_channels=c["in_channels"],
                                       in_index=c["in_index"], pool_scales=tuple(c["pool_scales"])))


def test_uper_head(hip_lib, wiring):
    meta, npz = wiring
    for c in meta["uper"]:
        _head_case("uper", c, npz, dict(type="DynamicUPerHead", in_channels=c["in_channels"],
                                        in_index=list(range(len(c["in_channels"]))),
                                        pool_scales=tuple(c["pool_scales"])))


def test_fcn_head(hip_lib, wiring):
    meta, npz = wiring
    for c in meta["fcn"]:
        _head_case("fcn", c, npz, dict(type="DynamicFCNHead", in_channels=c["in_channels"],
                                       in_index=c["in_index"], num_convs=c["num_convs"],
                                       kernel_size=c["kernel_size"], concat_input=c["concat_input"]))


def test_losses_fused_resize_ce(hip_lib, wiring):
    """The product's `losses` (fused resize + cross entropy + accuracy kernel, the full-resolution
    logits never exist) against the values the reference's own `losses` code produced
    (dynamic_fcn_head.py:137-159 / dynamic_psp_head.py:149-173 with its cross_entropy and accuracy)."""
    from gaia_seg_amd.hip.runtime import Act
    from gaia_seg_amd.models.losses import seg_loss_and_accuracy
    meta, npz = wiring
    for c in meta["losses"]:
        t = c["tag"]
        logits = torch.from_numpy(npz["losses_%s_logits" % t])
        n, k, h, w = logits.shape
        a = Act.empty(n, h, w, k, torch.device(DEV))
        a.t.copy_(logits.permute(0, 2, 3, 1))
        label = torch.from_numpy(npz["losses_%s_label" % t]).to(DEV)
        pw = torch.from_numpy(npz["losses_%s_pixel_weight" % t]).to(DEV) if c["sampler"] else None
        cw = torch.from_numpy(npz["losses_%s_class_weight" % t]).to(DEV) if c["class_weight"] else None
        loss, acc = seg_loss_and_accuracy(a.as_nchw(), label, pw, cw, 255, c["align_corners"],
                                          c["loss_weight"])
        want_l, want_a = npz["losses_%s_loss_seg" % t].item(), npz["losses_%s_acc_seg" % t].item()
        assert abs(float(loss) - want_l) < 2e-5 * max(1.0, abs(want_l)), t
        assert abs(float(acc) - want_a) < 100.0 * 2.01 / label.numel(), t    # argmax ties: <= 2 pixels


def test_inference_epilogue_fused_kernel(hip_lib, wiring):
    """The product segmentor's test-time methods (inference / simple_test / aug_test: ONE fused
    epilogue kernel per view, csrc/inference.hip, all windows batched) against what the reference's own
    encode_decode / slide_inference / whole_inference / inference / simple_test / aug_test computed
    (dynamic_distiller.py:252-262, 416-540).  extract_feat is patched to the identity and the decode
    head's forward_test to the fixture's stand-in, exactly as the generator did on the reference side."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_ref_wiring import standin_low_logits
    from util_models import fcn_head, model_cfg
    from gaia_seg_amd.core.config import ConfigDict
    from gaia_seg_amd.models import build_segmentor
    meta, npz = wiring
    inf = meta["inference"]
    w = torch.from_numpy(npz["inf_w1x1"]).to(DEV)
    model = build_segmentor(model_cfg(fcn_head(), aux=False)).to(DEV).eval()
    model.extract_feat = lambda img: img
    model._decode_head_forward_test = lambda x, metas: standin_low_logits(x, w)
    model.decode_head.num_classes = model.num_classes = inf["num_classes"]

    def check_labels(got, want, prob_ref=None):
        got = torch.as_tensor(np.stack(got))
        want = torch.as_tensor(want)
        differ = got != want
        n = int(differ.sum())
        if n and prob_ref is not None:     # a differing pixel must be an argmax tie within rounding
            top2 = torch.as_tensor(prob_ref).topk(2, dim=1).values
            assert float((top2[:, 0] - top2[:, 1])[differ].max()) < 1e-5
        assert n <= 2, n

    for c in inf["cases"]:
        model.align_corners = model.decode_head.align_corners = c["align_corners"]
        model.test_cfg = ConfigDict(dict(mode=c["mode"], crop_size=c["crop_size"], stride=c["stride"]))
        img = torch.from_numpy(npz["inf_%s_img" % c["tag"]]).to(DEV)
        metas = [dict(ori_shape=tuple(c["ori_shape"]) + (3,), flip=c["flip"],
                      flip_direction=c["flip_direction"])] * img.shape[0]
        key = "inf_%s_prob" % c["tag"]
        prob_ref = npz[key] if key in npz.files else None
        with torch.no_grad():
            if prob_ref is not None:
                prob = model.inference(img, metas, True)
                assert close(prob.cpu(), prob_ref, 2e-5), c["tag"]
            check_labels(model.simple_test(img, metas, True), npz["inf_%s_seg" % c["tag"]], prob_ref)
    a = inf["aug"]
    model.align_corners = model.decode_head.align_corners = False
    model.test_cfg = ConfigDict(dict(mode="whole"))
    imgs = [torch.from_numpy(npz["inf_aug_img%d" % i]).to(DEV) for i in range(len(a["views"]))]
    metas = [[dict(ori_shape=tuple(a["ori_shape"]) + (3,), flip=v["flip"], flip_direction=v["flip_direction"])]
             for v in a["views"]]
    with torch.no_grad():
        check_labels(model.aug_test(imgs, metas, True), npz["inf_aug_seg"])
