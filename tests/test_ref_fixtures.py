"""What the reference itself can still pin (CPU only; fixtures made by
tests/golden/make_ref_pure_fixtures.py from /root/reference, which is absent at test time):

* the repo's configs and model-sampler trees == the reference's own config files, exec'd;
* oracle.ops.cross_entropy == the reference's cross_entropy (losses/cross_entropy_loss.py:67-94),
  bit for bit;
* the "DL -> LD" arch-meta slicing of DynamicResNet.manipulate_stem / manipulate_body
  (dynamic_resnet.py:381-403) in the product backbone and in the oracle;
* the slide-inference window grid and accumulate / normalise arithmetic
  (dynamic_distiller.py:416-459) in the oracle and in the product's host-side window helper."""
import json
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")


def _j(o):
    """JSON normal form (tuples -> lists, ConfigDict -> dict)."""
    return json.loads(json.dumps(o, sort_keys=True, default=list))


@pytest.fixture(scope="module")
def ref_cfg():
    with open(os.path.join(GOLD, "ref_configs.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def ref_fn():
    with open(os.path.join(GOLD, "ref_pure_functions.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLD, "ref_pure_functions.npz"))


# ---- configs ---------------------------------------------------------------------------------
def test_sampler_config_equals_reference(ref_cfg):
    from gaia_seg_amd.core.config import Config
    cfg = Config.fromfile(os.path.join(ROOT, "configs/supernet/pspnet_ar50to101v2.py"))
    ref = ref_cfg["model_samplers/ar50to101v2.py"]
    assert _j(cfg.train_sampler) == ref["train_sampler"]
    assert _j(cfg.val_sampler) == ref["val_sampler"]


def test_model_and_schedule_config_equal_reference(ref_cfg):
    from gaia_seg_amd.core.config import Config
    cfg = Config.fromfile(os.path.join(ROOT, "configs/supernet/pspnet_ar50to101v2.py"))
    ref = ref_cfg["models/pspnet_ar50to101v2_gsync.py"]
    assert _j(cfg.model) == ref["model"]
    for key in ("optimizer", "lr_config", "runner", "checkpoint_config", "crop_size", "test_cfg",
                "train_cfg", "optimizer_config"):
        assert _j(cfg[key]) == ref[key], key
    assert cfg.data["samples_per_gpu"] == ref["data"]["samples_per_gpu"] == 2
    assert cfg.evaluation["interval"] == ref["evaluation"]["interval"]


def test_reference_configs_build_the_same_search_space(ref_cfg):
    """The reference's sampler dicts, fed to this build's model_space, give the reference's anchors
    in order and random draws that stay on the reference's ranges."""
    from gaia_seg_amd.core.model_space import build_model_sampler
    ref = ref_cfg["model_samplers/ar50to101v2.py"]
    val = build_model_sampler(ref["val_sampler"])
    assert [m["name"] for m in val.traverse()] == ["R50", "R77", "R101"]
    assert val.traverse()[0] == ref["R50"] and val.traverse()[2] == ref["R101"]
    train = build_model_sampler(ref["train_sampler"])
    train.seed(0)
    names = set()
    wr, dr, sr = ref["body_width_range"], ref["body_depth_range"], ref["stem_width_range"]
    for _ in range(400):
        m = train.sample()
        names.add(m.get("name", "random"))
        if "name" in m:          # anchors need not lie on the random grid (R50's depth 6 does not)
            assert m == ref[m["name"]]
            continue
        w, d, s = (m["arch.backbone.body.width"], m["arch.backbone.body.depth"],
                   m["arch.backbone.stem.width"])
        for i in range(4):
            assert wr["start"][i] <= w[i] <= wr["end"][i] and (w[i] - wr["start"][i]) % wr["step"][i] == 0
            assert dr["start"][i] <= d[i] <= dr["end"][i] and (d[i] - dr["start"][i]) % dr["step"][i] == 0
        assert sr["start"] <= s <= sr["end"] and (s - sr["start"]) % sr["step"] == 0
        assert all(a <= b for a, b in zip(w, w[1:]))          # ascending=True
    assert names == {"MAX", "MIN", "R101", "R77", "R50", "random"}


def test_v1c_extract_config_builds(ref_cfg):
    """The reference's OS8 / deep-stem model dict builds through this build's registries with the
    reference's parameter count structure (state_dict keys of SURVEY.md Appendix C)."""
    from gaia_seg_amd.models import build_segmentor
    m = ref_cfg["extract_subnet/psp_ar50to101_v1c_extract.py"]["model"]
    m = json.loads(json.dumps(m))
    m["backbone"]["body_depth"] = [2, 2, 2, 2]   # keep the CPU test light: same code path
    model = build_segmentor(m)
    keys = set(model.state_dict())
    for k in ("backbone.stem.0.weight", "backbone.stem.7.running_var", "backbone.layer3.0.downsample.0.weight",
              "decode_head.psp_modules.3.1.conv.weight", "decode_head.bottleneck.bn.weight",
              "auxiliary_head.convs.0.conv.weight", "decode_head.conv_seg.bias"):
        assert k in keys, k
    assert model.backbone.layer3[1].conv2.dilation == 2 and model.backbone.layer3[0].conv2.dilation == 1
    assert model.backbone.layer4[1].conv2.dilation == 4 and model.backbone.layer4[0].conv2.dilation == 2


# ---- cross_entropy -----------------------------------------------------------------------------
def test_oracle_cross_entropy_equals_reference_bit_for_bit(ref_fn):
    from oracle import ops as O
    _, g = ref_fn
    for i in range(3):
        pred = torch.from_numpy(g["ce%d_pred" % i])
        label = torch.from_numpy(g["ce%d_label" % i])
        pw = torch.from_numpy(g["ce%d_pixel_weight" % i])
        cw = torch.from_numpy(g["ce%d_class_weight" % i])
        c = pred.shape[1]

        def eq(name, got):
            assert torch.equal(got, torch.from_numpy(g["ce%d_%s" % (i, name)])), (i, name)
        eq("mean", O.cross_entropy(pred, label))
        eq("wmean", O.cross_entropy(pred, label, weight=pw))
        eq("cwmean", O.cross_entropy(pred, label, weight=pw, class_weight=cw))
        eq("none", O.cross_entropy(pred, label, reduction="none"))
        eq("sum", O.cross_entropy(pred, label, weight=pw, reduction="sum"))
        eq("avg", O.cross_entropy(pred, label, weight=pw, avg_factor=float(pw.sum())))
        eq("ignore0", O.cross_entropy(pred, label.clamp(max=c - 1), ignore_index=0))


# ---- DL -> LD ----------------------------------------------------------------------------------
class _Rec:
    def __init__(self):
        self.got = []

    def __call__(self, meta):
        self.got.append(meta)


def test_manipulate_stem_and_body_slice_like_the_reference(ref_fn):
    from gaia_seg_amd.models import build_backbone
    from oracle.model import ODynamicResNet
    meta, _ = ref_fn
    for case in meta["manipulate"]:
        deep = case["deep_stem"]
        cfg = dict(type="DynamicResNet", in_channels=3, stem_width=[32, 32, 64] if deep else 64,
                   body_depth=[4, 6, 29, 4], body_width=[80, 160, 320, 640], deep_stem=deep,
                   conv_cfg=dict(type="DynConv2d"), norm_cfg=dict(type="DynBN"))
        bk = build_backbone(cfg)
        recs = {}
        targets = {("stem", i): bk.stem[i] for i in (0, 3, 6)} if deep else {("conv1", 0): bk.conv1}
        for i, name in enumerate(bk.res_layers):
            targets[("layer", i)] = getattr(bk, name)
        for key, mod in targets.items():
            recs[key] = _Rec()
            mod.manipulate_arch = recs[key]          # instance attribute shadows the method
        bk.manipulate_stem(case["stem_meta"])
        bk.manipulate_body(case["body_meta"])
        assert bk.stem_state == case["stem_state"] and bk.body_state == case["body_state"]
        if deep:
            for i in (0, 3, 6):
                assert recs[("stem", i)].got == case["stem_children"][str(i)]
        else:
            assert recs[("conv1", 0)].got == case["conv1"]
        for i in range(4):
            assert recs[("layer", i)].got == case["layers"][i]
        # the oracle ends in the states those per-child dicts describe
        ob = ODynamicResNet(3, cfg["stem_width"], cfg["body_width"], cfg["body_depth"], deep_stem=deep)
        ob.manipulate_arch({"stem": case["stem_meta"], "body": case["body_meta"]})
        for i, name in enumerate(ob.res_layers):
            want = case["layers"][i][0]
            layer = getattr(ob, name)
            if "depth" in want:
                assert layer.depth_state == want["depth"]
            if "width" in want:
                assert all(b.conv2.width_state == want["width"] and b.conv3.width_state == 4 * want["width"]
                           for b in layer)
        if deep:
            assert [ob.stem[i].width_state for i in (0, 3, 6)] == case["stem_meta"]["width"]
        else:
            assert ob.conv1.width_state == case["stem_meta"]["width"]


# ---- slide windows -----------------------------------------------------------------------------
def test_slide_window_grid_equals_reference(ref_fn):
    from gaia_seg_amd.core.inference import slide_windows as product_windows
    from oracle.inference import slide_windows
    meta, _ = ref_fn
    assert len(meta["slide_grids"]) >= 6
    for gcase in meta["slide_grids"]:
        args = (gcase["h_img"], gcase["w_img"], tuple(gcase["crop_size"]), tuple(gcase["stride"]))
        want = [tuple(w) for w in gcase["windows"]]
        assert slide_windows(*args) == want
        assert [tuple(w) for w in product_windows(*args)] == want
    c5 = meta["slide_grids"][0]
    assert (c5["h_img"], c5["w_img"]) == (1024, 2048) and len(c5["windows"]) == 9   # config 5: 3 x 3


def test_oracle_slide_accumulation_equals_reference(ref_fn):
    from oracle import inference as OI
    meta, g = ref_fn
    checked = 0
    for k, gcase in enumerate(meta["slide_grids"]):
        if "slide%d_preds" % k not in g:
            continue
        h_img, w_img = gcase["h_img"], gcase["w_img"]
        yy, xx = torch.meshgrid(torch.arange(h_img), torch.arange(w_img), indexing="ij")
        img = torch.stack([yy.float(), xx.float(), torch.zeros(h_img, w_img)])[None]
        calls = []

        class M:
            class decode_head:
                align_corners = False

            @staticmethod
            def encode_decode(crop):
                calls.append(1)
                base = crop[:, :1] * 0.001 + crop[:, 1:2] * 0.002 + len(calls)
                return torch.cat([base, base * 0.5, -base], dim=1)
        got = OI.slide_inference(M, img, (h_img, w_img, 3), tuple(gcase["crop_size"]),
                                 tuple(gcase["stride"]), rescale=False)
        assert torch.equal(got, torch.from_numpy(g["slide%d_preds" % k])), k
        checked += 1
    assert checked >= 4
