"""Generates tests/golden/ref_configs.json and tests/golden/ref_pure_functions.npz by RUNNING pieces of
the reference that need none of its absent dependencies (mmcv / mmseg / gaiavision).  Run in the
build container only (the reference does not travel; these data files do):

    python tests/golden/make_ref_pure_fixtures.py

1. ref_configs.json — the reference's dependency-free python config files, exec'd as they are:
     configs/_dynamic_/model_samplers/ar50to101v2.py      search ranges, anchors, sampler trees
     configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py model dims, optimizer, schedule, test_cfg
     configs/local_examples/extract_subnet/psp_ar50to101_v1c_extract.py   the OS8 / deep-stem model
2. ref_pure_functions.npz — outputs of functions cut out of the reference by AST (the file text is
   never copied: the function node is compiled in memory with its real free names bound) and run:
     cross_entropy                gaiaseg/models/losses/cross_entropy_loss.py:67-94, with
                                  weight_reduce_loss from the reference's own losses/utils.py
     manipulate_stem / _body      gaiaseg/models/backbones/dynamic_resnet.py:381-403 ("DL to LD")
                                  on recording stand-ins for the child modules
     slide_inference              gaiaseg/models/segmentors/dynamic_distiller.py:416-459 with a
                                  coordinate-coded image and a synthetic encode_decode: window list,
                                  accumulated / normalised predictions
"""
import ast
import importlib.util
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _exec_config(rel):
    scope = {}
    with open(os.path.join(REF, rel)) as f:
        exec(compile(f.read(), rel, "exec"), scope)
    return {k: v for k, v in scope.items() if not k.startswith("__")}


def _jsonable(o):
    if isinstance(o, dict):
        return {str(k): _jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_jsonable(v) for v in o]
    if isinstance(o, range):
        return list(o)
    return o


def _load_module(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _extract(rel, func, cls=None, namespace=None):
    """Compile one function (or method of `cls`) of a reference file and return it."""
    with open(os.path.join(REF, rel)) as f:
        tree = ast.parse(f.read())
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == func)
    node.decorator_list = []
    mod = ast.Module(body=[node], type_ignores=[])
    ns = dict(namespace or {})
    exec(compile(mod, rel, "exec"), ns)
    return ns[func]


class _Recorder:
    def __init__(self):
        self.got = []

    def manipulate_arch(self, meta):
        self.got.append(meta)


def configs_fixture():
    out = {}
    s = _exec_config("configs/_dynamic_/model_samplers/ar50to101v2.py")
    out["model_samplers/ar50to101v2.py"] = _jsonable(s)
    m = _exec_config("configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py")
    keep = ("model", "train_cfg", "test_cfg", "crop_size", "img_norm_cfg", "optimizer",
            "optimizer_config", "lr_config", "runner", "checkpoint_config", "evaluation", "data",
            "train_pipeline", "test_pipeline")
    out["models/pspnet_ar50to101v2_gsync.py"] = _jsonable({k: m[k] for k in keep if k in m})
    e = _exec_config("configs/local_examples/extract_subnet/psp_ar50to101_v1c_extract.py")
    out["extract_subnet/psp_ar50to101_v1c_extract.py"] = _jsonable(
        {k: e[k] for k in ("model", "train_cfg", "test_cfg") if k in e})
    with open(os.path.join(HERE, "ref_configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote ref_configs.json:", {k: sorted(v)[:6] for k, v in out.items()})


def functions_fixture():
    out = {}
    meta = {}
    g = torch.Generator().manual_seed(4321)

    # ---- cross_entropy ----
    utils = _load_module(os.path.join(REF, "gaiaseg/models/losses/utils.py"), "ref_loss_utils2")
    ce = _extract("gaiaseg/models/losses/cross_entropy_loss.py", "cross_entropy",
                  namespace=dict(torch=torch, F=F, weight_reduce_loss=utils.weight_reduce_loss))
    for i, (n, c, h, w) in enumerate([(2, 19, 9, 13), (1, 5, 6, 6), (3, 19, 8, 8)]):
        pred = torch.randn(n, c, h, w, generator=g) * 3
        label = torch.randint(0, c, (n, h, w), generator=g)
        label[0, :2] = 255
        label[-1, :, -1] = 255
        pw = (torch.rand(n, h, w, generator=g) > 0.3).float()
        cw = torch.rand(c, generator=g) + 0.5
        out["ce%d_pred" % i] = pred.numpy()
        out["ce%d_label" % i] = label.numpy()
        out["ce%d_pixel_weight" % i] = pw.numpy()
        out["ce%d_class_weight" % i] = cw.numpy()
        out["ce%d_mean" % i] = ce(pred, label).numpy()
        out["ce%d_wmean" % i] = ce(pred, label, weight=pw).numpy()
        out["ce%d_cwmean" % i] = ce(pred, label, weight=pw, class_weight=cw).numpy()
        out["ce%d_none" % i] = ce(pred, label, reduction="none").numpy()
        out["ce%d_sum" % i] = ce(pred, label, weight=pw, reduction="sum").numpy()
        out["ce%d_avg" % i] = ce(pred, label, weight=pw, avg_factor=float(pw.sum())).numpy()
        out["ce%d_ignore0" % i] = ce(pred, label.clamp(max=c - 1), ignore_index=0).numpy()

    # ---- DL -> LD slicing of manipulate_stem / manipulate_body ----
    stem_fn = _extract("gaiaseg/models/backbones/dynamic_resnet.py", "manipulate_stem", cls="DynamicResNet")
    body_fn = _extract("gaiaseg/models/backbones/dynamic_resnet.py", "manipulate_body", cls="DynamicResNet")
    cases = []
    for deep, stem_meta, body_meta in [
            (False, {"width": 48}, {"width": [64, 128, 256, 512], "depth": [3, 4, 6, 3]}),
            (True, {"width": [16, 16, 32]}, {"width": [48, 96, 192, 384], "depth": [2, 2, 5, 2]}),
            (False, {"width": 64}, {"depth": [4, 6, 29, 4]}),
            (True, {"width": [32, 32, 64]}, {"width": [80, 160, 320, 640]})]:
        class Fake:
            pass
        fk = Fake()
        fk.deep_stem = deep
        fk.stem = [_Recorder() for _ in range(9)]
        fk.conv1 = _Recorder()
        fk.res_layers = ["layer1", "layer2", "layer3", "layer4"]
        for name in fk.res_layers:
            setattr(fk, name, _Recorder())
        stem_fn(fk, stem_meta)
        body_fn(fk, body_meta)
        cases.append(dict(deep_stem=deep, stem_meta=stem_meta, body_meta=body_meta,
                          stem_state=fk.stem_state, body_state=fk.body_state,
                          stem_children={str(i): r.got for i, r in enumerate(fk.stem) if r.got},
                          conv1=fk.conv1.got,
                          layers=[getattr(fk, n).got for n in fk.res_layers]))
    meta["manipulate"] = cases

    # ---- slide_inference: window grid + accumulate / normalise ----
    slide = _extract("gaiaseg/models/segmentors/dynamic_distiller.py", "slide_inference",
                     cls="DynamicDistiller" if _has_class("gaiaseg/models/segmentors/dynamic_distiller.py",
                                                          "DynamicDistiller") else None,
                     namespace=dict(torch=torch, F=F, resize=None))
    grids = []
    for k, (h_img, w_img, crop, stride) in enumerate([
            (1024, 2048, (512, 1024), (341, 683)),     # BASELINE config 5
            (96, 128, (64, 64), (40, 40)),
            (769, 769, (769, 769), (513, 513)),
            (50, 70, (64, 64), (32, 32)),              # crop larger than the image
            (100, 100, (33, 47), (17, 29)),
            (65, 129, (64, 64), (64, 64))]):
        class Cfg:
            pass
        wins = []

        class Seg:
            num_classes = 3
            test_cfg = Cfg()

            def encode_decode(self, crop_img, img_meta):
                y1, x1 = int(crop_img[0, 0, 0, 0]), int(crop_img[0, 1, 0, 0])
                hh, ww = crop_img.shape[2:]
                wins.append([y1, y1 + hh, x1, x1 + ww])
                # logits: a function of the window index and of the pixel, so that the accumulated
                # result depends on which windows cover a pixel
                base = crop_img[:, :1] * 0.001 + crop_img[:, 1:2] * 0.002 + len(wins)
                return torch.cat([base, base * 0.5, -base], dim=1)
        Seg.test_cfg.stride = stride
        Seg.test_cfg.crop_size = crop
        yy, xx = torch.meshgrid(torch.arange(h_img), torch.arange(w_img), indexing="ij")
        img = torch.stack([yy.float(), xx.float(), torch.zeros(h_img, w_img)])[None]
        preds = slide(Seg(), img, [dict(ori_shape=(h_img, w_img, 3))], False)
        grids.append(dict(h_img=h_img, w_img=w_img, crop_size=list(crop), stride=list(stride),
                          windows=wins))
        if h_img * w_img <= 100 * 130:
            out["slide%d_preds" % k] = preds.numpy()
    meta["slide_grids"] = grids

    np.savez_compressed(os.path.join(HERE, "ref_pure_functions.npz"), **out)
    with open(os.path.join(HERE, "ref_pure_functions.json"), "w") as f:
        json.dump(_jsonable(meta), f, indent=1, sort_keys=True)
    print("wrote ref_pure_functions.npz (%d arrays) and .json" % len(out))


def _has_class(rel, name):
    with open(os.path.join(REF, rel)) as f:
        tree = ast.parse(f.read())
    return any(isinstance(n, ast.ClassDef) and n.name == name for n in tree.body)


if __name__ == "__main__":
    configs_fixture()
    functions_fixture()
