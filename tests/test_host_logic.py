"""Host-side mirror of the reference interface (CPU only): registries, configs, arch plumbing,
state_dict names, samplers, parameter arena planning."""
import os
import random

import pytest
import torch

from gaia_seg_amd.core.config import Config, DictAction
from gaia_seg_amd.core.dynamic import fold_dict, unfold_dict
from gaia_seg_amd.core.model_space import arch_key, build_model_sampler
from gaia_seg_amd.models import BACKBONES, HEADS, LOSSES, SEGMENTORS, build_segmentor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "supernet", "pspnet_ar50to101v2.py")


def test_registered_names_match_reference():
    # names the reference configs use
    assert "DynamicEncoderDecoder" in SEGMENTORS
    assert "DynamicResNet" in BACKBONES
    for h in ("DynamicFCNHead", "DynamicPSPHead", "DynamicUPerHead"):
        assert h in HEADS
    assert "CrossEntropyLoss" in LOSSES
    from gaia_seg_amd.core.bricks import CONV_LAYERS, NORM_LAYERS
    assert "DynConv2d" in CONV_LAYERS
    for n in ("DynBN", "DynSyncBN", "SyncBN", "BN"):
        assert n in NORM_LAYERS


def test_config_base_merge_and_cfg_options():
    cfg = Config.fromfile(CFG)
    assert cfg.model.type == "DynamicEncoderDecoder"
    assert cfg.model.decode_head.type == "DynamicPSPHead"
    assert cfg.optimizer.lr == 0.01 and cfg.lr_config.policy == "poly"
    assert cfg.train_sampler.type == "concat"
    cfg.merge_from_dict(DictAction.parse(["optimizer.lr=0.02", "model.backbone.out_indices=(2,3)",
                                          "data.samples_per_gpu=4"]))
    assert cfg.optimizer.lr == 0.02 and cfg.optimizer.momentum == 0.9
    assert cfg.model.backbone.out_indices == (2, 3) and cfg.data.samples_per_gpu == 4


@pytest.fixture(scope="module")
def psp_model():
    cfg = Config.fromfile(CFG)
    return build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))


def test_supernet_dimensions_and_state_dict_names(psp_model):
    m = psp_model
    n_b = sum(p.numel() for p in m.backbone.parameters())
    n_d = sum(p.numel() for p in m.decode_head.parameters())
    n_a = sum(p.numel() for p in m.auxiliary_head.parameters())
    # 84.78 M + 26.49 M + 2.95 M = 114.2 M
    assert abs(n_b - 84.78e6) < 0.01e6 and abs(n_d - 26.49e6) < 0.01e6 and abs(n_a - 2.95e6) < 0.01e6
    sd = m.state_dict()
    for k in ["backbone.conv1.weight", "backbone.bn1.running_mean", "backbone.layer3.28.conv2.weight",
              "backbone.layer2.0.downsample.0.weight", "backbone.layer2.0.downsample.1.running_var",
              "decode_head.psp_modules.3.1.conv.weight", "decode_head.psp_modules.0.1.bn.weight",
              "decode_head.bottleneck.conv.weight", "decode_head.conv_seg.bias",
              "auxiliary_head.convs.0.conv.weight", "auxiliary_head.conv_seg.weight"]:
        assert k in sd, k
    # logical OIHW shapes at MAX size, physical HWIO storage
    w = m.backbone.layer4[0].conv2.weight
    assert tuple(w.shape) == (640, 640, 3, 3) and w.stride(0) == 1 and w.stride(1) == 640
    ws = m.decode_head.conv_seg.weight
    assert tuple(ws.shape) == (19, 512, 1, 1) and ws.stride(1) == 20   # class dim padded to 20
    assert tuple(sd["decode_head.conv_seg.weight"].contiguous().shape) == (19, 512, 1, 1)


def test_manipulate_arch_plumbing(psp_model):
    m = psp_model
    meta = {"name": "R50", "arch.backbone.stem.width": 64, "arch.backbone.body.width": [64, 128, 256, 512],
            "arch.backbone.body.depth": [3, 4, 6, 3]}
    m.manipulate_arch(fold_dict(meta)["arch"])
    b = m.backbone
    assert b.conv1.width_state == 64
    assert [getattr(b, n).depth_state for n in b.res_layers] == [3, 4, 6, 3]
    blk = b.layer3[28]                      # inactive blocks still get the width (reference fan-out)
    assert blk.conv1.width_state == 256 and blk.conv3.width_state == 1024
    assert b.layer2[0].downsample[0].width_state == 512
    n_active = sum(p.numel() for p in m.active_parameters())
    assert n_active < sum(p.numel() for p in m.parameters())
    with pytest.raises(AssertionError):
        b.layer1.manipulate_depth(0)
    with pytest.raises(KeyError):
        m.manipulate_arch({"roi_head": {}})
    m.manipulate_arch({"decode_head": {"anything": 1}})   # no-op like the reference


def test_fold_unfold_roundtrip():
    flat = {"name": "x", "arch.backbone.stem.width": 32, "arch.backbone.body.depth": [2, 2, 5, 2]}
    nested = fold_dict(flat)
    assert nested["arch"]["backbone"]["body"]["depth"] == [2, 2, 5, 2]
    assert unfold_dict(nested) == flat


def test_samplers_cover_search_space_and_are_seedable():
    cfg = Config.fromfile(CFG)
    s = build_model_sampler(cfg.train_sampler)
    s.seed(0)
    a = [s.sample() for _ in range(200)]
    s.seed(0)
    b = [s.sample() for _ in range(200)]
    assert a == b
    names = {m.get("name", "random") for m in a}
    assert {"MAX", "MIN", "R50", "R77", "R101", "random"} <= names
    for m in a:
        w, d = m["arch.backbone.body.width"], m["arch.backbone.body.depth"]
        assert all(x <= y for x, y in zip(w, w[1:]))                  # ascending
        assert 48 <= w[0] <= 80 and 384 <= w[3] <= 640 and w[2] % 64 == 0
        assert 2 <= d[0] <= 4 and 5 <= d[2] <= 29
        if "name" not in m:                                              # random draw: 5,7,..,29
            assert d[2] % 2 == 1
        assert m["arch.backbone.stem.width"] in (32, 48, 64)
    v = build_model_sampler(cfg.val_sampler)
    assert [v.anchor_name(i) for i in range(len(v))] == ["R50", "R77", "R101"]
    assert len(v.traverse()) == 3
    assert arch_key(a[0]) == arch_key(dict(a[0], name="other"))      # name is not part of the key


def test_range_sampler_traverse_counts_supernet_size():
    # 3 stem x 3^4 widths x (3*3*13*3) depths = 85293 subnets
    cfg = Config.fromfile(CFG)
    comp = build_model_sampler(cfg.train_sampler).model_samplers[1].model_sampler
    n = 1
    for c in comp.model_samplers:
        n *= len(c.traverse())
    assert n == 85293


def test_grad_reducer_plan_covers_active_ranges_once():
    from gaia_seg_amd.core.dist import GradReducer
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (64, 128, 64, 256, 64)]
    segs, off = {}, 0
    for p in params:
        segs[id(p)] = (off, p.numel())
        off += p.numel()
    flat = torch.zeros(off)
    red = GradReducer(flat, segs, bucket_bytes=4 * 200)
    active = [params[0], params[1], params[3], params[4]]      # params[2] is a skipped block
    plan = red._plan(active, key="k")
    covered = sorted(r for b in plan for r in b["runs"])
    assert covered == [(0, 192), (256, 576)]
    red.begin(active, "k")                                     # world size 1: nothing is armed
    assert not any(hasattr(p, "_gs_grad_ready") for p in active)
    red.finish()                                               # ... and nothing to wait for


def test_grad_reducer_pads_small_holes_into_one_collective():
    """Depth-skipped blocks leave holes in a bucket; holes worth < 5 % of the bucket are reduced along
    with it (their gradients are zero on every rank), smallest first, so the bucket is one collective."""
    from gaia_seg_amd.core.dist import GradReducer
    pad = GradReducer.pad_holes
    assert pad([(0, 1000)], 0.05) == [(0, 1000)]
    assert pad([(0, 1000), (1010, 2000)], 0.05) == [(0, 2000)]                  # hole 10 of 1990
    assert pad([(0, 1000), (1200, 2000)], 0.05) == [(0, 1000), (1200, 2000)]    # hole 200 > 90
    # the budget is shared: holes 30 + 40 fit into 5 % of 1930, the third (60) does not
    assert pad([(0, 500), (530, 1000), (1040, 1500), (1560, 2060)], 0.05) == [(0, 1500), (1560, 2060)]
    assert pad([(600, 700), (0, 500)], 0.0) == [(0, 500), (600, 700)]
    # through the planner: a 16-element skipped parameter inside a 1-bucket plan
    params = [torch.nn.Parameter(torch.zeros(n)) for n in (512, 16, 512)]
    segs, off = {}, 0
    for p in params:
        segs[id(p)] = (off, p.numel())
        off += p.numel()
    red = GradReducer(torch.zeros(off), segs, bucket_bytes=1 << 20)
    assert red._plan([params[0], params[2]], key=None)[0]["runs"] == [(0, 1040)]
    red0 = GradReducer(torch.zeros(off), segs, bucket_bytes=1 << 20, hole_frac=0.0)
    assert red0._plan([params[0], params[2]], key=None)[0]["runs"] == [(0, 512), (528, 1040)]
    assert not red._coalescing()      # CPU tensors / no process group: one call per run


def test_closed_form_flops_match_baseline_table(psp_model):
    """BASELINE.md §2 (OS32, 512x1024, per image): backbone GF (3x3 part) and params."""
    from gaia_seg_amd.core.flops import model_flops
    want = {"MIN": (34.8, 14.9, 9.73, 17.0, 7.3), "R50": (85.4, 38.7, 23.51, 19.4, 9.7),
            "R101": (163.0, 79.7, 42.50, 19.4, 9.7), "MAX": (324.2, 162.3, 84.78, 21.9, 12.1)}
    cfg = Config.fromfile(CFG)
    anchors = {a["name"]: a for a in build_model_sampler(cfg.train_sampler).model_samplers[0].anchors}
    for name, (b, k3, params, psp, aux) in want.items():
        psp_model.manipulate_arch(fold_dict(anchors[name])["arch"])
        f = model_flops(psp_model, 512, 1024)
        assert abs(f["backbone"] / 1e9 - b) < 0.06, (name, f["backbone"] / 1e9)
        assert abs(f["backbone_3x3"] / 1e9 - k3) < 0.06
        assert abs(f["backbone_params"] / 1e6 - params) < 0.02
        assert abs(f["decode"] / 1e9 - psp) < 0.06 and abs(f["aux"] / 1e9 - aux) < 0.06


def test_closed_form_flops_of_the_uper_head():
    """BASELINE config 4 (UPerNet, R101 anchor, 769 x 769; SURVEY.md section 8 a16: ~964 GF per image):
    the head's closed form against the same sum written out from the level sizes
    (dynamic_uper_head.py:81-131: PPM + bottleneck on C5, laterals and fpn convs on the other three
    levels, fpn_bottleneck and the classifier at level 0)."""
    from gaia_seg_amd.core.flops import model_flops
    from gaia_seg_amd.models import build_segmentor
    cfg = Config.fromfile(os.path.join(ROOT, "configs", "supernet", "upernet_ar50to101v2.py"))
    model = build_segmentor(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    anchors = {a["name"]: a for a in build_model_sampler(cfg.train_sampler).model_samplers[0].anchors}
    model.manipulate_arch(fold_dict(anchors["R101"])["arch"])
    f = model_flops(model, 769, 769)
    lv = [(256, 193), (512, 97), (1024, 49), (2048, 25)]            # (channels, side) of C2..C5
    ch = 512
    want = sum(2.0 * s * s * ch * 2048 for s in (1, 2, 3, 6))       # PPM 1x1 convs
    want += 2.0 * 25 * 25 * ch * (2048 + 4 * ch) * 9                # bottleneck 3x3
    want += sum(2.0 * n * n * ch * c for c, n in lv[:3])            # laterals 1x1
    want += sum(2.0 * n * n * ch * ch * 9 for _, n in lv[:3])       # fpn convs 3x3
    want += 2.0 * 193 * 193 * ch * (4 * ch) * 9                     # fpn_bottleneck 3x3
    want += 2.0 * 193 * 193 * 19 * ch                               # conv_seg
    assert abs(f["decode"] - want) / want < 1e-12, (f["decode"], want)
    assert abs(f["decode"] / 1e9 - 964) / 964 < 0.02                # SURVEY's rounded figure
    assert f["total"] == f["backbone"] + f["decode"] + f["aux"]


def test_bn_calibration_switches():
    """cfg.caliberate_bn: reset_stats before training (gaiaseg/apis/train.py:177-184) and
    use_minibatch_stats at test time (tools/test_supernet.py:190-198)."""
    import torch
    from gaia_seg_amd.apis.test import apply_bn_calibration
    from gaia_seg_amd.core.bricks import DynamicBatchNorm2d
    net = torch.nn.Sequential(DynamicBatchNorm2d(8), torch.nn.ReLU(), DynamicBatchNorm2d(4))
    for m in (net[0], net[2]):
        m.running_mean.normal_()
        m.running_var.uniform_(0.5, 2.0)
    assert apply_bn_calibration(net, None, "train") == 0
    assert apply_bn_calibration(net, dict(reset_stats=True), "test") == 0      # wrong phase: no-op
    assert apply_bn_calibration(net, dict(reset_stats=True), "train") == 2
    assert float(net[0].running_mean.abs().max()) == 0.0 and float((net[2].running_var - 1).abs().max()) == 0.0
    # an eval forward BEFORE the calibration fills the per-mode BNParams cache with the running
    # buffers; the calibration must drop it, or the BN keeps normalising with the old statistics
    net.eval()
    assert net[0].bn_params(8).running_mean is not None and "_bnp_cache" in net[0].__dict__
    assert apply_bn_calibration(net, dict(use_minibatch_stats=True), "test") == 2
    assert net[0].running_mean is None and net[0].track_running_stats is False
    assert "_bnp_cache" not in net[0].__dict__
    # eval-mode BN without running statistics normalises with the batch it sees
    assert net[0].eval().bn_params(8).running_mean is None


def test_arena_zero_grad_clears_unless_the_runner_vouches_for_it():
    """ParamArena.zero_grad skips the fill only for the caller that owns the "gradients are clean"
    invariant (IterBasedRunner: trust_clean=True); any other caller gets a real clear even while the
    flag is set (a manual backward outside the runner must not leak stale gradients)."""
    import torch
    from gaia_seg_amd.core.param_arena import ParamArena
    arena = ParamArena.__new__(ParamArena)        # (the constructor needs the GPU; zero_grad does not)
    arena.flat_grad = torch.full((256,), 3.0)
    arena.grads_clean = True
    arena.zero_grad(trust_clean=True)
    assert float(arena.flat_grad.abs().max()) == 3.0            # the runner's no-op
    first = [(0, 64), (128, 192)]
    arena.zero_grad(first)
    for a, b in first:
        assert float(arena.flat_grad[a:b].abs().max()) == 0.0   # only the ranges asked for
    assert float(arena.flat_grad.abs().max()) == 3.0
    arena.zero_grad()
    assert float(arena.flat_grad.abs().max()) == 0.0


def test_train_pipeline_random_decisions_follow_the_cpu_transforms():
    """draw_train_params consumes numpy's RandomState in the order mmseg's transforms do
    (Resize.random_sample_ratio, RandomCrop.get_crop_bbox, RandomFlip, PhotoMetricDistortion)."""
    import numpy as np
    from gaia_seg_amd.datasets.gpu_pipeline import draw_train_params, rescale_size
    assert rescale_size(1024, 2048, (2048, 1024)) == (1024, 2048)
    assert rescale_size(1024, 2048, (1024, 512)) == (512, 1024)
    assert rescale_size(300, 500, (int(2048 * 0.7), int(1024 * 0.7))) == (716, 1193)  # short edge binds
    cfg = dict(crop_size=(512, 1024), img_scale=(2048, 1024), ratio_range=(0.5, 2.0), cat_max_ratio=1.0)
    rng = np.random.RandomState(0)
    p = draw_train_params(rng, 1024, 2048, cfg)
    ref = np.random.RandomState(0)
    ratio = ref.random_sample() * 1.5 + 0.5
    rh, rw = rescale_size(1024, 2048, (int(2048 * ratio), int(1024 * ratio)))
    assert (p["res_h"], p["res_w"]) == (rh, rw)
    oy = ref.randint(0, max(rh - 512, 0) + 1)
    ox = ref.randint(0, max(rw - 1024, 0) + 1)
    assert (p["crop_y"], p["crop_x"]) == (oy, ox)
    assert p["flip"] == bool(ref.rand() < 0.5)
    assert p["pm_brightness"] == bool(ref.randint(2))
    for seed in range(50):          # invariants over many draws
        p = draw_train_params(np.random.RandomState(seed), 1024, 2048, cfg)
        assert 512 <= p["res_h"] <= 2048 and p["res_w"] == 2 * p["res_h"]
        assert p["crop_y"] + p["crop_h"] <= p["res_h"] and p["crop_x"] + p["crop_w"] <= p["res_w"]
        assert p["crop_h"] == 512 and p["crop_w"] == 1024
        assert -32 <= p["pm_delta"] <= 32 and 0.5 <= p["pm_alpha"] <= 1.5 and -18 <= p["pm_hue_delta"] < 18
        assert p["pm_contrast_first"] in (True, False)


def test_oracle_pipeline_identity_and_pad():
    import numpy as np
    from oracle.pipeline import bgr2hsv, hsv2bgr, train_sample
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(40, 60, 3)).astype(np.uint8)
    lab = rng.randint(0, 19, size=(40, 60)).astype(np.uint8)
    p = dict(res_h=40, res_w=60, crop_y=0, crop_x=0, crop_h=40, crop_w=60, flip=False, pm_enable=False)
    out, ol = train_sample(img, lab, p, crop_size=(48, 64), mean=(0, 0, 0), std=(1, 1, 1), to_rgb=False)
    assert np.array_equal(out[:, :40, :60], img.transpose(2, 0, 1).astype(np.float32))
    assert float(np.abs(out[:, 40:]).max()) == 0.0 and int((ol[40:] != 255).sum()) == 0
    assert np.array_equal(ol[:40, :60], lab)
    # 8-bit HSV round trip stays within the quantisation of H (2 degrees) and S
    back = hsv2bgr(bgr2hsv(img.astype(np.float32)))
    assert float(np.abs(back - img).max()) <= 6
    grey = np.full((2, 2, 3), 77, np.float32)
    assert np.array_equal(hsv2bgr(bgr2hsv(grey)), grey)


def test_training_step_leaves_no_reference_cycles():
    """A finished step must release its activations by reference counting alone: a cycle (r02 had
    Act.bnb tuples that contained their own activation) keeps device memory alive until Python's
    cyclic collector runs and made the caching allocator grow from 7.7 to 25 GB in 60 steps.
    tests/host_dry_run.py runs one R50 step of the FCN supernet with the C-ABI stubbed out (in a
    subprocess: the stubs patch the library module) and reports what gc.collect() finds."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for cfg in ("fcn", "pspnet", "upernet"):
        res = subprocess.run([sys.executable, os.path.join(here, "host_dry_run.py"), "2",
                              "configs/supernet/%s_ar50to101v2.py" % cfg],
                             capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stderr[-2000:]
        out = json.loads(res.stdout.strip().splitlines()[-1])
        assert out["unreachable_per_step"] == 0, (cfg, out)


def test_range_arithmetic_of_the_optimizer_instalments():
    """core/runner.py splits the active arena ranges into instalments: what the early step updated is
    subtracted from what remains.  Disjoint sorted [begin, end) lists."""
    from gaia_seg_amd.core.runner import _ranges_intersect, _ranges_subtract
    a = [(0, 10), (20, 30), (40, 50)]
    assert _ranges_subtract(a, []) == a
    assert _ranges_subtract(a, [(0, 50)]) == []
    assert _ranges_subtract(a, [(5, 25)]) == [(0, 5), (25, 30), (40, 50)]
    assert _ranges_subtract(a, [(2, 3), (4, 6), (28, 45)]) == [(0, 2), (3, 4), (6, 10), (20, 28), (45, 50)]
    assert _ranges_intersect(a, [(5, 25), (45, 60)]) == [(5, 10), (20, 25), (45, 50)]
    assert _ranges_intersect(a, []) == []
    # a partition: (a - b) and (a & b) tile a
    b = [(3, 22), (29, 41)]
    parts = sorted(_ranges_subtract(a, b) + _ranges_intersect(a, b))
    assert sum(e - s for s, e in parts) == sum(e - s for s, e in a)
    assert all(x[1] <= y[0] for x, y in zip(parts, parts[1:]))


def test_bench_plan_only_needs_no_gpu():
    """`bench.py --plan-only --gpus 8`: the data-parallel exchange plan from host arithmetic alone
    (bucket planner over the arena layout); one JSON object, no GPU touched."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--plan-only", "--gpus", "8",
                        "--steps", "4"], capture_output=True, text=True, cwd=root,
                       env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["plan_only"] and d["n_gpus"] == 8 and d["arena_bytes"] > 4e8
    per = d["per_step"]
    assert set(per) >= {"MIN", "R50", "MAX", "mix_of_4_draws_mean"}
    for name in ("MIN", "R50", "MAX"):
        p = per[name]
        assert p["collective_launches"] == p["buckets"] == len(p["bucket_bytes"])
        assert sum(p["bucket_bytes"]) == p["allreduce_bytes"] >= p["active_parameter_bytes"]
        assert p["padded_hole_bytes"] <= 0.05 * p["allreduce_bytes"] + 1
        assert p["syncbn_collectives"] == 2 * p["syncbn_layers"] and p["expected_allreduce_ms"] > 0
    assert per["MIN"]["allreduce_bytes"] < per["R50"]["allreduce_bytes"] < per["MAX"]["allreduce_bytes"]
    assert per["MAX"]["allreduce_bytes"] == d["arena_bytes"]      # MAX uses every parameter
