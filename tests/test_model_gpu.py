"""End-to-end parity of the MI355X model path (HIP kernels through the C-ABI) against the CPU
oracle on the same seeded inputs: features, losses, accuracy, parameter gradients, BN running
statistics — for the three decode heads, 7x7 and deep stems, OS32 and OS8 (dilated) backbones and
several subnets of one supernet.  Tolerance from BASELINE.json: 1e-3 relative (fp32)."""
import pytest
import torch

from conftest import rel_err
from util_models import (arch_meta, fcn_head, make_batch, make_pair, model_cfg, psp_head, uper_head)

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _run_pair(cfg, arch, deep_stem=False, size=(64, 96), check_grads=True):
    prod, orc = make_pair(cfg)
    prod = prod.cuda().train()
    orc.train()
    meta = arch_meta(arch, deep_stem)
    prod.manipulate_arch(meta)
    orc.manipulate_arch(meta)
    img, gt = make_batch(2, *size)
    losses_o = orc.forward_train(img, gt)
    loss_o, _ = orc.parse_losses(losses_o)
    loss_o.backward()
    metas = [dict(ori_shape=size + (3,), img_shape=size + (3,), flip=False) for _ in range(2)]
    out = prod.train_step(dict(img=img.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda()), None)
    out["loss"].backward()
    errs = {}
    for k, v in losses_o.items():
        errs[k] = abs(float(out["log_vars"][k]) - float(v)) / max(abs(float(v)), 1e-6)
    errs["loss"] = abs(float(out["loss"]) - float(loss_o)) / abs(float(loss_o))
    if check_grads:
        op = dict(orc.named_parameters())
        n_checked = 0
        for name, p in prod.named_parameters():
            go = op[name].grad
            if go is None or float(go.abs().max()) == 0.0:
                # unused by this subnet: the product must not have produced a gradient either
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
                continue
            assert p.grad is not None, name
            errs["grad:" + name] = rel_err(p.grad, go)
            n_checked += 1
        assert n_checked > 10
        ob = dict(orc.named_buffers())
        for name, b in prod.named_buffers():
            if name.endswith("running_mean") or name.endswith("running_var"):
                errs["buf:" + name] = rel_err(b, ob[name])
            elif name.endswith("num_batches_tracked"):
                assert int(b) == int(ob[name]), name
    bad = {k: v for k, v in errs.items() if not v < TOL}
    assert not bad, bad
    return errs


@pytest.mark.parametrize("arch", ["max", "sub", "min"])
def test_fcn_supernet_train_step(hip_lib, arch):
    _run_pair(model_cfg(fcn_head(), aux=True), arch)


def test_fcn_deep_stem_os8(hip_lib):
    _run_pair(model_cfg(fcn_head(), aux=True, deep_stem=True, os8=True), "sub", deep_stem=True)


@pytest.mark.parametrize("arch", ["max", "sub"])
def test_psp_supernet_train_step(hip_lib, arch):
    _run_pair(model_cfg(psp_head(), aux=True), arch)


def test_psp_os8(hip_lib):
    _run_pair(model_cfg(psp_head(), aux=True, os8=True), "sub", size=(64, 64))


@pytest.mark.parametrize("arch", ["max", "sub"])
def test_uper_supernet_train_step(hip_lib, arch):
    _run_pair(model_cfg(uper_head(), aux=False), arch, size=(97, 97))


def test_backbone_features_and_depth_prefix(hip_lib):
    """depth d == first d blocks; features equal the oracle's on every output level."""
    prod, orc = make_pair(model_cfg(fcn_head(), aux=False))
    prod = prod.cuda().eval()
    orc.eval()
    img, _ = make_batch(2, 64, 64)
    for arch in ["max", "sub", "min"]:
        meta = arch_meta(arch)
        prod.manipulate_arch(meta)
        orc.manipulate_arch(meta)
        with torch.no_grad():
            fp = prod.extract_feat(img.cuda())
            fo = orc.backbone(img)
        for a, b in zip(fp, fo):
            assert a.shape == b.shape
            assert rel_err(a, b) < TOL
