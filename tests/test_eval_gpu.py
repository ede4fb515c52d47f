"""Cross-arch evaluation: confusion-matrix kernel vs numpy, and the eval
hook's per-anchor mIoU against predictions computed through the oracle."""
import numpy as np
import pytest
import torch

from util_models import arch_meta, fcn_head, make_pair, model_cfg

pytestmark = pytest.mark.gpu


def test_confusion_matrix_kernel_exact(hip_lib):
    from gaia_seg_amd.core.evaluation import confusion_matrix, metrics_from_confusion
    rng = np.random.RandomState(0)
    for C in (19, 150):
        pred = rng.randint(0, C, size=(2, 97, 131))
        label = rng.randint(0, C, size=(2, 97, 131))
        label[:, :5] = 255
        want = np.zeros((C, C), dtype=np.int64)
        m = label != 255
        np.add.at(want, (label[m], pred[m]), 1)
        got = confusion_matrix(torch.from_numpy(pred).cuda(), torch.from_numpy(label).cuda(), C)
        got = confusion_matrix(torch.from_numpy(pred).cuda(), torch.from_numpy(label).cuda(), C, out=got)
        assert np.array_equal(got.cpu().numpy(), 2 * want)
        res = metrics_from_confusion(got)
        iou = np.diag(want) / (want.sum(0) + want.sum(1) - np.diag(want))
        assert abs(res["mIoU"] - np.nanmean(iou)) < 1e-9


def test_cross_arch_eval_hook_matches_oracle(hip_lib):
    from gaia_seg_amd.core.evaluation import CrossArchEvalHook, metrics_from_confusion
    from gaia_seg_amd.core.model_space import build_model_sampler
    from gaia_seg_amd.core.synthetic import make_batch
    prod, orc = make_pair(model_cfg(fcn_head(), aux=True))
    prod = prod.cuda()
    batches = [make_batch(2, 64, 96, seed=s, device="cuda", border=2) for s in range(2)]
    anchors = []
    for name in ("sub", "max"):
        a = arch_meta(name)["backbone"]
        anchors.append({"name": name, "arch.backbone.stem.width": a["stem"]["width"],
                        "arch.backbone.body.width": a["body"]["width"],
                        "arch.backbone.body.depth": a["body"]["depth"]})
    sampler = build_model_sampler(dict(type="anchor", anchors=anchors))

    from gaia_seg_amd.core.dist import GradReducer
    from gaia_seg_amd.core.param_arena import ParamArena
    from gaia_seg_amd.core.runner import IterBasedRunner
    arena = ParamArena(prod)
    runner = IterBasedRunner(prod, arena, GradReducer(arena.flat_grad, arena.segments))
    hook = CrossArchEvalHook(batches, sampler, interval=1, num_batches=2)
    out = hook.evaluate(runner)
    orc.eval()
    for name in ("sub", "max"):
        orc.manipulate_arch(arch_meta(name))
        conf = np.zeros((19, 19), dtype=np.int64)
        with torch.no_grad():
            for b in batches:
                pred = orc.encode_decode(b["img"].cpu()).argmax(1).numpy()
                lab = b["gt_semantic_seg"].squeeze(1).cpu().numpy()
                m = lab != 255
                np.add.at(conf, (lab[m], pred[m]), 1)
        want = metrics_from_confusion(torch.from_numpy(conf))
        assert abs(out[name]["mIoU"] - want["mIoU"]) < 2e-3 and abs(out[name]["aAcc"] - want["aAcc"]) < 2e-3
