"""Training-loop state on the GPU: frozen parameters, the subnet after a cross-arch evaluation,
optimizer state in checkpoints (r01 advisor findings, each with the reference behaviour it must
match)."""
import copy
import os

import pytest
import torch

from util_models import arch_meta, fcn_head, make_batch, model_cfg

pytestmark = pytest.mark.gpu


def _runner(model, **kw):
    from gaia_seg_amd.core.dist import GradReducer
    from gaia_seg_amd.core.param_arena import ParamArena
    from gaia_seg_amd.core.runner import ArenaOptimizerHook, IterBasedRunner
    arena = ParamArena(model)
    runner = IterBasedRunner(model, arena, GradReducer(arena.flat_grad, arena.segments), base_lr=0.05,
                             momentum=0.9, weight_decay=5e-4, max_iters=100, **kw)
    runner.register_hook(ArenaOptimizerHook())
    return runner, arena


def _batch(seed=0):
    img, gt = make_batch(2, 64, 96, seed=seed)
    metas = [dict(ori_shape=(64, 96, 3), img_shape=(64, 96, 3), flip=False) for _ in range(2)]
    return dict(img=img.cuda(), img_metas=metas, gt_semantic_seg=gt.cuda())


def _anchor(name):
    a = arch_meta(name)["backbone"]
    return {"name": name, "arch.backbone.stem.width": a["stem"]["width"],
            "arch.backbone.body.width": a["body"]["width"], "arch.backbone.body.depth": a["body"]["depth"]}


def test_frozen_stages_are_bit_unchanged_by_training(hip_lib):
    """frozen_stages=1 (gaiaseg/models/backbones/dynamic_resnet.py:304-321): stem + layer1 have
    requires_grad False and eval-mode BN.  torch.optim.SGD skips parameters without a gradient, so
    neither weight decay nor momentum may touch them, and their BN statistics must not move."""
    from gaia_seg_amd.models import build_segmentor
    cfg = copy.deepcopy(model_cfg(fcn_head(), aux=True))
    cfg["backbone"]["frozen_stages"] = 1
    torch.manual_seed(0)
    model = build_segmentor(cfg).cuda().train()
    runner, arena = _runner(model)
    runner.set_arch(_anchor("sub"))
    frozen = {k: v.detach().clone() for k, v in model.state_dict().items()
              if k.startswith(("backbone.conv1", "backbone.bn1", "backbone.layer1"))}
    others = {k: v.detach().clone() for k, v in model.state_dict().items()
              if k.startswith("backbone.layer2.0.conv1")}
    assert frozen and others
    for it in range(3):
        runner.train_iter(_batch(it))
    torch.cuda.synchronize()
    sd = model.state_dict()
    for k, v in frozen.items():
        assert torch.equal(sd[k], v), "frozen tensor %s changed" % k
    for k, v in others.items():
        assert not torch.equal(sd[k], v), "trainable tensor %s did not move" % k
    # the frozen segments are outside the zero / all-reduce / SGD ranges
    covered = set()
    for a, b in runner.active_ranges:
        covered.update(range(a, b, 64))
    for n, p in model.named_parameters():
        o, _ = arena.segments[id(p)]
        if n.startswith(("backbone.conv1", "backbone.bn1", "backbone.layer1.")):
            assert o not in covered, n


def test_eval_hook_leaves_the_training_subnet_in_place(hip_lib):
    """CrossArchEvalHook.evaluate walks the val anchors; afterwards the model, the active ranges and
    the reducer plan must again be those of the subnet that was training (manipulate_arch=False
    runs never re-apply it)."""
    from gaia_seg_amd.core.evaluation import CrossArchEvalHook
    from gaia_seg_amd.core.model_space import build_model_sampler
    from gaia_seg_amd.models import build_segmentor
    torch.manual_seed(0)
    model = build_segmentor(copy.deepcopy(model_cfg(fcn_head(), aux=True))).cuda().train()
    runner, arena = _runner(model)
    runner.set_arch(_anchor("sub"))
    ranges0 = list(runner.active_ranges)
    state0 = copy.deepcopy(model.backbone.state_dict_of_arch())
    sampler = build_model_sampler(dict(type="anchor", anchors=[_anchor("min"), _anchor("max")]))
    hook = CrossArchEvalHook([_batch(5)], sampler, interval=1, num_batches=1)
    hook.evaluate(runner)
    assert model.backbone.state_dict_of_arch() == state0
    assert runner.active_ranges == ranges0 and runner.arch_name == "sub"
    assert model.training
    # and a following step trains exactly the subnet: layer3.2 (depth 2 in 'sub') stays untouched
    before = model.backbone.layer3[2].conv1.weight.detach().clone()
    runner.train_iter(_batch(6))
    torch.cuda.synchronize()
    assert torch.equal(model.backbone.layer3[2].conv1.weight, before)

    # same with a run that never called set_arch(meta): the state, not the meta, is restored
    model.manipulate_arch(arch_meta("min"))
    runner.arch_meta, runner.arch_key = None, ("current",)
    runner.refresh_active()
    ranges1 = list(runner.active_ranges)
    hook.evaluate(runner)
    assert runner.active_ranges == ranges1
    assert model.backbone.state_dict_of_arch()["body"]["depth"] == arch_meta("min")["backbone"]["body"]["depth"]


def test_checkpoint_keeps_momentum_by_name_in_oihw(hip_lib, tmp_path):
    """`optimizer` in a checkpoint is keyed by parameter name in the logical layout; a foreign
    optimizer state (torch.optim.SGD of an mmcv checkpoint) is tolerated on resume."""
    from gaia_seg_amd.core.checkpoint import save_checkpoint
    from gaia_seg_amd.models import build_segmentor
    torch.manual_seed(0)
    cfg = model_cfg(fcn_head(), aux=True)
    model = build_segmentor(copy.deepcopy(cfg)).cuda().train()
    runner, arena = _runner(model)
    runner.set_arch(_anchor("max"))
    for it in range(2):
        runner.train_iter(_batch(it))
    path = os.path.join(str(tmp_path), "iter_2.pth")
    save_checkpoint(model, path, optimizer=arena, meta=dict(iter=2))
    ck = torch.load(path, map_location="cpu")
    mom = ck["optimizer"]["state"]
    w = model.backbone.layer2[0].conv2.weight
    assert tuple(mom["backbone.layer2.0.conv2.weight"].shape) == tuple(w.shape)      # OIHW
    assert mom["backbone.layer2.0.conv2.weight"].is_contiguous()
    assert float(mom["backbone.layer2.0.conv2.weight"].abs().max()) > 0

    torch.manual_seed(1)
    model2 = build_segmentor(copy.deepcopy(cfg)).cuda().train()
    runner2, arena2 = _runner(model2)
    runner2.resume(path)
    assert runner2.iter == 2
    assert torch.equal(arena2.flat_mom, arena.flat_mom)
    assert torch.equal(arena2.flat_param, arena.flat_param)
    # both continue identically
    runner2.set_arch(_anchor("max"))
    runner.train_iter(_batch(9))
    runner2.train_iter(_batch(9))
    torch.cuda.synchronize()
    assert torch.equal(arena2.flat_param, arena.flat_param)

    # a torch.optim.SGD state_dict (integer keys, no names) must not break resume
    ck["optimizer"] = {"state": {0: {"momentum_buffer": torch.zeros(3)}},
                       "param_groups": [{"lr": 0.01, "params": [0]}]}
    foreign = os.path.join(str(tmp_path), "foreign.pth")
    torch.save(ck, foreign)
    model3 = build_segmentor(copy.deepcopy(cfg)).cuda().train()
    runner3, arena3 = _runner(model3)
    with pytest.warns(UserWarning):
        runner3.resume(foreign)
    assert float(arena3.flat_mom.abs().max()) == 0.0


def test_step_graph_replay_equals_the_eager_step(hip_lib):
    """A recurring subnet's training step is captured into a HIP graph and replayed
    (IterBasedRunner.train_iter): same kernels in the same order, so parameters, momentum, BN
    statistics and the logged losses must be BIT-identical to the eager path — across alternating
    subnets, a changing learning rate (the SGD kernel reads it from device memory) and changing
    batches.  Dropout is switched off: the device RNG streams of a graph and of eager launches differ."""
    from gaia_seg_amd.core.runner import PolyLrUpdaterHook
    from gaia_seg_amd.models import build_segmentor

    def run(graphs):
        torch.manual_seed(0)
        model = build_segmentor(copy.deepcopy(model_cfg(fcn_head(), aux=True))).cuda().train()
        for h in (model.decode_head, model.auxiliary_head):
            h.dropout = None
        runner, arena = _runner(model)
        runner.graphs_enabled = graphs
        runner.register_hook(PolyLrUpdaterHook(power=0.9, min_lr=1e-4))
        runner.call_hook("before_run")
        logs = []
        for it, name in enumerate(["sub", "min", "sub", "min", "sub", "sub", "min"]):
            runner.set_arch(_anchor(name))
            out = runner.train_iter(_batch(it))
            logs.append({k: float(v) for k, v in out["log_vars"].items()})
        torch.cuda.synchronize()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        return sd, arena.flat_mom.detach().clone(), logs, dict(runner.graph_stats)

    sd_e, mom_e, logs_e, st_e = run(False)
    sd_g, mom_g, logs_g, st_g = run(True)
    assert st_e == {"captured": 0, "replayed": 0, "eager": 7}
    assert st_g == {"captured": 2, "replayed": 4, "eager": 1}, st_g   # (opt-in: GS_STEP_GRAPH=1)
    assert logs_e == logs_g
    assert torch.equal(mom_e, mom_g)
    for k in sd_e:
        assert torch.equal(sd_e[k], sd_g[k]), k
    assert int(sd_g["backbone.bn1.num_batches_tracked"]) == 7


def test_optimizer_instalment_inside_backward_is_bitwise_the_same_training(hip_lib):
    """ArenaOptimizerHook.EARLY (GS_EARLY_SGD=1): stages 3.. and the heads are updated on the optimizer
    stream when backward crosses the backbone's "stage2|stage3" mark, the rest at the end.  Same SGD
    arithmetic on the same gradients, only the queue and the moment differ: after four steps over two
    subnets every parameter, momentum buffer and BN statistic equals the default schedule's bit for
    bit, and the instalment really ran (runner.early_steps)."""
    from gaia_seg_amd.core.runner import ArenaOptimizerHook
    from gaia_seg_amd.models import build_segmentor
    results = []
    keep = ArenaOptimizerHook.EARLY
    try:
        for early in (False, True):
            ArenaOptimizerHook.EARLY = early
            torch.manual_seed(0)
            model = build_segmentor(copy.deepcopy(model_cfg(fcn_head(), aux=True))).cuda().train()
            runner, arena = _runner(model)
            for it, name in enumerate(("max", "sub", "max", "min")):
                runner.set_arch(_anchor(name))
                runner.train_iter(_batch(it))
            torch.cuda.synchronize()
            results.append((runner.early_steps, arena.flat_param.clone(), arena.flat_mom.clone(),
                            {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    finally:
        ArenaOptimizerHook.EARLY = keep
    (n0, p0, m0, b0), (n1, p1, m1, b1) = results
    assert n0 == 0 and n1 == 4
    assert torch.equal(p0, p1) and torch.equal(m0, m1)
    assert b0.keys() == b1.keys() and all(torch.equal(b0[k], b1[k]) for k in b0)
    assert float(m1.abs().max()) > 0


def test_high_priority_training_stream_is_bitwise_the_same_training(hip_lib):
    """IterBasedRunner.TRAIN_PRIORITY (the default): the step runs on a high-priority stream that the
    first train_iter makes current; weight-gradient, optimizer and exchange streams keep normal priority.
    Only the dispatcher's choice between ready workgroups changes, so after four steps over three
    subnets every parameter, momentum buffer and BN statistic must equal, bit for bit, the same
    training on the caller's (default) stream — a dependency that only the default stream happened to
    provide would show here."""
    from gaia_seg_amd.core.runner import IterBasedRunner
    from gaia_seg_amd.models import build_segmentor
    results = []
    keep = IterBasedRunner.TRAIN_PRIORITY
    default = torch.cuda.default_stream()
    try:
        for prio in (False, True):
            IterBasedRunner.TRAIN_PRIORITY = prio
            torch.cuda.synchronize()
            torch.cuda.set_stream(default)
            torch.manual_seed(0)
            model = build_segmentor(copy.deepcopy(model_cfg(fcn_head(), aux=True))).cuda().train()
            runner, arena = _runner(model)
            for it, name in enumerate(("max", "sub", "max", "min")):
                runner.set_arch(_anchor(name))
                runner.train_iter(_batch(it))
            used = torch.cuda.current_stream()
            torch.cuda.synchronize()
            results.append((used != default, used.priority, arena.flat_param.clone(), arena.flat_mom.clone(),
                            {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    finally:
        IterBasedRunner.TRAIN_PRIORITY = keep
        torch.cuda.synchronize()
        torch.cuda.set_stream(default)
    (own0, pr0, p0, m0, b0), (own1, pr1, p1, m1, b1) = results
    assert not own0 and own1 and pr1 < pr0        # (a smaller number is a higher priority)
    assert torch.equal(p0, p1) and torch.equal(m0, m1)
    assert b0.keys() == b1.keys() and all(torch.equal(b0[k], b1[k]) for k in b0)
    assert float(m1.abs().max()) > 0
