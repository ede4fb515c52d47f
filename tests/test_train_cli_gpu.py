"""The supernet training CLI runs end to end on one MI355X with the in-repo PSP supernet config
(the reference's config of record) and a reduced crop; loss is finite and
parameters move; a checkpoint round-trips in the reference format."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_supernet_cli_smoke(tmp_path):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_supernet.py"),
           os.path.join(ROOT, "configs", "supernet", "pspnet_ar50to101v2.py"),
           "--work-dir", str(tmp_path), "--seed", "0", "--no-validate", "--max-iters", "6",
           "--cfg-options", "data.train.size=(128,256)", "log_config.interval=2",
           "checkpoint_config.interval=6"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = res.stderr + res.stdout
    assert "Iter [6/6]" in out and "decode.loss_seg" in out and "finished 6 iterations" in out
    ck = torch.load(os.path.join(str(tmp_path), "iter_6.pth"), map_location="cpu")
    assert set(ck) >= {"meta", "state_dict", "optimizer"}
    w = ck["state_dict"]["backbone.layer3.28.conv2.weight"]
    assert tuple(w.shape) == (320, 320, 3, 3) and w.is_contiguous()
    assert torch.isfinite(ck["state_dict"]["decode_head.conv_seg.weight"]).all()


def test_train_supernet_cli_with_cross_arch_eval(tmp_path):
    """Same CLI without --no-validate: the cross-arch eval hook (gaiaseg/apis/train.py:150-170)
    runs every val anchor and logs one mIoU line per subnet."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "train_supernet.py"),
           os.path.join(ROOT, "configs", "supernet", "fcn_ar50to101v2.py"),
           "--work-dir", str(tmp_path), "--seed", "0", "--max-iters", "4",
           "--cfg-options", "data.train.size=(128,256)", "log_config.interval=2",
           "checkpoint_config.interval=100", "evaluation.interval=4", "evaluation.num_batches=1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    out = res.stderr + res.stdout
    assert out.count("mIoU") >= 3 and "R101" in out, out[-2000:]
