"""ORACLE — test infrastructure only.

A CPU restatement (plain PyTorch, fp32 / fp64) of the reference algorithm for the GAIA-seg
supernet hot path, used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
the checker.  Nothing under gaia_seg_amd/ imports it; the product path has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * PINNED by the reference itself run in the build container: weight_reduce_loss / reduce_loss
    (gaiaseg/models/losses/utils.py) and accuracy (gaiaseg/models/losses/accuracy.py) — fixtures
    tests/golden/ref_loss_utils.npz, generator tests/golden/make_ref_loss_fixtures.py.
  * PARITY UNPINNED: everything that depends on gaiavision / mmseg / mmcv source (DynConv2d,
    DynBN, DynamicBottleneck, DynamicConvModule, resize, OHEM, EncoderDecoder): those packages are
    absent from /root/reference and from the image and the reference has no tests, golden vectors
    or fixtures.  They follow the contracts reconstructed from the reference
    call sites and are anchored by analytic known-answer and metamorphic
    tests (tests/test_oracle_*.py).
"""
