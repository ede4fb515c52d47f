# Dynamic ResNet supernet backbone (max widths 64 / 80-160-320-640, max depths 4-6-29-4), OS32 with
# the 7x7 stem: the backbone of the reference's configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:11-24.
_conv = dict(type='DynConv2d')
_backbone = dict(
    type='DynamicResNet', in_channels=3, stem_width=64, body_depth=[4, 6, 29, 4],
    body_width=[80, 160, 320, 640], num_stages=4, out_indices=(0, 1, 2, 3), conv_cfg=_conv,
    norm_cfg=dict(type='DynSyncBN', requires_grad=True, group_size=1), style='pytorch')
_aux_head = dict(
    type='DynamicFCNHead', conv_cfg=_conv, in_channels=1280, in_index=2, channels=256, num_convs=1,
    concat_input=False, dropout_ratio=0.1, num_classes=19,
    norm_cfg=dict(type='SyncBN', requires_grad=True), align_corners=False,
    loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4))
crop_size = (512, 1024)
data = dict(samples_per_gpu=2, workers_per_gpu=2,
            train=dict(type='SyntheticSegDataset', size=crop_size, num_classes=19))
optimizer = dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0005)
optimizer_config = dict()
lr_config = dict(policy='poly', power=0.9, min_lr=1e-4, by_epoch=False)
runner = dict(type='IterBasedRunner', max_iters=80000)
checkpoint_config = dict(by_epoch=False, interval=8000)
evaluation = dict(interval=8000, metric='mIoU')
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook', by_epoch=False)])
dist_params = dict(backend='nccl')
log_level = 'INFO'
load_from = None
resume_from = None
workflow = [('train', 1)]
train_cfg = dict()
test_cfg = dict(mode='whole')
