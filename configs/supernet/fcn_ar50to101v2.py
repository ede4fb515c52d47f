# BASELINE config 2: FCN decode head (mmseg defaults: 2 convs, concat_input, 512 channels) on the
# dynamic R50..R101 supernet, 1024x512 crops, bs 2 / GPU, random-subnet sampling per step.
_base_ = ['../_dynamic_/models/backbone_ar50to101v2.py', '../_dynamic_/model_samplers/ar50to101v2.py']
model = dict(
    type='DynamicEncoderDecoder',
    backbone=dict(type='DynamicResNet', in_channels=3, stem_width=64, body_depth=[4, 6, 29, 4],
                  body_width=[80, 160, 320, 640], num_stages=4, out_indices=(0, 1, 2, 3),
                  conv_cfg=dict(type='DynConv2d'),
                  norm_cfg=dict(type='DynSyncBN', requires_grad=True, group_size=1),
                  style='pytorch'),
    decode_head=dict(type='DynamicFCNHead', conv_cfg=dict(type='DynConv2d'), in_channels=2560,
                     in_index=3, channels=512, num_convs=2, concat_input=True, dropout_ratio=0.1,
                     num_classes=19, norm_cfg=dict(type='SyncBN', requires_grad=True),
                     align_corners=False,
                     loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
    auxiliary_head=dict(type='DynamicFCNHead', conv_cfg=dict(type='DynConv2d'), in_channels=1280,
                        in_index=2, channels=256, num_convs=1, concat_input=False,
                        dropout_ratio=0.1, num_classes=19,
                        norm_cfg=dict(type='SyncBN', requires_grad=True), align_corners=False,
                        loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False,
                                         loss_weight=0.4)))
