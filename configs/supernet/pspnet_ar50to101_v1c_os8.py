# OS8 / "v1c" supernet (SURVEY.md section 8d "report both OS32 and OS8"): the model of the reference's
# configs/local_examples/extract_subnet/psp_ar50to101_v1c_extract.py:6-14 -- deep stem [32, 32, 64],
# strides (1, 2, 1, 1), dilations (1, 1, 2, 4), contract_dilation -- with the PSP decode head and the
# aux FCN head of the in-tree training config.  Stages 3 and 4 run at 1/8 resolution (64 x 128 for a
# 512 x 1024 crop): dilated 3x3 bottleneck convs at M = 16384 rows and a 348 GF/img PSP bottleneck.
_base_ = ['../_dynamic_/models/backbone_ar50to101v2.py', '../_dynamic_/model_samplers/ar50to101v2.py']
model = dict(
    type='DynamicEncoderDecoder',
    backbone=dict(type='DynamicResNet', in_channels=3, stem_width=[32, 32, 64], deep_stem=True,
                  avg_down=False, body_depth=[4, 6, 29, 4], body_width=[80, 160, 320, 640],
                  num_stages=4, dilations=(1, 1, 2, 4), strides=(1, 2, 1, 1), contract_dilation=True,
                  out_indices=(0, 1, 2, 3), conv_cfg=dict(type='DynConv2d'),
                  norm_cfg=dict(type='DynSyncBN', requires_grad=True, group_size=1),
                  style='pytorch'),
    decode_head=dict(type='DynamicPSPHead', conv_cfg=dict(type='DynConv2d'), in_channels=2560,
                     in_index=3, channels=512, pool_scales=(1, 2, 3, 6), dropout_ratio=0.1,
                     num_classes=19, norm_cfg=dict(type='SyncBN', requires_grad=True),
                     align_corners=False,
                     loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
    auxiliary_head=dict(type='DynamicFCNHead', conv_cfg=dict(type='DynConv2d'), in_channels=1280,
                        in_index=2, channels=256, num_convs=1, concat_input=False,
                        dropout_ratio=0.1, num_classes=19,
                        norm_cfg=dict(type='SyncBN', requires_grad=True), align_corners=False,
                        loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False,
                                         loss_weight=0.4)))
train_cfg = dict()
test_cfg = dict(mode='whole')
# the deep stem takes a width per stem conv (reference anchors of psp_ar50to101_v1c_extract.py:78-113)
stem_anchors = dict(MAX=[32, 32, 64], MIN=[16, 16, 32], R50=[32, 32, 64], R77=[32, 32, 64],
                    R101=[32, 32, 64])
