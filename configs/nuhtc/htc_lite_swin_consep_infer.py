# Inference-only config for the MI355X engine, using the reference's config schema (same key names and values as the
# model / test_cfg / test_pipeline sections of configs/nuhtc/htc_lite_swin_pytorch_fpn_CoNSeP_seasaw_CAS.py of
# boyden/NuHTC; training, dataset, optimizer and hook keys are not part of the inference surface).  The reference's own
# config files are accepted unchanged by nuhtc_amd.config.Config.fromfile.
thres = 0.965926
num_classes = 4
scale_factor = 2.0
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)


def _bbox_head(stds):
    return dict(type='Shared2FCBBoxHeadWithProb', in_channels=64, fc_out_channels=256, roi_feat_size=7, num_classes=num_classes,
                bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=stds),
                reg_class_agnostic=True, cls_predictor_cfg=dict(type='NormedLinear', tempearture=20))


def _extractor(size, sr, strides):
    return dict(type='AttentionRoIExtractor', start_level=2, thres=thres, roi_layer=dict(type='RoIAlign', output_size=size, sampling_ratio=sr),
                out_channels=64, featmap_strides=strides)


model = dict(
    type='HybridTaskCascade_Cus',
    backbone=dict(type='SwinTransformer', embed_dims=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7, mlp_ratio=4,
                  qkv_bias=True, patch_norm=True, out_indices=(0, 1, 2, 3)),
    neck=dict(type='FPN', in_channels=[96, 192, 384, 768], out_channels=64, num_outs=4),
    rpn_head=dict(type='RPNHead', in_channels=64, feat_channels=64,
                  anchor_generator=dict(type='AnchorGenerator', scales=[4], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32]),
                  bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0.0, 0.0, 0.0, 0.0], target_stds=[1.0, 1.0, 1.0, 1.0])),
    roi_head=dict(
        type='HybridTaskCascadeRoIHead_Lite', interleaved=True, mask_info_flow=True, num_stages=3, watershed_proposal=True,
        bbox_roi_extractor=_extractor(7, 2, [4, 8, 16, 32]),
        bbox_head=[_bbox_head([0.1, 0.1, 0.2, 0.2]), _bbox_head([0.05, 0.05, 0.1, 0.1]), _bbox_head([0.033, 0.033, 0.067, 0.067])],
        mask_roi_extractor=_extractor(14, 0, [4, 8, 16, 32]),
        mask_head=[dict(type='HTCMaskHead', with_conv_res=True, num_convs=4, in_channels=64, conv_out_channels=64, class_agnostic=True,
                        num_classes=num_classes)],
        semantic_roi_extractor=_extractor(14, 0, [4]),
        semantic_head=dict(type='FusedSemanticHead', num_ins=4, fusion_level=0, num_convs=4, in_channels=64, conv_out_channels=64, num_classes=1)),
    test_cfg=dict(
        rpn=dict(nms_pre=3000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=10),
        rcnn=dict(score_thr=0.35, nms=dict(type='nms', iou_threshold=0.5), max_per_img=300, mask_thr_binary=0.5)))

test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='MultiScaleFlipAug', scale_factor=scale_factor, flip=False,
         transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), dict(type='Normalize', **img_norm_cfg),
                     dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])]),
]
data = dict(test=dict(pipeline=test_pipeline))
