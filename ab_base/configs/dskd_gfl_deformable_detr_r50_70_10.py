# DSKD incremental Deformable-DETR R50, 70+10 split (BASELINE.json configs[1]), written in the
# reference's config schema (cf. /root/reference/configs/deformable_detr/
# chaosuan_gfl_deformable_detr_70_r50_8x4_1x_qoqo_il.py; the reference's own files also load
# unchanged through dskd_amd.config.Config).  Data is synthetic (dskd_amd/datasets.py).
num_prev, num_curr = 70, 10

model = dict(
    type='DeformableDETR_il',
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch', init_cfg=None),
    neck=dict(type='ChannelMapper', in_channels=[512, 1024, 2048], kernel_size=1, out_channels=256, act_cfg=None,
              norm_cfg=dict(type='GN', num_groups=32), num_outs=4),
    bbox_head=dict(
        type='GFLDeformableDETRHead_il', num_query=300, num_classes=80, in_channels=2048, sync_cls_avg_factor=True,
        as_two_stage=False,
        transformer=dict(
            type='DeformableDetrTransformer',
            encoder=dict(type='DetrTransformerEncoder', num_layers=6,
                         transformerlayers=dict(type='BaseTransformerLayer',
                                                attn_cfgs=dict(type='MultiScaleDeformableAttention', embed_dims=256),
                                                feedforward_channels=1024, ffn_dropout=0.1,
                                                operation_order=('self_attn', 'norm', 'ffn', 'norm'))),
            decoder=dict(type='DeformableDetrTransformerDecoder', num_layers=6, return_intermediate=True,
                         transformerlayers=dict(type='DetrTransformerDecoderLayer',
                                                attn_cfgs=[dict(type='MultiheadAttention', embed_dims=256, num_heads=8,
                                                                dropout=0.1),
                                                           dict(type='MultiScaleDeformableAttention', embed_dims=256)],
                                                feedforward_channels=1024, ffn_dropout=0.1,
                                                operation_order=('self_attn', 'norm', 'cross_attn', 'norm', 'ffn',
                                                                 'norm')))),
        positional_encoding=dict(type='SinePositionalEncoding', num_feats=128, normalize=True, offset=-0.5),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=2.0),
        loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.5),
        loss_bbox=dict(type='L1Loss', loss_weight=5.0),
        loss_iou=dict(type='GIoULoss', loss_weight=2.0),
        cates_distill='hard + teacher-first', locat_distill='', memory_distill='',
        feats_distill='corr + fg_info + decode_v1',
        loss_kd=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=1, T=2, reduction='mean'),
        loss_fg_feature=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=1, T=2, reduction='sum'),
        loss_corr=dict(type='MSELoss', loss_weight=1, reduction='mean')),
    train_cfg=dict(assigner=dict(type='GFLHungarianAssigner',
                                 cls_cost=dict(type='QualityFocalLossCost', weight=2.0),
                                 reg_cost=dict(type='BBoxL1Cost', weight=5.0, box_format='xywh'),
                                 iou_cost=dict(type='IoUCost', iou_mode='giou', weight=2.0))),
    test_cfg=dict(max_per_img=100, score_thr=0.0),
    teacher_test_cfg=dict(min_bbox_size=0, score_thr=0.3, max_per_img=100))

catsplit, catload = (num_prev, num_curr), (1, 0)
data = dict(samples_per_gpu=4, workers_per_gpu=0, cat_split_load='auto',
            train=dict(test_mode=False, catsplit=catsplit, catload=catload, catpred='prev-cur', catwise=True, imgpercent=1,
                       img_size=(800, 1333), n_gt=7, num_images=64))
task_nums = len(catsplit)
workflow = [('train', 1)]
optimizer = [dict(type='AdamW', lr=2e-4, weight_decay=0.0001,
                  paramwise_cfg=dict(custom_keys={'backbone': dict(lr_mult=0.1), 'sampling_offsets': dict(lr_mult=0.1),
                                                  'reference_points': dict(lr_mult=0.1)}))] * task_nums
optimizer_config = dict(grad_clip=dict(max_norm=0.1, norm_type=2))
lr_config = [dict(policy='step', warmup='linear', warmup_iters=1500, warmup_ratio=0.01, step=[8, 11])] * task_nums
runner = [dict(type='TaskEpochBasedRunner', max_epochs=12, max_tasks=task_nums, save_teacher=False)] * task_nums
checkpoint_config = dict(interval=1)
log_config = dict(interval=50)
