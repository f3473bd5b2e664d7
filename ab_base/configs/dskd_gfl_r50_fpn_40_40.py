# DSKD incremental GFL R50-FPN, 40+40 split (BASELINE.json configs[4]: "GFL R50 (configs/gfl) 40+40 incremental with
# DSKD feature-map loss only"), written in the reference's config schema: model = the reference's
# configs/gfl/gfl_r50_fpn_1x_coco.py (backbone / FPN / GFLHead / ATSS / test_cfg verbatim; tests/test_gfl.py checks that)
# plus the incremental-training keys of the DSKD configs.  The reference registers no incremental GFL head, so the
# distillation keys below (feats_distill, loss_fg_feature, teacher_test_cfg) are this repo's (dskd_amd/gfl_head.py).
num_prev, num_curr = 40, 40

model = dict(
    type='GFL',
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch', init_cfg=None),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
              add_extra_convs='on_output', num_outs=5),
    bbox_head=dict(
        type='GFLHead', num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
        anchor_generator=dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8, scales_per_octave=1,
                              strides=[8, 16, 32, 64, 128]),
        loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
        loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25),
        reg_max=16,
        loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
        feats_distill='fg_info + decode_v1',
        loss_fg_feature=dict(type='KnowledgeDistillationKLDivLoss', loss_weight=1, T=2, reduction='sum')),
    train_cfg=dict(assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1, pos_weight=-1, debug=False),
    test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.6), max_per_img=100),
    teacher_test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.3, nms=dict(type='nms', iou_threshold=0.6),
                          max_per_img=100))

catsplit, catload = (num_prev, num_curr), (1, 0)
data = dict(samples_per_gpu=4, workers_per_gpu=0, cat_split_load='auto',
            train=dict(test_mode=False, catsplit=catsplit, catload=catload, catpred='prev-cur', catwise=True, imgpercent=1,
                       img_size=(800, 1333), n_gt=7, num_images=64))
task_nums = len(catsplit)
workflow = [('train', 1)]
optimizer = [dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001)] * task_nums
optimizer_config = dict(grad_clip=None)
lr_config = [dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[8, 11])] * task_nums
runner = [dict(type='TaskEpochBasedRunner', max_epochs=12, max_tasks=task_nums, save_teacher=False)] * task_nums
checkpoint_config = dict(interval=1)
log_config = dict(interval=50)
