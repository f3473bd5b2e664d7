"""Synthetic COCO-shaped incremental dataset.

The reference's IL dataset class is missing from its checkout (configs
pass ``catsplit/catload/catpred/catwise/imgpercent`` and the driver reads
``ALL_CLASSES_IDS / cat2label / PRED_CLASSES / LOAD_CLASSES / TASK_CLASSES``,
/root/reference/tools/train_increment.py:268-272, but no class implements them).  This shim
provides exactly that attribute surface over synthetic tensors of the benchmark's shape
(SURVEY.md section 8d): img ~ N(0,1) fp32 [3,H,W]; per image ``n_gt`` boxes with
x1,y1 ~ U(0,0.6)*(W,H), w,h ~ U(8px, 0.35*(W,H)), labels ~ U over the CURRENT task's classes.
"""
import torch


class SyntheticILDataset(torch.utils.data.Dataset):
    def __init__(self, catsplit=(40, 40), catload=(1, 0), catpred="prev-cur", catwise=True, imgpercent=1,
                 test_mode=False, num_images=64, img_size=(800, 1333), n_gt=7, seed=111, num_classes=80, **kwargs):
        assert sum(catsplit) == num_classes and len(catload) == len(catsplit)
        self.catsplit, self.catload, self.test_mode = tuple(catsplit), tuple(catload), test_mode
        self.num_images, self.img_size, self.n_gt, self.seed = num_images, tuple(img_size), n_gt, seed
        names = [f"class_{i:02d}" for i in range(num_classes)]
        self.CLASSES = tuple(names)
        self.ALL_CLASSES_IDS = {n: i + 1 for i, n in enumerate(names)}        # CatName -> CatID (1-based like COCO)
        self.cat2label = {i + 1: i for i in range(num_classes)}                 # CatID -> label
        bounds = [0]
        for n in catsplit:
            bounds.append(bounds[-1] + n)
        self.TASK_CLASSES = [names[bounds[i]:bounds[i + 1]] for i in range(len(catsplit))]
        cur = max(i for i, v in enumerate(catload) if v) if any(catload) else 0
        self.LOAD_CLASSES = [c for i, v in enumerate(catload) if v for c in self.TASK_CLASSES[i]]
        self.PRED_CLASSES = [c for i in range(cur + 1) for c in self.TASK_CLASSES[i]] if catpred == "prev-cur" \
            else list(self.LOAD_CLASSES)
        self._load_labels = torch.tensor([self.cat2label[self.ALL_CLASSES_IDS[c]] for c in self.LOAD_CLASSES])
        self.flag = torch.zeros(num_images, dtype=torch.uint8).numpy()

    def __len__(self):
        return self.num_images

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed * 1000003 + idx)
        H, W = self.img_size
        img = torch.randn(3, H, W, generator=g)
        xy = torch.rand(self.n_gt, 2, generator=g) * torch.tensor([0.6 * W, 0.6 * H])
        lo = torch.tensor([8.0, 8.0])
        hi = torch.tensor([0.35 * W, 0.35 * H])
        wh = lo + torch.rand(self.n_gt, 2, generator=g) * (hi - lo).clamp(min=0)
        boxes = torch.cat([xy, torch.minimum(xy + wh, torch.tensor([float(W), float(H)]))], 1)
        labels = self._load_labels[torch.randint(0, len(self._load_labels), (self.n_gt,), generator=g)]
        meta = dict(img_shape=(H, W, 3), ori_shape=(H, W, 3), pad_shape=(H, W, 3), batch_input_shape=(H, W),
                    scale_factor=1.0, flip=False, filename=f"synthetic_{idx}")
        return dict(img=img, img_metas=meta, gt_bboxes=boxes, gt_labels=labels)


def collate(batch):
    return dict(img=torch.stack([b["img"] for b in batch]), img_metas=[b["img_metas"] for b in batch],
                gt_bboxes=[b["gt_bboxes"] for b in batch], gt_labels=[b["gt_labels"] for b in batch])


def build_dataset(cfg, default_args=None):
    args = dict(cfg)
    args.update(default_args or {})
    args.pop("type", None)
    for k in ("ann_file", "img_prefix", "pipeline"):
        args.pop(k, None)
    return SyntheticILDataset(**args)


def build_dataloader(dataset, samples_per_gpu, workers_per_gpu=0, dist=False, shuffle=True, seed=None, **kwargs):
    sampler = None
    if dist:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=shuffle, seed=seed or 0)
        shuffle = False
    g = torch.Generator()
    g.manual_seed(seed or 0)
    return torch.utils.data.DataLoader(dataset, batch_size=samples_per_gpu, shuffle=shuffle, sampler=sampler,
                                       num_workers=workers_per_gpu, collate_fn=collate, drop_last=True, generator=g)


def to_device(data, device):
    return dict(img=data["img"].to(device, non_blocking=True), img_metas=data["img_metas"],
                gt_bboxes=[b.to(device, non_blocking=True) for b in data["gt_bboxes"]],
                gt_labels=[l.to(device, non_blocking=True) for l in data["gt_labels"]])
