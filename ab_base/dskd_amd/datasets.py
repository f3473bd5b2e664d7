"""Class split of the incremental protocol + a synthetic COCO-shaped incremental dataset.

The reference's IL dataset class is missing from its checkout (configs
pass ``catsplit/catload/catpred/catwise/imgpercent`` and the driver reads
``ALL_CLASSES_IDS / cat2label / PRED_CLASSES / LOAD_CLASSES / TASK_CLASSES``,
/root/reference/tools/train_increment.py:268-272, but no class implements them).  This shim
provides exactly that attribute surface over synthetic tensors of the benchmark's shape
(SURVEY.md section 8d): img ~ N(0,1) fp32 [3,H,W]; per image ``n_gt`` boxes with
x1,y1 ~ U(0,0.6)*(W,H), w,h ~ U(8px, 0.35*(W,H)), labels ~ U over the CURRENT task's classes.
"""
import copy
import random
from collections import OrderedDict

import torch

# COCO 2017 detection categories, official (name, category id) pairs in id order.
_COCO_ID_ORDER = (
    ("person", 1), ("bicycle", 2), ("car", 3), ("motorcycle", 4), ("airplane", 5), ("bus", 6), ("train", 7), ("truck", 8),
    ("boat", 9), ("traffic light", 10), ("fire hydrant", 11), ("stop sign", 13), ("parking meter", 14), ("bench", 15),
    ("bird", 16), ("cat", 17), ("dog", 18), ("horse", 19), ("sheep", 20), ("cow", 21), ("elephant", 22), ("bear", 23),
    ("zebra", 24), ("giraffe", 25), ("backpack", 27), ("umbrella", 28), ("handbag", 31), ("tie", 32), ("suitcase", 33),
    ("frisbee", 34), ("skis", 35), ("snowboard", 36), ("sports ball", 37), ("kite", 38), ("baseball bat", 39),
    ("baseball glove", 40), ("skateboard", 41), ("surfboard", 42), ("tennis racket", 43), ("bottle", 44), ("wine glass", 46),
    ("cup", 47), ("fork", 48), ("knife", 49), ("spoon", 50), ("bowl", 51), ("banana", 52), ("apple", 53), ("sandwich", 54),
    ("orange", 55), ("broccoli", 56), ("carrot", 57), ("hot dog", 58), ("pizza", 59), ("donut", 60), ("cake", 61),
    ("chair", 62), ("couch", 63), ("potted plant", 64), ("bed", 65), ("dining table", 67), ("toilet", 70), ("tv", 72),
    ("laptop", 73), ("mouse", 74), ("remote", 75), ("keyboard", 76), ("cell phone", 77), ("microwave", 78), ("oven", 79),
    ("toaster", 80), ("sink", 81), ("refrigerator", 82), ("book", 84), ("clock", 85), ("vase", 86), ("scissors", 87),
    ("teddy bear", 88), ("hair drier", 89), ("toothbrush", 90))
# The incremental protocol orders the classes by NAME (/root/reference/mmdet/datasets/data_split.py:62-80,
# "pingyin" order); task t takes the next ``split[t]`` names of that order.
COCO_CATS_IDS = OrderedDict(sorted(_COCO_ID_ORDER))


def split_data_category(dataname="CocoDataset", split=(20, 20, 20, 20), order="pingyin", catofset="train|val|fine",
                        trainpart="cur-only", valpart="prev-only|cur-only|prev-cur"):
    """Class split of the incremental protocol (/root/reference/mmdet/datasets/data_split.py:100-158): cuts the
    name-ordered COCO classes into ``len(split)`` tasks and returns, per task, {name: category id} of
    the classes to TRAIN on (the task's own), to VALIDATE on (``valpart``: the previous task's /
    the task's own / everything seen so far) and of everything seen so far ('fine').  ``split``
    may be '40-40'; ``order='shuffle'`` permutes the names with Python's ``random`` (seed it for a
    reproducible protocol).  Returns what ``catofset`` names: 'train', 'val', 'fine', 'train|val'
    or all three.  Same errors as the reference for an unknown dataset / order / valpart."""
    if dataname != "CocoDataset":
        raise NotImplementedError(f"unknown dataset: {dataname}")
    if order == "shuffle":
        keys = list(COCO_CATS_IDS.keys())
        random.shuffle(keys)
        ordered = {k: COCO_CATS_IDS[k] for k in keys}
    elif order == "pingyin":
        ordered = copy.copy(COCO_CATS_IDS)
    else:
        raise ValueError("unsupported class order")
    if isinstance(split, str):
        split = [int(s) for s in split.split("-")]
    assert isinstance(split, (tuple, list))
    names, ids = list(ordered.keys()), list(ordered.values())
    start, trainsplit, valsplit, finesplit = 0, [], [], []
    for n in split:
        trainsplit.append(dict(zip(names[start:start + n], ids[start:start + n])))
        start += n
    seen = {}
    for t, own in enumerate(trainsplit):
        assert valpart in ("prev-only", "cur-only", "prev-cur"), f"bad validation mode: {valpart}"
        if valpart == "prev-only":
            seen = trainsplit[t - 1] if t >= 1 else {}
        elif valpart == "cur-only":
            seen = own
        else:
            seen.update(own)
        valsplit.append(copy.copy(seen))
    seen = {}
    for own in trainsplit:
        seen.update(own)
        finesplit.append(copy.copy(seen))
    if catofset == "train":
        return trainsplit
    if catofset == "val":
        return valsplit
    if catofset == "fine":
        return finesplit
    if catofset == "train|val":
        return trainsplit, valsplit
    return trainsplit, valsplit, finesplit


class SyntheticILDataset(torch.utils.data.Dataset):
    def __init__(self, catsplit=(40, 40), catload=(1, 0), catpred="prev-cur", catwise=True, imgpercent=1,
                 test_mode=False, num_images=64, img_size=(800, 1333), n_gt=7, seed=111, num_classes=80, **kwargs):
        assert sum(catsplit) == num_classes and len(catload) == len(catsplit)
        self.catsplit, self.catload, self.test_mode = tuple(catsplit), tuple(catload), test_mode
        self.num_images, self.img_size, self.n_gt, self.seed = num_images, tuple(img_size), n_gt, seed
        if num_classes == len(COCO_CATS_IDS):
            # the protocol's own classes: COCO names in name order with their COCO ids, label = rank in that
            # order, tasks cut by split_data_category (so 'prev' of a 70+10 run is labels 0..69)
            names = list(COCO_CATS_IDS)
            self.ALL_CLASSES_IDS = dict(COCO_CATS_IDS)                          # CatName -> CatID
            self.cat2label = {cid: i for i, cid in enumerate(COCO_CATS_IDS.values())}   # CatID -> label
            self.TASK_CLASSES = [list(t) for t in split_data_category(split=tuple(catsplit), catofset="train",
                                                                       valpart="prev-cur")]
        else:
            names = [f"class_{i:02d}" for i in range(num_classes)]
            self.ALL_CLASSES_IDS = {n: i + 1 for i, n in enumerate(names)}    # CatName -> CatID (1-based like COCO)
            self.cat2label = {i + 1: i for i in range(num_classes)}             # CatID -> label
            bounds = [0]
            for n in catsplit:
                bounds.append(bounds[-1] + n)
            self.TASK_CLASSES = [names[bounds[i]:bounds[i + 1]] for i in range(len(catsplit))]
        self.CLASSES = tuple(names)
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
