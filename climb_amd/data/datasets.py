"""The four vision-language datasets of the upstream CL sequence and their collate functions: the same
on-disk files, the same per-item dictionaries and the same batch dictionaries as the reference's
  REF/data/visionlanguage_datasets/vqa_dataset.py:35-253     (VQADataset, vqa_batch_collate, build_vqa_dataloader)
  REF/data/visionlanguage_datasets/nlvr2_dataset.py:30-190   (NLVR2Dataset, nlvr2_batch_collate, build_nlvr2_dataloader)
  REF/data/visionlanguage_datasets/snli_ve_dataset.py:35-219 (SnliVEDataset, snlive_batch_collate, build_snli_ve_dataloader)
  REF/data/visionlanguage_datasets/vcr_dataset.py:40-248     (VCRDataset, vcr_batch_collate, build_vcr_dataloader)
  REF/data/image_datasets/cocoimages_dataset.py:23-95, flickr30kimages_dataset.py:23-94, REF/data/image_collation.py:30-63
so that `Trainer(args, task_configs, model_config, device)` can build its own loaders exactly as the reference trainers do
(REF/train/visionlanguage_tasks/train_vqa.py:40-97).  This is host I/O: it decodes files and hands PIL images + strings to
`ViltEncoderWrapper.process_inputs`, whose image half then runs on the device (climb_amd/data/image_pipeline.py).

Differences, all outside the arithmetic:
  * torchvision is not a dependency: the one transform the 'pil-image' path uses, `T.Resize(size=384, max_size=640)` on a PIL image
    (bilinear, antialiased), is restated in `resize_shorter_edge` from torchvision's documented size rule;
  * jsonl files are read with the json module; parsed annotations are cached next to the data in the reference's own pickle files
    when the cache directory exists (and the reference's caches are read when present), so both code bases share them;
  * collate functions are module-level callables (picklable: DataLoader workers can be spawned, not only forked).
"""
from __future__ import annotations

import functools
import json
import logging
import os
import pickle as pkl
import random
from collections import defaultdict
from typing import Dict, List, Optional

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

logger = logging.getLogger(__name__)

ALLOWED_VISUAL_INPUT_TYPES = ["raw", "pil-image", "fast-rcnn"]


# ------------------------------------------------------------------------------------------------ image side
def resized_output_size(w: int, h: int, size: int = 384, max_size: Optional[int] = 640):
    """torchvision.transforms.functional._compute_resized_output_size for an int `size`: shorter edge -> size, aspect kept, and if the
    longer edge would exceed max_size, longer edge -> max_size instead."""
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long_ / short)
    if max_size is not None and new_long > max_size:
        new_short, new_long = int(max_size * new_short / new_long), max_size
    return (new_short, new_long) if w <= h else (new_long, new_short)          # (new_w, new_h)


def resize_shorter_edge(image, size: int = 384, max_size: Optional[int] = 640):
    """`T.Resize(size=384, max_size=640)(pil_image)`: PIL bilinear (antialiased by construction) to `resized_output_size`."""
    from PIL import Image
    nw, nh = resized_output_size(image.size[0], image.size[1], size, max_size)
    return image.resize((nw, nh), Image.BILINEAR)


def _load_rgb(path: str, transform):
    """REF cocoimages_dataset.py:68-80: RGB; pre-shrunk only when BOTH edges exceed 384 (min(size) > 384)."""
    from PIL import Image
    image = Image.open(path)
    image = image.convert("RGB")
    if min(list(image.size)) > 384:
        image = transform(image)
    return image


def _raw_tensor(path: str, image_size):
    """'raw' visual input (REF cocoimages_dataset.py:82-95): Resize((H, W)) -> [0,1] -> Normalize(0.5, 0.5) = [-1, 1], shape (3, H, W)."""
    from PIL import Image
    image = Image.open(path).convert("RGB")
    image = image.resize((image_size[1], image_size[0]), Image.BILINEAR)
    arr = torch.from_numpy(np.asarray(image, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    return (arr - 0.5) / 0.5


class _ImagesDataset(Dataset):
    """Image-side backbone shared by VL tasks that draw from one image collection: maps an integer image id to a file."""
    images_subdir = "images"
    exact_resize = False            # Flickr30K: T.Resize((384, 640)) -- an exact, aspect-changing resize (REF flickr30kimages_dataset.py:51)

    def __init__(self, root: str, visual_input_type: str, image_size=(384, 640)):
        assert visual_input_type in ALLOWED_VISUAL_INPUT_TYPES
        self.images_dir = os.path.join(root, self.images_subdir)
        self.image_size, self.visual_input_type = image_size, visual_input_type
        self.imageid2filename: Dict[int, str] = {}
        for fn in os.listdir(self.images_dir):
            self.imageid2filename[self.image_id_of(fn)] = os.path.join(self.images_dir, self.stored_name(fn))
        self.imageids = list(set(self.imageid2filename.keys()))

    def image_id_of(self, fn: str) -> int:
        raise NotImplementedError

    def stored_name(self, fn: str) -> str:
        return fn

    def __len__(self):
        return len(self.imageids)

    def pil_transform(self, image):
        from PIL import Image
        if self.exact_resize:
            return image.resize((self.image_size[1], self.image_size[0]), Image.BILINEAR)
        return resize_shorter_edge(image, 384, 640)

    def get_image_data(self, image_id: int):
        if self.visual_input_type == "pil-image":
            return self.get_pil_image(image_id)
        if self.visual_input_type == "raw":
            return self.get_raw_image_tensor(image_id)
        raise NotImplementedError("Fast-RCNN feature inputs are not implemented (nor are they in the reference)")

    def get_pil_image(self, image_id: int):
        assert image_id in self.imageid2filename
        return _load_rgb(self.imageid2filename[image_id], self.pil_transform)

    def get_raw_image_tensor(self, image_id: int) -> torch.Tensor:
        assert image_id in self.imageid2filename
        return _raw_tensor(self.imageid2filename[image_id], self.image_size)


class MSCOCOImagesDataset(_ImagesDataset):
    """REF/data/image_datasets/cocoimages_dataset.py:23-95.  File names `COCO_<split>2014_<id>.jpg` or `<id>.jpg`; the reference keys a
    file by the part after the last underscore AND stores that stripped name as the path (:42-45), i.e. it expects the un-prefixed
    files to exist -- kept."""

    def image_id_of(self, fn: str) -> int:
        return int(fn.split("_")[-1].strip(".jpg"))

    def stored_name(self, fn: str) -> str:
        return fn.split("_")[-1]


class Flickr30KImagesDataset(_ImagesDataset):
    """REF/data/image_datasets/flickr30kimages_dataset.py:23-94."""
    images_subdir = "flickr30k_images"
    exact_resize = True

    def image_id_of(self, fn: str) -> int:
        return int(fn.strip(".jpg"))


def image_collate(images: List, visual_input_type: str):
    """REF/data/image_collation.py:30-63."""
    if visual_input_type == "pil-image":
        return images
    if visual_input_type == "raw":
        return torch.stack(images, dim=0)
    if visual_input_type == "fast-rcnn":
        max_len = max(t.shape[0] for t in images)
        return torch.stack([torch.cat((t, torch.zeros(max_len - t.shape[0], t.shape[1])), dim=0) for t in images], dim=0)
    raise ValueError(visual_input_type)


# ------------------------------------------------------------------------------------------------ helpers
def _read_jsonl(path: str):
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                yield json.loads(line)


def _load_or_build_cache(cache_file: str, build):
    """The reference's parse caches (REF/data/visionlanguage_datasets/*: `pkl.load` if the file exists, else parse and `pkl.dump`).  Under N ranks
    every process builds its datasets at the same moment: a cache file is therefore written to a temporary name and renamed into place (a reader
    either sees no file or a whole one), and a file that cannot be unpickled -- somebody else's half-written one from before this rule, a
    truncated copy -- is parsed again instead of failing the run (found by the two-rank driver test: rank 1 read the file rank 0 was writing)."""
    if os.path.exists(cache_file):
        try:
            with open(cache_file, "rb") as f:
                return pkl.load(f)
        except Exception as e:          #  a cut can surface as EOFError / UnpicklingError, but also as AttributeError, ValueError, IndexError ... depending on where it lands
            logger.warning("parse cache %s could not be read (%s: %s); building it again", cache_file, type(e).__name__, e)
    data = build()
    if os.path.isdir(os.path.dirname(cache_file)):          # the reference's data trees ship these directories; never create them
        tmp = f"{cache_file}.tmp.{os.getpid()}"
        try:
            with open(tmp, "wb") as f:
                pkl.dump(data, f)
            os.replace(tmp, cache_file)
        except OSError as e:            # a full / read-only data tree must not fail the run, nor leave the temporary behind
            logger.warning("parse cache %s not written (%s)", cache_file, e)
        finally:
            if os.path.exists(tmp):
                try:
                    os.unlink(tmp)
                except OSError:
                    pass
    return data


def _pad_ids(input_ids: List[List[int]], pad_token: int = 0):
    max_len = max(len(x) for x in input_ids)
    ids = [x + [pad_token] * (max_len - len(x)) for x in input_ids]
    mask = [[1] * len(x) + [0] * (max_len - len(x)) for x in input_ids]
    return torch.tensor(ids, dtype=torch.long).reshape(len(ids), max_len), torch.tensor(mask, dtype=torch.long).reshape(len(ids), max_len)


def get_score(occurences: int) -> float:
    """REF/utils/vqa_utils.py:10-20: VQA soft score of an answer given by `occurences` annotators."""
    return {0: 0.0, 1: 0.3, 2: 0.6, 3: 0.9}.get(occurences, 1.0)


def target_tensor(num_labels: int, labels: List[int], scores: List[float]) -> torch.Tensor:
    """REF/utils/vqa_utils.py:51-56."""
    target = torch.zeros(num_labels)
    target[labels] = torch.tensor(scores)
    return target


def _tokenize(tokenizer, text: str) -> List[int]:
    return tokenizer.convert_tokens_to_ids(tokenizer.tokenize(text)) if tokenizer is not None else []


def _loader(dataset, args, batch_size: int, shuffle: bool, collate) -> DataLoader:
    """One process: the reference's plain DataLoader.  Under torch.distributed (one process per GPU): `batch_size`
    is the GLOBAL batch and the loader yields this rank's strided share of it (climb_amd/data/sharding.py) -- training shares are padded so
    that every rank runs every step, evaluation shares are exact."""
    from .sharding import ShardedDataLoader, dp_rank_world
    rank, world = dp_rank_world()
    if world > 1:
        return ShardedDataLoader(dataset, batch_size, shuffle, collate, num_workers=getattr(args, "num_workers", 0), rank=rank, world=world, pad=shuffle)
    return DataLoader(dataset, num_workers=getattr(args, "num_workers", 0), batch_size=batch_size, shuffle=shuffle, collate_fn=collate)


# ------------------------------------------------------------------------------------------------ VQAv2
class VQADataset(Dataset):
    """REF vqa_dataset.py:35-187.  One item per (question, image) pair with the soft target over `num_labels` answers."""

    def __init__(self, data_dir: str, images_dataset: MSCOCOImagesDataset, split: str, **kwargs):
        self.images_dataset, self.data_dir, self.split = images_dataset, data_dir, split
        self.tokenizer = kwargs.get("tokenizer")
        self.annotations_file = os.path.join(data_dir, "v2_mscoco_{}2014_annotations.json".format(split))
        self.questions_file = os.path.join(data_dir, "v2_OpenEnded_mscoco_{}2014_questions.json".format(split))
        self.ans2label_file = os.path.join(data_dir, "ans2label.pkl")
        with open(self.ans2label_file, "rb") as f:
            self.ans2label = pkl.load(f)
        self.label2ans = {v: k for k, v in self.ans2label.items()}
        self.num_labels = len(self.label2ans)
        self.num_answers = len(self.ans2label)
        self.cached_data_file = os.path.join(data_dir, "cached_vqa_data", "vqa_{}.pkl".format(split))
        self.data = _load_or_build_cache(self.cached_data_file, self._parse)
        self.n_examples = len(self.data)
        logger.info("Loaded VQAv2 {} dataset, with {} examples".format(self.split, len(self.data)))

    def _parse(self):
        with open(self.questions_file) as f:
            qid2qdata = {x["question_id"]: x for x in json.load(f)["questions"]}
        with open(self.annotations_file) as f:
            annotations = json.load(f)["annotations"]
        data = []
        for anno in annotations:
            qid, image_id = anno["question_id"], anno["image_id"]
            qdata = qid2qdata[qid]
            assert qdata["image_id"] == image_id
            answer_count = defaultdict(int)
            for a in anno["answers"]:
                answer_count[a["answer"]] += 1
            labels, scores, answers = [], [], []
            for answer in answer_count:                      # answers outside the label vocabulary carry no target mass
                if answer not in self.ans2label:
                    continue
                labels.append(self.ans2label[answer])
                scores.append(get_score(answer_count[answer]))
                answers.append(answer)
            data.append({"question_id": qid, "image_id": image_id, "question": qdata["question"],
                         "question_input_ids": _tokenize(self.tokenizer, qdata["question"]), "correct_answer": anno["multiple_choice_answer"],
                         "labels": labels, "answers": answers, "scores": scores})
        return data

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index: int):
        ex = self.data[index]
        return {"question": ex["question"], "input_ids": ex["question_input_ids"], "image": self.images_dataset.get_image_data(ex["image_id"]),
                "labels": ex["labels"], "target_scores": target_tensor(self.num_labels, ex["labels"], ex["scores"]), "question_id": ex["question_id"]}

    def convert_to_low_shot(self, low_shot_percentage: float):
        assert self.split == "train"
        self.data = random.sample(self.data, int(low_shot_percentage * self.n_examples))
        self.n_examples = len(self.data)


def vqa_batch_collate(batch: List[Dict], visual_input_type: str):
    """REF vqa_dataset.py:189-234."""
    input_ids, attn_mask = _pad_ids([x["input_ids"] for x in batch])
    return {"raw_texts": [x["question"] for x in batch], "input_ids": input_ids, "attn_mask": attn_mask,
            "images": image_collate([x["image"] for x in batch], visual_input_type),
            "target_scores": torch.stack([x["target_scores"] for x in batch], dim=0), "labels": [x["labels"] for x in batch]}


def build_vqa_dataloader(args, data_dir: str, images_dataset: MSCOCOImagesDataset, split: str, visual_input_type: str, **kwargs) -> DataLoader:
    """REF vqa_dataset.py:236-268."""
    logger.info("Creating VQAv2 {} dataloader with batch size of {}".format(split, args.batch_size))
    return _loader(VQADataset(data_dir, images_dataset, split, **kwargs), args, args.batch_size, split == "train",
                   functools.partial(vqa_batch_collate, visual_input_type=visual_input_type))


# ------------------------------------------------------------------------------------------------ NLVR2
class NLVR2Dataset(Dataset):
    """REF nlvr2_dataset.py:30-142.  One sentence + two images + a binary label."""

    def __init__(self, data_dir: str, split: str, **kwargs):
        self.data_dir, self.num_labels, self.split = data_dir, 2, split
        _split = {"train": "train", "val": "dev", "test": "test1"}[split]
        self.image_dir = os.path.join(data_dir, "images", _split)
        self.cached_data_file = os.path.join(data_dir, "cached_nlvr2_data", f"{_split}.pkl")
        annotations_file = os.path.join(data_dir, "data", f"{_split}.json")

        def parse():
            data = []
            for annotation in _read_jsonl(annotations_file):
                stem = "-".join(annotation["identifier"].split("-")[:-1])
                data.append({"id": annotation["identifier"], "image_id_0": os.path.join(self.image_dir, stem + "-img0.png"),
                             "image_id_1": os.path.join(self.image_dir, stem + "-img1.png"), "sentence": str(annotation["sentence"]),
                             "labels": 0 if str(annotation["label"]) == "False" else 1})
            return data
        self.data = _load_or_build_cache(self.cached_data_file, parse)
        self.n_examples = len(self.data)
        logger.info("Loaded NLVRv2 {} dataset, with {} examples".format(split, self.n_examples))

    def get_pil_image(self, image_fn: str):
        return _load_rgb(image_fn, resize_shorter_edge)

    def __len__(self):
        return self.n_examples

    def __getitem__(self, index: int):
        ex = self.data[index]
        return {"text": ex["sentence"], "image": [self.get_pil_image(ex["image_id_0"]), self.get_pil_image(ex["image_id_1"])], "label": ex["labels"]}

    def convert_to_low_shot(self, num_shots_per_class: int):
        assert self.split == "train"
        new_data = []
        for i in range(self.num_labels):
            new_data.extend(random.sample([d for d in self.data if d["labels"] == i], num_shots_per_class))
        self.data, self.n_examples = new_data, len(new_data)


def nlvr2_batch_collate(batch: List[Dict], visual_input_type: str):
    """REF nlvr2_dataset.py:136-157."""
    assert visual_input_type == "pil-image"
    return {"raw_texts": [x["text"] for x in batch], "images": [x["image"] for x in batch], "labels": torch.LongTensor([x["label"] for x in batch])}


def build_nlvr2_dataloader(args, data_dir: str, split: str, visual_input_type: str, **kwargs) -> DataLoader:
    """REF nlvr2_dataset.py:159-190: batch_size / 2 examples (two encoder sequences each)."""
    if visual_input_type != "pil-image":
        raise NotImplementedError("Have not implemented other inputs for NLVR2 images!")
    logger.info("Creating NLVR2 {} dataloader with batch size of {}".format(split, int(args.batch_size / 2)))
    return _loader(NLVR2Dataset(data_dir, split, **kwargs), args, int(args.batch_size / 2), split == "train",
                   functools.partial(nlvr2_batch_collate, visual_input_type=visual_input_type))


# ------------------------------------------------------------------------------------------------ SNLI-VE
class SnliVEDataset(Dataset):
    """REF snli_ve_dataset.py:35-144.  One hypothesis + one Flickr30K image + a 3-way label."""
    categories = ["entailment", "contradiction", "neutral"]

    def __init__(self, data_dir: str, images_dataset: Flickr30KImagesDataset, split: str, **kwargs):
        self.data_dir, self.images_dataset, self.split = data_dir, images_dataset, split
        self.image_dir = os.path.join(data_dir, "flickr30k_images")
        self.tokenizer = kwargs.get("tokenizer")
        self.annotations_file = os.path.join(data_dir, "snli_ve_{}.jsonl".format(split))
        self.cat2label = {cat: i for i, cat in enumerate(self.categories)}
        self.num_labels = len(self.categories)
        self.cached_data_file = os.path.join(data_dir, "cached_ve_data", "snli-ve_{}.pkl".format(split))

        def parse():
            return [{"image_id": int(line["Flickr30K_ID"]), "hypothesis": str(line["sentence2"]),
                     "hypothesis_input_ids": _tokenize(self.tokenizer, str(line["sentence2"])), "label": self.cat2label[line["gold_label"]]}
                    for line in _read_jsonl(self.annotations_file)]
        self.data = _load_or_build_cache(self.cached_data_file, parse)
        self.n_examples = len(self.data)
        logger.info("Loaded SNLI-VE {} dataset, with {} examples".format(self.split, len(self.data)))

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index: int):
        ex = self.data[index]
        return {"hypothesis": ex["hypothesis"], "input_ids": ex["hypothesis_input_ids"], "image": self.images_dataset.get_image_data(ex["image_id"]),
                "label": ex["label"]}

    def convert_to_low_shot(self, num_shots_per_class: int):
        assert self.split == "train"
        new_data = []
        for i in range(self.num_labels):
            new_data.extend(random.sample([d for d in self.data if d["label"] == i], num_shots_per_class))
        self.data, self.n_examples = new_data, len(new_data)


def snlive_batch_collate(batch: List[Dict], visual_input_type: str):
    """REF snli_ve_dataset.py:145-190."""
    input_ids, attn_mask = _pad_ids([x["input_ids"] for x in batch])
    return {"raw_texts": [x["hypothesis"] for x in batch], "input_ids": input_ids, "attn_mask": attn_mask,
            "images": image_collate([x["image"] for x in batch], visual_input_type),
            "labels": torch.tensor([x["label"] for x in batch], dtype=torch.long)}


def build_snli_ve_dataloader(args, data_dir: str, images_dataset: Flickr30KImagesDataset, split: str, visual_input_type: str, **kwargs) -> DataLoader:
    """REF snli_ve_dataset.py:192-219."""
    logger.info("Creating SNLI-VE {} dataloader with batch size of {}".format(split, args.batch_size))
    return _loader(SnliVEDataset(data_dir, images_dataset, split, **kwargs), args, args.batch_size, split == "train",
                   functools.partial(snlive_batch_collate, visual_input_type=visual_input_type))


# ------------------------------------------------------------------------------------------------ VCR
GENDER_NEUTRAL_NAMES = ["Casey", "Riley", "Jessie", "Jackie", "Avery", "Jaime", "Peyton", "Kerry", "Jody", "Kendall", "Skyler", "Frankie", "Pat",
                        "Quinn", "Morgan", "Finley", "Harley", "Robbie", "Sidney", "Tommie", "Ashley", "Carter", "Adrian", "Clarke", "Logan",
                        "Mickey", "Nicky", "Parker", "Tyler", "Reese", "Charlie", "Austin", "Denver", "Emerson", "Tatum", "Dallas", "Haven",
                        "Jordan", "Robin", "Rory", "Bellamy", "Salem", "Sutton", "Gray", "Shae", "Kyle", "Alex", "Ryan", "Cameron", "Dakota"]


def process_list(mytext, objects) -> str:
    """REF vcr_dataset.py:40-63: object references -> a gender-neutral name (persons) or 'the gray <class>'.  Quirks kept because they
    decide the text the model sees: a list of references contributes only its LAST element, and a bare int reference is resolved with
    the index left over from the most recent list (the reference's variable reuse)."""
    text = ""
    subelement = None
    for element in mytext:
        if type(element) == list:
            for subelement in element:
                if objects[int(subelement)] == "person":
                    temporal_text = GENDER_NEUTRAL_NAMES[int(subelement)]
                else:
                    temporal_text = "the gray " + str(objects[int(subelement)]).strip()
        elif type(element) == int:
            if objects[int(element)] == "person":
                temporal_text = GENDER_NEUTRAL_NAMES[int(subelement)]
            else:
                temporal_text = "the gray " + str(objects[int(subelement)])
        else:
            temporal_text = element
        text += temporal_text + " "
    return text


class VCRDataset(Dataset):
    """REF vcr_dataset.py:65-187.  Four choice texts ('question [SEP] answer' or, for QA->R, '... [SEP] rationale') + one drawn-box image."""

    def __init__(self, data_dir: str, split: str, task_type: str = "qa", **kwargs):
        self.data_dir, self.split, self.task_type = data_dir, split, task_type
        self.image_dir = os.path.join(data_dir, "vcr")
        self.tokenizer = kwargs.get("tokenizer")
        self.annotations_file = os.path.join(data_dir, "annotation/{}.jsonl".format(split))
        self.cached_data_file = os.path.join(data_dir, "cached_vcr_data", "vcr_" + str(task_type) + "_" + "{}.pkl".format(split))

        def parse():
            data = []
            for line in _read_jsonl(self.annotations_file):
                image_path = os.path.join("drawn_images/bbox/" + str(split) + "/" + str(task_type) + "/" + str(line["annot_id"]) + ".jpg")
                objects = line["objects"]
                question = process_list(line["question"], objects)
                texts = []
                if task_type == "qa":
                    for answer in line["answer_choices"]:
                        texts.append(question + " [SEP] " + process_list(answer, objects))
                    label = int(line["answer_label"])
                else:
                    answer = process_list(line["answer_choices"][int(line["answer_label"])], objects)
                    for rationale in line["rationale_choices"]:
                        texts.append(question + " [SEP] " + answer + " [SEP] " + process_list(rationale, objects))
                    label = int(line["rationale_label"])
                ids = [_tokenize(self.tokenizer, t) for t in texts] if self.tokenizer is not None else []
                data.append({"image_path": image_path, "texts": texts, "input_ids": ids, "label": label})
            return data
        self.data = _load_or_build_cache(self.cached_data_file, parse)
        self.n_examples = len(self.data)
        logger.info("Loaded VCR-{} {} dataset, with {} examples".format(self.task_type, self.split, len(self.data)))

    def __len__(self):
        return self.n_examples

    def __getitem__(self, index: int):
        ex = self.data[index]
        return {"texts": ex["texts"], "image": _load_rgb(os.path.join(self.data_dir, ex["image_path"]), resize_shorter_edge), "label": ex["label"]}

    def convert_to_low_shot(self, low_shot_percentage: float):
        assert self.split == "train"
        self.data = random.sample(self.data, int(low_shot_percentage * self.n_examples))
        self.n_examples = len(self.data)


def vcr_batch_collate(batch: List[Dict], visual_input_type: str):
    """REF vcr_dataset.py:189-210."""
    assert visual_input_type == "pil-image"
    return {"raw_texts": [x["texts"] for x in batch], "images": [x["image"] for x in batch], "labels": torch.LongTensor([x["label"] for x in batch])}


def build_vcr_dataloader(args, data_dir: str, split: str, task_type: str, visual_input_type: str, **kwargs) -> DataLoader:
    """REF vcr_dataset.py:212-248: batch_size / 4 examples (four encoder sequences each)."""
    assert visual_input_type == "pil-image"
    batch_size = int(args.batch_size / 4)
    logger.info("Creating VCR {} dataloader with batch size of {}".format(split, batch_size))
    return _loader(VCRDataset(data_dir, split, task_type, **kwargs), args, batch_size, split == "train",
                   functools.partial(vcr_batch_collate, visual_input_type=visual_input_type))
