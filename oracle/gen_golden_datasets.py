"""tests/golden/datasets.json: what the REFERENCE's own dataset classes and collate functions produce on the synthetic data tree of
tests/synth_data.py (row F1).  BUILD-CONTAINER ONLY (imports /root/reference).

    python oracle/gen_golden_datasets.py

The reference datasets import torchvision and jsonlines, which this image lacks.  They are replaced, for this run only, by throw-away
modules written to a temporary directory: `jsonlines.open` = iterate json lines, and `torchvision.transforms.Resize` = PIL bilinear
resize to torchvision's documented output-size rule (`_compute_resized_output_size`).  So the fixture pins annotation parsing, label /
score construction, text processing (incl. VCR's object-reference quirks), batch dictionaries and loader batch sizes to the reference;
the pre-shrink image sizes are pinned only to that documented rule."""
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_data as sd      # noqa: E402

REF_SRC = "/root/reference/src"

TV = '''
from PIL import Image
class Resize:
    def __init__(self, size, max_size=None):
        self.size, self.max_size = size, max_size
    def __call__(self, img):
        w, h = img.size
        if isinstance(self.size, (tuple, list)):
            return img.resize((self.size[1], self.size[0]), Image.BILINEAR)
        short, long_ = (w, h) if w <= h else (h, w)
        new_short, new_long = self.size, int(self.size * long_ / short)
        if self.max_size is not None and new_long > self.max_size:
            new_short, new_long = int(self.max_size * new_short / new_long), self.max_size
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        return img.resize((nw, nh), Image.BILINEAR)
class _T:
    def __init__(self, *a, **k): pass
    def __call__(self, x): return x
Compose = ToTensor = Normalize = _T
'''
JL = '''
import json, io
class _Reader:
    def __init__(self, path): self.f = io.open(path)
    def __iter__(self):
        for line in self.f:
            line = line.strip()
            if line: yield json.loads(line)
    def __enter__(self): return self
    def __exit__(self, *a): self.f.close()
def open(path, *a, **k): return _Reader(path)
'''


def main():
    assert os.path.isdir(REF_SRC), "needs /root/reference"
    stub = tempfile.mkdtemp(prefix="climb_ds_stubs_")
    os.makedirs(os.path.join(stub, "torchvision"))
    open(os.path.join(stub, "torchvision", "__init__.py"), "w").write("from . import transforms\n")
    open(os.path.join(stub, "torchvision", "transforms.py"), "w").write(TV)
    os.makedirs(os.path.join(stub, "jsonlines"))
    open(os.path.join(stub, "jsonlines", "__init__.py"), "w").write(JL)
    sys.path.insert(0, stub)
    sys.path.insert(0, REF_SRC)
    from data.image_datasets.cocoimages_dataset import MSCOCOImagesDataset
    from data.image_datasets.flickr30kimages_dataset import Flickr30KImagesDataset
    from data.visionlanguage_datasets.vqa_dataset import build_vqa_dataloader
    from data.visionlanguage_datasets.nlvr2_dataset import build_nlvr2_dataloader
    from data.visionlanguage_datasets.snli_ve_dataset import build_snli_ve_dataloader
    from data.visionlanguage_datasets.vcr_dataset import build_vcr_dataloader, process_list
    import transformers
    root = sd.make_climb_data_tree(tempfile.mkdtemp(prefix="climb_synth_"), n_train=8, n_val=4, seed=0)
    import shutil
    shutil.copytree(os.path.join(root, "vqav2"), os.path.join(root, "vqav2-tok"))       # a second copy: the parse cache stores the token ids
    vocab = sd.write_vocab(os.path.join(root, "vocab.txt"))
    tok = sd.make_tokenizer(vocab)
    args = types.SimpleNamespace(batch_size=4, num_workers=0, visual_input_type="pil-image")

    def dump(batch):
        out = {}
        for k, v in batch.items():
            if k == "images":
                out[k] = [[list(i.size) for i in im] if isinstance(im, list) else list(im.size) for im in v]
            elif k == "target_scores":
                out[k] = [[int(r), int(c), round(float(v[r, c]), 6)] for r, c in v.nonzero().tolist()]
                out["target_scores_shape"] = list(v.shape)
            elif hasattr(v, "tolist"):
                out[k] = v.tolist()
            else:
                out[k] = v
        return out
    coco = MSCOCOImagesDataset(os.path.join(root, "ms-coco"), "pil-image")
    flickr = Flickr30KImagesDataset(os.path.join(root, "flickr30k"), "pil-image")
    golden = {"seed": 0, "n_train": 8, "n_val": 4, "batch_size": 4, "loaders": {}}
    loaders = {
        "vqa/val": build_vqa_dataloader(args, os.path.join(root, "vqav2"), coco, "val", "pil-image"),
        "vqa/val/tokenized": build_vqa_dataloader(args, os.path.join(root, "vqav2-tok"), coco, "val", "pil-image", tokenizer=tok)
        if os.path.isdir(os.path.join(root, "vqav2-tok")) else None,
        "nlvr2/val": build_nlvr2_dataloader(args, os.path.join(root, "nlvr2"), "val", "pil-image"),
        "snli-ve/dev": build_snli_ve_dataloader(args, os.path.join(root, "snli-ve"), flickr, "dev", "pil-image"),
        "vcr/val": build_vcr_dataloader(args, os.path.join(root, "vcr") + "/", "val", "qa", "pil-image"),
    }
    for name, dl in loaders.items():
        if dl is None:
            continue
        golden["loaders"][name] = {"n_examples": len(dl.dataset), "n_batches": len(dl), "batches": [dump(b) for b in dl]}
    # the train splits: sizes only (their loaders shuffle)
    golden["train_sizes"] = {
        "vqa": len(build_vqa_dataloader(args, os.path.join(root, "vqav2"), coco, "train", "pil-image").dataset),
        "nlvr2": len(build_nlvr2_dataloader(args, os.path.join(root, "nlvr2"), "train", "pil-image").dataset),
        "snli-ve": len(build_snli_ve_dataloader(args, os.path.join(root, "snli-ve"), flickr, "train", "pil-image").dataset),
        "vcr": len(build_vcr_dataloader(args, os.path.join(root, "vcr") + "/", "train", "qa", "pil-image").dataset)}
    golden["process_list"] = [
        {"text": t, "objects": o, "out": process_list(t, o)} for t, o in [
            (["Why", "is", [0], "smiling", "at", [1], "?"], ["person", "person", "bottle"]),
            (["Is", [2], "next", "to", [0, 1], "?"], ["person", "person", "bottle"]),
            ([[1], "holds", [2, 0], "."], ["dog", "person", "bottle"])]]
    out = os.path.join(ROOT, "tests", "golden", "datasets.json")
    json.dump(golden, open(out, "w"), indent=1)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
