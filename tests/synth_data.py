"""A tiny synthetic copy of the reference's data tree (`--climb_data_dir`), in the reference's own file formats
(REF/data/visionlanguage_datasets/*.py, REF/data/image_datasets/*.py), for the dataset / trainer / driver tests.
Deterministic in `seed`; a few images are larger than 384 px on both edges so the datasets' pre-shrink runs."""
import json
import os
import pickle

import numpy as np

VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "what", "is", "the", "color", "of", "cat", "dog", "a", "on", "left", "right", "two", "there",
         "are", "man", "woman", "why", "smiling", "at", "gray", "person", "bottle", "yes", "no", "red", "blue", "casey", "riley", "jessie", "?", ".",
         "holding", "because", "they", "like", "it", "in", "both", "images", "one", "image", "shows", "sitting", "standing", "outside"]
WORDS = VOCAB[5:34] + VOCAB[36:]


def write_vocab(path):
    with open(path, "w") as f:
        f.write("\n".join(VOCAB) + "\n")
    return path


def _sentence(rng, n):
    return " ".join(rng.choice(WORDS, size=n).tolist())


def _image(rng, w, h):
    from PIL import Image
    return Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), "RGB")


def _size(rng, i):
    # every fourth image exceeds 384 on both edges (the datasets' `min(size) > 384` pre-shrink); the others are small and untouched
    return [(500, 400), (420, 900)][(i // 4) % 2] if i % 4 == 3 else (int(rng.integers(40, 120)), int(rng.integers(40, 120)))


def make_climb_data_tree(root, n_train=8, n_val=4, seed=0, num_answers=3129, easy_answer=None):
    """easy_answer=k: two extra annotators of EVERY VQA question answer `ans<k>` (soft score 0.6), so that a model that always predicts k
    scores above the random baseline of 0 -- the forgetting metric divides by that margin."""
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)
    # ---- MS-COCO images + VQAv2
    coco = os.path.join(root, "ms-coco", "images")
    os.makedirs(coco)
    n_img = n_train + n_val
    for i in range(n_img):
        _image(rng, *_size(rng, i)).save(os.path.join(coco, f"{1000 + i}.jpg"), quality=90)
    vqa = os.path.join(root, "vqav2")
    os.makedirs(os.path.join(vqa, "cached_vqa_data"))
    ans2label = {f"ans{i}": i for i in range(num_answers)}
    with open(os.path.join(vqa, "ans2label.pkl"), "wb") as f:
        pickle.dump(ans2label, f)
    qid = 0
    for split, lo, n in (("train", 0, n_train), ("val", n_train, n_val)):
        questions, annotations = [], []
        for j in range(n):
            image_id = 1000 + lo + j
            qid += 1
            questions.append({"question_id": qid, "image_id": image_id, "question": _sentence(rng, int(rng.integers(3, 9))) + " ?"})
            main = f"ans{int(rng.integers(0, num_answers))}"
            other = f"ans{int(rng.integers(0, num_answers))}"
            k = int(rng.integers(1, 11))                     # k annotators gave `main`, the rest `other` / an out-of-vocabulary answer
            answers = [{"answer": main}] * k + [{"answer": other}] * ((10 - k) // 2) + [{"answer": "not in vocabulary"}] * (10 - k - (10 - k) // 2)
            if easy_answer is not None:
                answers = answers + [{"answer": f"ans{easy_answer}"}] * 2
            annotations.append({"question_id": qid, "image_id": image_id, "multiple_choice_answer": main, "answers": answers})
        json.dump({"questions": questions}, open(os.path.join(vqa, f"v2_OpenEnded_mscoco_{split}2014_questions.json"), "w"))
        json.dump({"annotations": annotations}, open(os.path.join(vqa, f"v2_mscoco_{split}2014_annotations.json"), "w"))
    # ---- NLVR2
    nlvr = os.path.join(root, "nlvr2")
    os.makedirs(os.path.join(nlvr, "data"))
    os.makedirs(os.path.join(nlvr, "cached_nlvr2_data"))
    for split, n in (("train", n_train), ("dev", n_val)):
        os.makedirs(os.path.join(nlvr, "images", split))
        with open(os.path.join(nlvr, "data", f"{split}.json"), "w") as f:
            for j in range(n):
                stem = f"{split}-{100 + j}-{j % 3}"
                for side in (0, 1):
                    _image(rng, *_size(rng, 2 * j + side)).save(os.path.join(nlvr, "images", split, f"{stem}-img{side}.png"))
                f.write(json.dumps({"identifier": f"{stem}-0", "sentence": _sentence(rng, int(rng.integers(4, 10))) + " .",
                                    "label": "True" if rng.integers(0, 2) else "False"}) + "\n")
    # ---- Flickr30K images + SNLI-VE
    flickr = os.path.join(root, "flickr30k", "flickr30k_images")
    os.makedirs(flickr)
    for i in range(n_img):
        _image(rng, *_size(rng, i + 1)).save(os.path.join(flickr, f"{2000 + i}.jpg"), quality=90)
    ve = os.path.join(root, "snli-ve")
    os.makedirs(os.path.join(ve, "cached_ve_data"))
    for split, lo, n in (("train", 0, n_train), ("dev", n_train, n_val)):
        with open(os.path.join(ve, f"snli_ve_{split}.jsonl"), "w") as f:
            for j in range(n):
                f.write(json.dumps({"Flickr30K_ID": str(2000 + lo + j), "sentence2": _sentence(rng, int(rng.integers(3, 9))),
                                    "gold_label": ["entailment", "contradiction", "neutral"][int(rng.integers(0, 3))]}) + "\n")
    # ---- VCR
    vcr = os.path.join(root, "vcr")
    os.makedirs(os.path.join(vcr, "annotation"))
    os.makedirs(os.path.join(vcr, "cached_vcr_data"))
    objects_pool = ["person", "person", "bottle", "dog", "person"]
    for split, n in (("train", n_train), ("val", n_val)):
        os.makedirs(os.path.join(vcr, "drawn_images", "bbox", split, "qa"))
        with open(os.path.join(vcr, "annotation", f"{split}.jsonl"), "w") as f:
            for j in range(n):
                annot_id = f"{split}-{j}"
                _image(rng, *_size(rng, j + 2)).save(os.path.join(vcr, "drawn_images", "bbox", split, "qa", f"{annot_id}.jpg"), quality=90)
                nobj = int(rng.integers(2, 6))
                objects = objects_pool[:nobj]

                def toks(m):
                    out = []
                    for _ in range(m):
                        r = rng.random()
                        if r < 0.25:
                            out.append([int(v) for v in rng.integers(0, nobj, size=int(rng.integers(1, 3)))])       # reference(s) to detected objects
                        else:
                            out.append(str(rng.choice(WORDS)))
                    return out
                f.write(json.dumps({"annot_id": annot_id, "objects": objects, "question": toks(int(rng.integers(3, 7))) + ["?"],
                                    "answer_choices": [toks(int(rng.integers(2, 6))) + ["."] for _ in range(4)], "answer_label": int(rng.integers(0, 4)),
                                    "rationale_choices": [toks(int(rng.integers(2, 6))) + ["."] for _ in range(4)],
                                    "rationale_label": int(rng.integers(0, 4))}) + "\n")
    return root


def make_tokenizer(vocab_path, fast=False):
    """A BERT word-piece tokenizer over `vocab_path`.  transformers 5.x takes the file as `vocab=` (and silently ignores `vocab_file=`,
    leaving a 5-token vocabulary in which every word is [UNK]); 4.x takes `vocab_file=`."""
    import transformers
    cls = transformers.BertTokenizerFast if fast else transformers.BertTokenizer
    words = [w.strip() for w in open(vocab_path) if w.strip()]
    for kw in ({"vocab": vocab_path}, {"vocab_file": vocab_path}):
        try:
            tok = cls(do_lower_case=True, **kw)
        except Exception:      # noqa: BLE001
            continue
        if tok.convert_tokens_to_ids(words[5]) == 5:
            return tok
    raise RuntimeError("could not build a BERT tokenizer from a vocabulary file with this transformers version")
