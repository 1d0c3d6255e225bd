"""tests/golden/viltbert_train_dropout.json: what the one reference behaviour this build does NOT reproduce is worth.  BUILD-CONTAINER ONLY.

    python oracle/measure_bert_dropout.py

REF/modeling/viltbert.py:115-120 runs the frozen BERT under `torch.no_grad()` but never puts it in eval mode, so while the learner is in
train mode BERT's 37 dropouts (p = 0.1: embeddings, 12 x {attention probabilities, attention output, FFN output}) perturb the "frozen"
text features with torch's global RNG stream.  This build computes the deterministic eval-mode features.  Here the
reference's own ViltBertContinualLearner is run on the viltbert_vqa_b3 batch in eval mode and, for several torch seeds, in train mode;
recorded: how far the train-mode features are from the eval-mode ones and what that does to one step's loss and gradients."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import bert_oracle as bo          # noqa: E402
import ref_import as ri           # noqa: E402
import vilt_oracle as vo          # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "viltbert_train_dropout.json")


def main(tasks=("vqa", "nlvr2"), B=3, wseed=42, bseed=7, dseed=21, seeds=(0, 1, 2, 3, 4, 5, 6, 7)):
    assert ri.reference_available()
    tasks = list(tasks)
    P, PB = vo.init_params(tasks, wseed), bo.init_bert_params(bseed)
    enc = vo.synthetic_encodings(B, seed=dseed, ragged_text=True)
    target = vo.synthetic_vqa_targets(B, seed=dseed)
    model = ri.build_reference_viltbert_learner(tasks, P, PB)
    trainer = ri.make_trainer("vqa")
    import modeling.viltbert as ref_vb
    trainer.batch2inputs_converter = ref_vb.convert_batch_to_viltbert_input_dict
    model.viltbert_encoder.process_inputs = lambda images, texts: dict(enc)
    batch = {"raw_texts": [""] * B, "images": None, "target_scores": target}
    valid = enc["attention_mask"].bool()

    def step(train, seed):
        model.train(train)
        model.zero_grad()
        torch.manual_seed(seed)
        feats = model.viltbert_encoder.get_bert_outputs(**enc)
        torch.manual_seed(seed)            # the step below draws the same masks again
        loss, (pooled, logits), _, _ = trainer.train_step(model, batch, None, None, None)
        g = torch.cat([p.grad.reshape(-1) for n, p in model.named_parameters() if p.grad is not None]).double()
        return feats[valid].double(), float(loss), logits.detach().double(), g
    f0, l0, lg0, g0 = step(False, 0)
    rows = []
    for s in seeds:
        f, l, lg, g = step(True, s)
        rows.append(dict(seed=s, feature_rel_rms=float((f - f0).norm() / f0.norm()), loss=l, loss_rel=abs(l - l0) / abs(l0),
                         logits_rel_max=float((lg - lg0).abs().max() / lg0.abs().max()), grad_rel_l2=float((g - g0).norm() / g0.norm())))
        print(rows[-1])
    out = dict(case="viltbert_vqa_b3 batch (B=3, ragged text), reference ViltBertContinualLearner, random-init weights",
               eval_loss=l0, rows=rows, mean_feature_rel_rms=float(np.mean([r["feature_rel_rms"] for r in rows])),
               mean_loss_rel=float(np.mean([r["loss_rel"] for r in rows])), mean_grad_rel_l2=float(np.mean([r["grad_rel_l2"] for r in rows])),
               mean_train_loss=float(np.mean([r["loss"] for r in rows])))
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, {k: v for k, v in out.items() if k.startswith("mean")})


if __name__ == "__main__":
    main()
