"""ViLT-BERT behind the reference's names (REF/modeling/viltbert.py): a frozen BERT-base encodes the text, its last hidden state
replaces ViLT's word-embedding lookup (`inputs_embeds`), everything after that is the ViLT engine.

  ViltBertEncoderWrapper (:31-168)  ViltBertContinualLearner (:171-437)  load_viltbert_encoder (:459-493)
  create_viltbert_continual_learner_model (:495-522)  convert_batch_to_viltbert_input_dict (:524-530)

State-dict keys are the reference's (`viltbert_encoder.vilt.*`, `viltbert_encoder.bert.*`, `task_layer.*`), so checkpoints interchange.
The language-only helpers of the reference file (`reallocate_text_image`, the `ViltBertForSequenceClassification` /
`...ForMultipleChoice` low-shot heads, :56-84, :439-457) belong to the downstream low-shot drivers, outside SURVEY.md §8."""
from __future__ import annotations

import logging
from typing import Dict, List, Optional

import torch

from ..bert import BertParams
from .vilt import (ViltContinualLearner, ViltEncoderWrapper, ViltModelParams, _load_encoder_state, _make_processor, _offline_processor, init_like_hf)

logger = logging.getLogger(__name__)


class ViltBertEncoderWrapper(ViltEncoderWrapper):
    """REF/modeling/viltbert.py:31-168."""

    def __init__(self, processor, vilt: ViltModelParams, bert: BertParams, device: torch.device, precision: Optional[str] = None):
        super().__init__(processor, vilt, device, precision)
        self.bert = bert
        self.bert.precision = "fp32" if self.precision == "bf16x3" else self.precision      # (the frozen BERT has no split path: exact fp32 under the split mode)

    def get_bert_outputs(self, **encodings) -> torch.Tensor:
        """REF:115-121: BERT's last hidden state, no gradient.  [B, roundup(T, 32), 768] fp32, first T rows of a sequence valid.
        Like the reference, BERT is NOT put in eval mode here: while the learner trains, its dropouts are live (climb_amd/bert.py).
        `self.bert_dropout_masks` (test hook, consumed by one call) supplies the keep-masks instead of drawing them."""
        masks, self.bert_dropout_masks = getattr(self, "bert_dropout_masks", None), None
        with torch.no_grad():
            return self.bert(input_ids=encodings["input_ids"], attention_mask=encodings["attention_mask"], token_type_ids=encodings["token_type_ids"],
                             dropout_masks=masks)

    def prepare_encodings(self, enc: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """REF:145-147: `inputs_embeds` = BERT features, `input_ids` = None."""
        if enc.get("inputs_embeds") is not None:
            return enc
        enc = dict(enc)
        enc["inputs_embeds"] = self.get_bert_outputs(**enc)
        enc["input_ids"] = None
        return enc

    def create_optimizer(self, hparams):
        """REF:123-133 defines the optimizer factory on the encoder wrapper as well; the fused optimizer needs the whole learner's flat
        buffers, so this forwards to the learner that owns this encoder."""
        if self._host is None or getattr(self._host, "learner", None) is None:
            raise RuntimeError("create_optimizer: this encoder is not attached to a continual learner")
        return self._host.learner.create_optimizer(hparams)


class ViltBertContinualLearner(ViltContinualLearner):
    """REF/modeling/viltbert.py:171-437: identical to the ViLT learner but for the encoder attribute."""
    encoder_attr = "viltbert_encoder"

    def __init__(self, ordered_cl_tasks: List[str], encoder: ViltBertEncoderWrapper, encoder_dim: int, task_configs: Dict):
        super().__init__(ordered_cl_tasks, encoder, encoder_dim, task_configs)
        self._host.learner = self


def load_viltbert_encoder(checkpoint_name: str, device: torch.device, pretrained_vilt_name: str, precision: Optional[str] = None) -> ViltBertEncoderWrapper:
    """REF/modeling/viltbert.py:459-493.  `random-init[:seed]` builds both architectures with HF's initialisers (no network here)."""
    logger.info("-" * 100)
    logger.info("Loading ViLT encoder model: {}".format(checkpoint_name))
    device = torch.device(device)
    bert = BertParams()
    if checkpoint_name.startswith("random-init"):
        seed = int(checkpoint_name.split(":")[1]) if ":" in checkpoint_name else None
        vilt = ViltModelParams(2)
        init_like_hf(vilt, seed)
        bert.init_like_hf(None if seed is None else seed + 1)
        return ViltBertEncoderWrapper(_offline_processor(), vilt.to(device), bert.to(device), device, precision)
    processor = _make_processor(pretrained_vilt_name)
    try:                                                    # REF:477 BertModel.from_pretrained("bert-base-uncased")
        import transformers
        hf = transformers.BertModel.from_pretrained("bert-base-uncased")
        bert.load_state_dict({k: v for k, v in hf.state_dict().items() if k in bert.state_dict()})
    except Exception as e:      # noqa: BLE001
        raise OSError(f"bert-base-uncased weights are needed for ViLT-BERT and could not be loaded ({type(e).__name__}: {e})")
    rows = 3 if (checkpoint_name != pretrained_vilt_name and "nlvr2" in checkpoint_name) else 2
    vilt = ViltModelParams(rows)
    init_like_hf(vilt)
    enc = ViltBertEncoderWrapper(processor, vilt, bert, device, precision)
    if checkpoint_name == pretrained_vilt_name:
        _load_encoder_state(enc, checkpoint_name)          # HF ViLT checkpoint: `vilt.*` keys; BERT stays bert-base-uncased
    else:
        sd = torch.load(checkpoint_name, map_location="cpu")
        enc.load_state_dict({k: v for k, v in sd.items() if not k.endswith("position_ids")})
    enc.vilt.to(device)
    enc.bert.to(device)
    logger.info("Successfully loaded pretrained ViLT-BERT encoder")
    return enc


def create_viltbert_continual_learner_model(model_name_or_path: str, ordered_cl_tasks: List[str], model_config: Dict, task_configs: Dict,
                                            device: torch.device, precision: Optional[str] = None):
    """REF/modeling/viltbert.py:495-522."""
    encoder = load_viltbert_encoder(checkpoint_name=model_name_or_path, device=device, pretrained_vilt_name=model_name_or_path, precision=precision)
    cl_model = ViltBertContinualLearner(ordered_cl_tasks=ordered_cl_tasks, encoder=encoder, encoder_dim=model_config["encoder_dim"], task_configs=task_configs)
    logger.info("Successfully created and initialized ViLT-BERT Continual Learner model")
    return cl_model


def convert_batch_to_viltbert_input_dict(batch: Dict):
    """REF/modeling/viltbert.py:524-530."""
    return {"images": batch["images"], "texts": batch["raw_texts"] if "encodings" not in batch else batch["encodings"]}
