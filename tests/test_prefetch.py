"""PrefetchLoader (row F1, host logic): order, bounded look-ahead, error delivery, early exit; on the GPU the side-stream hand-off."""
import threading
import time

import pytest
import torch

from climb_amd.data import PrefetchLoader


def test_order_and_bounded_lookahead():
    produced, lock = [], threading.Lock()

    def prepare(i):
        with lock:
            produced.append(i)
        return i * i

    seen = []
    for v in PrefetchLoader(range(20), prepare, depth=2):
        time.sleep(0.01)                       # a slow consumer: the worker may run ahead by `depth` (+ the one it holds)
        with lock:
            ahead = len(produced) - len(seen)
        assert ahead <= 2 + 2, ahead
        seen.append(v)
    assert seen == [i * i for i in range(20)]


def test_worker_exception_surfaces_at_its_batch():
    def prepare(i):
        if i == 3:
            raise ValueError("bad batch 3")
        return i

    got = []
    with pytest.raises(ValueError, match="bad batch 3"):
        for v in PrefetchLoader(range(10), prepare, depth=3):
            got.append(v)
    assert got == [0, 1, 2]


def test_early_break_stops_the_worker():
    calls = []
    it = PrefetchLoader(range(10 ** 6), lambda i: calls.append(i) or i, depth=2)
    for v in it:
        if v == 5:
            break
    time.sleep(0.3)
    n = len(calls)
    time.sleep(0.3)
    assert len(calls) == n and n < 50           # the worker is not still consuming the source
    with pytest.raises(ValueError):
        PrefetchLoader([], lambda x: x, depth=0)


@pytest.mark.gpu
def test_side_stream_handoff_is_ordered():
    """Work enqueued by the worker on its side stream is visible to kernels the consumer enqueues afterwards on its own stream."""
    dev = torch.device("cuda:0")

    def prepare(i):
        x = torch.full((1 << 22,), float(i), device=dev)
        for _ in range(20):
            x = x * 1.0 + 0.0                  # keep the side stream busy so a missing dependency would be observable
        return x

    for i, x in enumerate(PrefetchLoader(range(8), prepare, depth=2, device=dev)):
        assert float(x.sum()) == float(i) * (1 << 22)


@pytest.mark.gpu
def test_trainer_steps_through_the_prefetcher_equal_inline_steps(tmp_path):
    """Row F1 wired in: `VLTaskTrainer.prefetched` hands `train_step` batches whose tokenisation, raw-byte staging,
    H2D copies and device image kernels already ran on the worker thread / side stream.  Same data order, same arithmetic: the losses of
    the first steps equal those of the inline path (fp32 mode: deterministic), and the evaluation score is the same."""
    import os
    import types
    from tests import synth_data
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.train.task_trainer import VQATrainer
    dev = torch.device("cuda:0")
    root = str(tmp_path / "data")
    synth_data.make_climb_data_tree(root, n_train=12, n_val=5, seed=1, easy_answer=7)
    os.environ["CLIMB_AMD_TOKENIZER_VOCAB"] = synth_data.write_vocab(str(tmp_path / "vocab.txt"))
    try:
        args = types.SimpleNamespace(climb_data_dir=root, batch_size=4, num_workers=0, cl_algorithm="sequential_ft", visual_input_type="pil-image")
        out = {}
        for mode in ("1", "0"):
            os.environ["CLIMB_AMD_PREFETCH"] = mode
            torch.manual_seed(3)
            model = create_continual_learner_map["vilt"](model_name_or_path="random-init:5", ordered_cl_tasks=["vqa"], model_config=model_configs["vilt"],
                                                         task_configs=task_configs, device=dev, precision="fp32")
            trainer = VQATrainer(args, task_configs, model_configs["vilt"], dev)
            opt = model.create_optimizer(trainer.hparams)
            model.train()
            losses, kinds = [], []
            for batch in trainer.prefetched(model, trainer.train_dataloader):
                kinds.append((isinstance(batch["images"], dict), "encodings" in batch))
                loss, _, _, _ = trainer.train_step(model, batch, opt)
                losses.append(float(loss))
            out[mode] = (losses, kinds, trainer.eval(model))
        assert all(k == (True, True) for k in out["1"][1]) and all(k == (False, False) for k in out["0"][1])
        assert len(out["1"][0]) == 3 and out["1"][0] == pytest.approx(out["0"][0], rel=1e-6)
        assert out["1"][2] == pytest.approx(out["0"][2], abs=1e-6)
    finally:
        os.environ.pop("CLIMB_AMD_PREFETCH", None)
        os.environ.pop("CLIMB_AMD_TOKENIZER_VOCAB", None)
