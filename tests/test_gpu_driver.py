"""The upstream continual-learning driver on the real engine.

tests/golden/driver_calls.json holds, per scenario, the calls the REFERENCE driver (REF/train/train_upstream_continual_learning.py,
run unchanged through integration/climb_shim by oracle/record_driver_calls.py) makes into this package.  Here the same scenarios run on
the MI355X through tests/upstream_driver.py -- a call-for-call restatement of the driver's main() -- under the same recorder, on the same
synthetic data tree (PIL images + strings through the real datasets, trainers, device image pipeline, plug-ins, checkpoints and CL
metrics), and must produce the same call sequence, the same files and (where predictions are pinned) the same scores.
Scenarios: BASELINE.json configs[2] (adapters VQA -> NLVR2) and configs[3] / [4] (EWC / ER over VQA -> NLVR2 -> SNLI-VE -> VCR, one GPU);
the low-shot transfer driver (REF/train/train_lowshot_multimodal.py) from scratch and from the checkpoints of an upstream run."""
import json
import os

import pytest
import torch

from tests import driver_scenarios as sc
from tests import driver_trace, lowshot_driver, synth_data, upstream_driver

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def data_tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("climb_data"))
    synth_data.make_climb_data_tree(os.path.join(root, "data"), n_train=sc.N_TRAIN, n_val=sc.N_VAL, seed=sc.SEED, easy_answer=sc.EASY_ANSWER)
    os.environ["CLIMB_AMD_TOKENIZER_VOCAB"] = synth_data.write_vocab(os.path.join(root, "vocab.txt"))
    yield os.path.join(root, "data")
    os.environ.pop("CLIMB_AMD_TOKENIZER_VOCAB", None)


def _pin_predictions(model):
    """What the recorder's stand-in C ABI did on the CPU: classification heads always answer index 7 (VQA) / class 0, so the scores
    the driver writes are the same numbers in both runs (VCR's one-logit-per-choice head cannot be pinned and is not compared)."""
    with torch.no_grad():
        for task, layer in model.task_layer.items():
            if task == "vqa":
                layer[3].bias[sc.EASY_ANSWER] = 50.0
            elif task in ("nlvr2", "snli-ve"):
                layer[3].bias[0] = 50.0


@pytest.mark.parametrize("name", list(sc.SCENARIOS))
def test_upstream_driver_scenario_matches_the_reference_drivers_call_sequence(name, data_tree, tmp_path, golden_dir):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    golden = json.load(open(os.path.join(golden_dir, "driver_calls.json")))["scenarios"][name]
    dev = torch.device("cuda:0")
    out_dir = str(tmp_path / "out")
    args = sc.namespace(name, data_tree, out_dir)
    sc.write_singletask_results(out_dir, args.ordered_cl_tasks)
    calls = driver_trace.install(upstream_driver.__file__)
    try:
        out = upstream_driver.run_upstream(args, dev, after_model_created=_pin_predictions)
        got = [dict(c) for c in calls]
    finally:
        driver_trace.uninstall()
    want = golden["calls"]
    assert [c["name"] for c in got] == [c["name"] for c in want]
    assert got == want, next((g, w) for g, w in zip(got, want) if g != w)
    run_dir = os.path.join(out_dir, golden["experiment_dir"])
    files = sorted(os.path.relpath(os.path.join(dp, f), run_dir) for dp, _, fs in os.walk(run_dir) for f in fs)
    assert files == golden["files"]
    results = json.load(open(os.path.join(run_dir, "results.json")))
    assert [r["task_key"] for r in results] == args.ordered_cl_tasks
    for r, g in zip(results, golden["results"]):
        assert r["task_key"] == g["task_key"] and 0.0 <= r["best_score"] <= 100.0
        if r["task_key"] != "vcr":
            assert r["best_score"] == pytest.approx(g["best_score"], abs=1e-4), (r, g)
    model = out["model"]
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())
    ev = out["eval_results"]
    assert set(ev["upstream_knowledge_transfer"]) == set(args.ordered_cl_tasks)
    for cur, prevs in ev["forgetting"].items():
        for prev, rec in prevs.items():
            assert 0.0 <= rec["absolute_transfer_score"] <= 100.0 and rec["forgetting"] == rec["forgetting"]       # not NaN
    n_tasks = len(args.ordered_cl_tasks)
    assert sum(len(v) for v in ev["forgetting"].values()) == n_tasks * (n_tasks - 1) // 2
    if name == "adapter":
        assert sum(p.numel() for p in model.parameters() if p.requires_grad) < 0.10 * out["total_params"]      # last task's adapter (1.8 M) + the two heads (8.3 M) of 123.5 M
        sd = torch.load(os.path.join(run_dir, "checkpoints", "task1_nlvr2", "model"))
        assert any(".adapters.nlvr2." in k for k in sd) and any(".adapters.vqa." in k for k in sd)
    if name == "freeze_encoder":
        assert all(not p.requires_grad for p in model.get_encoder().parameters())


@pytest.mark.parametrize("name", list(sc.LOWSHOT_SCENARIOS))
def test_lowshot_driver_scenario_matches_the_reference_drivers_call_sequence(name, data_tree, tmp_path, golden_dir):
    """SURVEY.md row F4, second half: REF/train/train_lowshot_multimodal.py was run unchanged through integration/climb_shim by
    oracle/record_driver_calls.py; the same scenarios on the MI355X (tests/lowshot_driver.py, LowShot*Trainer on sub-sampled training
    sets, checkpoints of an upstream run re-loaded) must make the same calls and write the same lowshot_results.json."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import copy
    from climb_amd.configs.task_configs import task_configs
    spec = sc.LOWSHOT_SCENARIOS[name]
    golden = json.load(open(os.path.join(golden_dir, "driver_calls.json")))["lowshot_scenarios"][name]
    dev = torch.device("cuda:0")
    out_dir = str(tmp_path / "out")
    saved = {k: copy.copy(task_configs[k]["low_shot_config"]) for k in sc.LOWSHOT_OVERRIDES}
    sc.apply_lowshot_overrides(task_configs)
    try:
        if spec["upstream"] is not None:           # the checkpoints the low-shot driver loads
            up = sc.namespace(spec["upstream"], data_tree, out_dir)
            sc.write_singletask_results(out_dir, up.ordered_cl_tasks)
            upstream_driver.run_upstream(up, dev, after_model_created=_pin_predictions)
        args = sc.lowshot_namespace(name, data_tree, out_dir)
        calls = driver_trace.install(lowshot_driver.__file__)
        try:
            out = lowshot_driver.run_lowshot(args, dev, after_model_created=_pin_predictions)
            got = [dict(c) for c in calls]
        finally:
            driver_trace.uninstall()
    finally:
        for k, v in saved.items():
            task_configs[k]["low_shot_config"] = v
    assert got == golden["calls"], next(((g, w) for g, w in zip(got, golden["calls"]) if g != w), (len(got), len(golden["calls"])))
    results = json.load(open(os.path.join(out_dir, golden["experiment_dir"], "lowshot_results.json")))
    assert len(results) == len(golden["results"])
    for r, g in zip(results, golden["results"]):
        assert {k: v for k, v in r.items() if k != "best_low_shot_score"} == {k: v for k, v in g.items() if k != "best_low_shot_score"}
        assert r["best_low_shot_score"] == pytest.approx(g["best_low_shot_score"], abs=1e-4), (r, g)
    assert all(bool(torch.isfinite(p).all()) for p in out["model"].parameters())
