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


# ------------------------------------------------------------------------------------------------ data-parallel driver runs
@pytest.fixture(scope="module")
def dp_tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("climb_dp"))
    # 19 training examples at a global batch of 8 / 4 / 2 (VQA, SNLI-VE / NLVR2 / VCR): last batches of 3, 3 and 1 -- uneven shares
    # (weights 4/3, 2/3) and a rank with no example of its own (weight 0)
    synth_data.make_climb_data_tree(os.path.join(root, "data"), n_train=19, n_val=sc.N_VAL, seed=sc.SEED, easy_answer=sc.EASY_ANSWER)
    return os.path.join(root, "data"), synth_data.write_vocab(os.path.join(root, "vocab.txt"))


def _rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.fixture(scope="module")
def dp_runs(dp_tree, tmp_path_factory):
    """The four driver runs the two tests below compare -- {EWC, ER} x {one process, two ranks} -- started TOGETHER (r06, the suite's time budget: each is a
    child process tree of its own that keeps the GPU busy a few percent of the time; one after the other they were 85 s of waiting).  fp32 arithmetic, 2 epochs
    per task: the same paths in a fraction of the time.  Checkpoints are removed when the module is done."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from concurrent.futures import ThreadPoolExecutor
    from tests.test_dp_training import _scenario
    tmp = tmp_path_factory.mktemp("dp_runs")
    env = {"CLIMB_AMD_PRECISION": "fp32"}
    keys = [(name, world) for name in ("ewc", "experience_replay") for world in (1, 2)]
    with ThreadPoolExecutor(max_workers=len(keys)) as ex:
        futs = {k: ex.submit(_scenario, k[1], k[0], dp_tree, tmp, abi="hip", env_extra=env, extra=["--pin", "--epochs", "2"]) for k in keys}
        runs = {}
        for k, f in futs.items():
            try:
                runs[k] = f.result()
            except BaseException as e:          # (reported by the test that needs this run)
                runs[k] = e
    yield runs
    for dp, _, fs in os.walk(str(tmp)):
        for f in fs:
            q = os.path.join(dp, f)
            try:
                if not os.path.islink(q) and os.path.getsize(q) > (4 << 20):
                    os.remove(q)
            except OSError:
                pass


@pytest.mark.parametrize("name", ["ewc", "experience_replay"])
def test_two_rank_data_parallel_driver_run_equals_the_single_process_run(name, dp_runs, golden_dir):
    """BASELINE.json configs[3] / configs[4] AS data-parallel jobs, on the real engine: two processes (gloo on device tensors; both on the
    box's one GPU) run the four-task EWC / ER driver scenario with rank-sharded loaders, the reducer the trainers attach, all-reduced
    evaluation, the replicated + broadcast Fisher pass, sharded replay batches and rank-0 checkpoints (fp32 arithmetic so that the
    comparison with ONE process on the global batches is tight).  Required: the reference driver's call sequence on every rank, replicas
    bit-identical at the end, the single-process results.json, and -- EWC -- theta* / Fisher of every finished task equal to the
    single-process ones up to fp32 summation order (global batch k of the two-rank run IS batch k of the single-process run)."""
    from tests.test_dp_training import _files, _pinned, _same
    for world in (1, 2):
        if isinstance(dp_runs[(name, world)], BaseException):
            raise dp_runs[(name, world)]
    (out1, rep1), (out2, rep2) = dp_runs[(name, 1)], dp_runs[(name, 2)]
    golden = json.load(open(os.path.join(golden_dir, "driver_calls.json")))["scenarios"][name]
    for rep in rep2 + rep1:
        assert rep["calls"] == golden["calls"]
    for rep in rep2:
        assert rep["reducer_attached"] and rep["replicas_in_sync"] and rep["collectives"] > 0
        assert rep["results"] == rep2[0]["results"] and rep["eval_results"] == rep2[0]["eval_results"]
        assert rep["params"] == rep2[0]["params"]
        _same(_pinned(rep["results"]), _pinned(rep1[0]["results"]))
    run1, run2 = os.path.join(out1, golden["experiment_dir"]), os.path.join(out2, golden["experiment_dir"])
    assert _files(run2) == _files(run1) == golden["files"]
    if name == "ewc":
        assert rep2[0]["fisher"] == rep2[1]["fisher"] and rep2[0]["theta_star"] == rep2[1]["theta_star"]      # broadcast / in-sync: identical bits
        e1 = torch.load(os.path.join(out1, "report_0.json.ewc.pt"))
        e2 = torch.load(os.path.join(out2, "report_0.json.ewc.pt"))
        worst = 0.0
        for task in ("vqa", "nlvr2", "snli-ve"):
            dt, df = _rel(e2["theta_star"][task], e1["theta_star"][task]), _rel(e2["fisher"][task], e1["fisher"][task])
            worst = max(worst, dt, df)
            assert dt < 1e-4 and df < 2e-3, (task, dt, df)
        print(f"two-rank EWC run vs single process: worst relative L2 difference of theta* / Fisher {worst:.2e}")
    else:
        assert rep2[0]["memory_idxs"] == rep2[1]["memory_idxs"] == rep1[0]["memory_idxs"]
        # up to the end of the third task nothing is stochastic (VCR's head dropout draws per-rank masks): compare that checkpoint
        a = torch.load(os.path.join(run1, "checkpoints", "task2_snli-ve", "encoder"))
        b = torch.load(os.path.join(run2, "checkpoints", "task2_snli-ve", "encoder"))
        # ... on the weight matrices only: a replay step runs a FRESH AdamW (REF experience_replay.py:61), whose first update is
        # lr * sign(g) element-wise -- tensors that sit near zero (biases) amplify the two runs' summation-order noise in g without bound
        worst = max(_rel(b[k].float(), a[k].float()) for k in a if a[k].dim() == 2 and a[k].numel() >= 768 * 768)
        print(f"two-rank ER run vs single process: worst relative difference of a weight matrix after three tasks {worst:.2e}")
        assert worst < 2e-2
