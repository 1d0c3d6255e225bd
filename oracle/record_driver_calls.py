"""tests/golden/driver_calls.json: every call the REFERENCE's upstream driver -- and its low-shot transfer driver -- makes into this
package.  BUILD-CONTAINER ONLY.

    python oracle/record_driver_calls.py

`REF/train/train_upstream_continual_learning.py` is executed UNCHANGED (runpy, its own argparse, its own main()) with
integration/climb_shim first on sys.path, i.e. exactly INTEGRATION.md's option 1, on the synthetic data tree of tests/synth_data.py,
for the scenarios of tests/driver_scenarios.py.  There is no GPU here, so the C-ABI layer is replaced IN THIS PROCESS by a recording
stand-in (`_lib.call` becomes a no-op that only gives task heads a constant prediction and losses a finite value): everything above
the C ABI -- model construction, trainers, dataloaders, plug-ins, checkpoints, results.json, CL metrics -- is the product's real host
code, driven by the reference's real driver.  Recorded per scenario: the ordered list of {name, nargs, kwargs, returns} of calls whose
caller is a frame of the driver file, and the results.json it wrote.  tests/test_gpu_driver.py replays the scenarios on the real
engine through tests/upstream_driver.py and requires the same call sequence."""
import json
import os
import runpy
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DRIVER = "/root/reference/src/train/train_upstream_continual_learning.py"
LOWSHOT_DRIVER = "/root/reference/src/train/train_lowshot_multimodal.py"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "integration", "climb_shim"))
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")


def install_fake_c_abi():
    import torch
    from climb_amd import _lib, engine

    def fake_call(name, *a):
        if (name == "climb_gemm_f32" and isinstance(a[6], torch.Tensor)) or (name == "climb_skinny_f32" and isinstance(a[5], torch.Tensor)):
            C, N = (a[6], a[9]) if name == "climb_gemm_f32" else (a[5], a[8])      # (r04: the head products of csrc/heads.hip)
            if N == 3129:                       # VQA head: always answer 7 (the synthetic tree gives it a score of 0.6)
                C.zero_()
                C[:, 7] = 50.0
            elif N in (2, 3):                   # NLVR2 / SNLI-VE heads: always class 0
                C.zero_()
                C[:, 0] = 50.0
            else:
                C.normal_()
        elif name == "climb_bce_logits":
            a[6].fill_(1.0)
        elif name == "climb_cross_entropy":
            a[5].fill_(1.0)
    _lib.load = lambda: None
    _lib.call = fake_call
    _lib.query = lambda name: 32
    _lib.query_arg = lambda name, *a: 256
    # everything ViltEngine asks the library about itself (r02's fp16 build added these).  tests/test_dp_training.py (a CPU test) runs whole
    # driver scenarios -- engines constructed, fused steps, plug-ins -- on this stand-in, so the recorder cannot silently rot again when the
    # host code grows another query
    _lib.select_h16 = lambda name: None
    _lib.h16 = lambda: os.environ.get("CLIMB_AMD_H16") or "bf16"
    _lib.torch_h16 = lambda: torch.float16 if _lib.h16() == "fp16" else torch.bfloat16
    engine._lib = _lib

    def allocate(self):
        self.flat = torch.zeros(self.layout.total, dtype=torch.float32)
        self.grad = torch.zeros(self.layout.total, dtype=torch.float32)
        self._ws.clear()
        self._shadow = None
        self._shadow_version = -1
    engine.ViltEngine.allocate = allocate
    # the grouped weight-gradient / batched-reduction launches build device tables through a host function of the library: no-ops here too
    engine.ViltEngine._dw_flush = lambda self, ws, pending: pending.clear()
    engine.ViltEngine._red_flush = lambda self, ws, pending: pending.clear()
    engine._stream = lambda: 0
    torch.cuda.current_stream = lambda *a, **k: types.SimpleNamespace(cuda_stream=0)


def main():
    assert os.path.exists(DRIVER), "needs /root/reference"
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import driver_scenarios as sc
    import driver_trace
    import synth_data as sd
    work = tempfile.mkdtemp(prefix="climb_driver_")
    data = sd.make_climb_data_tree(os.path.join(work, "data"), n_train=sc.N_TRAIN, n_val=sc.N_VAL, seed=sc.SEED, easy_answer=sc.EASY_ANSWER)
    os.environ["CLIMB_AMD_TOKENIZER_VOCAB"] = sd.write_vocab(os.path.join(work, "vocab.txt"))
    # the fork-only module the driver imports a name from (REF/.gitmodules:1-3: absent here)
    adapters = types.ModuleType("transformers.adapters")
    adapters.AdapterConfig = type("AdapterConfig", (dict,), {})
    import transformers  # noqa: F401
    sys.modules["transformers.adapters"] = adapters
    install_fake_c_abi()
    os.chdir(work)                       # the driver puts '.' first on sys.path
    golden = {"driver": "REF/train/train_upstream_continual_learning.py", "data": dict(n_train=sc.N_TRAIN, n_val=sc.N_VAL, seed=sc.SEED, easy_answer=sc.EASY_ANSWER),
              "scenarios": {}}
    for name in sc.SCENARIOS:
        out_dir = os.path.join(work, "out_" + name)
        sc.write_singletask_results(out_dir, sc.SCENARIOS[name]["ordered_cl_tasks"])
        calls = driver_trace.install(DRIVER)
        for m in [m for m in sys.modules if m.split(".")[0] in ("modeling", "cl_algorithms", "cl_evaluation", "configs", "utils")]:
            del sys.modules[m]           # the shim modules re-bind the (now wrapped) functions on import
        sys.argv = [DRIVER] + sc.argv(name, data, out_dir)
        print("=" * 30, name, " ".join(sys.argv[1:]))
        runpy.run_path(DRIVER, run_name="__main__")
        rec = [dict(c) for c in calls]
        driver_trace.uninstall()
        ns = sc.namespace(name, data, out_dir)
        run_dirs = [d for d in os.listdir(out_dir) if "singletask" not in d]
        assert len(run_dirs) == 1, run_dirs
        results = json.load(open(os.path.join(out_dir, run_dirs[0], "results.json")))
        golden["scenarios"][name] = {"experiment_dir": run_dirs[0], "calls": rec, "results": results,
                                     "files": sorted(os.path.relpath(os.path.join(dp, f), os.path.join(out_dir, run_dirs[0]))
                                                     for dp, _, fs in os.walk(os.path.join(out_dir, run_dirs[0])) for f in fs)}
        print(name, len(rec), "calls;", [(r["task_key"], r["best_score"]) for r in results])
    # ---- the low-shot transfer driver, same shim, same recorder; "after" scenarios reuse the upstream run's checkpoints
    from climb_amd.configs.task_configs import task_configs
    sc.apply_lowshot_overrides(task_configs)
    golden["lowshot_driver"] = "REF/train/train_lowshot_multimodal.py"
    golden["lowshot_overrides"] = sc.LOWSHOT_OVERRIDES
    golden["lowshot_scenarios"] = {}
    for name, spec in sc.LOWSHOT_SCENARIOS.items():
        out_dir = os.path.join(work, "out_" + (spec["upstream"] or name))
        calls = driver_trace.install(LOWSHOT_DRIVER)
        for m in [m for m in sys.modules if m.split(".")[0] in ("modeling", "cl_algorithms", "cl_evaluation", "configs", "utils")]:
            del sys.modules[m]
        sys.argv = [LOWSHOT_DRIVER] + sc.lowshot_argv(name, data, out_dir)
        print("=" * 30, name, " ".join(sys.argv[1:]))
        runpy.run_path(LOWSHOT_DRIVER, run_name="__main__")
        rec = [dict(c) for c in calls]
        driver_trace.uninstall()
        run_dirs = [d for d in os.listdir(out_dir) if os.path.exists(os.path.join(out_dir, d, "lowshot_results.json"))]
        assert len(run_dirs) == 1, run_dirs
        results = json.load(open(os.path.join(out_dir, run_dirs[0], "lowshot_results.json")))
        golden["lowshot_scenarios"][name] = {"experiment_dir": run_dirs[0], "calls": rec, "results": results}
        print(name, len(rec), "calls;", [(r.get("lowshot_task_key", r.get("task_key")), r["best_low_shot_score"]) for r in results])
    out = os.path.join(ROOT, "tests", "golden", "driver_calls.json")
    json.dump(golden, open(out, "w"), indent=1)
    print("wrote", out, os.path.getsize(out), "bytes")
    os.chdir(ROOT)
    import shutil
    shutil.rmtree(work, ignore_errors=True)      # 13 GB of checkpoints the scenarios wrote; nothing reads them after the recording


if __name__ == "__main__":
    main()
