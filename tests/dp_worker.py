"""One rank of a data-parallel run of an upstream-driver scenario (tests/driver_scenarios.py), started by the tests with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment (WORLD_SIZE=1: the single-process run the N-rank run must reproduce).

    python tests/dp_worker.py --abi {recording,hip} --scenario ewc --data <tree> --out <dir> --report <json> [--batch_size 8]
                              [--reference-driver /path/to/train_upstream_continual_learning.py]

--abi recording : CPU; the C ABI is the recording stand-in of oracle/record_driver_calls.py (host logic only: loaders, reducer on gloo,
                  plug-ins, checkpoints, metrics);
--abi hip       : the real engine on cuda:0 (every rank shares the test box's one GPU; gloo moves the device tensors).
With --reference-driver the REFERENCE's own driver file is executed unchanged through integration/climb_torchrun.py (build container
only); otherwise tests/upstream_driver.py, its call-for-call restatement.  Writes one JSON report per rank."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def checksum(t):
    import torch
    t = t.detach().double()
    return [float(t.sum()), float(t.abs().sum()), float((t * t).sum())]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--abi", choices=("recording", "hip"), required=True)
    ap.add_argument("--scenario", required=True)
    ap.add_argument("--data", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--report", required=True)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--vocab", required=True)
    ap.add_argument("--reference-driver", default=None)
    ap.add_argument("--epochs", type=int, default=0, help="override every task's num_epochs (the CPU tests: same host logic, a fraction of the time)")
    ap.add_argument("--pin", action="store_true", help="pin the classification heads' predictions like tests/test_gpu_driver.py does")
    a = ap.parse_args()
    os.environ["CLIMB_AMD_TOKENIZER_VOCAB"] = a.vocab
    import torch
    from tests import driver_scenarios as sc
    from tests import driver_trace
    if a.abi == "recording":
        import types
        adapters = types.ModuleType("transformers.adapters")
        adapters.AdapterConfig = type("AdapterConfig", (dict,), {})
        import transformers  # noqa: F401
        sys.modules["transformers.adapters"] = adapters
        from oracle.record_driver_calls import install_fake_c_abi
        install_fake_c_abi()
    from climb_amd import parallel
    if a.epochs:
        from climb_amd.configs.task_configs import task_configs
        for k in sc.FOUR:
            task_configs[k]["num_epochs"] = a.epochs
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.abi == "recording":
        torch.cuda.is_available = lambda: False
    rank, world, device = parallel.init_data_parallel("gloo")
    parallel.rank0_only_io()
    report = {"rank": rank, "world": world}
    if a.reference_driver:
        # the reference's driver itself, through the shim, exactly as integration/climb_torchrun.py runs it
        sys.path.insert(0, os.path.join(ROOT, "integration", "climb_shim"))
        calls = driver_trace.install(a.reference_driver)
        argv = sc.argv(a.scenario, a.data, a.out)
        argv[argv.index("--batch_size") + 1] = str(a.batch_size)
        sys.argv = [os.path.join(ROOT, "integration", "climb_torchrun.py"), a.reference_driver] + argv
        import runpy
        os.chdir(a.out)
        runpy.run_path(sys.argv[0], run_name="__main__")
        report["calls"] = [dict(c) for c in calls]
        driver_trace.uninstall()
        # the launcher held the IO patches for the driver call only (parallel.single_writer_io); what it logged: one writer per file
        import json as _json
        report["patches_gone"] = bool(parallel._real_torch_save is None and _json.dump.__module__ == "json")
        report["io_log"] = [list(x) for x in parallel.io_log]
        run_dirs = [d for d in os.listdir(a.out) if "singletask" not in d]
        report["results"] = json.load(open(os.path.join(a.out, run_dirs[0], "results.json")))
        json.dump(report, open(a.report, "w"))
        return
    from tests import upstream_driver
    args = sc.namespace(a.scenario, a.data, a.out)
    args.batch_size = a.batch_size
    calls = driver_trace.install(upstream_driver.__file__)
    try:
        pin = None
        if a.pin:
            from tests.test_gpu_driver import _pin_predictions as pin
        out = upstream_driver.run_upstream(args, device, after_model_created=pin)
        report["calls"] = [dict(c) for c in calls]
    finally:
        driver_trace.uninstall()
    model = out["model"]
    report["results"] = out["results"]
    report["eval_results"] = out["eval_results"]
    report["output_dir"] = out["output_dir"]
    red = model._host.ddp
    report["reducer_attached"] = red is not None
    report["replicas_in_sync"] = bool(red.replicas_in_sync()) if red is not None else True
    if red is not None:
        report["collectives"], report["bytes_reduced"] = red.collectives, red.bytes_reduced
    eng = model._host.engine()
    report["params"] = checksum(eng.flat)
    if a.abi == "hip":
        torch.cuda.synchronize()
    ewc, mem = out.get("ewc"), out.get("replay_memory")
    report["io_log"] = [list(x) for x in parallel.io_log]
    parallel.restore_io()              # the report files below are written by rank 0 alone, outside the collective save protocol
    if ewc is not None:
        report["fisher"] = {k: checksum(v) for k, v in ewc.fisher_flat.items()}
        report["theta_star"] = {k: checksum(v) for k, v in ewc.param_flat.items()}
        if rank == 0 and a.abi == "hip":
            torch.save({"fisher": {k: v.cpu() for k, v in ewc.fisher_flat.items()}, "theta_star": {k: v.cpu() for k, v in ewc.param_flat.items()}},
                       a.report + ".ewc.pt")
    if mem is not None:
        report["memory_idxs"] = {k: list(b.memory_idxs) for k, b in mem.memory_buffers.items()}
    if rank == 0 and a.abi == "hip":
        torch.save(eng.flat.cpu(), a.report + ".params.pt")
    json.dump(report, open(a.report, "w"))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
