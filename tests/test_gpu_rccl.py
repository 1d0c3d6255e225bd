"""Data parallelism on real collectives: the scenarios a multi-GPU lease has to answer on its first
call instead of its first debugging session.  Every test is parametrised on the backend: "gloo" runs on any GPU box (two processes on the one
GPU: the product's host logic with collectives through the host), "nccl" (= RCCL over xGMI, one process per GPU as climb_amd/parallel.py::
init_data_parallel binds them) runs when the box has >= 2 GPUs and is SKIPPED otherwise.  The two-rank equality with the single-process global
batch and the deferred un-cast path are the backend-parametrised tests of tests/test_gpu_parity.py (`test_data_parallel_two_ranks_...`,
`test_data_parallel_optimizer_reads_the_averaged_payload_in_place`); here: EWC's replicated Fisher pass + broadcast, the CU reserve taken and
given back around collectives that run under the backward, and `bench.py --gpus 2`."""
import json
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

from oracle import vilt_oracle as vo
from tests.test_gpu_parity import BACKENDS, H16, _close, _dev, _meta, _summary, collect_ranks, dp_join, enc_to_inputs, make_model, need_backend

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spawn(target, backend, *args):
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    return collect_ranks([ctx.Process(target=target, args=(r, 2, port, q, backend) + args) for r in range(2)], q)


# ------------------------------------------------------------------------------------------------ EWC: replicated Fisher pass, rank 0's F broadcast
def _fisher_worker(rank, world, port, q, backend, golden_dir, sharded=False):
    dist = dp_join(rank, world, port, backend)
    try:
        from climb_amd.cl_algorithms import EWC
        from climb_amd.configs.model_configs import model_configs
        from climb_amd.configs.task_configs import task_configs
        from climb_amd.parallel import GradientAllReducer
        from climb_amd.train import VQATrainer
        z = np.load(os.path.join(golden_dir, "fisher_3x2.npz"))
        m = _meta(z)
        B, nb = int(m["B"]), int(m["batches"])
        model, P = make_model(m["tasks"].split(","), int(m["wseed"]) + 5 * rank)          # rank 1 starts from OTHER weights: the reducer's broadcast fixes that
        ddp = GradientAllReducer(model)

        class Loader(list):
            collate_fn = None
        loader = Loader()
        for i in range(nb):
            e = vo.synthetic_encodings(B, seed=200 + i)
            images, texts = enc_to_inputs(e)
            tgt = vo.synthetic_vqa_targets(B, seed=200 + i)
            if sharded:          # this rank's strided share of every global batch (equal shares: weight 1), as climb_amd/data/sharding.py deals them
                images, texts = enc_to_inputs({k: v[rank::world] for k, v in e.items()})
                tgt = tgt[rank::world]
            loader.append({"raw_texts": [""] * (B // world if sharded else B), "encodings": texts, "images": images, "target_scores": tgt})
        loader.dataset = list(range(int(nb * B / 0.01)))
        trainer = VQATrainer(types.SimpleNamespace(cl_algorithm="ewc"), task_configs, model_configs["vilt"], _dev(), train_dataloader=loader, val_dataloader=loader)
        ewc = EWC(types.SimpleNamespace(ewc_fisher_sample_percentage=0.01, ewc_loss_weight=100.0, ewc_fisher_sharded=sharded))
        before = (ddp.bytes_reduced, ddp.collectives)
        ewc.save_task_parameters(task_key="vqa", model=model, task_trainer=trainer, device=_dev())
        torch.cuda.synchronize()
        quiet = (ddp.bytes_reduced, ddp.collectives) == before          # nothing on the wire during the pass (the step below does reduce)
        names = [str(n) for n in z["names"]]
        fisher = {k: v.detach().cpu() for k, v in ewc.fisher_dict["vqa"].items()}
        norms, heads = _summary(fisher, names)
        F = ewc.fisher_flat["vqa"].double()
        # one EWC training step afterwards: the penalty gradient is added AFTER the average, the replicas must stay identical
        opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        opt.zero_grad()
        model.train()
        e = vo.synthetic_encodings(2 * B, seed=31)
        images, texts = enc_to_inputs({k: v[rank::world] for k, v in e.items()})
        loss, *_ = model.fused_forward_backward("vqa", images, texts, vo.synthetic_vqa_targets(2 * B, seed=31)[rank::world], ewc=ewc, optimizer=opt)
        opt.step()
        q.put((rank, ddp.enabled, quiet, norms, heads, [float(F.sum()), float((F * F).sum())],
               ddp.replicas_in_sync(), float(loss)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", BACKENDS)
def test_fisher_pass_is_replicated_and_broadcast(golden_dir, backend):
    """REF/cl_algorithms/ewc.py:56-68 squares gradients that accumulate across batches in order, so the pass cannot be sharded: every rank runs it on
    the whole batches with the reducer suspended (nothing on the wire), rank 0's F is broadcast, and it is the reference's F (fisher_3x2.npz)."""
    need_backend(backend)
    res = _spawn(_fisher_worker, backend, golden_dir)
    z = np.load(os.path.join(golden_dir, "fisher_3x2.npz"))
    for r in res:
        assert r[1], "the reducer stayed suspended after the Fisher pass"
        assert r[2], "the Fisher pass put gradients on the wire"
        _close(r[3], z["fisher_norms"], 2e-3, f"fisher norms, rank {r[0]}")
        _close(r[4], z["fisher_heads"], 2e-3, f"fisher heads, rank {r[0]}")
        assert r[6], "replicas diverged in the EWC step after the Fisher pass"
    assert res[0][5] == res[1][5], "F differs between the ranks after the broadcast"


def test_fisher_pass_sharded_over_the_ranks_is_the_same_estimate(golden_dir):
    """r06 (SURVEY.md 8(e) "Fisher under DP"): the opt-in sharded pass -- each rank runs its share of every Fisher batch, one fp32
    all-reduce per batch rebuilds the batch gradient, every rank accumulates and squares the running sum itself -- gives the reference's estimate
    (fisher_3x2.npz, same tolerance as the replicated pass) on every rank, and the replicas stay identical through the EWC step that follows."""
    need_backend("gloo")
    res = _spawn(_fisher_worker, "gloo", golden_dir, True)
    z = np.load(os.path.join(golden_dir, "fisher_3x2.npz"))
    for r in res:
        assert r[1], "the reducer stayed suspended after the Fisher pass"
        _close(r[3], z["fisher_norms"], 2e-3, f"fisher norms, rank {r[0]}")
        _close(r[4], z["fisher_heads"], 2e-3, f"fisher heads, rank {r[0]}")
        assert r[6], "replicas diverged in the EWC step after the sharded Fisher pass"
    assert res[0][5] == res[1][5], "F differs between the ranks after the broadcast"


# ------------------------------------------------------------------------------------------------ CU reserve around overlapped collectives
def _reserve_worker(rank, world, port, q, backend, precision):
    dist = dp_join(rank, world, port, backend)
    try:
        from climb_amd import _lib
        from climb_amd.parallel import GradientAllReducer
        model, _ = make_model(["vqa"], 42 + rank, precision=precision)
        ddp = GradientAllReducer(model, overlap=True)
        ddp.reserve_cus = 32
        eng = model._host.engine()
        grid0 = int(_lib.query_arg("climb_get_option", 9)) if precision != "fp32" else 0
        seen = []
        hook = eng.grad_ready_hook

        def spy(lo, hi):
            hook(lo, hi)
            seen.append((eng._cu_reserve, int(_lib.query_arg("climb_get_option", 9)) if precision != "fp32" else 0))
        eng.grad_ready_hook = spy
        opt = model.create_optimizer({"lr": 1e-3, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        opt.zero_grad()
        model.train()
        enc = vo.synthetic_encodings(4, seed=21)
        images, texts = enc_to_inputs({k: v[rank::world] for k, v in enc.items()})
        tgt = vo.synthetic_vqa_targets(4, seed=21)[rank::world]
        for _ in range(2):
            model.fused_forward_backward("vqa", images, texts, tgt, optimizer=opt)
            after = (eng._cu_reserve, int(_lib.query_arg("climb_get_option", 9)) if precision != "fp32" else 0)
            opt.step()
            opt.zero_grad()
        q.put((rank, grid0, seen, after, ddp.replicas_in_sync(), ddp.collectives))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", BACKENDS)
def test_cu_reserve_is_taken_under_the_backward_and_given_back(backend):
    """GradientAllReducer.reserve_cus: from the first collective launched under the backward until finish() the persistent GEMM grids leave 32 CUs to
    the ring kernels; afterwards the library-wide grid is exactly what it was (engine.set_cu_reserve), on every rank, and the replicas agree."""
    need_backend(backend)
    res = _spawn(_reserve_worker, backend, H16)
    for rank, grid0, seen, after, sync, ncoll in res:
        assert ncoll > 0 and sync
        assert any(rs == 32 for rs, _ in seen), f"rank {rank}: no reserve was taken while collectives were in flight: {seen}"
        if H16 != "fp32":
            full = grid0 if grid0 > 0 else torch.cuda.get_device_properties(0).multi_processor_count
            assert all(g == max(8, (full - 32) // 8 * 8) for rs, g in seen if rs == 32), (grid0, seen)
        assert after == (0, grid0), f"rank {rank}: reserve / grid not restored after finish(): {after} vs grid {grid0}"


# ------------------------------------------------------------------------------------------------ bench.py --gpus 2 on ONE GPU (gloo): the N > 1 code of the bench itself
def test_bench_two_ranks_on_one_gpu_through_gloo():
    """Everything bench.py does only under world > 1 -- the launcher it becomes, rank-0-only stdout, the reducer, the warm-up trial that settles overlap /
    CU reserve / deferred collectives with an all-reduced decision, the timed region's barriers and max-over-ranks time, the A/B leg, the data-parallel
    fields of the JSON line -- with two ranks sharing the test box's GPU and collectives through the host (CLIMB_AMD_DP_BACKEND=gloo).  Not a
    performance number (two replicas time-share one GPU); a small batch keeps it short."""
    env = dict(os.environ, CLIMB_AMD_DP_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "8", "--no-cpu-baseline",
                        "--precision", H16],          # (the 16-bit build this suite runs on: CLIMB_AMD_H16 pins one per process)
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line on stdout (rank 0's)"
    j = json.loads(lines[-1])
    assert j["n_gpus"] == 2 and j["replicas_in_sync"] is True and j["config"]["global_batch"] == 16 and j["scaling"] == "weak"
    assert j["allreduce_MB_per_step"] > 100 and j["value"] > 0 and float(j["config"]["final_loss"]) == float(j["config"]["final_loss"]) and j["dp_payload"] in ("bf16", "none")
    assert set(j["dp_overlap_warmup_trial"]) >= {"overlap_ms", "deferred_ms", "chosen"} and "dp_overlap_ab" in j
    # r06: the fields that make a multi-GPU lease self-validating
    assert j["value_per_gpu"] == pytest.approx(j["value"] / 2, rel=1e-3)
    assert j["process_group"] == {"backend": "gloo", "world_size": 2, "reducer_world": 2}
    assert sorted(d["rank"] for d in j["rank_devices"]) == [0, 1] and all(d["uuid"] or d["pci"] for d in j["rank_devices"])
    assert j["distinct_devices"] == 1          # both ranks of THIS test share the box's GPU; under RCCL the launch binds one GPU per rank: N


# ------------------------------------------------------------------------------------------------ bench.py --gpus 2 (RCCL only: its ranks bind one GPU each)
def test_bench_two_gpus_smoke():
    """The driver's scaling command at N = 2, shortened: runs only where two GPUs exist (skipped on the 1-GPU test box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["replicas_in_sync"] is True and j["config"]["global_batch"] == 128
    assert j["allreduce_MB_per_step"] > 100 and j["value"] > 0
    assert j["process_group"]["backend"] == "nccl" and j["process_group"]["world_size"] == 2 and j["distinct_devices"] == 2
