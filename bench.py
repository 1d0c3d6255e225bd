#!/usr/bin/env python
"""bench.py -- ViLT VQAv2 fine-tune step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5                      # BASELINE configs[1]: bf16, bs=64/GPU, 384x384, 40 tokens
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = forward + BCE loss + backward + (N>1: gradient all-reduce over RCCL) + fused AdamW with the reference's
warm-up/decay schedule, on one synthetic batch already resident in HBM (SURVEY.md §8(d) input spec).  Rank 0 prints ONE
JSON line; `value` is whole-job image-text pairs/s.  `roofline` is for the dominant kernel (the GEMM), timed live with HIP
events on the launch stream; `cpu_baseline` is the CPU oracle (a port of the reference path) on this node's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 100.14e9          # forward 33.38 + backward 66.76 GFLOP at S=185, no padding counted
# of which the opt-in CLIMB_AMD_CLS_ONLY_LAST=1 step does not execute (ViltEngine.cls_only_last: out-projection + MLP of the last layer on 1 of S rows, fwd + bwd):
FLOP_NOT_RUN_CLS_ONLY = 3 * 184 * 2 * (768 * 768 + 2 * 768 * 3072)          # 5.86 GFLOP; the whole-step rate below counts EXECUTED flops
PEAK = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3,   # dense MFMA peak TFLOP/s for the operand dtype (MI355X_MICROARCH.md)
        "bf16x3": 2500.0 / 3}                             # split operands: three bf16 MFMA products per algorithmic multiply-add


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="sequences per GPU per step")
    ap.add_argument("--precision", default=os.environ.get("CLIMB_AMD_PRECISION", "bf16"), choices=["bf16", "fp16", "fp32", "bf16x3"],
                    help="bf16 (BASELINE configs[1], default); fp16 = the same code path on IEEE-half operands (DESIGN.md section 3); fp32 = parity mode; "
                         "bf16x3 = split (hi, lo) bf16 operands, three MFMA products per k-step: the fast mode inside the 1e-3 bar")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--real-input-only", action="store_true", help="only the `real_input` leg (the trainer loop on JPEG files), printed as JSON")
    ap.add_argument("--child-check", action="store_true", help=argparse.SUPPRESS)       # fp16_operand_line()'s child: run the reference checker leg only
    ap.add_argument("--graph", action="store_true", help="replay the step as a captured hipGraph (measured equal to eager launches at bs=64: the GPU, not the host, is the bottleneck)")
    ap.add_argument("--no-cls-only-leg", action="store_true", help="skip the extra K steps with CLIMB_AMD_CLS_ONLY_LAST=1 (profiling runs: the trace then holds the default step only)")
    ap.add_argument("--cpu-steps", type=int, default=10, help="timed B=2 CPU steps (BASELINE.md section 4: 10)")
    ap.add_argument("--spawn-check", action="store_true", help="N-rank launch plumbing only (no GPU needed): every rank joins a gloo group, one all-reduce, "
                                                                "rank 0 prints a JSON line -- what tests/test_host_logic.py runs for `--gpus 2`")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        # started as plain `python bench.py --gpus N`: become the launcher -- N ranks of this same command line under torch.distributed.run on
        # 127.0.0.1 and a free port; rank 0's JSON line is the last line of OUR stdout, the exit code is the job's
        sys.exit(self_launch(args.gpus))
    if args.spawn_check:
        return spawn_check(world, rank)
    if rank != 0:
        os.dup2(2, 1)       # only rank 0 owns stdout: whatever another rank's libraries print (RCCL's banner sits in a C stdio buffer until exit) goes to stderr
    # one process per GPU.  (CLIMB_AMD_DP_BACKEND=gloo is the test hook of tests/test_gpu_rccl.py: N ranks on whatever GPUs exist -- on the one-GPU test box
    # all of them on cuda:0 -- with collectives through the host, so that everything this file does only under world > 1 runs before a multi-GPU lease does)
    backend = os.environ.get("CLIMB_AMD_DP_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if args.real_input_only:
        print(json.dumps({"real_input": real_input_line(dev, args)}), flush=True)
        return
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    from climb_amd.train import polynomial_decay_schedule_with_warmup

    B, T = args.batch, 40
    tasks = ["vqa"]
    model = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=tasks, model_config=model_configs["vilt"],
                                                 task_configs=task_configs, device=dev, precision=args.precision)
    model.train()
    ddp = None
    if world > 1 or os.environ.get("CLIMB_AMD_FORCE_DDP") == "1":
        if not dist.is_initialized():       # single-rank smoke of the RCCL path
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29512")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from climb_amd.parallel import GradientAllReducer
        ddp = GradientAllReducer(model)       # broadcasts rank 0's weights, hooks bucketed all-reduce into the backward
        ddp.world = max(ddp.world, 2) if os.environ.get("CLIMB_AMD_FORCE_DDP") == "1" else ddp.world
    # synthetic batch, resident in HBM before the timed region (per-rank shard: different seed per rank)
    g = torch.Generator().manual_seed(1 + rank)
    texts = dict(input_ids=torch.randint(0, 30522, (B, T), generator=g).to(dev), token_type_ids=torch.zeros(B, T, dtype=torch.long, device=dev),
                 attention_mask=torch.ones(B, T, dtype=torch.long, device=dev))
    pixels = torch.randn(B, 3, 384, 384, generator=g).to(dev)
    target = torch.zeros(B, 3129)
    target[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
    target = target.to(dev)

    total_steps = args.steps + args.warmup
    opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
    sched = polynomial_decay_schedule_with_warmup(opt, max(1, int(0.1 * total_steps)), total_steps, 0.0, 1.0)
    opt.zero_grad()

    use_graph = args.graph and world == 1 and os.environ.get("CLIMB_AMD_FORCE_DDP") != "1"
    fwd_bwd = model.graphed_forward_backward if use_graph else model.fused_forward_backward

    def step(eager=False):
        if eager or not use_graph:       # the training step as the trainers run it (climb_amd/train/task_trainer.py::train_step): the optimizer is named, step() follows
            loss, _, _, _ = model.fused_forward_backward("vqa", pixels, texts, target, optimizer=opt)
        else:
            loss, _, _, _ = fwd_bwd("vqa", pixels, texts, target)
        opt.step()
        sched.step()
        opt.zero_grad()
        return loss

    for _ in range(args.warmup):
        loss = step()
    # Data parallel: whether the collectives should run UNDER the backward is a property of the fabric and of what shares the CUs with the
    # persistent GEMMs, so it is settled here, before the timed region: three untimed steps each way, the slowest rank's
    # time decides, every rank takes the same decision (CLIMB_AMD_DP_OVERLAP pins it instead).  The timed K steps then run one setting.
    dp_tuned = None
    if ddp is not None and (world > 1 or os.environ.get("CLIMB_AMD_FORCE_DDP") == "1") and os.environ.get("CLIMB_AMD_DP_OVERLAP") is None:
        trial = {}
        # overlap, overlap with 32 / 64 CUs left to RCCL (the persistent GEMM grids shrink while collectives are in flight: parallel.py), deferred
        settings = {"overlap": (True, 0), "overlap_reserve32": (True, 32), "overlap_reserve64": (True, 64), "deferred": (False, 0)}
        if os.environ.get("CLIMB_AMD_DP_RESERVE_CUS") is not None:       # pinned reserve: only overlap against deferred, as before
            settings = {"overlap": (True, ddp.reserve_cus), "deferred": (False, 0)}
        for setting, (ov, rs) in settings.items():
            ddp.overlap, ddp.reserve_cus = ov, rs
            step()
            dist.barrier()
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            for _ in range(3):
                step()
            dist.barrier()
            torch.cuda.synchronize()
            tt = torch.tensor([time.perf_counter() - t_], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            trial[setting] = float(tt.item()) / 3
        best = min(trial, key=trial.get)          # every rank holds the same (all-reduced) times: the same decision everywhere
        ddp.overlap, ddp.reserve_cus = settings[best]
        dp_tuned = dict({k + "_ms": round(v * 1e3, 3) for k, v in trial.items()}, chosen=best)
    eng = model._host.engine()
    dominant = "gemm_bf16_nt" if args.precision in ("bf16", "fp16") else ("gemm_split_nt" if args.precision == "bf16x3" else "gemm_f32")
    prof = {"kernel": dominant, "events": []}
    prof_every = 10         # HIP-event pairs around every launch of the dominant kernel on every 10th timed step (an instrumented step costs
    #                         +0.6 ms: 97 launches x 2 event records), created and recorded once BEFORE the timed region (event creation inside it
    #                         showed up as a 4 - 38 ms outlier step)
    n_prof_steps = (args.steps + prof_every - 1) // prof_every
    prof["pool"] = [torch.cuda.Event(enable_timing=True) for _ in range(2 * 128 * n_prof_steps)]
    for ev in prof["pool"]:
        ev.record()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # step boundaries on the compute stream (median step time)
    fence()
    if ddp is not None:
        ddp.bytes_reduced = 0             # the payload of the TIMED steps only (warm-up, the overlap trial and the A/B leg are not in the divisor either)
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        sampled = i % prof_every == 0          # event-instrumented steps launch eagerly (graph nodes cannot carry per-kernel events)
        eng.prof = prof if sampled else None
        loss = step(eager=sampled)
        marks[i + 1].record()
    eng.prof = None
    fence()
    dt = time.perf_counter() - t0
    bytes_timed = ddp.bytes_reduced if ddp is not None else 0
    step_ms_seq = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if os.environ.get("CLIMB_AMD_BENCH_STEPS"):
        print("per-step ms:", " ".join(f"{v:.2f}" for v in step_ms_seq), file=sys.stderr)
    step_ms = sorted(step_ms_seq)
    # data parallel: the same K steps again with the collectives deferred to after the backward (or overlapped, if the default was
    # deferred), so that one multi-GPU run answers whether overlap pays on this fabric
    dp_ab = None
    if ddp is not None and world > 1:
        first, first_rs = ddp.overlap, ddp.reserve_cus
        ddp.overlap, ddp.reserve_cus = not first, 0
        for _ in range(2):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        other = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(other, op=dist.ReduceOp.MAX)
        ddp.overlap, ddp.reserve_cus = first, first_rs
        dp_ab = {("overlap" if not first else "deferred") + "_ms_per_step": round(float(other.item()) / args.steps * 1e3, 3)}
    final_loss = float(loss.item())
    # the same K steps with the opt-in dead-row elimination of the last encoder layer (ViltEngine.cls_only_last, DESIGN.md section 5: the
    # rows of x_L that nothing reads are not computed; same loss and gradients) -- reported NEXT TO the headline, which executes every row
    cls_only = None
    if world == 1 and ddp is None and not eng.cls_only_last and not args.no_cls_only_leg:
        eng.cls_only_last = True
        for _ in range(3):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_cls = time.perf_counter() - t1
        eng.cls_only_last = False
        cls_only = {"ms_per_step": round(dt_cls / args.steps * 1e3, 3), "value": round(B * args.steps / dt_cls, 2), "unit": "samples/s",
                    "flops_not_executed_frac": round(FLOP_NOT_RUN_CLS_ONLY / FLOP_PER_SAMPLE, 4)}
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    events = prof["events"]
    in_sync = ddp.replicas_in_sync() if ddp is not None else True
    # r06: what makes a first multi-GPU lease self-validating -- the world size as the PROCESS GROUP reports it (not the launcher's
    # environment), and which physical device every rank computed on (one UUID / PCI address per rank: N distinct ones under RCCL, the same one N times in
    # the gloo test that time-shares a GPU)
    pg_info = rank_devices = None
    if dist.is_initialized():
        import socket
        pr = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "cuda_device": int(torch.cuda.current_device()), "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
                "pci": "%04x:%02x:%02x" % (int(getattr(pr, "pci_domain_id", 0)), int(getattr(pr, "pci_bus_id", 0)), int(getattr(pr, "pci_device_id", 0))),
                "host": socket.gethostname()}
        rank_devices = [None] * dist.get_world_size()
        dist.all_gather_object(rank_devices, mine)
        pg_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "reducer_world": int(ddp.world) if ddp is not None else 1}

    if rank == 0:
        ws = eng.workspace(B, T)
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        # dominant kernel: average launch duration and algorithmic rate (padding rows S..S_pad not counted)
        k_ms = [ev[0].elapsed_time(ev[1]) for ev in events]
        k_flops = sum(ev[2] for ev in events) * (ws.S / ws.S_pad)
        # the same launches by kind (output width N, reduction depth K, epilogue 0 none / 1 GELU / 2 fp32 residual / 3 x GELU', output type): which
        # member of the family is furthest below the roof; fractions are of the dense spec peak, like `frac`
        kinds = {}
        for ev, t_ms in zip(events, k_ms):
            kd = kinds.setdefault(ev[3], [0, 0.0, 0.0])
            kd[0] += 1
            kd[1] += t_ms
            kd[2] += ev[2] * (ws.S / ws.S_pad)
        per_kind = {k: {"launches_per_step": v[0] // max(1, n_prof_steps), "avg_us": round(1e3 * v[1] / v[0], 2), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1),
                        "frac": round(v[2] / (v[1] * 1e-3) / 1e12 / PEAK[args.precision], 4)} for k, v in sorted(kinds.items(), key=lambda kv: -kv[1][1])}
        k_time = sum(k_ms) * 1e-3
        achieved = k_flops / k_time / 1e12 if k_time > 0 else 0.0
        traffic = hbm_step = None
        try:        # HBM bytes per launch / per step from the committed PMC passes (rocprofv3 cannot be driven from inside the timed run);
            # the file names the hash of the kernel sources it was measured on: a stale figure is reported as null, not repeated
            # (tests/test_host_logic.py::test_committed_traffic_figures_belong_to_this_tree keeps a late edit from doing that unnoticed)
            tj = latest_traffic(args.precision)
            if tj.get("csrc_sha16") == csrc_hash() and args.precision in ("bf16", "fp16", "bf16x3") and B == 64:
                traffic = tj["gemm_bf16_nt"]["hbm_bytes_per_launch"]
                if tj.get("step_sha16", step_hash()) == step_hash():
                    hbm_step = tj.get("hbm_bytes_per_step")
        except Exception:
            traffic = hbm_step = None
        flop_run = FLOP_PER_SAMPLE - (FLOP_NOT_RUN_CLS_ONLY if eng.cls_only_last else 0)
        # what the matrix cores SUSTAIN on this box with N(0,1) operands and no memory traffic at all (the clock follows the power budget; DESIGN.md
        # section 8): measured here, after the timed region, by the library's diagnostic kernel -- reported next to the spec peak, which `frac` uses
        sustained = None
        if args.precision in ("bf16", "fp16"):
            try:
                from climb_amd import _lib
                nwg, it_ = 256, 20000
                src = torch.randn(nwg * 256 * 64, device=dev).to(eng.t16)
                outp = torch.empty(nwg * 256, device=dev)
                st_ = torch.cuda.current_stream().cuda_stream
                _lib.call("climb_mfma_sustained_probe", src, outp, nwg, 2000, st_)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                _lib.call("climb_mfma_sustained_probe", src, outp, nwg, it_, st_)
                e1.record()
                torch.cuda.synchronize()
                sustained = round(nwg * 4 * it_ * 16 * 32768.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
            except Exception as ex:          # an older library without the diagnostic: the field stays null
                print(f"sustained-MFMA probe skipped: {ex}", file=sys.stderr)
        roof = {"bound": "mfma", "kernel": dominant, "achieved": round(achieved, 2), "peak": PEAK[args.precision], "unit": "TFLOP/s",
                "frac": round(achieved / PEAK[args.precision], 4), "traffic": traffic,
                # the whole step against the OTHER roof: HBM-side bytes of one step (every kernel, same PMC passes) over this run's step time, as a
                # fraction of the 6.29 TB/s a streaming kernel achieves on this chip (MI355X_MICROARCH.md) -- the step is nearly as close to this roof
                "hbm_bytes_per_step": hbm_step,
                "hbm_frac_of_6.29TBps": round(hbm_step / (dt / args.steps) / 6.29e12, 4) if hbm_step else None,
                "per_kind": per_kind,
                "launches_per_step": len(events) // max(1, n_prof_steps), "event_sampled_steps": n_prof_steps, "avg_launch_us": round(1e3 * sum(k_ms) / max(1, len(k_ms)), 2),
                "kernel_time_frac_of_step": round(k_time / n_prof_steps / (dt / args.steps), 4),
                "sustained_mfma_tflops_random_operands": sustained,
                "frac_of_sustained": round(achieved / sustained, 4) if sustained else None,
                "whole_step_tflops": round(flop_run * B * args.steps / dt / 1e12, 2),
                "whole_step_frac": round(flop_run * B * args.steps / dt / 1e12 / PEAK[args.precision], 4)}
        out = {"metric": "image-text pairs/sec on ViLT VQAv2 fine-tune step", "value": round(value, 2), "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "median_ms_per_step": round(step_ms[len(step_ms) // 2], 3),
               "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.precision, "data": "synthetic", "hip_graph": use_graph,
               "config": {"workload": "BASELINE.json configs[1]: ViLT sequential-FT VQAv2 step (fwd+BCE+bwd+AdamW), 384x384 image + 40 tokens, "
                                      "12-layer ViLT-B/32 random-init + VQA head", "batch_per_gpu": B, "global_batch": B * world, "seq_len": ws.S,
                          "seq_len_padded": ws.S_pad, "parallelism": f"dp{world}", "final_loss": round(final_loss, 3),
                          "last_layer": "[CLS] rows only after its attention (CLIMB_AMD_CLS_ONLY_LAST=1)" if eng.cls_only_last else "every row"},
               "roofline": roof}
        if cls_only is not None:
            out["cls_only_last_layer"] = cls_only
        out["value_per_gpu"] = round(value / world, 2)          # BASELINE.json's metric is quoted per GPU; `value` is the whole job
        if pg_info is not None:
            out["process_group"] = pg_info
            out["rank_devices"] = rank_devices
            out["distinct_devices"] = len({(d["host"], d["uuid"] or d["pci"], d["cuda_device"]) for d in rank_devices})
        if ddp is not None:
            out["replicas_in_sync"] = in_sync
            out["dp_overlap"] = bool(ddp.overlap)
            out["dp_reserved_cus"] = int(ddp.reserve_cus) if ddp.overlap else 0
            if dp_tuned:
                out["dp_overlap_warmup_trial"] = dp_tuned
            out["dp_payload"] = ddp.compress
            out["allreduce_MB_per_step"] = round(bytes_timed / 1e6 / args.steps, 1)
            if dp_ab:
                out["dp_overlap_ab"] = dict(dp_ab, **{("overlap" if ddp.overlap else "deferred") + "_ms_per_step": round(ms, 3)})
        if world == 1 and args.child_check:
            out["fp16_vs_ref"] = bf16_vs_reference(dev, args.precision)
        if world == 1 and not args.no_cpu_baseline:
            out["bf16_vs_ref" if args.precision != "fp16" else "fp16_vs_ref"] = bf16_vs_reference(dev, args.precision) if args.precision != "fp32" else None
            if args.precision == "bf16":
                out["parity_fast_mode"] = mode_line(dev, args, "bf16x3")
            if args.precision == "bf16":
                out["fp16_operands"] = fp16_operand_line(args)
            if args.precision == "bf16":
                out["real_input"] = real_input_line(dev, args)
            if args.precision == "bf16":
                out["fp32_parity_mode"] = mode_line(dev, args, "fp32")
                out["reference_stack_on_this_gpu"] = torch_stack_line(dev, args)
                if isinstance(out["reference_stack_on_this_gpu"].get("fp32"), dict):
                    rs = out["reference_stack_on_this_gpu"]
                    rs["headline_over_fp32"] = round(value / rs["fp32"]["value"], 2)
                    rs["headline_over_bf16_autocast"] = round(value / rs["bf16_autocast"]["value"], 2)
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        # the JSON line must be the LAST line on stdout: RCCL prints its version banner through C stdio, which is block-buffered on a pipe and would
        # otherwise come out at process exit, after this line (seen with the forced single-rank RCCL run).  Flush it now, print, then hand fd 1 to stderr.
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def fp16_operand_line(args):
    """Reported next to the bf16 headline, never as `value`: the SAME step on the IEEE-half build of the library (fp16 GEMM / attention operands,
    scaled loss gradient) -- its throughput and its errors against the reference fixture.  A process holds one build, so
    this is a child process of rank 0."""
    import subprocess
    env = dict(os.environ, CLIMB_AMD_H16="fp16")
    env.pop("CLIMB_AMD_LIB", None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--precision", "fp16", "--no-cpu-baseline", "--child-check", "--steps", str(args.steps),
                            "--warmup", str(args.warmup), "--batch", str(args.batch)], capture_output=True, text=True, timeout=900, env=env)
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        return {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "dtype": "fp16", "vs_ref": j.get("fp16_vs_ref"),
                "note": "same kernels, IEEE-half operands (libclimb_hip_f16.so), static power-of-two loss scale; not the BASELINE dtype"}
    except Exception as e:      # the headline must not depend on the extra line
        return {"error": repr(e)[:200]}


NT_SOURCES = ("common.h", "gemm_bf16.hip", "gemm_bf16_nt.h", "gemm_bf16_nt2p.hip", "gemm_bf16_nt4.hip", "gemm_bf16_ntp.hip", "gemm_bf16_phase.h")


MODE_NOTES = {
    "fp32": "the parity mode (exact-fp32 MFMA, fp32 activations): the arithmetic that meets 1e-3 / argmax-exact against the reference; not the BASELINE dtype",
    "bf16x3": "split operands (r06): the fp32 mode's data flow with every encoder GEMM on (hi, lo) bf16 plane pairs, three MFMA products per k-step, fp32 "
              "accumulate -- the FAST mode inside north_star's 1e-3 / argmax-exact bar; not the BASELINE dtype"}


def mode_line(dev, args, precision):
    """Reported next to the bf16 headline, never as `value` (VERDICT r4 next #3 iii, r5 next #1): the SAME step in one of the two modes whose outputs
    meet north_star's 1e-3 / argmax-exact bar against the reference -- "fp32": exact-fp32 MFMA (`v_mfma_f32_32x32x2_f32`), fp32 activations;
    "bf16x3": split (hi, lo) bf16 operands, three MFMA products per k-step -- timed at the benchmark's batch, with that mode's errors against
    the reference's own B = 64 outputs."""
    import torch
    try:
        from climb_amd.modeling import create_continual_learner_map
        from climb_amd.configs.task_configs import task_configs
        from climb_amd.configs.model_configs import model_configs
        from climb_amd.train import polynomial_decay_schedule_with_warmup
        B, T, steps, warm = args.batch, 40, (5 if precision == "fp32" else 10), 2
        model = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=["vqa"], model_config=model_configs["vilt"],
                                                     task_configs=task_configs, device=dev, precision=precision)
        model.train()
        g = torch.Generator().manual_seed(1)
        texts = dict(input_ids=torch.randint(0, 30522, (B, T), generator=g).to(dev), token_type_ids=torch.zeros(B, T, dtype=torch.long, device=dev),
                     attention_mask=torch.ones(B, T, dtype=torch.long, device=dev))
        pixels = torch.randn(B, 3, 384, 384, generator=g).to(dev)
        target = torch.zeros(B, 3129)
        target[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
        target = target.to(dev)
        opt = model.create_optimizer({"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8})
        sched = polynomial_decay_schedule_with_warmup(opt, 1, steps + warm, 0.0, 1.0)
        opt.zero_grad()

        def step():
            model.fused_forward_backward("vqa", pixels, texts, target, optimizer=opt)
            opt.step()
            sched.step()
            opt.zero_grad()
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        del model, opt
        torch.cuda.empty_cache()
        return {"value": round(B / dt, 1), "unit": "samples/s", "ms_per_step": round(dt * 1e3, 2), "dtype": precision, "steps": steps, "batch_per_gpu": B,
                ("whole_step_frac_of_fp32_mfma_peak" if precision == "fp32" else "whole_step_frac_of_bf16_mfma_peak_over_3"): round(FLOP_PER_SAMPLE * B / dt / 1e12 / PEAK[precision], 4),
                "vs_ref": bf16_vs_reference(dev, precision),
                "note": MODE_NOTES[precision]}
    except Exception as e:      # the headline must not depend on the extra line
        return {"error": repr(e)[:300]}


def torch_stack_line(dev, args):
    """Reported next to the headline, never as `value`: the step CLiMB itself would run on this GPU -- transformers' `ViltModel` (the module
    REF/modeling/vilt.py:17 imports and delegates the encoder to), the reference's VQA head (REF/modeling/vilt.py:179-203), BCE-with-logits x 3129
    (REF/train/visionlanguage_tasks/train_vqa.py:95,157) and torch.optim.AdamW, in plain PyTorch-ROCm on the same MI355X, same batch shape, random-init
    weights: fp32 as the reference runs it (it has no autocast / half anywhere), and under bf16 autocast as the strongest setting that stack offers.
    Third-party library code only (no file of /root/reference, no oracle); a failure here never touches the headline."""
    import torch
    try:
        from transformers import ViltConfig, ViltModel
        B, T = args.batch, 40
        torch.manual_seed(0)
        enc = ViltModel(ViltConfig()).to(dev)          # ViLT-B/32 defaults: 12 x 768, 384 x 384 / 32, 40 text positions
        head = torch.nn.Sequential(torch.nn.Linear(768, 1536), torch.nn.LayerNorm(1536), torch.nn.GELU(), torch.nn.Linear(1536, 3129)).to(dev)
        enc.train()
        head.train()
        params = list(enc.parameters()) + list(head.parameters())
        g = torch.Generator().manual_seed(1)
        ids = torch.randint(0, 30522, (B, T), generator=g).to(dev)
        pixels = torch.randn(B, 3, 384, 384, generator=g).to(dev)
        mask = torch.ones(B, 384, 384, dtype=torch.long, device=dev)
        target = torch.zeros(B, 3129)
        target[torch.arange(B), torch.randint(0, 3129, (B,), generator=g)] = 1.0
        target = target.to(dev)
        out = {}
        for mode in ("fp32", "bf16_autocast"):
            opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=1e-2, betas=(0.9, 0.98))

            def step():
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=(mode != "fp32")):
                    pooled = enc(input_ids=ids, token_type_ids=torch.zeros_like(ids), attention_mask=torch.ones_like(ids), pixel_values=pixels, pixel_mask=mask).pooler_output
                    logits = head(pooled)
                loss = torch.nn.functional.binary_cross_entropy_with_logits(logits.float(), target) * target.shape[1]
                loss.backward()
                opt.step()
                opt.zero_grad()
            steps, warm = (6, 2) if mode == "fp32" else (10, 3)
            for _ in range(warm):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[mode] = {"value": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 2)}
        del enc, head, params
        torch.cuda.empty_cache()
        out["unit"] = "samples/s"
        out["note"] = ("transformers ViltModel + CLiMB's VQA head + BCE + torch.optim.AdamW in plain PyTorch-ROCm on this GPU (what the reference's own code path "
                       "would execute here): fp32 as the reference runs it, and under bf16 autocast; batch %d, 384x384 + 40 tokens, random-init" % B)
        return out
    except Exception as e:
        return {"error": repr(e)[:300]}


STEP_SOURCES_GLOB = ("*.hip", "*.h")


def step_hash():
    """sha256 over EVERY kernel source: the key `hbm_bytes_per_step` (all kernels of a step) is valid for."""
    import glob
    import hashlib
    d = os.path.join(ROOT, "climb_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(set(sum((glob.glob(os.path.join(d, g)) for g in STEP_SOURCES_GLOB), []))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def latest_traffic(precision="bf16"):
    import glob
    pat = "r[0-9][0-9]_traffic.json" if precision in ("bf16", "fp16") else f"r[0-9][0-9]_traffic_{precision}.json"      # (r06: the split-operand mode has its own passes)
    return json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))[-1]))      # the latest round's passes


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N ranks of the same command under torch.distributed.run, rendezvous on 127.0.0.1 (the container
    hostname may not resolve) and a port the kernel hands out.  Children inherit stdout / stderr: rank 0's JSON line is the last line on stdout."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def spawn_check(world, rank):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    who = [None] * dist.get_world_size()          # the same gather the real run prints as `rank_devices` (here: which process is which rank)
    dist.all_gather_object(who, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()})
    if rank == 0:
        print(json.dumps({"spawn_check": True, "n_gpus": world, "sum_of_ranks_plus_one": float(t.item()), "master_addr": os.environ.get("MASTER_ADDR"),
                          "process_group": {"backend": dist.get_backend(), "world_size": dist.get_world_size()}, "rank_devices": who,
                          "value_per_gpu_is_value_over": world}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def csrc_hash():
    """sha256 over the sources of the kernel family `roofline` is about (the bf16 NT GEMMs): the key profiles/rNN_traffic.json is valid for.
    Written ONLY by tools/summarize_profile.py, next to the PMC tables it summarises; edits to other kernels do not touch it."""
    import hashlib
    d = os.path.join(ROOT, "climb_amd", "csrc")
    h = hashlib.sha256()
    for f in NT_SOURCES:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def bf16_vs_reference(dev, precision="bf16"):
    """CHECKER leg (rank 0, N = 1, after the timed region; the only place besides cpu_baseline that touches oracle/): one step of the
    timed arithmetic mode on tests/golden/vqa_b64.npz -- the REFERENCE's own outputs for 64 seeded sequences (oracle/gen_golden.py) --
    and its errors against them (max|d| / max|ref|), plus the share of the 64 rows whose argmax answer equals the reference's."""
    import numpy as np
    import torch
    from oracle import vilt_oracle as vo
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    z = np.load(os.path.join(ROOT, "tests", "golden", "vqa_b64.npz"))
    m = dict(kv.split("=", 1) for kv in str(z["meta"][0]).split(";"))
    tasks, B = m["tasks"].split(","), int(m["B"])
    model = create_continual_learner_map["vilt"](model_name_or_path="random-init:0", ordered_cl_tasks=tasks, model_config=model_configs["vilt"],
                                                 task_configs=task_configs, device=dev, precision=precision)
    model.load_state_dict(vo.init_params(tasks, int(m["wseed"])), strict=True)
    model.to(dev)
    model.train()
    enc = vo.synthetic_encodings(B, seed=int(m["dseed"]))
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    loss, (pooled, logits), _, _ = model.fused_forward_backward("vqa", enc["pixel_values"], texts, vo.synthetic_vqa_targets(B, seed=int(m["dseed"])))

    def rel(a, b):
        a, b = a.detach().double().cpu(), torch.as_tensor(np.asarray(b)).double()
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    G = {n: p.grad.detach().double().cpu() for n, p in model.named_parameters() if p.grad is not None}
    names = [str(n) for n in z["grad_names"]]
    norms = np.array([float(G[n].norm()) for n in names])
    big = z["grad_norms"] > 1e-3 * z["grad_norms"].max()
    gerr = np.abs(norms - z["grad_norms"])[big] / z["grad_norms"][big]
    r = {"fixture": "tests/golden/vqa_b64.npz (reference train_step, B=64)", "pooled": round(rel(pooled, z["pooled"]), 6),
         "logits": round(rel(logits, z["logits"]), 6), "loss": round(rel(loss, z["loss"]), 8),
         "argmax_agreement": float((logits.argmax(-1).cpu().numpy() == z["logits"].argmax(-1)).mean()),
         "grad_norm_rel_err_median": round(float(np.median(gerr)), 6), "grad_norm_rel_err_max": round(float(gerr.max()), 6)}
    # north_star's bar, clause by clause, for THIS arithmetic mode (1e-3 relative; argmax bit-exact): the headline's bf16 mode meets the loss and the
    # typical gradient norm and misses the element-wise clauses by the 2^-9 operand rounding; fp32 mode meets all of them
    r["meets_1e-3"] = {"loss": r["loss"] <= 1e-3, "grad_norm_median": r["grad_norm_rel_err_median"] <= 1e-3, "grad_norm_max": r["grad_norm_rel_err_max"] <= 1e-3,
                       "pooled": r["pooled"] <= 1e-3, "logits": r["logits"] <= 1e-3, "argmax": r["argmax_agreement"] == 1.0}
    return r


def real_input_line(dev, args):
    """Reported next to the synthetic headline, never as `value` (VERDICT r2 missing #5; median steady-state step): the VQA TRAINER LOOP end to end -- JPEG files on
    disk -> the reference-format dataset + collate in DataLoader worker processes -> tokeniser + raw-byte staging + H2D + device image
    kernels on the prefetch thread / side stream (climb_amd/data/prefetch.py) -> the same fused step + AdamW -- at 64 examples per step on
    COCO-sized (640 x 480 / 480 x 640) images.  The reference runs all of the host half inline on the training thread
    (REF/modeling/vilt.py:83-96).  Images here are variable-resolution (padded canvases, packed patches), so a step is not the
    fixed-384 step of the headline."""
    import shutil
    import tempfile
    import types
    import numpy as np
    import torch
    try:
        from PIL import Image
        from tests import synth_data
        from climb_amd.configs.model_configs import model_configs
        from climb_amd.configs.task_configs import task_configs
        from climb_amd.modeling import create_continual_learner_map
        from climb_amd.train.task_trainer import VQATrainer
        work = tempfile.mkdtemp(prefix="climb_real_input_")
        try:
            B, steps_per_epoch, n_files = args.batch, 28, 128
            n = B * steps_per_epoch
            root = synth_data.make_climb_data_tree(os.path.join(work, "data"), n_train=n, n_val=2, seed=0, easy_answer=7)
            rng = np.random.default_rng(0)
            coco = os.path.join(root, "ms-coco", "images")
            for i in range(n):                                   # COCO-sized photograph stand-ins: smooth content + grain, JPEG quality 90;
                dst = os.path.join(coco, f"{1000 + i}.jpg")      # n_files distinct files, the other ids are links to them
                os.remove(dst)
                if i >= n_files:
                    os.symlink(os.path.join(coco, f"{1000 + i % n_files}.jpg"), dst)
                    continue
                w, h = (640, 480) if i % 4 else (480, 640)
                low = rng.integers(0, 256, size=(h // 16, w // 16, 3), dtype=np.uint8)
                img = np.asarray(Image.fromarray(low, "RGB").resize((w, h), Image.BICUBIC), dtype=np.int16) + rng.integers(-12, 13, size=(h, w, 3), dtype=np.int16)
                Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), "RGB").save(dst, quality=90)
            prev_vocab = os.environ.get("CLIMB_AMD_TOKENIZER_VOCAB")
            os.environ["CLIMB_AMD_TOKENIZER_VOCAB"] = synth_data.write_vocab(os.path.join(work, "vocab.txt"))
            workers = max(1, min(24, (os.cpu_count() or 8) // 2))
            a = types.SimpleNamespace(climb_data_dir=root, batch_size=B, num_workers=workers, cl_algorithm="sequential_ft", visual_input_type="pil-image")
            model = create_continual_learner_map["vilt"](model_name_or_path="random-init:42", ordered_cl_tasks=["vqa"], model_config=model_configs["vilt"],
                                                         task_configs=task_configs, device=dev, precision=args.precision)
            trainer = VQATrainer(a, task_configs, model_configs["vilt"], dev)
            opt = model.create_optimizer(trainer.hparams)
            model.train()
            out = {}
            skip = 10                                            # steady state: worker start-up, workspace allocation and the first look-ahead excluded
            for mode in ("1", "0"):                              # prefetched (the product's default), then inline on the training thread
                os.environ["CLIMB_AMD_PREFETCH"] = mode
                stamps = []
                for batch in trainer.prefetched(model, trainer.train_dataloader):
                    trainer.train_step(model, batch, opt)
                    torch.cuda.synchronize()
                    stamps.append(time.perf_counter())
                gaps = sorted(b - a_ for a_, b in zip(stamps[skip:-1], stamps[skip + 1:]))
                out[mode] = gaps[len(gaps) // 2]
                if mode == "1":                                   # the same step with its (last) prepared batch resident: what the loop costs without any input work
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(10):
                        trainer.train_step(model, batch, opt)
                    torch.cuda.synchronize()
                    out["resident"] = (time.perf_counter() - t0) / 10
                    out["seq"] = int(model._host.engine().last_ws.S)
            os.environ.pop("CLIMB_AMD_PREFETCH", None)
            if prev_vocab is None:
                os.environ.pop("CLIMB_AMD_TOKENIZER_VOCAB", None)
            else:
                os.environ["CLIMB_AMD_TOKENIZER_VOCAB"] = prev_vocab
            return {"value": round(B / out["1"], 1), "unit": "samples/s", "ms_per_step": round(out["1"] * 1e3, 2),
                    "inline_on_training_thread": {"value": round(B / out["0"], 1), "ms_per_step": round(out["0"] * 1e3, 2)},
                    "same_step_inputs_resident_ms": round(out["resident"] * 1e3, 2), "fraction_of_input_resident_rate": round(out["resident"] / out["1"], 3),
                    "tokens_per_sequence": out["seq"],
                    "dataloader_workers": workers, "examples_per_step": B,
                    "note": "VQATrainer loop on 640x480 / 480x640 JPEGs: disk -> dataset/collate workers -> prefetch thread (tokeniser, raw-byte staging, "
                            "H2D, device resize/normalise/pad) -> fused step + AdamW; variable-resolution canvases, so not the fixed-384 step of `value`"}
        finally:
            shutil.rmtree(work, ignore_errors=True)
    except Exception as e:      # the headline must not depend on the extra line
        return {"error": repr(e)[:300]}


def cpu_baseline(steps: int):
    """The CPU oracle (oracle/vilt_oracle.py: a plain-PyTorch fp32 port of the reference path, pinned to the reference by
    tests/golden) timed on this node's host cores, BASELINE.md section 4's protocol: `steps` (10) training steps at B=2 (BASELINE.json
    configs[0]) after 1 warm-up, median; and at B=64 -- the batch the GPU line is quoted on -- 1 untimed warm-up step (first-touch
    allocations of the new shape) and the median of 3 timed steps."""
    import torch
    from oracle import vilt_oracle as vo
    # torch's default intra-op thread count already honours the cgroup / affinity limits of this container
    # (os.cpu_count() does not, and oversubscribing OpenMP threads makes the baseline meaningless)
    cores = min(torch.get_num_threads(), int(os.environ.get("CLIMB_CPU_THREADS", "16")))      # measured on the EPYC 9575F host at B=2: 16 threads 5.6 samples/s, 32 -> 4.1, 64 -> 2.0, 128 -> 0.7
    torch.set_num_threads(cores)
    B = 2
    P = vo.init_params(["vqa"], 42)
    state = {}
    times = []
    for s in range(steps + 1):
        enc = vo.synthetic_encodings(B, seed=100 + s)
        tgt = vo.synthetic_vqa_targets(B, seed=100 + s)
        t0 = time.perf_counter()
        vo.train_step(P, "vqa", enc, tgt, opt_state=state, lr=1e-4)
        times.append(time.perf_counter() - t0)
    times = sorted(times[1:])
    med = times[len(times) // 2]
    B64, n64 = 64, int(os.environ.get("CLIMB_CPU_B64_STEPS", "3"))
    t64s = []
    for s in range(n64 + 1):
        enc = vo.synthetic_encodings(B64, seed=7 + s)
        tgt = vo.synthetic_vqa_targets(B64, seed=7 + s)
        t0 = time.perf_counter()
        vo.train_step(P, "vqa", enc, tgt, opt_state=state, lr=1e-4)
        t64s.append(time.perf_counter() - t0)
    t64 = sorted(t64s[1:])[len(t64s[1:]) // 2]
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": round(B64 / t64, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "port_pinned_by": "oracle/vilt_oracle.py = a restatement of the reference path, verified against the reference itself (REF + transformers, imported in "
                              "the build container) at <= 2e-5 by oracle/gen_golden.py while it wrote tests/golden/*; tests/test_oracle_golden.py re-checks it",
            "sample": f"fp32 training steps (fwd+BCE+bwd+AdamW), 384x384 + 40 tokens: median of {n64} steps at batch {B64} = the GPU line's batch "
                      f"({t64:.2f}s per step, 1 warm-up excluded); and {steps} steps at batch {B} (BASELINE configs[0]), median step {med:.3f}s, 1 warm-up excluded",
            "value_b2": round(B / med, 3), "cpu": model}


if __name__ == "__main__":
    main()
