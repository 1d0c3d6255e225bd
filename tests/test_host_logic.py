"""CPU tests of the host side: C-ABI library loads and exports every symbol of include/climb_hip.h, flat layout
invariants, reference-compatible names/grouping, and the plugin / trainer host logic (no device compute)."""
import ctypes
import os
import random
import types

import numpy as np
import pytest
import torch

from oracle import vilt_oracle as vo


def test_abi_library_exports_every_header_symbol():
    from climb_amd import _lib
    from climb_amd.build import build_library
    lib = build_library(verbose=False)
    so = ctypes.CDLL(lib)
    protos = _lib.parse_header()
    assert len(protos) >= 30
    missing = [n for n in protos if not hasattr(so, n)]
    assert not missing, missing
    _lib.load()
    assert _lib.query("climb_version") >= 100
    assert _lib.load().climb_arch() == b"gfx950"
    assert _lib.error_string(-1).startswith("climb")
    # exported but undeclared entry points would be an undocumented ABI
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T climb_" in l}
    assert exported == set(protos), exported ^ set(protos)
    # the IEEE-half build of the same sources: the same ABI, and it says which 16-bit type it computes in
    from climb_amd.build import LIB_F16
    assert os.path.exists(LIB_F16)
    out16 = subprocess.run(["nm", "-D", "--defined-only", LIB_F16], capture_output=True, text=True).stdout
    assert {l.split()[-1] for l in out16.splitlines() if " T climb_" in l} == set(protos)
    so16 = ctypes.CDLL(LIB_F16)
    so16.climb_h16.restype = ctypes.c_char_p
    assert so16.climb_h16() == b"fp16" and _lib.h16() == os.environ.get("CLIMB_AMD_H16", "bf16")


def test_one_process_holds_one_16_bit_operand_type():
    """engine precision "fp16" needs libclimb_hip_f16.so; a process that already loaded the bf16 build must be told so, not silently
    compute in the wrong type."""
    from climb_amd import _lib
    _lib.load()
    other = "fp16" if _lib.h16() == "bf16" else "bf16"
    with pytest.raises(RuntimeError, match="one 16-bit operand type"):
        _lib.select_h16(other)
    _lib.select_h16(_lib.h16())          # asking for the loaded one is fine
    with pytest.raises(ValueError):
        _lib.select_h16("fp8")


def test_engine_refuses_cpu():
    from climb_amd.engine import ViltEngine
    from climb_amd.layout import FlatLayout, TASK_ARITH
    eng = ViltEngine(FlatLayout(["vqa"], TASK_ARITH), torch.device("cpu"), "fp32")
    with pytest.raises(RuntimeError, match="no CPU path"):
        eng.allocate()


def test_flat_layout_invariants():
    from climb_amd.layout import FlatLayout, TASK_ARITH, ENC, ALIGN
    lay = FlatLayout(["vqa", "nlvr2"], TASK_ARITH)
    assert list(lay.shapes.keys()) == list(vo.param_shapes(["vqa", "nlvr2"]).keys())
    assert all(tuple(lay.shapes[n]) == tuple(s) for n, s in vo.param_shapes(["vqa", "nlvr2"]).items())
    assert all(o % ALIGN == 0 for o in lay.offset.values())
    segs = lay.segments()
    assert all(a[1] + a[2] == b[1] for a, b in zip(segs, segs[1:])) and segs[-1][1] + segs[-1][2] == lay.total
    assert all(lay.numel(n) <= ln for n, _, ln in segs)
    H = 768
    for i in range(12):
        l = f"{ENC}encoder.layer.{i}.attention.attention."
        assert lay.offset[l + "key.weight"] == lay.offset[l + "query.weight"] + H * H      # fused [2304,768] QKV is a view
        assert lay.offset[l + "value.weight"] == lay.offset[l + "key.weight"] + H * H
        assert lay.offset[l + "key.bias"] == lay.offset[l + "query.bias"] + H
        lo, hi = lay.layer_range[i]
        assert all(lo <= lay.offset[n] < hi for n in lay.shapes if f".layer.{i}." in n)
    enc_names = [n for n in lay.shapes if n.startswith(ENC)]
    assert max(lay.offset[n] for n in enc_names) < lay.encoder_end <= min(lay.offset[n] for n in lay.shapes if not n.startswith(ENC))
    assert lay.embed_range[1] == lay.layer_range[0][0] and lay.layer_range[-1][1] == lay.top_range[0]
    assert FlatLayout(["vqa"], TASK_ARITH).shapes[ENC + "embeddings.token_type_embeddings.weight"] == (2, H)
    assert lay.shapes[ENC + "embeddings.token_type_embeddings.weight"] == (3, H)


def test_decay_grouping_matches_reference_quirk():
    from climb_amd.layout import no_decay
    for n in vo.param_shapes(["vqa", "nlvr2", "snli-ve", "vcr"]):
        assert no_decay(n) == vo.no_decay(n)


def test_schedule_matches_transformers():
    from transformers import get_polynomial_decay_schedule_with_warmup
    from climb_amd.train import polynomial_decay_schedule_with_warmup
    p1, p2 = torch.nn.Parameter(torch.zeros(1)), torch.nn.Parameter(torch.zeros(1))
    o1, o2 = torch.optim.AdamW([p1], lr=1e-4), torch.optim.AdamW([p2], lr=1e-4)
    s1 = get_polynomial_decay_schedule_with_warmup(o1, num_warmup_steps=3, num_training_steps=30, lr_end=0, power=1)
    s2 = polynomial_decay_schedule_with_warmup(o2, 3, 30, 0.0, 1.0)
    for _ in range(33):
        assert abs(o1.param_groups[0]["lr"] - o2.param_groups[0]["lr"]) < 1e-15
        o1.step(); o2.step(); s1.step(); s2.step()


def _cpu_model(tasks):
    from climb_amd.modeling import create_continual_learner_map
    from climb_amd.configs.task_configs import task_configs
    from climb_amd.configs.model_configs import model_configs
    return create_continual_learner_map["vilt"](model_name_or_path="random-init:0", ordered_cl_tasks=tasks, model_config=model_configs["vilt"],
                                                task_configs=task_configs, device=torch.device("cpu"), precision="fp32")


def test_model_surface_on_cpu_and_loud_failure():
    model = _cpu_model(["vqa", "nlvr2"])
    P = vo.init_params(["vqa", "nlvr2"], 42)
    assert [n for n, _ in model.named_parameters()] == list(P.keys())
    model.load_state_dict(P)
    assert list(model.get_encoder().state_dict().keys()) == [k[len("vilt_encoder."):] for k in P if k.startswith(vo.ENC)]
    assert model.get_encoder().vilt.embeddings.token_type_embeddings.weight.shape[0] == 3      # nlvr2 => third modality row
    opt_groups = [
        {n for n, p in model.named_parameters() if not vo.no_decay(n)},
        {n for n, p in model.named_parameters() if vo.no_decay(n)},
    ]
    assert len(opt_groups[0]) + len(opt_groups[1]) == len(P)
    enc = vo.synthetic_encodings(2, seed=1)
    texts = dict(input_ids=enc["input_ids"], token_type_ids=enc["token_type_ids"], attention_mask=enc["attention_mask"])
    with pytest.raises(RuntimeError, match="no CPU path"):          # the product path never silently falls back to CPU
        model(task_key="vqa", images=enc["pixel_values"], texts=texts)
    model.get_encoder().freeze_bottom_k_layers(9)
    assert model._host.frozen_prefix() == (9, False)
    model.get_encoder().freeze_all_weights()
    assert model._host.any_encoder_grad() is None


def test_multi_image_and_multi_choice_batching():
    """One encoder call of b*n sequences: row order and image_token_type_idx must reproduce the reference's per-pass loop
    (REF/modeling/vilt.py:281-304 and :331-347)."""
    model = _cpu_model(["nlvr2", "vcr"])
    b = 3
    e = vo.synthetic_encodings(2 * b, seed=3)
    enc = dict(input_ids=e["input_ids"][:b], token_type_ids=e["token_type_ids"][:b], attention_mask=e["attention_mask"][:b],
               pixel_values=e["pixel_values"], pixel_mask=e["pixel_mask"])
    out, kind = model._expand("nlvr2", enc)
    assert kind == ("images", 2)
    assert out["input_ids"].shape[0] == 2 * b and torch.equal(out["input_ids"][0], out["input_ids"][1]) and torch.equal(out["input_ids"][2], enc["input_ids"][1])
    assert out["image_token_type_idx"].tolist() == [1, 2, 1, 2, 1, 2]
    pooled = torch.arange(2 * b * 4, dtype=torch.float32).view(2 * b, 4)
    shaped = model._shape_pooled(pooled, kind)
    assert torch.equal(shaped, torch.cat([pooled[0::2], pooled[1::2]], dim=-1))
    e = vo.synthetic_encodings(4 * b, seed=4)
    enc = dict(input_ids=e["input_ids"], token_type_ids=e["token_type_ids"], attention_mask=e["attention_mask"],
               pixel_values=e["pixel_values"][:b], pixel_mask=e["pixel_mask"][:b])
    out, kind = model._expand("vcr", enc)
    assert kind == ("choice", 4) and out["pixel_values"].shape[0] == 4 * b
    assert torch.equal(out["pixel_values"][5], enc["pixel_values"][1])
    shaped = model._shape_pooled(torch.arange(4 * b * 2, dtype=torch.float32).view(4 * b, 2), kind)
    assert shaped.shape == (b, 4, 2) and shaped[1, 2, 0] == (4 * 1 + 2) * 2


def test_replay_memory_host_logic():
    from climb_amd.cl_algorithms import ExperienceReplayMemory
    from climb_amd.configs.task_configs import task_configs
    calls = {}

    class FakeTrainer:
        hparams = {"lr": 1e-4, "weight_decay": 1e-2, "adam_epsilon": 1e-8}

        def __init__(self):
            self.loader = types.SimpleNamespace(dataset=list(range(1000)), collate_fn=lambda items: {"items": items})

        def get_train_dataloader(self):
            return self.loader

        def get_collate_fn(self):
            return self.loader.collate_fn

        def train_step(self, model, batch, optimizer=None, scheduler=None, ewc=None):
            calls["batch"], calls["opt"], calls["sched"] = batch, optimizer, scheduler
            return torch.tensor(1.5), None, None, None

    class FakeModel:
        def create_optimizer(self, hp):
            calls["hp"] = hp
            return "fresh-optimizer"
    random.seed(0)
    mem = ExperienceReplayMemory()
    assert not mem.do_replay()
    for key, bs in (("vqa", 64), ("nlvr2", 32), ("vcr", 16)):
        mem.add_task_memory_buffer(args=types.SimpleNamespace(batch_size=64), task_key=key, task_config=task_configs[key], task_trainer=FakeTrainer(),
                                   memory_percentage=0.01, sampling_strategy="random")
        assert len(mem.memory_buffers[key]) == 10 and mem.memory_buffers[key].batch_size == bs      # REF experience_replay.py:93-98
    assert mem.do_replay() and mem.sample_replay_task() in ("vqa", "nlvr2", "vcr")
    mem.memory_buffers["vqa"].batch_size = 4
    loss = mem.run_replay_step("vqa", FakeModel())
    assert float(loss) == 1.5 and calls["opt"] == "fresh-optimizer" and calls["sched"] is None and calls["hp"]["lr"] == 1e-4
    assert len(calls["batch"]["items"]) == 4 and set(calls["batch"]["items"]) <= set(mem.memory_buffers["vqa"].memory_idxs)
    with pytest.raises(AssertionError):
        mem.add_task_memory_buffer(args=types.SimpleNamespace(batch_size=64), task_key="snli-ve", task_config=task_configs["snli-ve"],
                                   task_trainer=FakeTrainer(), memory_percentage=0.01, sampling_strategy="random-balanced")


def test_task_configs_carry_reference_hyperparameters():
    from climb_amd.configs.task_configs import task_configs
    for k, lr, ep in (("vqa", 1e-4, 10), ("nlvr2", 1e-4, 10), ("snli-ve", 5e-5, 5), ("vcr", 1e-4, 10)):
        c = task_configs[k]
        assert (c["lr"], c["num_epochs"], c["weight_decay"], c["adam_epsilon"]) == (lr, ep, 1e-2, 1e-8)
        assert c["num_labels"] == vo.TASKS[k]["num_labels"] and c["model_type"] == vo.TASKS[k]["model_type"]


def test_background_suites_start_wait_and_stop(monkeypatch):
    """tests/_background.py (r06): a child suite started at collection time is waited for by its test, one that was not started runs when asked,
    and children still alive at session end are terminated by PID."""
    from tests import _background as bg
    monkeypatch.setitem(bg.SPECS, "quick", (["-c", "import os; print('child', os.environ['MARK'], os.environ.get('CLIMB_AMD_BACKGROUND_CHILD'), '1 passed')"], {"MARK": "x"}, (), 60))
    monkeypatch.setitem(bg.SPECS, "slow", (["-c", "import time; time.sleep(600)"], {}, (), 600))
    bg.start("quick")
    assert "quick" in bg._JOBS
    r = bg.result("quick")
    assert r.returncode == 0 and "child x 1 1 passed" in r.stdout and "quick" not in bg._JOBS
    r = bg.result("quick")                      # not started: runs now
    assert r.returncode == 0 and " passed" in r.stdout
    bg.start("slow")
    proc = bg._JOBS["slow"][0]
    assert proc.poll() is None
    bg.stop_all()
    assert proc.poll() is not None and not bg._JOBS


def test_shipped_libraries_have_no_store_data_hazard():
    """r06: on gfx950 a vector instruction that writes a data register of a `buffer_store_dwordx4` in the slot right after it can reach memory instead of the
    stored value, and the compiler inserts no wait state behind a store with an SGPR offset (tools/check_store_hazard.py, DESIGN.md section 0: found as
    negative second moments out of an optimizer epilogue).  Every gfx950 code object of the BUILT libraries is disassembled and searched for that pair
    (seconds, no compiler run); the same search over the pre-fix build of gemm_bf16_tnp.hip reports its six places."""
    import importlib.util
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_store_hazard", os.path.join(ROOT, "tools", "check_store_hazard.py"))
    ch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ch)
    libs = [os.path.join(ROOT, "climb_amd", "csrc", f) for f in ("libclimb_hip.so", "libclimb_hip_f16.so")]
    libs = [l for l in libs if os.path.exists(l)]
    if not libs or not os.path.exists(ch.HIPCC):
        pytest.skip("needs the built libraries and the ROCm llvm tools")
    for lib in libs:
        found, nobj, nins = ch.scan_library(lib)
        assert nobj >= 10 and nins > 100000, (lib, nobj, nins)          # (the search saw the library's kernels at all)
        assert not [f for f in found if (f[0], f[4]) not in ch.KNOWN], found[:3]


def test_fp32_gemm_does_not_spill():
    """Compile-time guard (hipcc resource report, no GPU): the parity mode's GEMM must keep its accumulators in registers.  Adding
    epilogue cases to its runtime switch once pushed the 128x128 instantiation into scratch and halved its rate unnoticed."""
    import importlib.util
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_spills", os.path.join(ROOT, "tools", "check_spills.py"))
    cs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cs)
    if not os.path.exists(cs.HIPCC):
        pytest.skip("hipcc not available")
    rows = cs.report(os.path.join(ROOT, "climb_amd", "csrc", "gemm_f32.hip"))
    assert rows, "no resource report parsed"
    for name, vgpr, scratch in rows:
        if not any(k in name for k in cs.KNOWN):
            assert scratch == 0, (name, vgpr, scratch)


@pytest.mark.parametrize("layers,nwg", [(12, 256), (6, 256), (4, 256), (1, 256), (12, 64)])
def test_grouped_weight_gradient_plan_covers_every_reduction_tile_once(layers, nwg):
    """climb_tn_grouped_plan (host function of the library): for ViLT-B's four weight gradients per layer (+ the patch projection's, with
    its own token count) every (tile, 64-token reduction tile) unit appears in exactly one item, items hold >= 2 reduction tiles, whole
    tiles are never marked partial (they use the plain read-modify-write), and no workgroup carries more than its share + one tile."""
    from climb_amd import _lib
    lib = _lib.load()
    shapes = [(12288, 768, 3072), (12288, 3072, 768), (12288, 768, 768), (12288, 2304, 768)] * layers + [(9216, 768, 3072)]
    M, N, K = (np.ascontiguousarray([s[i] for s in shapes], dtype=np.int32) for i in range(3))
    tiles = int(sum((n // 256) * (k // 256) for _, n, k in shapes))
    cap = tiles + nwg + 1
    items = np.zeros((cap, 8), dtype=np.int32)
    first = np.zeros(nwg + 1, dtype=np.int32)
    n = lib.climb_tn_grouped_plan(len(shapes), M.ctypes.data, N.ctypes.data, K.ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data)
    assert 0 < n <= cap and first[0] == 0 and first[-1] == n and (np.diff(first) >= 0).all()
    items = items[:n]
    seen = {}
    for p, tn, tk, k0, k1, partial, _, _ in items:
        nkt = shapes[p][0] // 64
        assert 0 <= tn < shapes[p][1] // 256 and 0 <= tk < shapes[p][2] // 256 and 0 <= k0 < k1 <= nkt and k1 - k0 >= 2
        assert bool(partial) == (k0 != 0 or k1 != nkt)
        cover = seen.setdefault((p, tn, tk), np.zeros(nkt, dtype=np.int32))
        cover[k0:k1] += 1
    assert len(seen) == tiles and all((c == 1).all() for c in seen.values())
    load = np.array([sum(int(k1 - k0) for _, _, _, k0, k1, _, _, _ in items[first[b]:first[b + 1]]) for b in range(nwg)])
    total = sum(s[0] // 64 * (s[1] // 256) * (s[2] // 256) for s in shapes)
    assert load.sum() == total and load.max() <= total / nwg + 192 + 2          # balanced to within one tile's reduction
    assert lib.climb_tn_grouped_plan(1, M.ctypes.data, N.ctypes.data, K.ctypes.data, 100, items.ctypes.data, cap, first.ctypes.data) == -1      # nwg % 8
    odd = np.array([12288, 12288, 2048], dtype=np.int32), np.array([768, 48, 264], dtype=np.int32), np.array([48, 768, 200], dtype=np.int32)
    assert lib.climb_tn_grouped_plan(3, odd[0].ctypes.data, odd[1].ctypes.data, odd[2].ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data) > 0    # ragged N, K: multiples of 8
    bad = np.array([12288 + 64], dtype=np.int32)
    assert lib.climb_tn_grouped_plan(1, bad.ctypes.data, N.ctypes.data, K.ctypes.data, 256, items.ctypes.data, cap, first.ctypes.data) == -1   # M % 128
    assert lib.climb_tn_grouped_plan(len(shapes), M.ctypes.data, N.ctypes.data, K.ctypes.data, nwg, items.ctypes.data, 8, first.ctypes.data) == -2


def test_bench_gpus_n_launches_itself_and_prints_one_json_line_last():
    """the driver starts the scaling run as plain `python bench.py --gpus N` -- without a launcher around it bench.py must become
    the launcher (N ranks of its own command line under torch.distributed.run on 127.0.0.1 / a free port) and rank 0's JSON line must be the last
    line on stdout.  `--spawn-check` runs exactly that plumbing without a GPU (gloo group, one all-reduce across the ranks)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--spawn-check"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    j = json.loads(last)
    assert j["spawn_check"] is True and j["n_gpus"] == 2 and j["sum_of_ranks_plus_one"] == 3.0 and j["master_addr"] == "127.0.0.1"
    # r06: the world size as the process group reports it, and one entry per rank from distinct processes
    assert j["process_group"] == {"backend": "gloo", "world_size": 2}
    assert sorted(d["rank"] for d in j["rank_devices"]) == [0, 1] and len({d["pid"] for d in j["rank_devices"]}) == 2


def test_nt4_tile_walk_is_a_bijection():
    """The four-wave NT kernel (csrc/gemm_bf16_nt4.hip) walks its tiles without a division: launch position id -> XCD id % 8, whose share is one
    supertile of nbm / 8 M-tiles x all N-tiles, walked M first, `step` = grid / 8 positions per persistent iteration.  The same arithmetic in
    Python: every tile exactly once, for the benchmark's shapes, a reduced grid (CUs left to RCCL) and tile counts below the grid."""
    def walk(nbm, nbn, G):
        nwg = nbm * nbn
        G = min(G // 8 * 8, nwg)
        gm, step = nbm // 8, G // 8
        seen = []
        for b in range(G):
            xcd, idx = b & 7, b >> 3
            tn, tml = idx // gm, idx % gm
            for _ in range((nwg - b + G - 1) // G):
                seen.append((xcd * gm + tml, tn))
                tml += step
                while tml >= gm:
                    tml -= gm
                    tn += 1
        return seen
    for nbm, nbn, G in [(64, 4, 256), (64, 12, 256), (64, 16, 256), (64, 16, 224), (96, 16, 256), (8, 1, 256), (16, 3, 256), (48, 4, 256), (64, 12, 64)]:
        seen = walk(nbm, nbn, G)
        assert len(seen) == nbm * nbn and set(seen) == {(m, n) for m in range(nbm) for n in range(nbn)}, (nbm, nbn, G)


@pytest.mark.parametrize("stagger", [0, 2, 4, 8])
def test_grouped_weight_gradient_plan_covers_every_tile_once_with_staggered_epilogues(stagger):
    """climb_tn_grouped_plan is host code (no device needed): for the benchmark's 49 weight-gradient problems on 256 workgroups -- and a small case where
    everything is stream-K tail -- every output tile's reduction range [0, tokens / 64) is covered exactly once by its items, cut points are even with
    >= 2 reduction tiles per item, every workgroup gets the same number of whole tiles and the same share of the tail (to one cut), and with
    climb_set_option(22, G) phase group g = idx % G of an XCD runs g / (G - 1) of its tail share BEFORE its whole tiles (r05: the optimizer-carrying
    epilogues of the 256 workgroups no longer arrive as one burst)."""
    import numpy as np
    from climb_amd import _lib
    lib = _lib.load()
    assert lib.climb_set_option(22, stagger) == 0
    try:
        layer = [(12288, 768, 3072), (12288, 3072, 768), (12288, 768, 768), (12288, 2304, 768)]
        for probs, nwg in [(layer * 12 + [(9216, 768, 3072)], 256), (layer, 256), (layer * 3, 64)]:
            Ms, Ns, Ks = (np.ascontiguousarray([p[i] for p in probs], dtype=np.int32) for i in range(3))
            ntiles = int(sum(((n + 255) // 256) * ((k + 255) // 256) for _, n, k in probs))
            cap = ntiles + 2 * nwg + 1
            items, first = np.zeros((cap, 8), dtype=np.int32), np.zeros(nwg + 1, dtype=np.int32)
            n = lib.climb_tn_grouped_plan(len(probs), Ms.ctypes.data, Ns.ctypes.data, Ks.ctypes.data, nwg, items.ctypes.data, cap, first.ctypes.data)
            assert n > 0 and first[nwg] == n and first[0] == 0 and (np.diff(first) >= 0).all()
            items = items[:n]
            cover = {}
            for prob, tn, tk, k0, k1, partial, _, _ in items:
                nkt = probs[prob][0] // 64
                assert 0 <= k0 < k1 <= nkt and k1 - k0 >= 2 and k0 % 2 == 0 and k1 % 2 == 0
                assert partial == int(k0 != 0 or k1 != nkt)
                assert tn < (probs[prob][1] + 255) // 256 and tk < (probs[prob][2] + 255) // 256
                cover.setdefault((prob, tn, tk), []).append((k0, k1))
            assert len(cover) == ntiles
            for (prob, tn, tk), rs in cover.items():
                rs.sort()
                assert rs[0][0] == 0 and rs[-1][1] == probs[prob][0] // 64 and all(a[1] == b[0] for a, b in zip(rs[:-1], rs[1:])), (prob, tn, tk, rs)
            rounds = ntiles // nwg
            per_xcd = nwg // 8
            G = stagger if stagger >= 2 and rounds >= 1 else 1
            tails = []
            for b in range(nwg):
                mine = items[first[b]:first[b + 1]]
                whole = [i for i, it in enumerate(mine) if it[5] == 0]
                assert len(whole) >= rounds
                units = int(sum(it[4] - it[3] for it in mine if it[5] == 1))
                tails.append(units)
                if G > 1 and rounds >= 1 and units:
                    rank = (b % 8) * per_xcd + b // 8
                    g = (rank % per_xcd) % G
                    before = int(sum(it[4] - it[3] for it in mine[:whole[0]] if it[5] == 1)) if whole else units
                    # (a share whose pieces cannot be cut at the wanted place keeps them whole: within one piece of the target)
                    assert abs(before - units * g / (G - 1)) <= max(2, max(it[4] - it[3] for it in mine if it[5] == 1)), (b, g, before, units)
            if G > 1 and rounds >= 1 and max(tails) >= 8:
                befores = [int(sum(it[4] - it[3] for it in items[first[b]:first[b + 1]][:next((i for i, it in enumerate(items[first[b]:first[b + 1]]) if it[5] == 0), 0)]))
                           for b in range(nwg)]
                assert len(set(befores)) >= 2, "no stagger: every workgroup runs the same amount of tail work before its whole tiles"
    finally:
        lib.climb_set_option(22, 0)


def test_committed_traffic_figures_belong_to_this_tree():
    """bench.py reports `roofline.traffic` / `roofline.hbm_bytes_per_step` from the latest profiles/rNN_traffic.json only when the file was measured on
    THIS tree's kernel sources (it names their hashes); a kernel edit after the last PMC pass silently nulled the figure in the round-4 driver line.
    This test is the guard: after editing anything under climb_amd/csrc, re-run tools/profile_step.sh + tools/summarize_profile.py and commit."""
    import bench
    tj = bench.latest_traffic()
    assert tj["csrc_sha16"] == bench.csrc_hash(), "the NT GEMM sources changed since the committed PMC passes: roofline.traffic would be null"
    assert tj.get("step_sha16") == bench.step_hash(), "a kernel source changed since the committed PMC passes: roofline.hbm_bytes_per_step would be null"
    assert tj["gemm_bf16_nt"]["hbm_bytes_per_launch"] > 0 and tj["hbm_bytes_per_step"] > 0
