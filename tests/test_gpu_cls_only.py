"""The opt-in dead-row elimination of the last encoder layer (CLIMB_AMD_CLS_ONLY_LAST=1) against the REFERENCE's fixtures:
the engine reads the knob at construction, so the step-level parity tests run in a subprocess with it set -- every task's single-image /
two-image / four-choice step, variable resolution and packed patches, the EWC penalty, the replay step, the 16-bit mode.  (The whole of tests/test_gpu_parity.py and tests/test_gpu_driver.py passes that way -- 67 tests,
9 minutes, ViLT-BERT, hipGraph replay, the Fisher pass, the autograd path and the two-rank runs included; this selection is what fits the suite.)
tests/test_gpu_parity.py::test_last_layer_on_cls_rows_only_equals_every_row compares the two steps directly, gradient by gradient."""
import os
import subprocess
import sys

import pytest

from tests import _background

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_fixtures_with_the_last_layer_on_cls_rows_only():
    # (r06: the suite's time budget -- the split mode has no row pruning, and the variable-resolution / EWC steps add nothing to what prunes here)
    # selection and environment: tests/_background.py (the child is started when collection ends; this test waits for it)
    r = _background.result("test_reference_fixtures_with_the_last_layer_on_cls_rows_only")
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
    print(r.stdout.strip().splitlines()[-1])
