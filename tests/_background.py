"""Subprocess suites of the GPU run, started when collection ends instead of when their test is reached (r06).

Three GPU tests are `pytest` runs of their own in a child process, because the engine reads what they vary once per process: the 16-bit operand
type of the library (tests/test_gpu_fp16_build.py) and the dead-row elimination knob (tests/test_gpu_cls_only.py).  Run inline they were 80 s of
a 580 s suite during which this process only waits.  conftest.py starts the selected ones in `pytest_collection_finish` and moves their tests to the
end of the run; the tests call `result(name)`, which waits for the child (or, when the job was not started -- a test run by node id on a box
where collection saw no GPU --, runs it then and there).  Nothing about WHAT is checked changes: same command lines, same assertions on the
child's exit code and output; only the waiting overlaps with the rest of the suite.  Children still alive when the session ends (an `-x` abort)
are terminated by PID."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PYTEST = ["-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"]

# test function name -> (argv after the interpreter, environment additions, environment removals, timeout)
SPECS = {
    "test_fp16_build_passes_the_kernel_suite": (_PYTEST + ["tests/test_gpu_kernels.py"], {"CLIMB_AMD_H16": "fp16"}, ("CLIMB_AMD_LIB",), 1500),
    "test_fp16_mode_passes_the_step_level_parity_tests": (
        _PYTEST + ["tests/test_gpu_parity.py", "-k", "training_curve or reference_style_autograd or fisher_accumulating or ewc_penalty or hipgraph"],
        {"CLIMB_AMD_H16": "fp16"}, ("CLIMB_AMD_LIB",), 2400),
    "test_reference_fixtures_with_the_last_layer_on_cls_rows_only": (
        _PYTEST + ["tests/test_gpu_parity.py", "-k", "(single_image or nlvr2_two_images or vcr_four or replay_step or bf16_mode_step) and not bf16x3"],
        {"CLIMB_AMD_CLS_ONLY_LAST": "1"}, (), 2400),
}
_JOBS = {}


class Result:
    def __init__(self, returncode, stdout):
        self.returncode, self.stdout, self.stderr = returncode, stdout, ""


def _spawn(name):
    argv, add, drop, _ = SPECS[name]
    env = dict(os.environ, **add)
    for k in drop:
        env.pop(k, None)
    env["CLIMB_AMD_BACKGROUND_CHILD"] = "1"          # (a child never starts grandchildren at ITS collection)
    log = tempfile.TemporaryFile(mode="w+")
    return subprocess.Popen([sys.executable] + argv, cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT, text=True), log


def start(name):
    if name in SPECS and name not in _JOBS:
        _JOBS[name] = _spawn(name)


def result(name):
    proc, log = _JOBS.pop(name, None) or _spawn(name)
    try:
        proc.wait(timeout=SPECS[name][3])
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.wait()
    log.seek(0)
    out = log.read()
    log.close()
    return Result(proc.returncode, out)


def stop_all():
    for name, (proc, log) in list(_JOBS.items()):
        if proc.poll() is None:
            proc.terminate()
            try:
                proc.wait(timeout=20)
            except subprocess.TimeoutExpired:
                proc.kill()
                proc.wait()
        log.close()
        _JOBS.pop(name, None)
