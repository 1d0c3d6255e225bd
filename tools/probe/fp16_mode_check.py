"""One step of the 16-bit throughput mode on tests/golden/vqa_b64.npz (the reference's outputs at the benchmark's batch size), in the
operand type the environment selects:  CLIMB_AMD_H16=fp16 python tools/probe/fp16_mode_check.py   (bench.py's checker, reused)"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
prec = os.environ.get("CLIMB_AMD_H16", "bf16")
print(prec, json.dumps(bench.bf16_vs_reference(torch.device("cuda:0"), prec)))
