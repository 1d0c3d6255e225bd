#!/bin/bash
# Where do the bf16 attention kernels spend their time?  Builds measurement variants of attention_bf16.hip next to the product library:
#   AB_PROBE bits  1 no inner-loop arithmetic, 2 no staging stream, 4 no output stores   (timed with tools/attn_bench.py)
#   AB_TRACE       per-workgroup wall-clock stamps of the forward kernel                 (tools/attn_trace.py)
# build:  bash tools/attn_probe.sh build      (here; hipcc cross-compiles)
# run:    bash tools/attn_probe.sh            (on the GPU box)
set -e
cd "$(dirname "$0")/.."
OUT=tools/probe/build
mkdir -p $OUT
variant() {   # name, flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc $2 -c climb_amd/csrc/attention_bf16.hip -o $OUT/attention_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libclimb_hip_$1.so $OUT/attention_$1.o $(ls climb_amd/csrc/build/*.o | grep -v attention_bf16)
}
if [ "$1" = build ]; then
  python -m climb_amd.build > /dev/null
  for p in 1 2 3 4 5 6 7; do variant p$p -DAB_PROBE=$p; done
  variant tr -DAB_TRACE
  exit 0
fi
echo "== product"; python tools/attn_bench.py
for p in 1 2 3 4 5 6 7; do
  echo "== AB_PROBE=$p"; CLIMB_AMD_LIB=$PWD/$OUT/libclimb_hip_p$p.so python tools/attn_bench.py
done
echo "== forward timeline"; CLIMB_AMD_LIB=$PWD/$OUT/libclimb_hip_tr.so python tools/attn_trace.py
