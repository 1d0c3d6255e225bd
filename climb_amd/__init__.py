"""climb_amd: MI355X-native (gfx950) implementation of CLiMB's ViLT continual-fine-tuning step.

Python host code mirrors the reference's `modeling` / `cl_algorithms` surface; all device work goes through the
C-ABI library built from `climb_amd/csrc` (see include/climb_hip.h).  There is no CPU fallback."""
__version__ = "0.1.0"
