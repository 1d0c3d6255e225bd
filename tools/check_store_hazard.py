"""Store-data hazard report for every kernel of the library (r06).

Measured on gfx950 (tools/probe/split_fused_nan3.py, DESIGN.md section 0 "findings"): a VALU instruction that writes one of the data registers of a
`buffer_store_dwordx4` in the issue slot RIGHT AFTER it can reach memory instead of the stored value -- the AdamW epilogue of the split grouped
weight-gradient launch stored `m` / `v` elements holding the select of two `p` values that the quad transpose had just written into the same register
(80 + 304 of 589 824 elements of one matrix, in lanes 12-15 / 28-31 of a half-wave, not every launch).  The compiler's hazard table inserts the wait
state only for MUBUF stores WITHOUT an SGPR offset operand (the pre-gfx9 rule); the epilogues here use the SGPR offset for the uniform part of
every address.  One instruction of distance was enough in every case seen (the write two slots behind the same store never showed).

Usage: python tools/check_store_hazard.py [--f16 | -DNAME ...] [file.hip ... | lib.so ...]  (a .so: the shipped binary's code objects are disassembled, seconds); compiles each file to gfx950 assembly and lists every MUBUF store of more than 64 bits
with an SGPR offset whose data registers are written by the next vector instruction, and exits non-zero if there is one.  (Found and fixed in r06:
the grouped weight-gradient launch's optimizer epilogues -- 32 places in the split instantiation, 1 and 5 in the 16-bit and EWC ones, now every register-side
step precedes the 16-byte stores and a wait state follows them -- and the persistent NT kernel's fp32 store, whose first data register the next
accumulator read-back landed in: no wrong element was ever seen there, the store now keeps its registers live over one `s_nop`.)"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "climb_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
NOT_VALU = ("buffer_", "ds_", "global_", "flat_", "s_", "scratch_")
KNOWN = ()


EXTRA = []          # (--f16: the IEEE-half build's flags, climb_amd/build.py; -D...: any other build variant)


def _analyse(tag, ins):
    found = []
    for i, (k, l) in enumerate(ins[:-1]):
        m = re.match(r"buffer_store_dwordx[34] v\[(\d+):(\d+)\], \S+ s\[\d+:\d+\], s\d+", l)
        if not m:
            continue
        regs = set(range(int(m.group(1)), int(m.group(2)) + 1))
        nx = ins[i + 1][1]
        if nx.startswith(NOT_VALU):
            continue
        mm = re.match(r"(\S+)\s+(?:v\[(\d+):(\d+)\]|v(\d+))", nx)
        if not mm:
            continue
        dst = set(range(int(mm.group(2)), int(mm.group(3)) + 1)) if mm.group(2) else {int(mm.group(4))}
        if dst & regs:
            found.append((tag, k, l, nx, mm.group(1)))
    return found


def scan(path):
    """one source file, compiled to assembly"""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "--cuda-device-only"] + EXTRA + ["-S", path, "-o", out],
                           capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(r.stderr[-2000:])
        kern, ins = None, []
        for l in open(out):
            l = l.strip()
            if not l or l.startswith((";", ".")):
                continue
            lab = re.match(r"(\S+):(\s|$)", l)
            if lab:
                if lab.group(1).startswith("_Z"):
                    kern = lab.group(1)
                continue
            ins.append((kern, l))
    return _analyse(os.path.basename(path), ins)


def scan_library(lib):
    """the SHIPPED binary: every gfx950 code object bundled into the shared library's .hip_fatbin section, disassembled (seconds, no compiler run) --
    what tests/test_host_logic.py checks after every build"""
    llvm = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib", "llvm", "bin")
    found, nobj, nins = [], 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        offs = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)] + [len(data)]
        for i in range(len(offs) - 1):
            b, co = os.path.join(tmp, f"b{i}.bin"), os.path.join(tmp, f"co{i}.o")
            open(b, "wb").write(data[offs[i]:offs[i + 1]])
            subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={b}", f"--output={co}"],
                           check=True, capture_output=True)
            dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
            kern, ins = None, []
            for l in dis.splitlines():
                lab = re.match(r"^[0-9a-f]+ <(\S+)>:$", l)
                if lab:
                    kern = lab.group(1)
                    continue
                l = l.split("//")[0].strip()
                if l and kern and not l.endswith(":") and not l.startswith(("Disassembly", lib)):
                    ins.append((kern, l))
            nobj += 1
            nins += len(ins)
            found += _analyse(os.path.basename(lib), ins)
    return found, nobj, nins


def main(files):
    bad = 0
    libs = [f for f in files if f.endswith(".so")]
    srcs = [f for f in files if not f.endswith(".so")]
    results = []
    for lib in libs:
        res, nobj, nins = scan_library(lib)
        print(f"{os.path.basename(lib)}: {nobj} code objects, {nins} instructions")
        results.append(res)
    if srcs:
        with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
            results += list(ex.map(scan, srcs))
    for res in results:
        for f, k, st, nx, op in res:
            known = (f, op) in KNOWN
            print(f"{'known ' if known else 'HAZARD'}  {f}  {(k or '?')[:70]}\n        {st}\n        {nx}")
            bad += 0 if known else 1
    print("no store-data hazards" if not bad else f"{bad} store-data hazard(s)")
    return bad


if __name__ == "__main__":
    for a in [a for a in sys.argv[1:] if a.startswith("-")]:
        sys.argv.remove(a)
        EXTRA.extend(["-DCLIMB_H16_F16=1"] if a == "--f16" else [a])
    fs = sys.argv[1:] or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
    sys.exit(1 if main(fs) else 0)
