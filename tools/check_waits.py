#!/usr/bin/env python3
"""Do the strip kernels of the built library still keep three blocks of loads in flight -- and wait for them at all?

The march of k_strip (seabreeze_param_amd/csrc/sb_strip_kernel.hip) stages block i from registers whose loads were
issued three steps earlier.  hipcc places the s_waitcnt vmcnt(N) for them: N counts the vector-memory operations issued
since, so N >= 8 in the steady state of a kernel that loads four values per row (theta, z, sigma, land-side word) and
N >= 4 where it loads two (t0, land-side word).  Two ways this has gone wrong, both silent, both seen:
  * a change of the control flow makes the compiler's count collapse to vmcnt(0..2): every step then sits out the full
    memory latency;
  * hipcc 7.2 emitted NO wait for the staged registers (fp64 kernels, an unconditional 16-bit load at the end of the
    step): the results then depend on timing.
This script disassembles the gfx950 code objects of the library (llvm-objdump from the ROCm tree) and reports, per
k_strip variant, the histogram of vmcnt waits; it fails if a variant has fewer than three waits at the expected depth.

usage: check_waits.py [path/to/libseabreeze_hip.so]      (exit code 0: fine)
"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def strip_kernel_waits(lib):
    """{kernel name: Counter of vmcnt values} for every k_strip instance in the library's gfx950 code objects"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([OBJDUMP, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([OBJDUMP, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            name = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:$", line)
                if m:
                    name = m.group(1) if m.group(1).startswith(("_Z7k_strip", "_Z9k_strip32")) else None
                    if name:
                        out[name] = collections.Counter()
                    continue
                if name:
                    w = re.search(r"s_waitcnt[^/]*vmcnt\((\d+)\)", line)
                    if w:
                        out[name][int(w.group(1))] += 1
    return out


def expected_depth(name):
    """fly (t0 derived while staging: four loads a row) -> 8, else 4; k_strip32 stages two pieces of a row: 16 and 8"""
    m = re.match(r"_Z9k_strip32ILb([01])E", name)
    if m:
        return 16 if m.group(1) == "1" else 8
    m = re.match(r"_Z7k_stripI([fd])Lb([01])E", name)
    return 8 if m.group(2) == "1" else 4


def check(lib):
    waits = strip_kernel_waits(lib)
    bad = []
    for name, hist in sorted(waits.items()):
        need = expected_depth(name)
        deep = sum(v for k, v in hist.items() if k >= need)
        print(f"{name}: waits at depth >= {need}: {deep}; histogram {sorted(hist.items())}")
        if deep < 3:
            bad.append(name)
    if not waits:
        print("no k_strip kernel found")
        return 2
    if bad:
        print("FAILED:", ", ".join(bad))
        return 1
    return 0


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    sys.exit(check(sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "seabreeze_param_amd", "libseabreeze_hip.so")))
