"""Build libmtr_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  No fallback: if hipcc is
missing or the build fails this raises."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmtr_hip.so")
SOURCES = ["mtr_abi.hip", "mtr_common.h", "device_util.hip.inc", "k1_ranges.hip.inc", "k2_units.hip.inc", "k3_staged.hip.inc",
           "dp_wrap.hip.inc", "dp_quad.hip.inc", "gather.hip.inc", "min_missing_table.h", os.path.join("..", "..", "include", "mtr_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fno-fast-math",      # fp64 DI values and float ratios must round exactly like the reference's C
         "-fhip-fp32-correctly-rounded-divide-sqrt",
         # The per-read kernel is one 9 000-line function at 128 VGPRs: hoisting loop invariants (mostly "pointer + lane"
         # addresses) to its top made 275 of them spill and be reloaded inside the loops (19 GB of HBM traffic per launch,
         # DESIGN.md 4.5).  No machine-level hoisting + sinking of what the IR level hoisted: 61 spilled VGPRs, +4.7 % reads/s.
         "-mllvm", "-sink-insts-to-avoid-spills", "-mllvm", "-disable-machine-licm"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libmtr_hip.so cannot be built (there is no CPU fallback)")


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


LIB_PROF = os.path.join(HERE, "libmtr_hip_prof.so")      # same code + the per-phase shader-clock timers (-DMTR_PROFILE)


def build(force: bool = False, verbose: bool = False, profile: bool = False) -> str:
    """libmtr_hip.so (product).  profile=True builds libmtr_hip_prof.so as well: the same kernels with the phase timers
    compiled in (select it with MTR_LIB=.../libmtr_hip_prof.so; tests/dev/gpu_phase.py does)."""
    targets = [(LIB, [])] + ([(LIB_PROF, ["-DMTR_PROFILE"])] if profile else [])
    running = []
    for lib, extra in targets:
        stale = force or not os.path.exists(lib) or any(os.path.getmtime(os.path.join(CSRC, s)) > os.path.getmtime(lib) for s in SOURCES)
        if not stale:
            continue
        cmd = [hipcc(), *FLAGS, *extra, "-o", lib, os.path.join(CSRC, "mtr_abi.hip")]
        running.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=CSRC)))    # (the two builds side by side)
    failure = None
    for cmd, p in running:
        if failure is not None:             # one build failed: the other is not left behind writing its library
            p.kill()
            p.communicate()
            continue
        out, err = p.communicate()
        if p.returncode != 0:
            failure = "hipcc failed (" + " ".join(cmd) + "):\n" + out + err
        elif verbose:
            print(" ".join(cmd))
    if failure is not None:
        raise RuntimeError(failure)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, profile="--profile" in sys.argv))
