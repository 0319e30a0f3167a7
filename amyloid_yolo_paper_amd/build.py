"""Build libamyloid_yolo_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libamyloid_yolo_hip.so")
SOURCES = ["ay_layout.hip", "ay_conv_bf16.hip", "ay_conv3x3_m16.hip", "ay_conv_f32.hip", "ay_conv_f32_mfma.hip", "ay_yolo.hip", "ay_nms.hip", "ay_train_f32.hip", "ay_stem_fused.hip", "ay_stem_train.hip", "ay_train_bf16.hip", "ay_wgrad_bf16.hip", "ay_ingest.hip", "ay_resblock_bf16.hip", "ay_stats.hip", "ay_merge.hip", "ay_plan.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]
if os.environ.get("AY_PHASE_CLOCK"):  # instrumented build: in-kernel phase clock of the ring convolution (AY_DBG=8 at run time)
    FLAGS.append("-DAY_PHASE_CLOCK")
FLAGS += os.environ.get("AY_CXXFLAGS", "").split()  # experiment builds (e.g. -DAY_BUF_STORE=0)


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=True):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "ay_common.h"), os.path.join(CSRC, "ay_conv_common.h"), os.path.join(HERE, "..", "include", "amyloid_yolo.h")]
    objs, procs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
