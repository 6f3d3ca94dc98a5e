"""In-tree build of the HIP extension: gloc3d_amd/lib/libgloc3d.so (gfx950 only).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels to the GPU
box with the snapshot.  `python -m gloc3d_amd.build` or `gloc3d_amd.build.build()`.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libgloc3d.so")
SOURCES = ["common.hip", "comm.hip", "knn.hip", "scan_store.hip", "reg.hip", "vlad.hip", "bev.hip", "ground.hip", "coarse.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the exact kernels must reproduce the reference's un-fused fp32 arithmetic
EXTRA = os.environ.get("GLOC3D_EXTRA_FLAGS", "").split()
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# knn.hip: MFMA accumulators in architectural VGPRs -- the split-bf16 distance kernel adds its accumulators to running
# totals every 64 k, and from AGPRs that is a v_accvgpr_read per register and step (measured: 121 of the main loop's
# ~330 vector instructions); without AGPRs the kernel also fits three work-groups per CU (161 registers, not 192)
PER_SOURCE = {"knn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(HERE, "..", "include", "gloc3d.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([HIPCC] + FLAGS + PER_SOURCE.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        # Linked WITHOUT a NEEDED entry for libamdhip64: the host process decides which HIP runtime
        # it runs on (PyTorch bundles its own; two runtimes in one process cannot share the GPU).
        # gloc3d_amd.capi preloads one with RTLD_GLOBAL; the C++ command lines link /opt/rocm's.
        run(["g++", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


def build_test_variant(force=False, verbose=False):
    """lib/libgloc3d_smallq.so: the same library with the culled 1-NN kernel evaluating its work queue before EVERY test
    step, incomplete rounds included (-DGLOC_NN_EAGER; the shipped kernel evaluates full rounds between steps and the
    rest at the end of a chunk): bounds tighten between all steps of a chunk and tail rounds of every size occur.  Test
    infrastructure (tests/test_reg_variant_gpu.py runs the bit-identity tests through it via GLOC3D_LIB_PATH); only
    reg.hip is compiled again."""
    build(force=force, verbose=verbose)
    out = os.path.join(LIBDIR, "libgloc3d_smallq.so")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    src = os.path.join(CSRC, "reg.hip")
    obj = os.path.join(LIBDIR, "reg_smallq.o")
    if force or _stale(obj, [src] + headers):
        cmd = [HIPCC] + FLAGS + ["-DGLOC_NN_EAGER", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    objs = [os.path.join(LIBDIR, s_.replace(".hip", ".o")) for s_ in SOURCES if s_ != "reg.hip"] + [obj]
    if force or _stale(out, objs):
        cmd = ["g++", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_cli(force=False, verbose=False):
    """The drop-in command lines (registration/global_localization, global_registration)."""
    build(force=force, verbose=verbose)
    bindir = os.path.join(HERE, "bin")
    os.makedirs(bindir, exist_ok=True)
    outs = []
    for name in ("global_localization", "global_registration"):
        src = os.path.join(CSRC, "cli", name + ".cpp")
        if not os.path.exists(src):
            continue
        out = os.path.join(bindir, name)
        deps = [src, LIB] + [os.path.join(CSRC, "host", f) for f in os.listdir(os.path.join(CSRC, "host"))]
        if force or _stale(out, deps):
            cmd = ["g++", "-O2", "-std=c++17", "-I" + os.path.join(HERE, "..", "include"),
                   "-I" + os.path.join(CSRC, "host"), src, "-o", out, "-L" + LIBDIR, "-lgloc3d",
                   "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,$ORIGIN/../lib",
                   "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        outs.append(out)
    return outs


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_cli(force="--force" in sys.argv, verbose=True)
    build_test_variant(force="--force" in sys.argv, verbose=True)
    print(LIB)
