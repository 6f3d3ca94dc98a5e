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
MFMA_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
PER_SOURCE = {"knn.hip": MFMA_VGPR_FORM}
_probe = {}


def mfma_vgpr_form_supported():
    """Does this hipcc's LLVM know -amdgpu-mfma-vgpr-form?  Probed once by compiling an empty translation unit; a compiler
    without it still builds the library (192 registers and AGPR copies in the split-bf16 kernel: slower, same results) and
    bench.py records which it was (build_flags.mfma_vgpr_form)."""
    if "ok" not in _probe:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            src = os.path.join(d, "probe.hip")
            with open(src, "w") as f:
                f.write("#include <hip/hip_runtime.h>\n__global__ void probe_kernel() {}\n")
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "--cuda-device-only"] + MFMA_VGPR_FORM + ["-c", src, "-o", os.path.join(d, "probe.o")],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _probe["ok"] = r.returncode == 0
    return _probe["ok"]


def per_source_flags(src):
    fl = PER_SOURCE.get(src, [])
    if fl == MFMA_VGPR_FORM and not mfma_vgpr_form_supported():
        print(f"[gloc3d build] warning: {HIPCC} rejects {' '.join(MFMA_VGPR_FORM)}: {src} is built without it "
              "(the split-bf16 distance kernel keeps its accumulators in AGPRs: same results, slower)", file=sys.stderr)
        return []
    return fl


FLAG_NOTE = os.path.join(LIBDIR, "build_flags.json")


def _stale(out, deps, cmd=None):
    """Out of date by time stamps -- or (round 6) by FLAGS: every object has a side-car `<obj>.cmd` holding the command line
    that compiled it; another command line (other -D switches, a compiler without -amdgpu-mfma-vgpr-form, GLOC3D_EXTRA_FLAGS
    set or unset) rebuilds it, so that build_flags.json below describes what the objects were REALLY compiled with."""
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    if cmd is not None:
        try:
            with open(out + ".cmd") as f:
                return f.read() != _cmd_text(cmd)
        except OSError:
            return True
    return False


def _cmd_text(cmd):
    """The command line with the checkout's own path taken out (the snapshot on the GPU box lives elsewhere)."""
    return " ".join(cmd).replace(os.path.dirname(HERE), "$ROOT")


def _compile(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(cmd[-1] + ".cmd", "w") as f:      # (-o <obj> is last)
        f.write(_cmd_text(cmd))


def _recorded_flags(obj):
    try:
        with open(obj + ".cmd") as f:
            return f.read().split()
    except OSError:
        return None


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(HERE, "..", "include", "gloc3d.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        cmd = [HIPCC] + FLAGS + per_source_flags(src) + ["-c", s, "-o", o]
        if force or _stale(o, [s] + headers, cmd):
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(lambda c: _compile(c, verbose), jobs))
    if jobs or force or _stale(LIB, objs) or not os.path.exists(FLAG_NOTE):
        # Linked WITHOUT a NEEDED entry for libamdhip64: the host process decides which HIP runtime
        # it runs on (PyTorch bundles its own; two runtimes in one process cannot share the GPU).
        # gloc3d_amd.capi preloads one with RTLD_GLOBAL; the C++ command lines link /opt/rocm's.
        run(["g++", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
        import json
        # the note is made from what each object was compiled with (its side-car), not from a fresh probe of the compiler
        per_obj = {os.path.basename(o): _recorded_flags(o) for o in objs}
        knn = per_obj.get("knn.o") or []
        defines = sorted({f for fl in per_obj.values() for f in (fl or []) if f.startswith("-D")})
        with open(FLAG_NOTE, "w") as f:   # travels with the .so (git-ignored like it); bench.py copies it into its line
            json.dump({"mfma_vgpr_form": "-amdgpu-mfma-vgpr-form" in knn, "flags": FLAGS, "extra_flags": EXTRA, "defines": defines,
                       "per_object": {k: (" ".join(v) if v else None) for k, v in per_obj.items()}, "hipcc": HIPCC}, f)
    return LIB


def build_flags():
    """What the library in lib/ was built with (None when the note is missing: a library built by hand)."""
    import json
    try:
        with open(FLAG_NOTE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


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
    cmd = [HIPCC] + FLAGS + ["-DGLOC_NN_EAGER", "-c", src, "-o", obj]
    if force or _stale(obj, [src] + headers, cmd):
        _compile(cmd, verbose)
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
