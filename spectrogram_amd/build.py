"""Build recipe for libspectro.so (hipcc, gfx950 only).

    python -m spectrogram_amd.build            # incremental
    python -m spectrogram_amd.build --force

hipcc cross-compiles without a GPU.  The library is built IN-TREE at
spectrogram_amd/lib/libspectro.so so that it travels with the repository
snapshot to the GPU box; objects go to spectrogram_amd/lib/obj/.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libspectro.so")
DIAG_LIB = os.path.join(LIBDIR, "libsgdiag.so")
STREAM_DEMO = os.path.join(LIBDIR, "stream_latency")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))
ARCH = "gfx950"
LOG2N_RANGE = range(5, 16)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm; this package has no CPU build)")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def device_flags():
    """The hipcc flags every unit of libspectro.so is compiled with (tests/test_kernel_resources.py compiles single
    instantiations with exactly these, so that the register budgets it guards are those of the shipped binary)."""
    flags = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-I" + INCLUDE,
             "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
    # Wave-uniform branches (scalar compare + s_cbranch on run / frame / ring-position state) are left as the compare
    # trees they are: StructurizeCFG otherwise threads every uniform switch through a chain of mask flags and skip
    # branches (the LDS-ring window dispatch: 10-20 executed branches per frame instead of 4).
    if not os.environ.get("SG_STRUCTURIZE_UNIFORM"):  # tuning A/B only
        flags += ["-mllvm", "-structurizecfg-skip-uniform-regions"]
    flags += os.environ.get("SG_EXTRA_HIPCC_FLAGS", "").split()  # tuning experiments only
    return flags


def build(force=False, verbose=False, jobs=None):
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(INCLUDE, "spectro.h"))
    flags = device_flags()
    units = []
    for l2 in LOG2N_RANGE:
        units.append((os.path.join(CSRC, "stft_inst.hip"), os.path.join(OBJDIR, "stft_n%d.o" % l2),
                      ["-DSG_LOG2N=%d" % l2]))
    units.append((os.path.join(CSRC, "aux_kernels.hip"), os.path.join(OBJDIR, "aux_kernels.o"), []))
    units.append((os.path.join(CSRC, "spectro_api.hip"), os.path.join(OBJDIR, "spectro_api.o"), []))
    units.append((os.path.join(CSRC, "wav.cpp"), os.path.join(OBJDIR, "wav.o"), ["-x", "c++"]))  # host only
    units.append((os.path.join(CSRC, "axis.cpp"), os.path.join(OBJDIR, "axis.o"), ["-x", "c++"]))  # host only
    # the per-size kernel units see only csrc/*.hpp; the C ABI header is a dependency of the host-side units alone
    # (a change to a comment in include/spectro.h used to recompile eleven kernel units: three minutes)
    kernel_headers = [h for h in headers if not h.endswith("spectro.h")]
    todo = [(s, o, d) for s, o, d in units
            if force or _newer(o, [s] + (kernel_headers if os.path.basename(s) == "stft_inst.hip" else headers))]

    def compile_one(u):
        s, o, d = u
        out = _run([cc] + flags + d + ["-c", s, "-o", o])
        if verbose and out.strip():
            print(out)
        return o

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            list(ex.map(compile_one, todo))
    objs = [o for _, o, _ in units]
    if todo or force or _newer(LIB, objs):
        _run([cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs +
             ["-Wl,-rpath,/opt/rocm/lib", "-Wl,--no-undefined"])
    # measurement aid for bench.py's roofline object (not the product): libsgdiag.so
    diag_src = os.path.join(CSRC, "diag_stream.hip")
    if force or _newer(DIAG_LIB, [diag_src]):
        _run([cc] + flags + ["-shared", diag_src, "-o", DIAG_LIB, "-Wl,-rpath,/opt/rocm/lib"])
    # examples/stream_latency.c: config 4's per-hop loop in plain C over the C ABI (bench.py runs it beside its own loop)
    demo_src = os.path.normpath(os.path.join(HERE, "..", "examples", "stream_latency.c"))
    if os.path.exists(demo_src) and (force or _newer(STREAM_DEMO, [demo_src, LIB, os.path.join(INCLUDE, "spectro.h")])):
        _run([shutil.which("gcc") or "gcc", "-O2", "-std=c11", "-Wall", "-I" + INCLUDE, demo_src, "-o", STREAM_DEMO,
              "-L" + LIBDIR, "-lspectro", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
