"""Build spectro.node (the N-API shim) with plain gcc against node's headers -- no node-gyp."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "spectro.node")


def node_include():
    for d in ("/usr/include/node", "/usr/local/include/node"):
        if os.path.exists(os.path.join(d, "node_api.h")):
            return d
    return None


def build(force=False):
    inc = node_include()
    if inc is None or not shutil.which("gcc"):
        return None  # no node headers on this machine: the C ABI and the Python layer stand alone
    src = os.path.join(HERE, "napi_shim.c")
    lib = os.path.normpath(os.path.join(HERE, "..", "lib", "libspectro.so"))
    if (not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(src)
            and os.path.getmtime(OUT) >= os.path.getmtime(lib)):
        return OUT
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-Wall", "-I" + inc,
                           "-DNODE_GYP_MODULE_NAME=spectro", src, "-o", OUT,
                           "-L" + os.path.dirname(lib), "-lspectro",
                           "-Wl,-rpath,$ORIGIN/../lib", "-Wl,-rpath,/opt/rocm/lib"])
    return OUT


if __name__ == "__main__":
    print(build(force=True))
