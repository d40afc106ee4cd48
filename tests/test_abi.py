"""CPU: the C-ABI library builds for gfx950, loads, and exports exactly what
include/spectro.h declares.  No compute calls (no GPU here) -- but the product path must
fail loudly, not fall back, when there is no device."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "spectro.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sg_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_header():
    from spectrogram_amd import _lib, build
    path = build.build()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "libspectro.so does not export %s" % s
    # and the ctypes table binds exactly the header's set
    assert sorted(_lib.SYMBOLS) == syms


def test_code_object_is_gfx950_only():
    from spectrogram_amd import build
    path = build.build()
    blob = open(path, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx90a", b"gfx942", b"sm_80", b"sm_90"):
        assert other not in blob


def test_config_default_and_version():
    import spectrogram_amd as S
    from spectrogram_amd import _lib as L
    lib = S.load()
    assert lib.sg_version() == 400
    cfg = L.SgConfig()
    assert lib.sg_config_default(ctypes.byref(cfg)) == 0
    assert (cfg.struct_size, cfg.fft_size, cfg.hop) == (ctypes.sizeof(L.SgConfig), 1024, 256)
    assert (cfg.min_db, cfg.max_db, cfg.floor_db) == (-100.0, -30.0, -200.0)
    assert (cfg.logfreq_align, cfg.history_samples) == (L.LOGFREQ_PIXEL_CENTRE, 0)
    assert lib.sg_config_default(None) == L.ERR_INVALID


def test_bad_configs_rejected_before_touching_the_device():
    import spectrogram_amd as S
    from spectrogram_amd import _lib as L
    for kw in (dict(fft_size=1000), dict(fft_size=16), dict(fft_size=65536), dict(hop=0),
               dict(min_db=-10.0, max_db=-20.0), dict(smoothing=1.5), dict(fft_size=2048, history_samples=1024)):
        with pytest.raises(S.SpectroError) as e:
            S.Engine(**kw)
        assert e.value.code == L.ERR_INVALID, kw
    with pytest.raises(ValueError):
        S.Analyser(fftSize=1000)


def test_library_never_reads_the_environment():
    """Tuning / test switches go through sg_debug_set (include/spectro.h); the product library has no getenv at all
    (six SG_* switches used to be read from the environment at sg_create)."""
    import subprocess
    from spectrogram_amd import build
    und = subprocess.check_output(["nm", "-D", "--undefined-only", build.build()], text=True)
    assert not re.search(r"\b(secure_)?getenv\b", und), "libspectro.so imports getenv"
    src = open(os.path.join(ROOT, "spectrogram_amd", "csrc", "spectro_api.hip")).read()
    src = re.sub(r"#if SG_STAMP.*?#endif", "", src, flags=re.S)   # the diagnostic stamp build (never the product) aside
    assert "getenv" not in src


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import spectrogram_amd as S
    from spectrogram_amd import _lib as L
    with pytest.raises(S.SpectroError) as e:
        S.Engine()
    assert e.value.code == L.ERR_HIP


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under spectrogram_amd/ may reference it
    pkg = os.path.join(ROOT, "spectrogram_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".c", ".js", ".h")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
                assert "liboracle" not in text and "stft_oracle" not in text, f


def _build_c_demo(tmp):
    import subprocess
    from spectrogram_amd import build
    lib = build.build()
    exe = os.path.join(tmp, "stft_demo")
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "stft_demo.c"), "-o", exe,
                           "-L" + os.path.dirname(lib), "-lspectro", "-Wl,-rpath," + os.path.dirname(lib),
                           "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    return exe


def test_c_abi_demo_compiles_and_links(tmp_path):
    """A plain C caller builds against include/spectro.h + libspectro.so (no torch, no Python)."""
    exe = _build_c_demo(str(tmp_path))
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_c_abi_demo_runs(tmp_path):
    import subprocess
    exe = _build_c_demo(str(tmp_path))
    out = subprocess.check_output([exe], text=True)
    assert out.strip().endswith("ok"), out


def test_stream_latency_driver_is_built_beside_the_library():
    """examples/stream_latency.c (config 4's per-hop loop in plain C) is compiled by spectrogram_amd/build.py next to
    libspectro.so, so that bench.py can run it on the GPU box without a compiler step of its own."""
    from spectrogram_amd import build
    build.build()
    assert os.path.exists(build.STREAM_DEMO) and os.access(build.STREAM_DEMO, os.X_OK)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", ["0", "1"])
def test_stream_latency_driver_runs(fused):
    import json
    import subprocess
    from spectrogram_amd import build
    build.build()
    out = subprocess.check_output([build.STREAM_DEMO, "200", "64", "2048", "512", "256", "1", fused], text=True, timeout=120)
    j = json.loads(out.strip().splitlines()[-1])
    assert j["hops"] == 200 and j["channels"] == 64 and j["row_bytes"] == 256 and j["pcm"] == "s16"
    assert j["fused_push_poll"] == (fused == "1")
    assert 0 < j["host_wall"]["p50"] < j["budget_ms"] and j["checksum"] > 0


# ---- which HIP runtime the process ends on: a stated policy, checked in fresh processes ---------------

def _fresh(code, env=None):
    import subprocess
    import sys
    e = dict(os.environ)
    e.pop("SPECTRO_HIP_RUNTIME", None)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", "import sys, json\nsys.path.insert(0, %r)\n%s" % (ROOT, code)],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_runtime_policy_three_load_orders():
    """engine first (torch installed, not imported): torch's runtime is loaded first because its SONAME is the one
    libspectro.so needs, and a later `import torch` shares it; torch first: nothing is preloaded, same runtime; "system"
    (argument or SPECTRO_HIP_RUNTIME): /opt/rocm's.  sg_runtime_info -- dladdr inside the library -- says where each ended."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None:
        pytest.skip("no torch in this environment")
    torch_lib = os.path.realpath(os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so"))
    a = _fresh("from spectrogram_amd import _lib as L\nassert 'torch' not in sys.modules\ni = L.runtime_info()\n"
               "import torch\ni['torch_hip'] = torch.version.hip\nprint(json.dumps(i))")
    assert os.path.realpath(a["library_path"]) == torch_lib and a["preloaded"] and a["policy"] == "auto"
    assert a["torch_hip"] and str(a["runtime_version"])[:2] == a["torch_hip"].replace(".", "")[:2]   # the wheel's runtime, e.g. 7.0
    b = _fresh("import torch\nfrom spectrogram_amd import _lib as L\nprint(json.dumps(L.runtime_info()))")
    assert os.path.realpath(b["library_path"]) == torch_lib and b["preloaded"] is None and "imported first" in b["note"]
    for env, arg in (({"SPECTRO_HIP_RUNTIME": "system"}, ""), ({}, "'system'")):
        c = _fresh("from spectrogram_amd import _lib as L\nL.load(%s)\nprint(json.dumps(L.runtime_info()))" % arg, env)
        assert c["library_path"].startswith("/opt/rocm") and c["preloaded"] is None and c["policy"] == "system"
        assert c["runtime_version"] == c["built_with_hip_version"]          # the runtime it was built against
    assert a["built_with_hip_version"] == c["built_with_hip_version"] != a["runtime_version"]   # 7.2 code objects on the wheel's 7.0: visible


def test_runtime_policy_refuses_a_runtime_of_another_soname(tmp_path, monkeypatch):
    """the candidate's SONAME is compared with libspectro's DT_NEEDED entry; a mismatch is skipped with a warning
    under "auto" and is an error under "torch"."""
    import warnings
    from spectrogram_amd import _lib as L
    soname, needed = L.elf_dynamic(L.LIB_PATH)
    want = [n for n in needed if n.startswith("libamdhip64")]
    assert want == ["libamdhip64.so.%s" % want[0].rsplit(".", 1)[1]] and L._needed_hip_soname(L.LIB_PATH) == want[0]
    # a fake torch whose "runtime" carries another SONAME: libspectro.so itself (SONAME None / not libamdhip64)
    fake = tmp_path / "torch"
    (fake / "lib").mkdir(parents=True)
    (fake / "__init__.py").write_text("")
    import shutil
    shutil.copy(os.path.join(ROOT, "spectrogram_amd", "lib", "libsgdiag.so"), fake / "lib" / "libamdhip64.so")
    code = ("sys.path.insert(0, %r)\nimport warnings\nfrom spectrogram_amd import _lib as L\n"
            "with warnings.catch_warnings(record=True) as w:\n    warnings.simplefilter('always')\n    L._apply_runtime_policy('auto', L.LIB_PATH)\n"
            "ok = False\ntry:\n    L._apply_runtime_policy('torch', L.LIB_PATH)\nexcept RuntimeError as e:\n    ok = 'SONAME' in str(e)\n"
            "print(json.dumps({'warned': [str(x.message) for x in w], 'state': L.runtime_policy(), 'torch_policy_raises': ok}))" % str(tmp_path))
    r = _fresh(code)
    assert r["torch_policy_raises"] and r["state"]["preloaded"] is None
    assert len(r["warned"]) == 1 and "not preloaded" in r["warned"][0]


def test_node_binds_the_system_runtime():
    """node without torch: nothing loads a runtime before libspectro.so does -- /opt/rocm's, through its RUNPATH."""
    import shutil
    import subprocess
    node = shutil.which("node")
    addon = os.path.join(ROOT, "spectrogram_amd", "js", "spectro.node")
    if node is None or not os.path.exists(addon):
        pytest.skip("node or the N-API addon is not available")
    import json
    out = subprocess.check_output([node, "-e", "console.log(JSON.stringify(require(%r).runtimeInfo()))" % addon], text=True, timeout=60)
    i = json.loads(out.strip().splitlines()[-1])
    assert i["libraryPath"].startswith("/opt/rocm") and i["runtimeVersion"] == i["builtWithHipVersion"] > 0


# ---- the host side under AddressSanitizer + UBSan, no GPU ------------------------

def test_host_side_under_sanitizers():
    """make -C oracle check-shim-asan: csrc/spectro_api.hip compiled host-only with -fsanitize=address,undefined on top of a
    host-memory stand-in for the device (tests/host/hip_stub.cpp, launch_stub.cpp) -- (1) tests/host/shim_driver.cpp over the
    C ABI: argument checks, staging growth, padded outputs, ragged pushes read back sample for sample through the history ring,
    the fused call's bookkeeping, the byte ring's row order, sg_set_fft_size under injected allocation failures, nothing left
    allocated, LeakSanitizer on; (2) js/napi_shim.c behind js/analyser.js under node with the sanitizer runtime preloaded,
    driven by tests/js_shim_args.js."""
    import shutil
    import subprocess
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm's clang is not installed")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "check-shim-asan"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "shim driver ok:" in r.stdout and "ERROR: AddressSanitizer" not in r.stdout, r.stdout[-3000:]
    assert "runtime error" not in r.stdout, r.stdout[-3000:]
    if shutil.which("node") and os.path.exists("/usr/include/node/node_api.h"):
        assert '{"ok":true' in r.stdout, r.stdout[-2000:]


def test_the_sanitizer_build_really_catches_a_host_side_overrun(tmp_path):
    """A check of the checker: the same build with one line of the stub changed so that a history row is one float short must
    die in AddressSanitizer inside the push path -- the green run above is not green because nothing is instrumented."""
    import subprocess
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm's clang is not installed")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "shim_asan"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    src = open(os.path.join(ROOT, "tests", "host", "hip_stub.cpp")).read()
    mutated = src.replace("*p = std::malloc(n ? n : 1);\n  if (!*p) return g_last = hipErrorOutOfMemory;\n  std::memset(*p, 0xCD, n);",
                          "*p = std::malloc(n > 4 ? n - 4 : 1);\n  if (!*p) return g_last = hipErrorOutOfMemory;\n  std::memset(*p, 0xCD, n > 4 ? n - 4 : 0);")
    assert mutated != src
    stub = tmp_path / "hip_stub_short.cpp"
    stub.write_text(mutated)
    shim = os.path.join(ROOT, "oracle", "_shim")
    exe = str(tmp_path / "shim_short")
    san = ["-O1", "-g", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17"] + san + ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                    os.path.join(ROOT, "tests", "host", "shim_driver.cpp"), str(stub)] +
                   [os.path.join(shim, o) for o in ("spectro_api.o", "launch_stub.o", "wav.o", "axis.o")], check=True, timeout=600)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode != 0 and "AddressSanitizer: heap-buffer-overflow" in r.stdout, r.stdout[-1500:]


def test_the_device_model_really_catches_work_on_the_wrong_device(tmp_path):
    """A check of the multi-device section of tests/host/shim_driver.cpp (eight handles on eight stub devices, every call made
    from a caller sitting on another device; no 8-GPU node is available to the builder): with a stub whose hipSetDevice never
    switches -- a library that forgot to enter its handle's device looks exactly like that -- the driver must fail on the
    first call that touches device memory, with the stub naming the foreign access."""
    import subprocess
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("ROCm's clang is not installed")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "shim_asan"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    src = open(os.path.join(ROOT, "tests", "host", "shim_driver.cpp")).read()
    # the driver itself moves the CALLER between devices through hipSetDevice: give it a switch of its own and take the
    # library's away (the library reaches the stub's hipSetDevice, the driver the renamed one)
    stub_src = open(os.path.join(ROOT, "tests", "host", "hip_stub.cpp")).read()
    mutated = stub_src.replace("hipError_t hipSetDevice(int d) {\n  if (d < 0 || d >= g_device_count) return g_last = hipErrorInvalidDevice;\n  g_device = d;",
                               "extern \"C\" int caller_set_device(int d) { g_device = d; return 0; }\n"
                               "hipError_t hipSetDevice(int d) {\n  if (d < 0 || d >= g_device_count) return g_last = hipErrorInvalidDevice;\n  (void)d;")
    assert mutated != stub_src
    drv = src.replace('extern "C" int hipSetDevice(int d);', 'extern "C" int caller_set_device(int d);\n#define hipSetDevice caller_set_device')
    assert drv != src
    (tmp_path / "hip_stub_stuck.cpp").write_text(mutated)
    (tmp_path / "shim_driver_stuck.cpp").write_text(drv.replace('#include "../../include/spectro.h"', '#include "%s"' % os.path.join(ROOT, "include", "spectro.h")))
    shim = os.path.join(ROOT, "oracle", "_shim")
    exe = str(tmp_path / "shim_stuck")
    san = ["-O1", "-g", "-fPIC", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-std=c++17"] + san + ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", exe,
                    str(tmp_path / "shim_driver_stuck.cpp"), str(tmp_path / "hip_stub_stuck.cpp")] +
                   [os.path.join(shim, o) for o in ("spectro_api.o", "launch_stub.o", "wav.o", "axis.o")], check=True, timeout=600)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode != 0 and ("device that is not current" in r.stdout or "another device" in r.stdout
                                  or "sg_stub_blocks_on_device" in r.stdout), r.stdout[-1500:]
