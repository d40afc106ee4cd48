"""CPU: register / scratch budgets of the kernels the BASELINE configs run (hipcc cross-compiles gfx950 without a GPU).

Round 5 found the config-4 poll kernel at 278 registers -- one wave per SIMD, +3.5 us per hop -- after a change that passed every
parity test: a wave-uniform RUN-TIME test had put two bodies of code side by side in one kernel.  Occupancy is part of what these
kernels promise, so the promise is tested: each instantiation below is compiled on its own (explicit
template instantiation, seconds each) and its resource summary read from the assembly."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")

# (log2n, load variant, output kind, max VGPRs, min waves per SIMD): scratch must be 0 for all of them
BUDGETS = [
    (10, 2, 0, 128, 4),         # the headline kernel: N = 1024, hop N/4, float32 power (configs 2 and 5)
    (10, 2, 3, 128, 4),         # ... bytes (the reference's encoding)
    (11, 2, 0, 128, 4),         # config 3: N = 2048, hop N/4
    (12, 2, 0, 128, 4),         # N = 4096 batch
    (13, 2, 0, 128, 4),         # N = 8192 batch
    (10, 1, 0, 128, 4),         # LDS sample ring, N = 1024 (any hop)
    (11, 1, 3, 128, 4),         # LDS sample ring, N = 2048, bytes: the reference's own cadence (hop 735)
    (12, 0, 8 | 3, 256, 2),     # config 4: poll-mode kernel (OUT_ANALYSER | OUT_DB_U8): both workgroups of a CU resident
    (12, 0, 32 | 8 | 3, 256, 2),  # ... fused with the push (OUT_PEND)
    (10, 2, 64 | 4 | 3, 128, 4),  # the reference's 256 byte columns per row at N = 1024 (OUT_LFFAST | OUT_LOGFREQ | OUT_DB_U8)
    (14, 5, 0, 128, 4),         # N = 16384, plain rows on the generic loads (variant 5): no scratch under the 128-register cap
]


@pytest.mark.skipif(HIPCC is None, reason="hipcc not installed")
@pytest.mark.parametrize("log2n,shift,out,max_vgpr,min_occ", BUDGETS)
def test_kernel_fits_its_register_budget(log2n, shift, out, max_vgpr, min_occ):
    csrc = os.path.join(ROOT, "spectrogram_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "one.hip")
        with open(src, "w") as f:
            f.write('#include "stft_kernel.hpp"\nnamespace sg { template __global__ void stft_kernel<%d, %d, %d>(const StftParams); }\n'
                    % (log2n, shift, out))
        asm = os.path.join(d, "one.s")
        # the build recipe's own flag list (spectrogram_amd/build.py device_flags): a change there changes what is tested here
        from spectrogram_amd import build as B
        r = subprocess.run([HIPCC] + B.device_flags() + ["-I" + csrc, "--cuda-device-only", "-S", src, "-o", asm],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-3000:]
        text = open(asm).read()

    def field(name):
        m = re.search(r"^; %s: (\d+)" % name, text, re.M)
        assert m, name
        return int(m.group(1))

    vgpr, agpr, scratch, occ = field("NumVgprs"), field("NumAgprs"), field("ScratchSize"), field("Occupancy")
    what = "stft_kernel<%d,%d,%d>: %d VGPRs + %d AGPRs, %d B scratch, %d waves/SIMD" % (log2n, shift, out, vgpr, agpr, scratch, occ)
    assert scratch == 0, what
    assert vgpr + agpr <= max_vgpr and occ >= min_occ, what
