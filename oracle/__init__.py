"""CPU oracle for the STFT hot path -- TEST INFRASTRUCTURE ONLY (parity unpinned).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  See oracle/stft_oracle.h for what it restates and why.
"""
from .oracle import (  # noqa: F401
    OracleConfig, build, lib, window, fft_radix2, num_frames, stft, stft_f32, stft_f32js,
    analyser_poll, time_domain_bytes, logfreq_resample,
)
