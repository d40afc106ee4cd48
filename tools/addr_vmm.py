#!/usr/bin/env python3
"""Round 6: the config-2 launch is fastest into output allocations whose pages lie scattered in physical memory and slowest into
physically contiguous ones (tools/addr_alloc.py).  Can the caller ask for the scattered kind?  Outputs built with the HIP
virtual-memory API: one virtual range, physical chunks of 2 MiB (or more) created one by one and mapped in order or in a
shuffled order -- against plain hipMalloc.   usage (GPU box): tools/addr_vmm.py"""
import ctypes as C, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spectrogram_amd as S
import bench

FR, BINS, NS = bench.FRAMES, bench.BINS, bench.N_SAMPLES
OUT_B = FR * BINS * 4
dev = torch.device("cuda", 0)
torch.cuda.init()
hip = C.CDLL("libamdhip64.so")
MiB = 1 << 20


class Loc(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class AllocFlags(C.Structure):
    _fields_ = [("compressionType", C.c_ubyte), ("gpuDirectRDMACapable", C.c_ubyte), ("usage", C.c_ushort)]


class Prop(C.Structure):
    _fields_ = [("type", C.c_int), ("requestedHandleType", C.c_int), ("location", Loc), ("win32HandleMetaData", C.c_void_p),
                ("allocFlags", AllocFlags)]


class Access(C.Structure):
    _fields_ = [("location", Loc), ("flags", C.c_int)]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (what, rc))


prop = Prop()
prop.type = 1                 # hipMemAllocationTypePinned
prop.requestedHandleType = 0
prop.location = Loc(1, 0)     # hipMemLocationTypeDevice, device 0
gran_min, gran_rec = C.c_size_t(), C.c_size_t()
ck(hip.hipMemGetAllocationGranularity(C.byref(gran_min), C.byref(prop), 0), "granularity(min)")
ck(hip.hipMemGetAllocationGranularity(C.byref(gran_rec), C.byref(prop), 1), "granularity(recommended)")
print("allocation granularity: minimum %d bytes, recommended %d bytes" % (gran_min.value, gran_rec.value))


def vmm_buffer(size, chunk, shuffle, seed=0):
    chunk = max(chunk, gran_min.value)
    n = (size + chunk - 1) // chunk
    va = C.c_void_p()
    ck(hip.hipMemAddressReserve(C.byref(va), C.c_size_t(n * chunk), C.c_size_t(0), C.c_void_p(0), C.c_ulonglong(0)), "reserve")
    handles = []
    for i in range(n):
        h = C.c_void_p()
        ck(hip.hipMemCreate(C.byref(h), C.c_size_t(chunk), C.byref(prop), C.c_ulonglong(0)), "create")
        handles.append(h)
    order = list(range(n))
    if shuffle:
        random.Random(seed).shuffle(order)
    for slot, i in enumerate(order):
        ck(hip.hipMemMap(C.c_void_p(va.value + slot * chunk), C.c_size_t(chunk), C.c_size_t(0), handles[i], C.c_ulonglong(0)), "map")
    acc = Access(Loc(1, 0), 3)
    ck(hip.hipMemSetAccess(va, C.c_size_t(n * chunk), C.byref(acc), C.c_size_t(1)), "set access")
    return va.value


def plain(size):
    p = C.c_void_p()
    ck(hip.hipMalloc(C.byref(p), C.c_size_t(size)), "hipMalloc")
    return p.value


K = 8
ins = [(torch.rand(1, NS, device=dev) * 2 - 1).contiguous() for _ in range(K)]
eng = S.Engine(fft_size=1024, hop=256)
bufs = []
for i in range(3):
    bufs.append(("hipMalloc exact", plain(OUT_B)))
for i in range(2):
    bufs.append(("hipMalloc 1 GiB", plain(1 << 30)))
for chunk, shuffle in ((2 * MiB, False), (2 * MiB, True), (64 * 1024, True), (64 * 1024, False), (256 * 1024, True), (16 * MiB, True), (64 * MiB, False)):
    try:
        bufs.append(("vmm %5d KiB chunks %s" % (chunk // 1024, "shuffled" if shuffle else "in order"), vmm_buffer(OUT_B, chunk, shuffle, seed=len(bufs))))
    except RuntimeError as e:
        print("vmm %d MiB: %s" % (chunk // MiB, e))
# inputs of the scattered kind as well: does the read side care?
vin = None
try:
    vin = vmm_buffer(NS * 4, 2 * MiB, True, seed=99)
    ck(hip.hipMemcpy(C.c_void_p(vin), C.c_void_p(ins[0].data_ptr()), C.c_size_t(NS * 4), 3), "copy")
except RuntimeError as e:
    print("vmm input: %s" % e)
stream = torch.cuda.current_stream(dev).cuda_stream


def launch(i, ptr, inp=None):
    x = ins[i % K].data_ptr() if inp is None else inp
    eng.stft_device_ptr(x, 1, NS, NS, 1, ptr, FR * BINS, stream)


for i in range(900):
    launch(i, bufs[i % len(bufs)][1])
torch.cuda.synchronize()
# correctness: the same input into a plain and into a shuffled buffer gives the same rows
ref = torch.empty(1, FR, BINS, device=dev)
eng.stft(ins[0], out=ref)
for label, p in bufs:
    if label.startswith("vmm"):
        launch(0, p)
        got = torch.empty(1, FR, BINS, device=dev)
        torch.cuda.synchronize()
        ck(hip.hipMemcpy(C.c_void_p(got.data_ptr()), C.c_void_p(p), C.c_size_t(OUT_B), 3), "copy back")
        assert torch.equal(got, ref), label
REPS = 24
n = len(bufs) * REPS
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    launch(i, bufs[i % len(bufs)][1])
    ev[i + 1].record()
torch.cuda.synchronize()
acc = [[] for _ in bufs]
for i in range(n):
    acc[i % len(bufs)].append(ev[i].elapsed_time(ev[i + 1]) * 1e3)
for (label, p), a in zip(bufs, acc):
    a.sort()
    print("%-30s %#x   median %.1f us  (min %.1f)" % (label, p, a[len(a) // 2], a[0]))
eng.close()
