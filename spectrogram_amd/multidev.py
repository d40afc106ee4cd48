"""Several GPUs from ONE process: one host thread, one handle and one HIP stream per device.

The multi-process form (one rank per GPU, bench.py / torch.distributed) is the deployment shape; this is the
single-process alternative SURVEY.md section 7.7 names for callers that cannot fork, e.g. an addon living inside
one node process.  Channels are split into contiguous blocks (shard.split_channels); every block is transformed
on its device by its own thread (ctypes releases the GIL during the call); there is no exchange between devices
and results stay where they were produced.  The C ABI makes this safe by construction: a handle is bound to its
device (sg_config.device_id), every entry point switches to that device for the duration of the call and
restores the caller's, and distinct handles share no state.
"""
import threading

from . import shard
from .engine import Engine


class MultiDeviceEngine:
    def __init__(self, devices, **engine_kw):
        """devices: HIP ordinals, one worker each (an ordinal may repeat: two workers on one GPU)."""
        self.devices = list(devices)
        if not self.devices:
            raise ValueError("need at least one device")
        self.engines = [Engine(device_id=d, **engine_kw) for d in self.devices]

    def close(self):
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def stft(self, pcm_host):
        """pcm_host: CPU float32 (or int16) torch tensor [channels, samples].  Returns a list of
        (ChannelShard, rows) with rows a CUDA tensor [count, frames, bins] on the worker's device."""
        import torch
        n_ch = pcm_host.shape[0]
        world = len(self.devices)
        results = [None] * world
        errors = []

        def work(i):
            try:
                sh = shard.split_channels(n_ch, world, i)
                if sh.count == 0:
                    results[i] = (sh, None)
                    return
                dev = torch.device("cuda", self.devices[i])
                stream = torch.cuda.Stream(dev)
                with torch.cuda.stream(stream):
                    x = pcm_host[sh.start:sh.start + sh.count].to(dev, non_blocking=True)
                    rows = self.engines[i].stft(x)
                stream.synchronize()
                results[i] = (sh, rows)
            except Exception as e:  # surfaced in the caller's thread
                errors.append(e)

        threads = [threading.Thread(target=work, args=(i,)) for i in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results
