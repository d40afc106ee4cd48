"""Chunked runner for batches that do not fit as one resident call (BASELINE.json configs[4]:
1e6 x 10 s mono 16 kHz utterances = 640 GB in, 1.27 TB out, sharded over the GPUs of a node).

Each rank (one process per GPU) owns a contiguous block of utterances (shard.split_channels) and
walks it in chunks of `chunk_utts`: the H2D copy of chunk i+1 runs on a copy stream while the
STFT of chunk i runs on the compute stream (two device input buffers, events both ways).  No
collective: utterances are independent, results stay on the producing device for the consumer
(`sink`) to use before the next chunk overwrites the output buffer.  The reference analogue is the
draw loop run N times (UI/spectrogram.js:153-161); nothing there is chunked.
"""
from collections import namedtuple

from . import shard

Chunk = namedtuple("Chunk", "index utt_start utt_count")


def plan_chunks(n_utts, world, rank, chunk_utts):
    """This rank's chunks: its channel block cut into pieces of at most chunk_utts utterances."""
    if chunk_utts < 1:
        raise ValueError("chunk_utts must be >= 1")
    sh = shard.split_channels(n_utts, world, rank)
    out, done = [], 0
    while done < sh.count:
        c = min(chunk_utts, sh.count - done)
        out.append(Chunk(len(out), sh.start + done, c))
        done += c
    return out


class ChunkedRunner:
    """Double-buffered chunk loop around one Engine.

    source(chunk) -> tensor [utt_count, utt_samples] of the engine's input dtype: a CUDA tensor is
    used in place (device-resident data), a CPU tensor (ideally pinned) is copied H2D on the copy
    stream into one of the two staging buffers.  sink(chunk, rows) is called on the compute stream's
    timeline with rows = [utt_count, frames, bins] (valid until the next chunk is enqueued).
    """

    def __init__(self, engine, utt_samples, chunk_utts, device):
        import torch
        self.torch = torch
        self.e = engine
        self.utt_samples = int(utt_samples)
        self.chunk_utts = int(chunk_utts)
        self.device = device
        self.frames = engine.num_frames(utt_samples)
        if engine.cfg.pcm_format == 2:
            raise NotImplementedError("packed int24 utterances: convert to int16 / float32 chunks first (rows of 3-byte "
                                      "samples have no [utterances, samples] staging tensor here)")
        in_dt = {1: torch.int16}.get(engine.cfg.pcm_format, torch.float32)
        out_dt = torch.uint8 if engine.out_dtype.__name__ == "uint8" else torch.float32
        self.stage = [torch.empty((self.chunk_utts, self.utt_samples), dtype=in_dt, device=device) for _ in range(2)]
        self.out = torch.empty((self.chunk_utts, self.frames, engine.bins), dtype=out_dt, device=device)
        self.copy_stream = torch.cuda.Stream(device)
        self.compute_stream = torch.cuda.Stream(device)
        self.ready = [torch.cuda.Event() for _ in range(2)]   # staging buffer b holds its chunk
        self.free = [torch.cuda.Event() for _ in range(2)]    # staging buffer b has been consumed

    def run(self, source, chunks, sink=None):
        """Process the chunks in order; returns the number of frames produced."""
        torch = self.torch
        total = 0
        pending = None
        # device-resident sources were produced on the caller's current stream: order the private streams behind it
        # (and tell the caching allocator about them), so that run() needs no synchronize() from the caller
        cur = torch.cuda.current_stream(self.device)
        self.compute_stream.wait_stream(cur)
        self.copy_stream.wait_stream(cur)

        def fetch(i):
            ch = chunks[i]
            t = source(ch)
            if t.is_cuda:
                t.record_stream(self.compute_stream)
                return t, None
            b = i & 1
            with torch.cuda.stream(self.copy_stream):
                self.copy_stream.wait_event(self.free[b])          # the compute that read this buffer is done
                self.stage[b][:ch.utt_count].copy_(t, non_blocking=True)
                self.ready[b].record(self.copy_stream)
            return self.stage[b][:ch.utt_count], b

        if chunks:
            pending = fetch(0)
        for i, ch in enumerate(chunks):
            x, b = pending
            pending = fetch(i + 1) if i + 1 < len(chunks) else None  # next copy overlaps this compute
            with torch.cuda.stream(self.compute_stream):
                if b is not None:
                    self.compute_stream.wait_event(self.ready[b])
                rows = self.e.stft(x, out=self.out)[:ch.utt_count]
                if b is not None:
                    self.free[b].record(self.compute_stream)
                if sink is not None:
                    sink(ch, rows)
            total += ch.utt_count * self.frames
        self.compute_stream.synchronize()
        self.copy_stream.synchronize()
        return total
