"""Sharding of the STFT path across the GPUs of a node: partitions only, no collective.

Frames and channels are independent, so each rank (one process per GPU)
computes its own slice and results stay on the producing device.  The reference has no analogue
(one browser tab, one AnalyserNode: UI/player.js:7).

  * batched configs (BASELINE.json configs 3 and 5): contiguous blocks of channels per rank;
  * one long stream (config 2): contiguous frame ranges, each rank reading its samples plus an
    (N - hop)-sample halo, so that concatenating the ranks' rows reproduces the unsharded output
    bit for bit.
"""
from collections import namedtuple

ChannelShard = namedtuple("ChannelShard", "start count")
FrameShard = namedtuple("FrameShard", "frame_start frame_count sample_start sample_count")


def _block(total, world, rank):
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def split_channels(n_channels, world, rank):
    """Channels [start, start+count) for this rank; the first (n % world) ranks get one extra."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return ChannelShard(*_block(n_channels, world, rank))


def num_frames(n_samples, fft_size, hop):
    return 0 if n_samples < fft_size else (n_samples - fft_size) // hop + 1


def split_frames(n_samples, fft_size, hop, world, rank):
    """Frame range of one long stream for this rank and the sample range (with halo) it must read."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    frames = num_frames(n_samples, fft_size, hop)
    f0, nf = _block(frames, world, rank)
    if nf == 0:
        return FrameShard(f0, 0, 0, 0)
    return FrameShard(f0, nf, f0 * hop, (nf - 1) * hop + fft_size)


def stft_sharded(engine, pcm, world, rank):
    """Run this rank's share of a [channels, samples] device tensor through `engine`.

    More than one channel: channel blocks.  A single channel: frame ranges with halo.
    Returns (rows, meta): the local output tensor and the ChannelShard / FrameShard it covers.
    """
    if pcm.dim() != 2:
        raise ValueError("expected [channels, samples]")
    n_ch, n_samples = pcm.shape
    if n_ch > 1:
        sh = split_channels(n_ch, world, rank)
        if sh.count == 0:
            return None, sh
        return engine.stft(pcm[sh.start:sh.start + sh.count]), sh
    sh = split_frames(n_samples, engine.fft_size, engine.cfg.hop, world, rank)
    if sh.frame_count == 0:
        return None, sh
    return engine.stft(pcm[:, sh.sample_start:sh.sample_start + sh.sample_count]), sh
