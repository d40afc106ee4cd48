"""CPU: the partitions of spectrogram_amd/shard.py and chunked.py as properties over arbitrary sizes.  hypothesis is a test
dependency of this module only: on a box without it these sweeps are skipped and tests/test_shard.py still runs."""
import numpy as np
import pytest

pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

from spectrogram_amd import chunked, shard  # noqa: E402


@settings(max_examples=300, deadline=None)
@given(n_samples=st.integers(0, 40_000_000), log2n=st.integers(5, 15), hop=st.integers(1, 70_000), world=st.integers(1, 16))
def test_frame_split_properties(n_samples, log2n, hop, world):
    """Whatever the sizes: the ranks' frame ranges tile [0, frames) in rank order, differ by at most one frame, and each
    rank's sample range (a) starts at its first frame, (b) stays inside the stream, (c) yields exactly its own frames
    when transformed on its own -- so concatenating the ranks' rows is the unsharded output."""
    n = 1 << log2n
    frames = shard.num_frames(n_samples, n, hop)
    parts = [shard.split_frames(n_samples, n, hop, world, r) for r in range(world)]
    assert sum(p.frame_count for p in parts) == frames
    counts = [p.frame_count for p in parts]
    assert max(counts) - min(counts) <= 1 and counts == sorted(counts, reverse=True)
    nxt = 0
    for p in parts:
        if p.frame_count == 0:
            assert p.sample_count == 0
            continue
        assert p.frame_start == nxt and p.sample_start == nxt * hop
        nxt += p.frame_count
        assert p.sample_start + p.sample_count <= n_samples
        assert shard.num_frames(p.sample_count, n, hop) == p.frame_count
        # the last frame of the shard ends exactly where its sample range ends
        assert (p.frame_start + p.frame_count - 1) * hop + n == p.sample_start + p.sample_count


@settings(max_examples=300, deadline=None)
@given(total=st.integers(0, 5_000_000), world=st.integers(1, 64))
def test_channel_split_properties(total, world):
    parts = [shard.split_channels(total, world, r) for r in range(world)]
    assert parts[0].start == 0 and sum(p.count for p in parts) == total
    for a, b in zip(parts, parts[1:]):
        assert b.start == a.start + a.count and 0 <= a.count - b.count <= 1
    with pytest.raises(ValueError):
        shard.split_channels(total, world, world)


@settings(max_examples=200, deadline=None)
@given(n_utts=st.integers(0, 2_000_000), world=st.integers(1, 16), chunk=st.integers(1, 100_000))
def test_chunk_plan_properties(n_utts, world, chunk):
    from spectrogram_amd import chunked
    nxt = 0
    for rank in range(world):
        own = shard.split_channels(n_utts, world, rank)
        plan = chunked.plan_chunks(n_utts, world, rank, chunk)
        assert sum(c.utt_count for c in plan) == own.count
        for i, c in enumerate(plan):
            assert c.index == i and c.utt_start == nxt and 1 <= c.utt_count <= chunk
            assert c.utt_count == chunk or i == len(plan) - 1
            nxt += c.utt_count
    assert nxt == n_utts
