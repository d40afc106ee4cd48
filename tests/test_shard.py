"""CPU: the multi-GPU path is a partition with no collective.  Checks the partition arithmetic
and, with two gloo ranks, that per-rank results (computed by the oracle standing in for a GPU)
reassemble to the unsharded answer exactly."""
import os
import socket

import numpy as np
import pytest

from spectrogram_amd import shard


def test_channel_split_covers_everything():
    for n, w in ((2048, 8), (7, 3), (3, 8), (1000000, 8), (0, 4)):
        parts = [shard.split_channels(n, w, r) for r in range(w)]
        assert sum(p.count for p in parts) == n
        pos = 0
        for p in parts:
            assert p.start == pos
            pos += p.count
        assert max(p.count for p in parts) - min(p.count for p in parts) <= 1


@pytest.mark.parametrize("n_samples,n,hop", [(28_800_000, 1024, 256), (160_000, 1024, 256), (5000, 2048, 512),
                                             (1023, 1024, 256), (1024, 1024, 256)])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_frame_split_with_halo(n_samples, n, hop, world):
    frames = shard.num_frames(n_samples, n, hop)
    parts = [shard.split_frames(n_samples, n, hop, world, r) for r in range(world)]
    assert sum(p.frame_count for p in parts) == frames
    nxt = 0
    for p in parts:
        if p.frame_count == 0:
            continue
        assert p.frame_start == nxt
        nxt += p.frame_count
        assert p.sample_start == p.frame_start * hop
        assert p.sample_start + p.sample_count <= n_samples
        # the shard's own frame count, computed from its sample range, is exactly its share
        assert shard.num_frames(p.sample_count, n, hop) == p.frame_count
    # config 2 at 8 GPUs: the halo is N - hop samples per cut
    if n_samples == 28_800_000 and world == 8:
        read = sum(p.sample_count for p in parts)
        assert read - n_samples == 7 * (n - hop)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, hop = 256, 64
    cfg = O.OracleConfig.make(n, hop)
    rng = np.random.default_rng(11)
    long_x = rng.uniform(-1, 1, 5000).astype(np.float32)          # one long stream
    many = rng.uniform(-1, 1, (5, 1200)).astype(np.float32)       # a batch of channels
    fs = shard.split_frames(len(long_x), n, hop, world, rank)
    a, _ = O.stft(cfg, long_x[fs.sample_start:fs.sample_start + fs.sample_count])
    cs = shard.split_channels(many.shape[0], world, rank)
    b = [O.stft(cfg, many[c])[0] for c in range(cs.start, cs.start + cs.count)]
    # no collective on the data path; the test gathers only to compare
    gathered = [None] * world
    dist.all_gather_object(gathered, (fs, a, cs, b))
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


def test_two_rank_gloo_reassembly():
    import torch.multiprocessing as mp
    import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    n, hop = 256, 64
    cfg = O.OracleConfig.make(n, hop)
    rng = np.random.default_rng(11)
    long_x = rng.uniform(-1, 1, 5000).astype(np.float32)
    many = rng.uniform(-1, 1, (5, 1200)).astype(np.float32)
    full, _ = O.stft(cfg, long_x)
    joined = np.concatenate([g[1] for g in gathered], axis=0)
    assert np.array_equal(joined, full)            # frame shards with halo: bit-identical
    chans = [row for g in gathered for row in g[3]]
    assert len(chans) == 5
    for c in range(5):
        assert np.array_equal(chans[c], O.stft(cfg, many[c])[0])


# ---- chunk plan of the config-5 runner (BASELINE.json configs[4]) ------------------------------

@pytest.mark.parametrize("n_utts,world,chunk", [(1_000_000, 8, 7813), (1_000_000, 1, 62_500), (1003, 4, 100),
                                                (7, 8, 3), (0, 2, 5)])
def test_chunk_plan_covers_every_utterance_once(n_utts, world, chunk):
    from spectrogram_amd import chunked
    seen = 0
    nxt = 0
    for rank in range(world):
        plan = chunked.plan_chunks(n_utts, world, rank, chunk)
        for i, ch in enumerate(plan):
            assert ch.index == i and 1 <= ch.utt_count <= chunk
            assert ch.utt_start == nxt            # ranks own contiguous blocks, chunks walk them in order
            nxt += ch.utt_count
            seen += ch.utt_count
        if plan:
            assert all(c.utt_count == chunk for c in plan[:-1])
    assert seen == n_utts
    # config 5 at 8 GPUs: 125,000 utterances = 16 chunks per rank
    if n_utts == 1_000_000 and world == 8:
        assert len(chunked.plan_chunks(n_utts, 8, 3, chunk)) == 16


def _chunk_worker(rank, world, port, q):
    """Each rank walks ITS chunks with the CPU oracle standing in for the engine; nothing is exchanged on
    the data path, the gather below exists only so that the test can compare."""
    import torch.distributed as dist
    import oracle as O
    from spectrogram_amd import chunked
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, hop, n_utts, utt_samples = 128, 32, 23, 700
    cfg = O.OracleConfig.make(n, hop)
    mine = []
    for ch in chunked.plan_chunks(n_utts, world, rank, 4):
        for u in range(ch.utt_start, ch.utt_start + ch.utt_count):
            x = np.random.default_rng(0x5EED0005 + u).uniform(-1, 1, utt_samples).astype(np.float32)  # seed + utterance id
            mine.append((u, O.stft(cfg, x)[0]))
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    dist.barrier()
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


def test_two_rank_gloo_chunked_plan():
    import torch.multiprocessing as mp
    import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_chunk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered = q.get()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rows = [r for g in gathered for r in g]
    assert [u for u, _ in rows] == list(range(23))       # rank order = utterance order, none twice, none missing
    cfg = O.OracleConfig.make(128, 32)
    for u, got in rows[::5]:
        x = np.random.default_rng(0x5EED0005 + u).uniform(-1, 1, 700).astype(np.float32)
        assert np.array_equal(got, O.stft(cfg, x)[0])


# ---- bench.py's collectives at world size 2 and 8 (gloo, CPU): the N-rank line's value and per-rank figures ------------

def _bench_worker(rank, world, port, q):
    import importlib.util
    import time
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SG_BENCH_DIST_TIMEOUT_S="120")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    # the same join bench.py's main() performs (deadline, one all-reduce of ones that must count `world` ranks)
    b.join_process_group(torch, dist, True, None, rank, world)

    class FakeCuda:                       # no GPU here: synchronize is the only torch.cuda call these helpers make
        @staticmethod
        def synchronize():
            pass

    class T:                              # torch, with cuda replaced
        cuda = FakeCuda
        tensor = staticmethod(torch.tensor)
        zeros_like = staticmethod(torch.zeros_like)
        float64 = torch.float64

    barrier, max_over_ranks = b.make_collectives(T, dist, True, True, None, world)
    ident = max_over_ranks.gather([float(rank % 3), 0.0, float(5 + rank % 3), 0.0])   # (device, PCI domain / bus / device) per rank
    busy = 0.02 * (rank + 1)              # the last rank is the slow one
    dt = b.timed_region(lambda: time.sleep(busy), barrier, max_over_ranks, T)
    rows = max_over_ranks.gather([100.0 + rank, b.timed_region.own_dt])
    if rank == 0:
        q.put((dt, rows, ident, dist.get_world_size()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_collectives_gloo(world):
    """`value` of an N-rank line is computed from the SLOWEST rank's own interval (no collective inside it), and every
    rank's own figures reach rank 0 in rank order -- so that a bent scaling curve can be attributed to a rank.  World size 8
    is the driver's scaling run, which no builder box can rehearse on devices (one GPU, at most six processes on it): its
    rendezvous, gathers and max-reduce run here over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    try:
        dt, rows, ident, seen = q.get(timeout=240)     # a rank that died before its put() fails the test instead of hanging it
    except queue.Empty:
        for p in procs:
            p.kill()
        raise AssertionError("rank 0 never reported: exit codes %r" % [p.exitcode for p in procs])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert seen == world and len(rows) == world == len(ident)
    assert [r[0] for r in rows] == [100.0 + r for r in range(world)]   # rank order
    assert [v[0] for v in ident] == [float(r % 3) for r in range(world)]
    own = [r[1] for r in rows]
    # each rank's own sleep (20 ms x (rank + 1)), not another's: lower bounds only -- on a quota-throttled box a sleep may
    # overrun by tens of ms (cgroup cpu.max stalls), so the upper bounds are generous
    assert all(own[r] >= 0.02 * (r + 1) for r in range(world)) and max(own) < 5.0, own
    assert dt == max(own)                                          # the line's interval is the slowest rank's


def test_bench_rank_that_cannot_join_exits_nonzero_with_a_reason():
    """a rendezvous that never completes must end as rc != 0 and one line on stderr within the
    stated limit -- here a lone rank 0 of a two-rank world whose peer never starts."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import importlib.util, os, sys, torch, torch.distributed as dist\n"
            "spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(%r, 'bench.py'))\n"
            "b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "b.join_process_group(torch, dist, True, None, 0, 2)\n"
            "print('joined')\n" % root)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="2",
               SG_BENCH_DIST_TIMEOUT_S="4")
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode in (3, 4) and "joined" not in r.stdout, (r.returncode, r.stdout, r.stderr[-500:])
    assert "did not join the process group" in r.stderr and "nothing was measured" in r.stderr
    assert time.time() - t0 < 90
