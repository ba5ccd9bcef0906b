"""Headline benchmark: compress + decompress throughput (GB/s of raw int16) of the chunked delta + DEFLATE
hot path on MI355X, inputs and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over the workload: compress every chunk (K1 delta+transpose ->
bit-exact zlib level-6 DEFLATE) and decompress it again (INFLATE -> K2 cumsum+transpose).  Workload at
N=1: BASELINE.json configs[1] (385 ch @ 30 kHz, 60 s, 1 s chunks, level 6).  With N ranks the chunks of an
N x 60 s recording are sharded round-robin (chunk i -> rank i mod N, no data-path collective; only the
compressed sizes are gathered), so per-GPU work is fixed: weak scaling.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process -- before it has made any HIP call --
starts the N ranks itself (`python -m torch.distributed.run --nproc-per-node N bench.py ...`, the command the driver uses),
relays rank 0's JSON line and exits with the child's status; fewer than N visible devices is an error, never a silent
downgrade.  (`--oversubscribe` lets the N ranks share the visible devices: a smoke test of the multi-rank path on a one-GPU
box, not a measurement.)

The path has no data-path collective (chunks are independent zlib streams): the ranks' only exchange is the host-side gather of
the compressed sizes, so the process group is gloo on CPU tensors by default and torch is used for torch.distributed alone --
device memory, copies and waits go through libmtscomp_hip.so (mts_dev_alloc / mts_dev_copy / mts_dev_sync): ONE HIP runtime per
process, whatever the import order.  `--dist-backend nccl` additionally brings up an RCCL group (init + one all-reduce outside
the timed region, under a watchdog that exits non-zero if it hangs); the size exchange stays on the host group.

The JSON line carries, next to the contract's keys:
  roofline             the dominant kernel (k_match5): algorithmic bytes (R + C of the batch) / its launch time
  roofline_compress    (R + C) / time of the whole compress direction, roofline_decompress likewise (C + R)
  cpu_baseline         the reference's ThreadPool path restated (numpy + stdlib zlib: the oracle) on ALL host cores and on 1
  byte_identical_chunks  how many of the recording's chunks are byte for byte what that CPU path produced (+ the .cbin sha1s)
  extras (N = 1 only, after the timed region; --no-extras skips them)
    file_to_file       the drop-in calls compress() / decompress(out=...) on the 60 s recording as a file on tmpfs (PCIe, SHA-1
                       and file I/O included; check_after_* off like the reference's benchmark.py:29,36), next to the CPU port's
    random_read        BASELINE configs[2]: a 600 s file, 1000 windows of 1 s at splitmix(i) starts through Reader[a:b]
    level_sweep        BASELINE configs[4] shape (1024 ch, 0.25 s chunks) at levels 1, 2, 3, 6, 9: ratio and GB/s
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured copy rate
RATE = 30000
CONFIG3_CHUNKS_PER_RANK = 450     # BASELINE configs[3]: 385 ch, 3600 s, chunks round robin over 8 GPUs
STRESS_CHUNKS = 2400              # BASELINE configs[4]: 1024 ch, 600 s in chunks of 0.25 s
STRESS_ROWS = 7500


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--seconds', type=int, default=0,
                   help='recording length per GPU in chunks of 1 s (default: 60 at N = 1 = BASELINE configs[1]; %d at N > 1 = one '
                        'rank\'s shard of configs[3], 3600 s over 8 GPUs)' % CONFIG3_CHUNKS_PER_RANK)
    p.add_argument('--config', default='headline', choices=('headline', 'stress'),
                   help='headline: 385 ch, 1 s chunks, level 6 (configs[1] / configs[3]); stress: BASELINE configs[4], 1024 ch, 0.25 s chunks, '
                        '600 s over the ranks, timed at level 6, levels 1 and 9 measured beside it')
    p.add_argument('--channels', type=int, default=0, help='default: 385 (headline), 1024 (stress)')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-chunks', type=int, default=0, help='chunks in the all-cores CPU sample (default: one per host cpu)')
    p.add_argument('--no-extras', action='store_true', help='skip the configs[2] / configs[4] measurements')
    p.add_argument('--extras-seconds', type=int, default=600, help='length of the random-read file (configs[2])')
    p.add_argument('--dist-backend', default='gloo', choices=('nccl', 'gloo'),
                   help='gloo (default): the host-side size exchange is the path\'s only exchange; nccl: also bring up an RCCL group (init check)')
    p.add_argument('--init-timeout', type=int, default=180, help='seconds the process group may take to come up before the rank exits non-zero')
    p.add_argument('--oversubscribe', action='store_true',
                   help='let ranks share devices (rank -> device rank mod visible): smoke test of the N > 1 path on fewer GPUs, not a measurement')
    p.add_argument('--master-port', type=int, default=0, help='rendezvous port of the self-launched ranks (default: a free one)')
    a = p.parse_args(argv)
    resolve_workload(a)
    return a


def resolve_workload(a):
    """Fills in what the flags left open: channels, rows per chunk, chunks per GPU and the sentence that names the workload.
    N = 1: BASELINE configs[1] (60 chunks).  N > 1: every rank owns what a rank of configs[3] owns on 8 GPUs (450 chunks: at N = 8
    the job IS configs[3], 3600 s; at N = 2 / 4 it is 900 / 1800 s: weak scaling, per-GPU work fixed).  --config stress: configs[4],
    its 2400 chunks dealt over the ranks (one rank alone takes an 8-GPU rank's 300)."""
    n = max(a.gpus, 1)
    if a.config == 'stress':
        a.channels = a.channels or 1024
        a.chunk_rows = STRESS_ROWS
        a.n_chunks = a.seconds or (STRESS_CHUNKS // n if n > 1 else STRESS_CHUNKS // 8)
        a.levels_beside = (1, 9)
        a.workload = ('%d ch @ 30 kHz, chunk = 0.25 s (%d samples), %d chunks per GPU (%d GPUs: %.0f s of the 600 s of BASELINE configs[4]), '
                      'timed at zlib level 6; levels 1 and 9 beside it (level_curve)' % (a.channels, STRESS_ROWS, a.n_chunks, n, a.n_chunks * n * 0.25))
    else:
        a.channels = a.channels or 385
        a.chunk_rows = RATE
        a.n_chunks = a.seconds or (60 if n == 1 else CONFIG3_CHUNKS_PER_RANK)
        a.levels_beside = ()
        which = 'BASELINE configs[1]' if n == 1 and a.n_chunks == 60 else \
            'BASELINE configs[3]: 3600 s over 8 GPUs' if n == 8 and a.n_chunks == CONFIG3_CHUNKS_PER_RANK else \
            "a rank's shard of BASELINE configs[3] (450 chunks) on every GPU: %d s in all" % (a.n_chunks * n) if a.n_chunks == CONFIG3_CHUNKS_PER_RANK else \
            'BASELINE configs[1] shape, %d s per GPU' % a.n_chunks
        a.workload = '%d ch @ 30 kHz, %d s synthetic AR int16 per GPU (%d chunks per rank), chunk=1 s, zlib level 6 (%s)' % (a.channels, a.n_chunks, a.n_chunks, which)


# ------------------------------------------------------------------------------------------------
# the N > 1 path: ranks, shards, the one exchange (tests/test_distributed.py drives these same functions under gloo)
# ------------------------------------------------------------------------------------------------
def shard_ids(rank, world, n_total):
    """Global chunk ids a rank owns: chunk i -> rank i mod N (the reference's batches, mtscomp.py:399-423)."""
    return list(range(rank, n_total, world))


def gather_chunk_offsets(local_sizes, rank, world, dist=None, group=None):
    """The only cross-rank step of the path (mtscomp.py:474-483): every rank's compressed sizes -> chunk_offsets of the whole
    recording (exclusive prefix sum in global chunk order i = rank + k * world).  All ranks own the same number of chunks.
    A host-side gather (CPU tensors over `group`, a gloo group): no device collective anywhere on the path."""
    local_sizes = np.asarray(local_sizes, dtype=np.int64)
    if world == 1:
        return np.concatenate(([0], np.cumsum(local_sizes)))
    import torch
    t = torch.from_numpy(np.ascontiguousarray(local_sizes))
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    allsz = torch.stack(out, dim=1).reshape(-1).numpy()          # [k][rank] -> global order
    return np.concatenate(([0], np.cumsum(allsz)))


def visible_gpus():
    """Devices torch sees, WITHOUT initialising the GPU in this process (device_count() does not, on this image)."""
    import torch
    return int(torch.cuda.device_count())


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def launch_ranks(n, script, script_args, port=0, env=None, timeout=None, capture=False):
    """Start `n` ranks of `script` on this node the way the driver does (one process per GPU, torch.distributed.run,
    rendezvous on 127.0.0.1) from a process that has NOT touched the GPU; relays the children's output and returns their
    exit status (capture=True: returns (status, stdout, stderr) instead)."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port or free_port()), str(script)] + list(script_args)
    e = dict(os.environ if env is None else env)
    e.setdefault('MASTER_ADDR', '127.0.0.1')
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if capture:
        r = subprocess.run(cmd, env=e, timeout=timeout, capture_output=True, text=True)
        return r.returncode, r.stdout, r.stderr
    return subprocess.run(cmd, env=e, timeout=timeout).returncode


def self_launch(args, argv):
    """`python bench.py --gpus N` (N > 1, not yet under torch.distributed): become the launcher."""
    have = visible_gpus()
    if have < args.gpus and not args.oversubscribe:
        sys.stderr.write('bench.py: --gpus %d asked for, %d device(s) visible; refusing to measure fewer GPUs than asked '
                         '(--oversubscribe shares devices between ranks for a smoke test)\n' % (args.gpus, have))
        return 2
    if have < 1:
        sys.stderr.write('bench.py: no GPU visible\n')
        return 2
    return launch_ranks(args.gpus, Path(__file__).resolve(), argv, port=args.master_port)


def cpu_baseline(x, nc, n_have, n_sample):
    """The reference's ThreadPool path restated on numpy + stdlib zlib (oracle.ref_*), timed on this box's host cores
    over a bounded sample of the benchmarked recording: one chunk per host cpu and at least the recording's own chunks
    (cycled), all of them in flight at once like the reference's batch of n_threads chunks; and one core alone on two chunks.
    Returns the record and the CPU path's compressed chunks 0 .. n_have-1 (the identity check of the GPU output)."""
    from multiprocessing.dummy import Pool as ThreadPool
    import zlib
    from oracle import oracle as O
    ncpu = os.cpu_count() or 1
    n_sample = n_sample or max(ncpu, n_have)
    cores = min(ncpu, n_sample)
    chunk = lambda i: x[(i % n_have) * RATE:(i % n_have + 1) * RATE]  # noqa: E731
    ids = list(range(n_sample))
    with ThreadPool(cores) as pool:
        t0 = time.perf_counter()
        cc = pool.map(lambda i: O.ref_compress_chunk(chunk(i)), ids)
        t1 = time.perf_counter()
        back = pool.map(lambda i: O.ref_decompress_chunk(cc[i], RATE, nc, 'int16'), ids)
        t2 = time.perf_counter()
    assert all(np.array_equal(back[i], chunk(i)) for i in (0, n_sample - 1))
    del back
    gb = n_sample * RATE * nc * 2 / 1e9
    # one core
    m1 = 2
    a0 = time.perf_counter()
    c1 = [O.ref_compress_chunk(chunk(i)) for i in range(m1)]
    a1 = time.perf_counter()
    b1 = [O.ref_decompress_chunk(c1[i], RATE, nc, 'int16') for i in range(m1)]
    a2 = time.perf_counter()
    assert np.array_equal(b1[0], chunk(0))
    gb1 = m1 * RATE * nc * 2 / 1e9
    return {
        'value': gb / (t2 - t0), 'unit': 'GB/s', 'cores': cores, 'kind': 'port',
        'sample': '%d chunks (%d ch x %d samples int16 each, %.0f MB; the recording\'s %d chunks cycled), numpy diff/tobytes + stdlib '
                  'zlib %s level 6, ThreadPool(%d) of %d host cpus, every chunk in flight at once: compress %.3f GB/s, decompress %.3f GB/s '
                  '(the threads share the GIL for the numpy part, as the reference\'s ThreadPool does: %.1f MB/s per thread against %.1f for one core alone); '
                  'one core on %d chunks: compress %.4f GB/s, decompress %.3f GB/s'
                  % (n_sample, nc, RATE, gb * 1e3, n_have, zlib.ZLIB_RUNTIME_VERSION, cores, ncpu, gb / (t1 - t0), gb / (t2 - t1),
                     gb / (t1 - t0) / cores * 1e3, gb1 / (a1 - a0) * 1e3, m1, gb1 / (a1 - a0), gb1 / (a2 - a1)),
        'compress_gbps': gb / (t1 - t0), 'decompress_gbps': gb / (t2 - t1),
        'one_core': {'value': gb1 / (a2 - a0), 'compress_gbps': gb1 / (a1 - a0), 'decompress_gbps': gb1 / (a2 - a1), 'cores': 1, 'chunks': m1},
    }, (cc[:n_have] if n_sample >= n_have else None)


# stage of the library's timing hooks -> the kernel that stage is (one launch per step)
STAGE_KERNEL = {'match': 'k_match5', 'hash_sort': 'k_hash_sort', 'delta_transpose': 'k_delta_rows', 'block_pack': 'k_block_pack',
                'block_trees': 'k_block_trees'}


def measured_traffic(kernel, n_chunks, nc):
    """HBM bytes per launch of `kernel` from the committed PMC passes of the newest profile (profiles/rN_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this script -- a counter pass cannot run inside the timed
    process).  Only valid for the profiled workload (60 chunks x 385 ch) and for a kernel the profile lists; null otherwise."""
    if n_chunks != 60 or nc != 385:
        return None, None
    for name in sorted((q.name for q in (ROOT / 'profiles').glob('r*_traffic.json')), key=lambda n: -int(n[1:].split('_')[0])):      # newest round first
        p = ROOT / 'profiles' / name
        d = json.loads(p.read_text())
        per = d.get('kernels', {d.get('kernel'): d.get('traffic_bytes_per_launch')})
        if per.get(kernel) is not None:
            return per[kernel], 'profiles/' + name
        return None, None                     # the newest profile does not know this kernel: say nothing rather than something stale
    return None, None


def splitmix(i):
    m = (1 << 64) - 1
    z = (i + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def build_synth_file(hip, dev, seconds, tmp, nc=385):
    """A `seconds` s recording of the synthetic generator as data.cbin + data.ch under `tmp`, compressed on the device a minute
    at a time (the raw file is never materialised).  Returns (n_samples, compressed bytes)."""
    piece = 60
    cb = (hip.compress_bound(RATE * nc * 2) + 255) // 256 * 256
    raw = hip.DevBuffer(piece * RATE * nc * 2, dev)
    cbuf = hip.DevBuffer(piece * cb, dev)
    flags = hip.make_flags(True, False, 'F')
    offsets = [0]
    with open(tmp / 'data.cbin', 'wb') as f:
        for p0 in range(0, seconds, piece):
            n = min(piece, seconds - p0)
            hip.dev_synth_int16(raw, 0, p0 * RATE, (p0 + n) * RATE, nc, 0)
            bounds = np.arange(n + 1, dtype=np.int64) * RATE
            slots = np.arange(n, dtype=np.int64) * cb
            sizes = np.zeros(n, dtype=np.int64)
            hip.dev_compress_chunks(raw, nc, 2, bounds, flags, 6, cbuf, slots, sizes)
            host = cbuf.download(0, n * cb)
            for k in range(n):
                f.write(host[k * cb:k * cb + int(sizes[k])].tobytes())
                offsets.append(offsets[-1] + int(sizes[k]))
    raw.free()
    cbuf.free()
    n_samples = seconds * RATE
    header = {'version': '1.0', 'algorithm': 'zlib', 'comp_level': -1, 'do_time_diff': True, 'do_spatial_diff': False, 'dtype': 'int16',
              'n_channels': nc, 'sample_rate': float(RATE), 'chunk_bounds': list(range(0, n_samples + 1, RATE)), 'chunk_offsets': offsets,
              'chunk_order': 'F', 'sha1_compressed': None, 'sha1_uncompressed': None, 'shape': [n_samples, nc]}
    (tmp / 'data.ch').write_text(json.dumps(header))
    return n_samples, offsets[-1]


def synth_host(hip, dev, t0, t1, nc):
    """Rows [t0, t1) of the synthetic recording as a host array (generated on the device)."""
    buf = hip.DevBuffer((t1 - t0) * nc * 2, dev)
    hip.dev_synth_int16(buf, 0, t0, t1, nc, 0)
    out = buf.download(dtype=np.int16).reshape(t1 - t0, nc)
    buf.free()
    return out


def window_starts(n_samples, n_windows):
    """BASELINE configs[2]'s windows: start = splitmix64(i) mod (n_samples - 30000)  (SURVEY 8d)."""
    return [int(splitmix(i) % (n_samples - RATE)) for i in range(n_windows)]


def extra_random_read(hip, dev, seconds, n_windows=1000, partial_decode=False):
    """BASELINE configs[2]: a `seconds` s 385-channel file (compressed on the device, written to tmpfs with its header), then
    Reader[s:s+30000] at s = splitmix(i) mod (n_samples - 30000): first pass (chunks decoded on first touch, then resident in
    the decoded-chunk cache in HBM) and second pass (every chunk resident); windows checked against the generator."""
    import tempfile
    import mtscomp_amd
    nc = 385
    tmp = Path(tempfile.mkdtemp(prefix='mtsbench_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None))
    os.environ.setdefault('HOME', str(tmp))
    t_gen = time.perf_counter()
    n_samples, cbytes = build_synth_file(hip, dev, seconds, tmp, nc)
    offsets = [0, cbytes]
    t_gen = time.perf_counter() - t_gen
    r = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch')
    starts = window_starts(n_samples, n_windows)
    passes = []
    for _ in range(2):
        t0 = time.perf_counter()
        nb = 0
        for s in starts:
            nb += r[s:s + RATE].nbytes
        passes.append((time.perf_counter() - t0, nb))
    # a few windows against the generator, and a column subset through the device gather
    ok = True
    r_last = synth_host(hip, dev, starts[99], starts[99] + RATE, nc)
    for s in starts[:3]:
        want = synth_host(hip, dev, s, s + RATE, nc)
        ok = ok and np.array_equal(r[s:s + RATE], want) and np.array_equal(r[s:s + RATE, 10:40], want[:, 10:40])
    t0 = time.perf_counter()
    got = r.read_slices([(slice(s, s + RATE), slice(0, 32)) for s in starts[:256]])
    t_cols = time.perf_counter() - t0
    r.close()
    # the same column windows one at a time from a Reader that has nothing resident.  Default settings inflate and check whole chunks
    # like the reference; with partial_decode=True (the caller's explicit choice) the chunks are inflated only as far as
    # the leading 32 channels reach, from a prefix of their bytes (mts_cache_read_slices_leading)
    cold = {}
    for cols, key, kw in ((32, 'cols32', {}), (32, 'cols32_partial', {'partial_decode': True}), (nc, 'all', {})):
        rc_ = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch', **kw)
        t0 = time.perf_counter()
        for s in starts[:100]:
            w = rc_[s:s + RATE, 0:cols]
        cold[key] = (time.perf_counter() - t0) / 100 * 1e3
        ok = ok and np.array_equal(w, r_last[:, 0:cols])
        rc_.close()
    # ONE cold 1 s window (two chunks nothing has touched), read-ahead off: what a caller waits for a single Reader[a:b] on a file
    # it has not read before -- pread, copy in, inflate of a two-chunk batch, rows out (the reference: two serial read_chunk calls,
    # mtscomp.py:798-856, 602-635)
    from mtscomp_amd import api as _api
    ra_keep, _api.READ_AHEAD_MAX = _api.READ_AHEAD_MAX, 0
    try:
        rc_ = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch')
        singles = []
        for k in range(1, min(seconds // 3, 60)):
            s0 = (3 * k) * RATE + RATE // 2
            t0 = time.perf_counter()
            w = rc_[s0:s0 + RATE]
            singles.append((time.perf_counter() - t0) * 1e3)
        ok = ok and np.array_equal(w, synth_host(hip, dev, s0, s0 + RATE, nc))
        rc_.close()
    finally:
        _api.READ_AHEAD_MAX = ra_keep
    for p in (tmp / 'data.cbin', tmp / 'data.ch'):
        p.unlink()
    try:
        (tmp / '.mtscomp').unlink()
    except OSError:
        pass
    tmp.rmdir()
    return {'workload': '385 ch @ 30 kHz, %d s file (%.2f GB raw, %.2f GB .cbin on tmpfs), %d windows of 1 s, start = splitmix64(i) mod (n_samples - 30000)'
                        % (seconds, n_samples * nc * 2 / 1e9, offsets[-1] / 1e9, n_windows),
            'first_pass_ms_per_window': passes[0][0] / n_windows * 1e3, 'first_pass_gbps': passes[0][1] / passes[0][0] / 1e9,
            'resident_ms_per_window': passes[1][0] / n_windows * 1e3, 'resident_gbps': passes[1][1] / passes[1][0] / 1e9,
            'columns_0_32_of_256_windows_one_call_ms': t_cols * 1e3, 'columns_bytes_returned': int(sum(g.nbytes for g in got)),
            'cold_ms_per_window_columns_0_32': cold['cols32'], 'cold_ms_per_window_columns_0_32_partial_decode': cold['cols32_partial'],
            'cold_ms_per_window_all_columns': cold['all'],
            'cold_single_window_ms': float(np.median(singles)) if singles else None, 'cold_single_window_ms_mean': float(np.mean(singles)) if singles else None,
            'cold_single_window_note': 'one Reader[a:b] of 1 s over two chunks never touched before, read-ahead OFF, median / mean of %d windows' % len(singles),
            'verified': bool(ok), 'build_file_s': t_gen,
            'reader': 'Reader[a:b] through the decoded-chunk cache in HBM (MTSCOMP_DEVICE_CACHE_GB, default 32); pread + H2D + decode on first touch, one D2H of the rows after'}


def extra_level_sweep(hip, dev, seconds=60, levels=(1, 2, 3, 6, 9)):
    """BASELINE configs[4] shape: 1024 ch @ 30 kHz, chunk = 0.25 s (7500 rows); `seconds` s of it, compressed on the device at
    levels 1, 2, 3 (deflate_fast), 6 and 9; chunk 0 of every level checked against stdlib zlib."""
    import zlib
    from oracle import oracle as O
    nc, rows = 1024, 7500
    n = seconds * 4
    raw = hip.DevBuffer(n * rows * nc * 2, dev)
    hip.dev_synth_int16(raw, 0, 0, n * rows, nc, 0)
    cb = (hip.compress_bound(rows * nc * 2) + 255) // 256 * 256
    cbuf = hip.DevBuffer(n * cb, dev)
    back = hip.DevBuffer(raw.nbytes, dev)
    bounds = np.arange(n + 1, dtype=np.int64) * rows
    slots = np.arange(n, dtype=np.int64) * cb
    sizes = np.zeros(n, dtype=np.int64)
    nrows = np.full(n, rows, dtype=np.int64)
    ooffs = np.arange(n, dtype=np.int64) * rows * nc * 2
    status = np.zeros(n, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')
    stream0 = O.delta_transpose(raw.download(0, rows * nc * 2, np.int16).reshape(rows, nc), flags).tobytes()
    out = {}
    for level in levels:
        best = None
        for rep in range(2):                                  # (the first call at a level also allocates its workspace)
            hip.dev_sync(dev)
            t0 = time.perf_counter()
            hip.dev_compress_chunks(raw, nc, 2, bounds, flags, level, cbuf, slots, sizes)
            hip.dev_sync(dev)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        dt_d = None
        for rep in range(2):                                  # (the first call at a new shape also allocates the inflate workspace)
            t0 = time.perf_counter()
            hip.dev_decompress_chunks(cbuf, slots, sizes, nrows, nc, 2, flags, back, ooffs, status)
            hip.dev_sync(dev)
            dt = time.perf_counter() - t0
            assert not status.any()
            dt_d = dt if dt_d is None else min(dt_d, dt)
        ident = cbuf.download(0, int(sizes[0])).tobytes() == zlib.compress(stream0, level)
        out[str(level)] = {'ratio': float(sizes.sum()) / (n * rows * nc * 2), 'compress_gbps': n * rows * nc * 2 / best / 1e9,
                           'decompress_gbps': n * rows * nc * 2 / dt_d / 1e9, 'round_trip_ok': back.diff(raw)[0] == 0,
                           'byte_identical_chunk0': bool(ident)}
    for b in (raw, cbuf, back):
        b.free()
    out['workload'] = '1024 ch @ 30 kHz, %d s, chunk = 0.25 s (%d chunks of 15.36 MB), device resident' % (seconds, n)
    return out


def cpu_file_to_file(raw_path, tmp, nc, n_threads):
    """The reference's file-to-file calls restated (Writer.write mtscomp.py:461-495, Reader.tofile :717-738) on numpy + stdlib
    zlib: batches of n_threads chunks through a ThreadPool, in-order write, both SHA-1s, JSON header; then the way back."""
    import hashlib
    from multiprocessing.dummy import Pool as ThreadPool
    from oracle import oracle as O
    data = np.memmap(raw_path, dtype=np.int16, mode='r').reshape(-1, nc)
    n = data.shape[0]
    bounds = list(range(0, n, RATE)) + ([n] if n % RATE else [n])
    bounds = sorted(set(bounds))
    ids = list(range(len(bounds) - 1))
    out, outmeta, back = tmp / 'cpu.cbin', tmp / 'cpu.ch', tmp / 'cpu_back.bin'
    s_raw, s_c = hashlib.sha1(), hashlib.sha1()
    offs = [0]
    t0 = time.perf_counter()
    with open(out, 'wb') as f, ThreadPool(n_threads) as pool:
        for b0 in range(0, len(ids), n_threads):
            batch = ids[b0:b0 + n_threads]
            cc = pool.map(lambda i: O.ref_compress_chunk(data[bounds[i]:bounds[i + 1]]), batch)
            for i, c in zip(batch, cc):
                f.write(c)
                offs.append(offs[-1] + len(c))
                s_raw.update(np.ascontiguousarray(data[bounds[i]:bounds[i + 1]]))
                s_c.update(c)
    outmeta.write_text(json.dumps({'chunk_offsets': offs, 'chunk_bounds': bounds, 'sha1_compressed': s_c.hexdigest(),
                                   'sha1_uncompressed': s_raw.hexdigest()}, indent=2, sort_keys=True))
    t1 = time.perf_counter()
    fd = os.open(out, os.O_RDONLY)
    with open(back, 'wb') as f, ThreadPool(n_threads) as pool:
        for b0 in range(0, len(ids), n_threads):
            batch = ids[b0:b0 + n_threads]
            arrs = pool.map(lambda i: O.ref_decompress_chunk(os.pread(fd, offs[i + 1] - offs[i], offs[i]), bounds[i + 1] - bounds[i], nc, 'int16'), batch)
            for a in arrs:
                f.write(a)
    os.close(fd)
    t2 = time.perf_counter()
    same = np.array_equal(np.memmap(back, dtype=np.int16, mode='r'), np.memmap(raw_path, dtype=np.int16, mode='r'))
    res = {'compress_gbps': data.nbytes / (t1 - t0) / 1e9, 'decompress_gbps': data.nbytes / (t2 - t1) / 1e9, 'threads': n_threads,
           'sha1_compressed': s_c.hexdigest(), 'round_trip_ok': bool(same)}
    for q in (out, outmeta, back):
        q.unlink()
    return res


def extra_file_to_file(x, nc, with_cpu):
    """The drop-in calls on a file: mtscomp_amd.compress(raw, .cbin, .ch) and mtscomp_amd.decompress(.cbin, .ch, out) of the
    benchmarked recording written to tmpfs -- pread, PCIe both ways, the two SHA-1s the .ch needs, file writes: everything
    the reference's benchmark.py:26-45 times, check_after_* off as there (:29, :36).  Next to it the CPU port's same calls."""
    import hashlib
    import tempfile
    import mtscomp_amd
    tmp = Path(tempfile.mkdtemp(prefix='mtsbench_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None))
    os.environ.setdefault('HOME', str(tmp))
    raw, out, outmeta, back = tmp / 'data.bin', tmp / 'data.cbin', tmp / 'data.ch', tmp / 'back.bin'
    try:
        x.tofile(raw)
        nbytes = raw.stat().st_size
        best_c = None
        d_calls = []
        # calls 0..2 write a NEW file each (Reader.tofile unlinks an existing one first, as the reference does, mtscomp.py:711-717;
        # call 0 also allocates the engine's staging buffers); calls 3 and 4 write over the existing file in place -- the library's
        # opt-in (MTSCOMP_TOFILE_IN_PLACE=1), NOT the reference's behaviour: reported beside, never as decompress_gbps
        for rep in range(5):
            t0 = time.perf_counter()
            ratio = mtscomp_amd.compress(raw, out, outmeta, sample_rate=float(RATE), n_channels=nc, dtype=np.int16, check_after_compress=False)
            tc = time.perf_counter() - t0
            if rep >= 3:
                os.environ['MTSCOMP_TOFILE_IN_PLACE'] = '1'
            try:
                t1 = time.perf_counter()
                r = mtscomp_amd.decompress(out, outmeta, back, overwrite=True, check_after_decompress=False)
                r.close()
                t2 = time.perf_counter()
            finally:
                os.environ.pop('MTSCOMP_TOFILE_IN_PLACE', None)
            d_calls.append(nbytes / (t2 - t1) / 1e9)
            best_c = tc if best_c is None else min(best_c, tc)
        meta = json.loads(outmeta.read_text())
        sha_c = hashlib.sha1(out.read_bytes()).hexdigest()
        same = np.array_equal(np.memmap(back, dtype=np.int16, mode='r'), np.memmap(raw, dtype=np.int16, mode='r'))
        res = {'workload': '%d ch @ 30 kHz, %.0f s int16 file on tmpfs (%.2f GB), chunk = 1 s, level 6; check_after_* off' % (nc, x.shape[0] / RATE, nbytes / 1e9),
               'compress_gbps': nbytes / best_c / 1e9, 'decompress_gbps': max(d_calls[1:3]), 'ratio': ratio,
               'decompress_semantics': "new file per call (an existing one unlinked first): the reference's open(out, 'wb'), mtscomp.py:711-717",
               'decompress_gbps_by_call': {'new_file_first_call': d_calls[0], 'new_file': d_calls[1:3], 'in_place_opt_in': d_calls[3:]},
               'header_sha1_matches_file': bool(meta['sha1_compressed'] == sha_c), 'round_trip_file_identical': bool(same),
               'sha1_compressed': sha_c}
        if with_cpu:
            cpu = cpu_file_to_file(raw, tmp, nc, os.cpu_count() or 1)
            res['cpu_port'] = cpu
            res['cbin_identical_to_cpu_port'] = bool(cpu['sha1_compressed'] == sha_c)
        return res
    finally:
        for q in (raw, out, outmeta, back, tmp / '.mtscomp'):
            try:
                q.unlink()
            except OSError:
                pass
        try:
            tmp.rmdir()
        except OSError:
            pass


def extra_host_api(hip, dev, x, nc, reps=3):
    """The drop-in boundary itself: mts_compress_chunks / mts_decompress_chunks (include/mtscomp_hip.h -- HOST pointers in and out,
    what a reference-side binding calls instead of pool.map(_compress_chunk) / zlib.decompress, mtscomp.py:399-423, 619) on the
    benchmarked recording, once from ordinary pageable memory and once from page-locked memory of mts_host_alloc; PCIe both ways
    is inside the timed call.  Best of `reps` calls after one warm-up; the results are checked (round trip on the host, chunk 0
    against zlib)."""
    import ctypes as C
    L = hip.lib()
    rows = RATE
    n = x.shape[0] // rows
    row_b = nc * x.itemsize
    raw_bytes = n * rows * row_b
    flags = hip.make_flags(True, False, 'F')
    bounds = np.arange(n + 1, dtype=np.int64) * rows
    cb = (hip.compress_bound(rows * row_b) + 15) // 16 * 16
    slots = np.arange(n, dtype=np.int64) * cb
    rws = np.full(n, rows, dtype=np.int64)
    ooffs = np.arange(n, dtype=np.int64) * (rows * row_b)
    out = {}
    for kind in ('pageable', 'page_locked'):
        held = []
        if kind == 'pageable':
            src = np.ascontiguousarray(x[:n * rows]).reshape(-1).view(np.uint8)
            comp = np.zeros(n * cb + 16, dtype=np.uint8)
            back = np.zeros(raw_bytes + 256, dtype=np.uint8)
        else:
            held = [hip.HostBuffer(raw_bytes), hip.HostBuffer(n * cb + 16), hip.HostBuffer(raw_bytes + 256)]
            src, comp, back = (h.array for h in held)
            src[:] = np.ascontiguousarray(x[:n * rows]).reshape(-1).view(np.uint8)
            comp[:] = 0
        sizes = np.zeros(n, dtype=np.int64)
        status = np.zeros(n, dtype=np.int32)
        packed = None
        tc, td = [], []
        for rep in range(reps + 1):
            t0 = time.perf_counter()
            rc = L.mts_compress_chunks(dev, src.ctypes.data_as(C.c_void_p), nc, x.itemsize, lp(bounds), n, flags, 6,
                                       comp.ctypes.data_as(C.c_void_p), lp(slots), lp(sizes))
            t1 = time.perf_counter()
            assert rc == 0, L.mts_last_error()
            # (the decoder takes the chunks where the encoder left them: slot offsets and sizes, as a .cbin reader would pass file offsets)
            rc = L.mts_decompress_chunks(dev, comp.ctypes.data_as(C.c_void_p), lp(slots), lp(sizes), lp(rws), n, nc, x.itemsize, flags,
                                         back.ctypes.data_as(C.c_void_p), lp(ooffs), status.ctypes.data_as(C.POINTER(C.c_int)))
            t2 = time.perf_counter()
            assert rc == 0 and not status.any(), (L.mts_last_error(), status)
            if rep:
                tc.append(t1 - t0)
                td.append(t2 - t1)
        ok = bool(np.array_equal(back[:raw_bytes], src[:raw_bytes]))
        from oracle import oracle as O
        same0 = comp[:int(sizes[0])].tobytes() == O.ref_compress_chunk(x[:rows])
        out[kind] = {'compress_gbps': raw_bytes / min(tc) / 1e9, 'decompress_gbps': raw_bytes / min(td) / 1e9,
                     'compress_gbps_calls': [raw_bytes / t / 1e9 for t in tc], 'decompress_gbps_calls': [raw_bytes / t / 1e9 for t in td],
                     'round_trip_ok': ok, 'byte_identical_chunk0': bool(same0)}
        assert ok and same0
        del src, comp, back
        for h in held:
            h.free()
    csize = int(sizes.sum())
    out['workload'] = '%d chunks of %d x %d int16 (%.2f GB raw, %.2f GB compressed) per call, host memory in and out, level 6' % (n, rows, nc, raw_bytes / 1e9, csize / 1e9)
    out['entry_points'] = 'mts_compress_chunks / mts_decompress_chunks (include/mtscomp_hip.h): PCIe both ways inside the call'
    out['pcie_note'] = 'compress moves R in + C out, decompress C in + R out over PCIe Gen5 x16 (~55 GB/s measured one way from page-locked memory)'
    return out


def extra_in_process_multi_gpu(hip, x, nc, n_dev):
    """What a drop-in user of an N-GPU node gets without torch.distributed: HipCodec(devices=range(N)), one host thread per
    GPU, chunk i -> GPU i mod N, host buffers in and out (PCIe included).  Asserted: the lanes sit on N DISTINCT devices, lane g
    was handed chunks g, g + N, ... and every one of the N devices reports kernel time for the call."""
    from mtscomp_amd.api import HipCodec
    codec = HipCodec(devices=list(range(n_dev)))
    assert len(set(codec.devices)) == n_dev == codec.n_lanes, codec.devices
    chunks = [x[i * RATE:(i + 1) * RATE] for i in range(x.shape[0] // RATE)]
    shards = [list(sh) for sh in codec._shards(len(chunks))]
    assert [sh for sh in shards if sh] == [list(range(g, len(chunks), n_dev)) for g in range(n_dev) if g < len(chunks)], shards
    flags = hip.make_flags(True, False, 'F')
    best_c = best_d = None
    for rep in range(2):
        t0 = time.perf_counter()
        cc = codec.compress(chunks, flags, 6)
        t1 = time.perf_counter()
        busy_c = {d: sum(ms for _, ms in hip.last_stage_times(d)) for d in codec.devices}
        st, arrs = codec.decompress(cc, [RATE] * len(cc), nc, np.int16, flags)
        t2 = time.perf_counter()
        busy_d = {d: sum(ms for _, ms in hip.last_stage_times(d)) for d in codec.devices}
        best_c = t1 - t0 if best_c is None else min(best_c, t1 - t0)
        best_d = t2 - t1 if best_d is None else min(best_d, t2 - t1)
    idle = [d for d in codec.devices[:min(n_dev, len(chunks))] if not (busy_c[d] > 0 and busy_d[d] > 0)]
    assert not idle, 'devices %s ran no kernel in the in-process multi-GPU call' % idle
    ok = all(s == 0 for s in st) and all(np.array_equal(a, c) for a, c in zip(arrs, chunks))
    nb = len(chunks) * RATE * nc * 2
    codec.close()
    return {'devices': n_dev, 'distinct_devices': sorted(set(codec.devices)), 'chunks': len(chunks), 'chunks_per_lane': [len(sh) for sh in shards],
            'device_ms_compress': {str(d): busy_c[d] for d in codec.devices}, 'device_ms_decompress': {str(d): busy_d[d] for d in codec.devices},
            'compress_gbps': nb / best_c / 1e9, 'decompress_gbps': nb / best_d / 1e9, 'round_trip_ok': bool(ok),
            'path': 'HipCodec(devices=range(N)): host arrays in, bytes out, one host thread per GPU (rank 0 of the bench, the other ranks idle)'}


def init_ranks(args, rank, world, dev):
    """The process groups of a multi-rank run, under a watchdog: a rendezvous or RCCL bring-up that does not finish within
    --init-timeout seconds ends this rank with a message and a non-zero status instead of hanging the node.
    Returns (dist, host_group): host_group is a gloo group -- every exchange of the run goes over it."""
    import datetime
    import threading
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    done = threading.Event()

    def watchdog():
        if not done.wait(args.init_timeout):
            sys.stderr.write('bench.py: rank %d: the %s process group did not come up within %d s; giving up\n' % (rank, args.dist_backend, args.init_timeout))
            sys.stderr.flush()
            os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    timeout = datetime.timedelta(seconds=args.init_timeout)
    try:
        if args.dist_backend == 'gloo':
            dist.init_process_group('gloo', timeout=timeout)
            host_group = dist.group.WORLD
        else:
            import torch
            torch.cuda.set_device(dev)
            dist.init_process_group('nccl', device_id=torch.device('cuda', dev), timeout=timeout)
            host_group = dist.new_group(backend='gloo', timeout=timeout)
            t = torch.ones(1, device='cuda')
            dist.all_reduce(t)                              # RCCL up and counting: outside the timed region, never on the data path
            torch.cuda.synchronize()
            assert int(t.item()) == world, 'RCCL all-reduce over %d ranks gave %r' % (world, t.item())
        dist.barrier(group=host_group)
    except Exception as e:  # noqa: BLE001
        sys.stderr.write('bench.py: rank %d: process group (%s) failed: %r\n' % (rank, args.dist_backend, e))
        sys.stderr.flush()
        os._exit(3)
    done.set()
    return dist, host_group


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # not under torch.distributed yet: start the ranks (nothing in this process has touched the GPU)
        sys.exit(self_launch(args, argv))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != max(args.gpus, 1):
        sys.stderr.write('bench.py: --gpus %d but WORLD_SIZE=%d\n' % (args.gpus, world))
        sys.exit(2)
    if world > 1:
        import torch  # noqa: F401  (torch.distributed only; imported before the library so that, if torch's HIP runtime is ever
        #                            initialised -- --dist-backend nccl --, the library resolves to that same runtime)
    from mtscomp_amd import hip
    have = hip.device_count()
    if local_rank >= have and not args.oversubscribe:
        sys.stderr.write('bench.py: rank %d has no device (%d visible)\n' % (local_rank, have))
        sys.exit(2)
    hip.require_device()
    dev = local_rank % max(have, 1)
    if args.oversubscribe and world > have and args.dist_backend == 'nccl':
        # RCCL refuses two ranks on one device ("Duplicate GPU detected"); the shared-device smoke test runs over gloo
        if rank == 0:
            sys.stderr.write('bench.py: %d ranks share %d device(s): process group over gloo instead of nccl\n' % (world, have))
        args.dist_backend = 'gloo'
    dist = host_group = None
    if world > 1:
        dist, host_group = init_ranks(args, rank, world, dev)
    L = hip.lib()

    nc, rate = args.channels, args.chunk_rows   # (rate: samples per chunk -- 30000 = 1 s, 7500 = 0.25 s)
    n_chunks = args.n_chunks                    # chunks this rank owns
    level = 6
    row = nc * 2
    chunk_bytes = rate * row
    raw_bytes = n_chunks * chunk_bytes
    sh = None                                   # the library's default stream on `dev`

    # synthetic recording, generated on device: this rank owns global chunks rank, rank + world, ... of a recording of world x n_chunks chunks
    mine = shard_ids(rank, world, n_chunks * world)
    raw = hip.DevBuffer(raw_bytes, dev)
    for k, g in enumerate(mine):
        hip.dev_synth_int16(raw, k * chunk_bytes, g * rate, (g + 1) * rate, nc, 0)
    hip.dev_sync(dev)

    bound = (hip.compress_bound(chunk_bytes) + 255) // 256 * 256
    cbuf = hip.DevBuffer(n_chunks * bound, dev)
    back = hip.DevBuffer(raw_bytes, dev)
    bounds = np.arange(n_chunks + 1, dtype=np.int64) * rate
    slots = np.arange(n_chunks, dtype=np.int64) * bound
    sizes = np.zeros(n_chunks, dtype=np.int64)
    rows = np.full(n_chunks, rate, dtype=np.int64)
    ooffs = np.arange(n_chunks, dtype=np.int64) * chunk_bytes
    status = np.zeros(n_chunks, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')

    def compress(lv=level):
        rc = L.mts_dev_compress_chunks(dev, sh, raw.at(), nc, 2, lp(bounds), n_chunks, flags, lv, cbuf.at(), lp(slots), lp(sizes))
        assert rc == 0, L.mts_last_error()

    def decompress():
        rc = L.mts_dev_decompress_chunks(dev, sh, cbuf.at(), lp(slots), lp(sizes), lp(rows), n_chunks, nc, 2, flags, back.at(), lp(ooffs),
                                         status.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0, L.mts_last_error()
        assert not status.any(), status

    def gather_sizes():
        return gather_chunk_offsets(sizes, rank, world, dist, host_group)

    stage = {}

    def add_stages():
        for name, ms in hip.last_stage_times(dev):
            stage.setdefault(name, []).append(ms)

    def barrier():                                  # every rank's device idle, then every rank here (the contract's barrier + synchronize)
        hip.dev_sync(dev)
        if world > 1:
            dist.barrier(group=host_group)
        hip.dev_sync(dev)

    for _ in range(args.warmup):
        compress(); decompress(); gather_sizes()
    stage.clear()
    t_c = t_d = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        a = time.perf_counter()
        compress()
        add_stages()
        b = time.perf_counter()
        decompress()
        add_stages()
        offsets = gather_sizes()
        c = time.perf_counter()
        t_c += b - a
        t_d += c - b
    barrier()
    elapsed = time.perf_counter() - t0
    per_rank_ms = None
    if world > 1:
        import torch
        # every rank's own clock for its steps (compress + decompress + the exchange, without the closing barrier), so that an
        # imbalance between the ranks shows; the job's time is the slowest rank's, barrier to barrier
        own = torch.tensor([(t_c + t_d) / max(args.steps, 1) * 1e3], dtype=torch.float64)
        every = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(every, own, group=host_group)
        per_rank_ms = [float(v.item()) for v in every]
        t = torch.tensor([elapsed, t_c, t_d], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=host_group)
        elapsed, t_c, t_d = t.tolist()

    # correctness of what was timed (outside the timed region): device round trip on every rank + oracle spot check
    n_diff, first_diff = back.diff(raw)
    assert n_diff == 0, 'round trip mismatch: %d bytes differ, the first at %d' % (n_diff, first_diff)
    csize_local = int(sizes.sum())
    csize = int(offsets[-1]) if args.steps else csize_local
    ok_oracle = None
    if rank == 0:
        from oracle import oracle as O
        first = raw.download(0, chunk_bytes, np.int16).reshape(rate, nc)
        z0 = cbuf.download(0, int(sizes[0])).tobytes()
        ok_oracle = z0 == O.ref_compress_chunk(first)
        assert ok_oracle, 'chunk 0 is not byte-identical to zlib level 6'

    # --config stress: levels 1 and 9 beside the timed level 6 (BASELINE configs[4]: "levels {1,6,9} sweep -- ratio vs GB/s curve"):
    # one warm-up and one timed pass per level on every rank, the slowest rank's time, the ratio over all ranks' chunks; after the
    # timed region, round trip checked on the device, chunk 0 of rank 0 against zlib at that level
    level_curve = None
    if args.levels_beside:
        level_curve = {'6': {'compress_gbps': raw_bytes * world * args.steps / t_c / 1e9, 'decompress_gbps': raw_bytes * world * args.steps / t_d / 1e9,
                             'ratio': (int(offsets[-1]) if args.steps else int(sizes.sum())) / (raw_bytes * world)}}
        for lv in args.levels_beside:
            compress(lv); decompress()
            barrier()
            a = time.perf_counter()
            compress(lv)
            hip.dev_sync(dev)
            b = time.perf_counter()
            decompress()
            hip.dev_sync(dev)
            c = time.perf_counter()
            offs_lv = gather_sizes()
            tc_lv, td_lv = b - a, c - b
            if world > 1:
                import torch
                t = torch.tensor([tc_lv, td_lv], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=host_group)
                tc_lv, td_lv = t.tolist()
            nd, fd = back.diff(raw)
            assert nd == 0, 'level %d: round trip mismatch (%d bytes, the first at %d)' % (lv, nd, fd)
            same = None
            if rank == 0:
                from oracle import oracle as O
                same = cbuf.download(0, int(sizes[0])).tobytes() == O.ref_compress_chunk(raw.download(0, chunk_bytes, np.int16).reshape(rate, nc), level=lv)
                assert same, 'level %d: chunk 0 is not byte-identical to zlib' % lv
            level_curve[str(lv)] = {'compress_gbps': raw_bytes * world / tc_lv / 1e9, 'decompress_gbps': raw_bytes * world / td_lv / 1e9,
                                    'ratio': int(offs_lv[-1]) / (raw_bytes * world), 'byte_identical_chunk0': same}
        compress(); decompress()                                 # (the buffers hold the level-6 result again: the checks below look at it)

    if rank == 0:
        total_raw = raw_bytes * world * args.steps
        ms_step = elapsed / args.steps * 1e3
        sm = {k: float(np.mean(v)) for k, v in stage.items()}
        algo = n_chunks * chunk_bytes + csize_local      # R + C per launch (a launch = this rank's batch of chunks)
        # the dominant kernel = the longest single-kernel stage
        dom_stage = max((k for k in sm if k in STAGE_KERNEL), key=lambda k: sm[k], default='match')
        dom_kernel, dom_ms = STAGE_KERNEL.get(dom_stage, 'k_match5'), sm.get(dom_stage, 0.0)
        achieved = algo / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(dom_kernel, n_chunks, nc)
        comp_names = ('delta_transpose', 'hash_sort', 'match', 'parse_fixpoint', 'parse_emit', 'block_trees', 'block_pack')
        comp_ms = sum(sm.get(k, 0.0) for k in comp_names)
        dec_ms = sum(v for k, v in sm.items() if k not in comp_names and not k.startswith('hash_sort'))
        dec_dom = max(((k, v) for k, v in sm.items() if k.startswith('inflate') or k in ('adler32', 'cumsum_transpose')), key=lambda kv: kv[1],
                      default=('', 0.0))

        def direction(ms):
            ach = algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS,
                    'algorithmic_bytes_per_step': algo, 'device_ms_per_step': ms}

        def transform(stage_name, kernel):                    # K1 / K2 alone: read R + write R
            ms = sm.get(stage_name, 0.0)
            ach = 2 * raw_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {'bound': 'hbm', 'kernel': kernel, 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS,
                    'algorithmic_bytes_per_launch': 2 * raw_bytes, 'launch_ms': ms}
        res = {
            'metric': 'compress + decompress GB/s (raw int16)', 'value': total_raw / elapsed / 1e9, 'unit': 'GB/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int16',
            'data': 'synthetic',
            'config': {'workload': args.workload,
                       'n_channels': nc, 'chunks_per_gpu': n_chunks, 'chunk_rows': rate, 'chunk_bytes': chunk_bytes,
                       'sharding': 'chunk i -> rank i mod N (round robin), no collective on the data path; the compressed sizes are '
                                   'gathered on the host (gloo, CPU tensors) and prefix-summed into chunk_offsets; process group: %s' % args.dist_backend,
                       'rank_chunks': {str(r): shard_ids(r, world, n_chunks * world) for r in range(world)} if world > 1 else None,
                       'oversubscribed': bool(args.oversubscribe and world > have)},
            'compress_gbps': raw_bytes * world * args.steps / t_c / 1e9,
            'decompress_gbps': raw_bytes * world * args.steps / t_d / 1e9,
            'ratio': csize / (raw_bytes * world), 'byte_identical_chunk0': ok_oracle,
            'stage_ms': sm,
            'roofline': {'bound': 'hbm', 'kernel': dom_kernel, 'stage': dom_stage, 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': algo, 'launch_ms': dom_ms},
            'roofline_compress': direction(comp_ms),
            'roofline_decompress': dict(direction(dec_ms), dominant_stage=dec_dom[0], dominant_stage_ms=dec_dom[1]),
            'roofline_k1': transform('delta_transpose', 'k_delta_rows'),
            'roofline_k2': transform('cumsum_transpose', 'k_cumsum_rows'),
        }
        if per_rank_ms is not None:
            res['per_rank_ms_per_step'] = {'min': min(per_rank_ms), 'max': max(per_rank_ms), 'ranks': per_rank_ms}
        if level_curve is not None:
            res['level_curve'] = level_curve
        x_host = None
        cpu_chunks = None
        headline = args.config == 'headline'
        if (not args.no_cpu_baseline or not args.no_extras) and world == 1 and headline:
            x_host = raw.download(dtype=np.int16).reshape(n_chunks * rate, nc)
        if not args.no_cpu_baseline and world == 1 and headline:             # (the CPU comparison is taken once, at N = 1)
            try:
                res['cpu_baseline'], cpu_chunks = cpu_baseline(x_host, nc, n_chunks, args.cpu_chunks)
            except Exception as e:  # noqa: BLE001  -- the headline line must come out whatever happens here
                res['cpu_baseline'] = {'error': repr(e)}
        if cpu_chunks is not None:
            # the whole recording, not chunk 0 alone: every chunk the timed region produced against the CPU path's bytes
            import hashlib
            host = cbuf.download()
            mine_c = [host[int(slots[k]):int(slots[k]) + int(sizes[k])].tobytes() for k in range(n_chunks)]
            same = [a == b for a, b in zip(mine_c, cpu_chunks)]
            res['byte_identical_chunks'] = '%d/%d' % (sum(same), n_chunks)
            res['cbin_sha1'] = hashlib.sha1(b''.join(mine_c)).hexdigest()
            res['cbin_sha1_cpu'] = hashlib.sha1(b''.join(cpu_chunks)).hexdigest()
            del host, mine_c
            assert all(same), 'chunks %s differ from zlib level 6' % [i for i, ok in enumerate(same) if not ok][:8]
        if not args.no_extras and world == 1 and headline:
            for b in (back, cbuf, raw):             # (room for the extras' buffers)
                b.free()
            extras = {}
            for name, fn in (('host_api', lambda: extra_host_api(hip, dev, x_host, nc)),
                             ('file_to_file', lambda: extra_file_to_file(x_host, nc, not args.no_cpu_baseline)),
                             ('random_read', lambda: extra_random_read(hip, dev, args.extras_seconds)),
                             ('level_sweep', lambda: extra_level_sweep(hip, dev))):
                t1 = time.perf_counter()
                try:
                    extras[name] = fn()
                except Exception as e:  # noqa: BLE001
                    extras[name] = {'error': repr(e)}
                extras[name]['wall_s'] = time.perf_counter() - t1
            res['extras'] = extras
        if not args.no_extras and world > 1 and not args.oversubscribe and headline:
            t1 = time.perf_counter()
            try:
                n_take = min(n_chunks, 64)                       # (host arrays through PCIe: a sample of the shard is enough to see every lane work)
                res['extras'] = {'in_process_multi_gpu': extra_in_process_multi_gpu(hip, raw.download(0, n_take * chunk_bytes, np.int16).reshape(n_take * rate, nc), nc, world)}
            except Exception as e:  # noqa: BLE001
                res['extras'] = {'in_process_multi_gpu': {'error': repr(e)}}
            res['extras']['in_process_multi_gpu']['wall_s'] = time.perf_counter() - t1
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier(group=host_group)              # (the other ranks keep their devices free until rank 0 is done)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
