"""Headline benchmark: compress + decompress throughput (GB/s of raw int16) of the chunked delta + DEFLATE
hot path on MI355X, inputs and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over the workload: compress every chunk (K1 delta+transpose ->
bit-exact zlib level-6 DEFLATE) and decompress it again (INFLATE -> K2 cumsum+transpose).  Workload at
N=1: BASELINE.json configs[1] (385 ch @ 30 kHz, 60 s, 1 s chunks, level 6).  With N ranks the chunks of an
N x 60 s recording are sharded round-robin (chunk i -> rank i mod N, no data-path collective; only the
compressed sizes are gathered), so per-GPU work is fixed: weak scaling.

The JSON line carries, next to the contract's keys:
  roofline             the dominant kernel (k_match5): algorithmic bytes (R + C of the batch) / its launch time
  roofline_compress    (R + C) / time of the whole compress direction, roofline_decompress likewise (C + R)
  cpu_baseline         the reference's ThreadPool path restated (numpy + stdlib zlib: the oracle) on ALL host cores and on 1
  extras (N = 1 only, after the timed region; --no-extras skips them)
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


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--seconds', type=int, default=60, help='recording length per GPU (1 s chunks)')
    p.add_argument('--channels', type=int, default=385)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-chunks', type=int, default=0, help='chunks in the all-cores CPU sample (default: one per host cpu)')
    p.add_argument('--no-extras', action='store_true', help='skip the configs[2] / configs[4] measurements')
    p.add_argument('--extras-seconds', type=int, default=600, help='length of the random-read file (configs[2])')
    return p.parse_args()


def cpu_baseline(x, nc, n_have, n_sample):
    """The reference's ThreadPool path restated on numpy + stdlib zlib (oracle.ref_*), timed on this box's host cores
    over a bounded sample of the benchmarked recording: one chunk per host cpu (the recording's chunks, cycled), all of them
    in flight at once like the reference's batch of n_threads chunks; and one core alone on two chunks."""
    from multiprocessing.dummy import Pool as ThreadPool
    import zlib
    from oracle import oracle as O
    ncpu = os.cpu_count() or 1
    n_sample = n_sample or ncpu
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
                  'zlib %s level 6, ThreadPool(%d) of %d host cpus, every chunk in flight at once: compress %.3f GB/s, decompress %.3f GB/s; '
                  'one core on %d chunks: compress %.4f GB/s, decompress %.3f GB/s'
                  % (n_sample, nc, RATE, gb * 1e3, n_have, zlib.ZLIB_RUNTIME_VERSION, cores, ncpu, gb / (t1 - t0), gb / (t2 - t1),
                     m1, gb1 / (a1 - a0), gb1 / (a2 - a1)),
        'compress_gbps': gb / (t1 - t0), 'decompress_gbps': gb / (t2 - t1),
        'one_core': {'value': gb1 / (a2 - a0), 'compress_gbps': gb1 / (a1 - a0), 'decompress_gbps': gb1 / (a2 - a1), 'cores': 1, 'chunks': m1},
    }


def measured_traffic(n_chunks, nc):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of THIS round's profile
    (profiles/r2_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this script; a counter pass cannot
    run inside the timed process).  Only valid for the profiled workload (60 chunks x 385 ch); null otherwise."""
    for name in ('r2_traffic.json', 'r1_traffic.json'):
        p = ROOT / 'profiles' / name
        if p.exists() and n_chunks == 60 and nc == 385:
            d = json.loads(p.read_text())
            return d['traffic_bytes_per_launch'], 'profiles/' + name
    return None, None


def splitmix(i):
    m = (1 << 64) - 1
    z = (i + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    return z ^ (z >> 31)


def lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_long))


def extra_random_read(torch, hip, L, dev, seconds, n_windows=1000):
    """BASELINE configs[2]: a `seconds` s 385-channel file (compressed on the device, written to tmpfs with its header), then
    Reader[s:s+30000] at s = splitmix(i) mod (n_samples - 30000): first pass (chunks decoded on first touch, then resident in
    the decoded-chunk cache in HBM) and second pass (every chunk resident); windows checked against the generator."""
    import tempfile
    import mtscomp_amd
    nc, piece = 385, 60
    tmp = Path(tempfile.mkdtemp(prefix='mtsbench_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None))
    os.environ.setdefault('HOME', str(tmp))
    cb = (hip.compress_bound(RATE * nc * 2) + 255) // 256 * 256
    raw = torch.empty((piece * RATE, nc), dtype=torch.int16, device='cuda')
    cbuf = torch.empty(piece * cb, dtype=torch.uint8, device='cuda')
    flags = hip.make_flags(True, False, 'F')
    offsets = [0]
    t_gen = time.perf_counter()
    with open(tmp / 'data.cbin', 'wb') as f:
        for p0 in range(0, seconds, piece):
            n = min(piece, seconds - p0)
            for k in range(n):
                rc = L.mts_dev_synth_int16(dev, None, C.c_void_p(raw[k * RATE:].data_ptr()), (p0 + k) * RATE, (p0 + k + 1) * RATE, nc, 0)
                assert rc == 0
            bounds = np.arange(n + 1, dtype=np.int64) * RATE
            slots = np.arange(n, dtype=np.int64) * cb
            sizes = np.zeros(n, dtype=np.int64)
            rc = L.mts_dev_compress_chunks(dev, None, C.c_void_p(raw.data_ptr()), nc, 2, lp(bounds), n, flags, 6, C.c_void_p(cbuf.data_ptr()),
                                           lp(slots), lp(sizes))
            assert rc == 0, L.mts_last_error()
            host = cbuf[:n * cb].cpu().numpy()
            for k in range(n):
                f.write(host[k * cb:k * cb + int(sizes[k])].tobytes())
                offsets.append(offsets[-1] + int(sizes[k]))
    del raw, cbuf
    n_samples = seconds * RATE
    header = {'version': '1.0', 'algorithm': 'zlib', 'comp_level': -1, 'do_time_diff': True, 'do_spatial_diff': False, 'dtype': 'int16',
              'n_channels': nc, 'sample_rate': float(RATE), 'chunk_bounds': list(range(0, n_samples + 1, RATE)), 'chunk_offsets': offsets,
              'chunk_order': 'F', 'sha1_compressed': None, 'sha1_uncompressed': None, 'shape': [n_samples, nc]}
    (tmp / 'data.ch').write_text(json.dumps(header))
    t_gen = time.perf_counter() - t_gen
    r = mtscomp_amd.decompress(tmp / 'data.cbin', tmp / 'data.ch')
    starts = [int(splitmix(i) % (n_samples - RATE)) for i in range(n_windows)]
    passes = []
    for _ in range(2):
        t0 = time.perf_counter()
        nb = 0
        for s in starts:
            nb += r[s:s + RATE].nbytes
        passes.append((time.perf_counter() - t0, nb))
    # a few windows against the generator, and a column subset through the device gather
    ok = True
    chk = torch.empty((RATE, nc), dtype=torch.int16, device='cuda')
    for s in starts[:3]:
        assert L.mts_dev_synth_int16(dev, None, C.c_void_p(chk.data_ptr()), s, s + RATE, nc, 0) == 0
        want = chk.cpu().numpy()
        ok = ok and np.array_equal(r[s:s + RATE], want) and np.array_equal(r[s:s + RATE, 10:40], want[:, 10:40])
    t0 = time.perf_counter()
    got = r.read_slices([(slice(s, s + RATE), slice(0, 32)) for s in starts[:256]])
    t_cols = time.perf_counter() - t0
    r.close()
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
            'verified': bool(ok), 'build_file_s': t_gen,
            'reader': 'Reader[a:b] through the decoded-chunk cache in HBM (MTSCOMP_DEVICE_CACHE_GB, default 32); pread + H2D + decode on first touch, one D2H of the rows after'}


def extra_level_sweep(torch, hip, L, dev, seconds=60, levels=(1, 2, 3, 6, 9)):
    """BASELINE configs[4] shape: 1024 ch @ 30 kHz, chunk = 0.25 s (7500 rows); `seconds` s of it, compressed on the device at
    levels 1, 2, 3 (deflate_fast), 6 and 9; chunk 0 of every level checked against stdlib zlib."""
    import zlib
    from oracle import oracle as O
    nc, rows = 1024, 7500
    n = seconds * 4
    raw = torch.empty((n * rows, nc), dtype=torch.int16, device='cuda')
    for k in range(n):
        assert L.mts_dev_synth_int16(dev, None, C.c_void_p(raw[k * rows:].data_ptr()), k * rows, (k + 1) * rows, nc, 0) == 0
    cb = (hip.compress_bound(rows * nc * 2) + 255) // 256 * 256
    cbuf = torch.empty(n * cb, dtype=torch.uint8, device='cuda')
    back = torch.empty_like(raw)
    bounds = np.arange(n + 1, dtype=np.int64) * rows
    slots = np.arange(n, dtype=np.int64) * cb
    sizes = np.zeros(n, dtype=np.int64)
    nrows = np.full(n, rows, dtype=np.int64)
    ooffs = np.arange(n, dtype=np.int64) * rows * nc * 2
    status = np.zeros(n, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')
    stream0 = O.delta_transpose(raw[:rows].cpu().numpy(), flags).tobytes()
    out = {}
    for level in levels:
        best = None
        for rep in range(2):                                  # (the first call at a level also allocates its workspace)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = L.mts_dev_compress_chunks(dev, None, C.c_void_p(raw.data_ptr()), nc, 2, lp(bounds), n, flags, level, C.c_void_p(cbuf.data_ptr()),
                                           lp(slots), lp(sizes))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert rc == 0, L.mts_last_error()
            best = dt if best is None else min(best, dt)
        dt_d = None
        for rep in range(2):                                  # (the first call at a new shape also allocates the inflate workspace)
            t0 = time.perf_counter()
            rc = L.mts_dev_decompress_chunks(dev, None, C.c_void_p(cbuf.data_ptr()), lp(slots), lp(sizes), lp(nrows), n, nc, 2, flags,
                                             C.c_void_p(back.data_ptr()), lp(ooffs), status.ctypes.data_as(C.POINTER(C.c_int)))
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            assert rc == 0 and not status.any()
            dt_d = dt if dt_d is None else min(dt_d, dt)
        ident = cbuf[:int(sizes[0])].cpu().numpy().tobytes() == zlib.compress(stream0, level)
        out[str(level)] = {'ratio': float(sizes.sum()) / (n * rows * nc * 2), 'compress_gbps': n * rows * nc * 2 / best / 1e9,
                           'decompress_gbps': n * rows * nc * 2 / dt_d / 1e9, 'round_trip_ok': bool(torch.equal(back, raw)),
                           'byte_identical_chunk0': bool(ident)}
    out['workload'] = '1024 ch @ 30 kHz, %d s, chunk = 0.25 s (%d chunks of 15.36 MB), device resident' % (seconds, n)
    return out


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from mtscomp_amd import hip

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)
    hip.require_device()
    L = hip.lib()

    nc, rate = args.channels, RATE
    n_chunks = args.seconds                     # 1 s chunks
    row = nc * 2
    chunk_bytes = rate * row
    raw_bytes = n_chunks * chunk_bytes
    stream = torch.cuda.current_stream()
    sh = C.c_void_p(stream.cuda_stream)

    # synthetic recording, generated on device: this rank owns global chunks rank, rank+world, ...
    raw = torch.empty((n_chunks * rate, nc), dtype=torch.int16, device='cuda')
    for k in range(n_chunks):
        g = rank + k * world
        rc = L.mts_dev_synth_int16(dev, sh, C.c_void_p(raw[k * rate:].data_ptr()), g * rate, (g + 1) * rate, nc, 0)
        assert rc == 0, hip.lib().mts_last_error()
    torch.cuda.synchronize()

    bound = (hip.compress_bound(chunk_bytes) + 255) // 256 * 256
    cbuf = torch.empty(n_chunks * bound, dtype=torch.uint8, device='cuda')
    back = torch.empty_like(raw)
    bounds = np.arange(n_chunks + 1, dtype=np.int64) * rate
    slots = np.arange(n_chunks, dtype=np.int64) * bound
    sizes = np.zeros(n_chunks, dtype=np.int64)
    rows = np.full(n_chunks, rate, dtype=np.int64)
    ooffs = np.arange(n_chunks, dtype=np.int64) * chunk_bytes
    status = np.zeros(n_chunks, dtype=np.int32)
    flags = hip.make_flags(True, False, 'F')

    def compress():
        rc = L.mts_dev_compress_chunks(dev, sh, C.c_void_p(raw.data_ptr()), nc, 2, lp(bounds), n_chunks, flags, 6,
                                       C.c_void_p(cbuf.data_ptr()), lp(slots), lp(sizes))
        assert rc == 0, L.mts_last_error()

    def decompress():
        rc = L.mts_dev_decompress_chunks(dev, sh, C.c_void_p(cbuf.data_ptr()), lp(slots), lp(sizes), lp(rows), n_chunks,
                                         nc, 2, flags, C.c_void_p(back.data_ptr()), lp(ooffs),
                                         status.ctypes.data_as(C.POINTER(C.c_int)))
        assert rc == 0, L.mts_last_error()
        assert not status.any(), status

    def gather_sizes():
        # the only cross-rank step of the path: compressed sizes -> chunk_offsets (host-side prefix sum)
        if world == 1:
            return sizes.copy()
        t = torch.from_numpy(sizes).cuda()
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        allsz = torch.stack(out, dim=1).reshape(-1).cpu().numpy()     # global chunk order: i = rank + k*world
        return np.concatenate(([0], np.cumsum(allsz)))

    stage = {}

    def add_stages():
        for name, ms in hip.last_stage_times(dev):
            stage.setdefault(name, []).append(ms)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

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
        gather_sizes()
        c = time.perf_counter()
        t_c += b - a
        t_d += c - b
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed, t_c, t_d], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, t_c, t_d = t.tolist()

    # correctness of what was timed (outside the timed region): device round trip + oracle spot check
    assert torch.equal(back, raw), 'round trip mismatch'
    csize = int(sizes.sum())
    ok_oracle = None
    if rank == 0:
        from oracle import oracle as O
        first = raw[:rate].cpu().numpy()
        z0 = cbuf[:int(sizes[0])].cpu().numpy().tobytes()
        ok_oracle = z0 == O.ref_compress_chunk(first)
        assert ok_oracle, 'chunk 0 is not byte-identical to zlib level 6'

    if rank == 0:
        total_raw = raw_bytes * world * args.steps
        ms_step = elapsed / args.steps * 1e3
        sm = {k: float(np.mean(v)) for k, v in stage.items()}
        match_ms = sm.get('match', 0.0)
        algo = n_chunks * chunk_bytes + csize            # R + C per launch of the match kernel
        achieved = algo / (match_ms * 1e-3) / 1e9 if match_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(n_chunks, nc)
        comp_names = ('delta_transpose', 'hash_sort', 'match', 'parse_fixpoint', 'parse_emit', 'block_trees', 'block_pack')
        comp_ms = sum(sm.get(k, 0.0) for k in comp_names)
        dec_ms = sum(v for k, v in sm.items() if k not in comp_names and not k.startswith('hash_sort'))
        dec_dom = max(((k, v) for k, v in sm.items() if k.startswith('inflate') or k in ('adler32', 'cumsum_transpose')), key=lambda kv: kv[1],
                      default=('', 0.0))

        def direction(ms):
            ach = algo / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBPS,
                    'algorithmic_bytes_per_step': algo, 'device_ms_per_step': ms}
        res = {
            'metric': 'compress + decompress GB/s (raw int16)', 'value': total_raw / elapsed / 1e9, 'unit': 'GB/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'int16',
            'data': 'synthetic',
            'config': {'workload': '%d ch @ 30 kHz, %d s synthetic AR int16 per GPU, chunk=1 s, zlib level 6 '
                                   '(BASELINE configs[1])' % (nc, args.seconds),
                       'n_channels': nc, 'chunks_per_gpu': n_chunks, 'chunk_bytes': chunk_bytes,
                       'sharding': 'chunk i -> rank i mod N (round robin), no collective on the data path'},
            'compress_gbps': raw_bytes * world * args.steps / t_c / 1e9,
            'decompress_gbps': raw_bytes * world * args.steps / t_d / 1e9,
            'ratio': csize / raw_bytes, 'byte_identical_chunk0': ok_oracle,
            'stage_ms': sm,
            'roofline': {'bound': 'hbm', 'kernel': 'k_match5', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': algo, 'launch_ms': match_ms},
            'roofline_compress': direction(comp_ms),
            'roofline_decompress': dict(direction(dec_ms), dominant_stage=dec_dom[0], dominant_stage_ms=dec_dom[1]),
        }
        if not args.no_cpu_baseline and world == 1:             # (the CPU comparison is taken once, at N = 1)
            try:
                res['cpu_baseline'] = cpu_baseline(raw.cpu().numpy(), nc, n_chunks, args.cpu_chunks)
            except Exception as e:  # noqa: BLE001  -- the headline line must come out whatever happens here
                res['cpu_baseline'] = {'error': repr(e)}
        if not args.no_extras and world == 1:
            back = cbuf = None                      # (room for the extras' buffers)
            extras = {}
            for name, fn in (('random_read', lambda: extra_random_read(torch, hip, L, dev, args.extras_seconds)),
                             ('level_sweep', lambda: extra_level_sweep(torch, hip, L, dev))):
                t1 = time.perf_counter()
                try:
                    extras[name] = fn()
                except Exception as e:  # noqa: BLE001
                    extras[name] = {'error': repr(e)}
                extras[name]['wall_s'] = time.perf_counter() - t1
            res['extras'] = extras
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
