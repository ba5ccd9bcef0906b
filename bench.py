"""Headline benchmark: compress + decompress throughput (GB/s of raw int16) of the chunked delta + DEFLATE
hot path on MI355X, inputs and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over the workload: compress every chunk (K1 delta+transpose ->
bit-exact zlib level-6 DEFLATE) and decompress it again (INFLATE -> K2 cumsum+transpose).  Workload at
N=1: BASELINE.json configs[1] (385 ch @ 30 kHz, 60 s, 1 s chunks, level 6).  With N ranks the chunks of an
N x 60 s recording are sharded round-robin (chunk i -> rank i mod N, no data-path collective; only the
compressed sizes are gathered), so per-GPU work is fixed: weak scaling.
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


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=3)
    p.add_argument('--warmup', type=int, default=1)
    p.add_argument('--seconds', type=int, default=60, help='recording length per GPU (1 s chunks)')
    p.add_argument('--channels', type=int, default=385)
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--cpu-chunks', type=int, default=32, help='chunks in the CPU baseline sample')
    return p.parse_args()


def cpu_baseline(x, nc, rate, n_chunks):
    """The reference's ThreadPool path restated on numpy + stdlib zlib (oracle.ref_*), timed on this box's
    host cores over a bounded sample of the same workload (the first chunks of the benchmarked recording)."""
    from oracle import oracle as O
    cores = min(os.cpu_count() or 1, n_chunks)
    bounds = [i * rate for i in range(n_chunks + 1)]
    t0 = time.perf_counter()
    cc = O.ref_compress_array(x, bounds, n_threads=cores)
    t1 = time.perf_counter()
    back = O.ref_decompress_array(cc, bounds, nc, 'int16', n_threads=cores)
    t2 = time.perf_counter()
    assert all(np.array_equal(back[i], x[bounds[i]:bounds[i + 1]]) for i in (0, n_chunks - 1))
    gb = x.nbytes / 1e9
    return {
        'value': gb / (t2 - t0), 'unit': 'GB/s', 'cores': cores, 'kind': 'port',
        'sample': 'first %d chunks of the benchmarked recording (%d ch x %d samples int16 each, %.0f MB), numpy '
                  'diff/tobytes + stdlib zlib %s level 6 + ThreadPool(%d) of %d host cpus: compress %.3f GB/s, '
                  'decompress %.3f GB/s'
                  % (n_chunks, nc, rate, x.nbytes / 1e6, __import__('zlib').ZLIB_RUNTIME_VERSION, cores,
                     os.cpu_count() or 1, gb / (t1 - t0), gb / (t2 - t1)),
        'compress_gbps': gb / (t1 - t0), 'decompress_gbps': gb / (t2 - t1),
    }


def measured_traffic(n_chunks, nc):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r1_traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this script).  Only valid for the profiled
    workload (60 chunks x 385 ch); null otherwise."""
    p = ROOT / 'profiles' / 'r1_traffic.json'
    if not p.exists() or n_chunks != 60 or nc != 385:
        return None
    return json.loads(p.read_text())['traffic_bytes_per_launch']


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

    nc, rate = args.channels, 30000
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
    lp = lambda a: a.ctypes.data_as(C.POINTER(C.c_long))  # noqa: E731

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
        match_ms = float(np.mean(stage.get('match', [0.0])))
        algo = n_chunks * chunk_bytes + csize            # R + C per launch of the match kernel
        achieved = algo / (match_ms * 1e-3) / 1e9 if match_ms > 0 else 0.0
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
            'stage_ms': {k: float(np.mean(v)) for k, v in stage.items()},
            'roofline': {'bound': 'hbm', 'kernel': 'k_match5', 'achieved': achieved, 'peak': HBM_PEAK_GBPS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBPS, 'traffic': measured_traffic(n_chunks, nc),
                         'algorithmic_bytes_per_launch': algo, 'launch_ms': match_ms},
        }
        if not args.no_cpu_baseline and world == 1:             # (the CPU comparison is taken once, at N = 1)
            m = max(1, min(args.cpu_chunks, n_chunks))
            res['cpu_baseline'] = cpu_baseline(raw[:m * rate].cpu().numpy(), nc, rate, m)
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
