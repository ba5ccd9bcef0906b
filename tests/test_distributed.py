"""N > 1 path on CPU: bench.py's launcher starts two, three and four ranks under gloo; chunks are sharded round robin (chunk i ->
rank i mod N, bench.shard_ids) with no collective on the data path; only the compressed sizes are gathered and prefix-summed
into chunk_offsets (bench.gather_chunk_offsets).  The per-rank codec is the test-only oracle (no GPU here); the assembled
.cbin must equal the single-process file byte for byte -- which is what makes the sharding a pure scheduling choice."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import bench
from mtscomp_amd.synth import synth_int16
from tests.codec_oracle import OracleCodec

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize('world', [2, 3, 4])
def test_ranks_round_robin_equal_single_process(tmp_path, world):
    nc, rate, per_rank = 16, 500, 3
    n_chunks = per_rank * world
    x = synth_int16(0, rate * n_chunks, nc, 0)
    want = OracleCodec().compress([x[i * rate:(i + 1) * rate] for i in range(n_chunks)], 5, 6)
    out = tmp_path / 'sharded.cbin'
    env = dict(os.environ, MTS_ROOT=str(ROOT), MTS_OUT=str(out), MTS_PER_RANK=str(per_rank))
    rc = bench.launch_ranks(world, ROOT / 'tests' / 'dist_worker.py', [], env=env, timeout=300)
    assert rc == 0
    assert out.read_bytes() == b''.join(want)
    meta = json.loads((tmp_path / 'sharded.cbin.json').read_text())
    assert meta['world'] == world
    assert meta['offsets'] == [0] + list(np.cumsum([len(b) for b in want]))


def test_shards_and_offsets_single_rank():
    assert bench.shard_ids(1, 4, 10) == [1, 5, 9]
    assert sorted(sum((bench.shard_ids(r, 3, 11) for r in range(3)), [])) == list(range(11))
    assert list(bench.gather_chunk_offsets([5, 7, 1], 0, 1)) == [0, 5, 12, 13]


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` never measures fewer GPUs than asked for: without enough devices it fails, loudly."""
    if bench.visible_gpus() >= 64:
        pytest.skip('this box has 64 GPUs')
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '64', '--steps', '1', '--warmup', '0'], capture_output=True, text=True,
                       timeout=300, env={k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')})
    assert r.returncode == 2 and 'refusing' in r.stderr
    assert r.stdout.strip() == ''


def test_default_workloads_land_on_baseline_configs():
    """What `python bench.py --gpus N` measures when the driver passes nothing else: N = 1 is BASELINE configs[1] (60 chunks);
    N = 8 is configs[3] (3600 s = 450 chunks per rank) and N = 2 / 4 keep that per-rank shard (weak scaling); --config stress is
    configs[4] (1024 ch, 0.25 s chunks, 2400 chunks over the ranks, levels 1 and 9 beside the timed level 6).  Every global chunk
    of those jobs has exactly one owner."""
    a = bench.parse_args([])
    assert (a.n_chunks, a.channels, a.chunk_rows, a.levels_beside) == (60, 385, 30000, ()) and 'configs[1]' in a.workload
    for n in (2, 4, 8):
        a = bench.parse_args(['--gpus', str(n)])
        assert (a.n_chunks, a.channels, a.chunk_rows) == (450, 385, 30000) and '450 chunks per rank' in a.workload
        owners = sorted(sum((bench.shard_ids(r, n, a.n_chunks * n) for r in range(n)), []))
        assert owners == list(range(450 * n))
    a = bench.parse_args(['--gpus', '8'])
    assert 'configs[3]: 3600 s over 8 GPUs' in a.workload and a.n_chunks * 8 == 3600
    a = bench.parse_args(['--gpus', '8', '--config', 'stress'])
    assert (a.n_chunks * 8, a.channels, a.chunk_rows, a.levels_beside) == (2400, 1024, 7500, (1, 9)) and 'configs[4]' in a.workload
    a = bench.parse_args(['--config', 'stress'])
    assert (a.n_chunks, a.channels, a.chunk_rows) == (300, 1024, 7500)
    a = bench.parse_args(['--gpus', '2', '--seconds', '2'])          # (the smoke test's size: flags still win)
    assert a.n_chunks == 2
