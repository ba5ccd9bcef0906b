"""N > 1 path on CPU: bench.py's launcher starts two (and three) ranks under gloo; chunks are sharded round robin (chunk i ->
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


@pytest.mark.parametrize('world', [2, 3])
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
