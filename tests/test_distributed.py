"""N > 1 path on CPU: two processes (gloo), chunks sharded round robin (chunk i -> rank i mod N) with no
collective on the data path; only the compressed sizes are gathered and prefix-summed into chunk_offsets.
The per-rank codec is the test-only oracle (no GPU here); the assembled .cbin must equal the single-process
file byte for byte -- which is what makes the sharding a pure scheduling choice."""
import hashlib
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys, json, hashlib
import numpy as np
sys.path.insert(0, os.environ['MTS_ROOT'])
import torch, torch.distributed as dist
from mtscomp_amd.synth import synth_int16
from tests.codec_oracle import OracleCodec
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
nc, rate, n_chunks = 16, 500, 7
mine = list(range(rank, n_chunks, world))                      # chunk i -> rank i mod N
chunks = [synth_int16(i * rate, (i + 1) * rate, nc, 0) for i in mine]
cbufs = OracleCodec().compress(chunks, 5, 6)
sizes = torch.zeros(n_chunks, dtype=torch.int64)
for i, b in zip(mine, cbufs):
    sizes[i] = len(b)
dist.all_reduce(sizes)                                          # the only exchange: sizes
offsets = np.concatenate(([0], np.cumsum(sizes.numpy())))
out = os.environ['MTS_OUT']
# every rank writes its chunks at the gathered offsets of a shared file
with open(out, 'r+b') as f:
    for i, b in zip(mine, cbufs):
        f.seek(int(offsets[i])); f.write(b)
dist.barrier()
if rank == 0:
    json.dump({'offsets': [int(v) for v in offsets]}, open(out + '.json', 'w'))
dist.destroy_process_group()
'''


def test_two_rank_round_robin_equals_single_process(tmp_path):
    from mtscomp_amd.synth import synth_int16
    from tests.codec_oracle import OracleCodec
    nc, rate, n_chunks = 16, 500, 7
    x = synth_int16(0, rate * n_chunks, nc, 0)
    want = OracleCodec().compress([x[i * rate:(i + 1) * rate] for i in range(n_chunks)], 5, 6)
    total = sum(map(len, want))
    out = tmp_path / 'sharded.cbin'
    out.write_bytes(bytes(total))
    worker = tmp_path / 'worker.py'
    worker.write_text(WORKER)
    env = dict(os.environ, MTS_ROOT=str(ROOT), MTS_OUT=str(out), MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29533', str(worker)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.read_bytes() == b''.join(want)
    offs = json.loads((tmp_path / 'sharded.cbin.json').read_text())['offsets']
    assert offs == [0] + list(np.cumsum([len(b) for b in want]))
