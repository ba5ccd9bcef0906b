"""One rank of tests/test_distributed.py: bench.py's own sharding and size exchange (shard_ids, gather_chunk_offsets) under
gloo on CPU, started by bench.py's own launcher (launch_ranks).  The per-rank codec is the test-only oracle (no GPU here)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.environ['MTS_ROOT'])
import bench  # noqa: E402
from mtscomp_amd.synth import synth_int16  # noqa: E402
from tests.codec_oracle import OracleCodec  # noqa: E402

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist, host_group = bench.init_ranks(bench.parse_args(['--gpus', str(world)]), rank, world, 0)      # bench.py's own bring-up (gloo, watchdog)
assert (dist.get_rank(), dist.get_world_size()) == (rank, world)
nc, rate, per_rank = 16, 500, int(os.environ['MTS_PER_RANK'])
mine = bench.shard_ids(rank, world, per_rank * world)                      # chunk i -> rank i mod N
chunks = [synth_int16(i * rate, (i + 1) * rate, nc, 0) for i in mine]
cbufs = OracleCodec().compress(chunks, 5, 6)
offsets = bench.gather_chunk_offsets([len(b) for b in cbufs], rank, world, dist, host_group)      # the only exchange: sizes
out = os.environ['MTS_OUT']
if rank == 0:
    with open(out, 'wb') as f:
        f.truncate(int(offsets[-1]))
dist.barrier()
with open(out, 'r+b') as f:                                                 # every rank writes its chunks where they belong
    for i, b in zip(mine, cbufs):
        f.seek(int(offsets[i]))
        f.write(b)
dist.barrier()
if rank == 0:
    json.dump({'offsets': [int(v) for v in offsets], 'world': world}, open(out + '.json', 'w'))
dist.destroy_process_group()
