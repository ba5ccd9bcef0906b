"""Randomised parity under -m gpu: the three fuzzers of tools/ (which found both real bugs of rounds 3-4: the host copy
shares, k_match6's look-back race) run for a fixed budget each, with seeds that CHANGE from day to day and are printed, so
that the driver's GPU run sees fresh cases every round and a failure can be replayed:

    MTS_FUZZ_SEED=<seed printed by the failing run> MTS_FUZZ_SECONDS=30 python -m pytest tests/test_gpu_fuzz.py -m gpu -s
    (or directly: python tools/fuzz_gpu.py <seed> <seconds>, FUZZ_LEVELS=2 for levels 1..9)

What is compared (all through the C ABI, mtscomp_amd/hip.py):
  * fuzz_gpu.py          random dtypes / shapes / flags / contents: compressed bytes == the reference's statement sequence on
                         numpy + stdlib zlib (zlib.compress(chunkd.tobytes(order)), mtscomp.py:394), status 0, decoded bytes equal
                         (mtscomp.py:619-635); at level 6 only, at levels 4..9 (deflate_slow) and at levels 1..9 (deflate_fast too)
  * fuzz_inflate_gpu.py  streams of every block type and zlib level, clean and with flipped bits: the device's verdict == zlib's
                         (corrupt / wrong size / fine), as Reader.read_chunk maps it to IOError (mtscomp.py:619-621)
  * fuzz_reader_gpu.py   files written by compress(), read back through Reader[...] with random slices, steps and column
                         picks, cache on / tiny / off: equal to numpy indexing of the raw array (the reference's own randomised
                         round trips: tests.py:212-243, 381-410)
"""
import datetime
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu

BASE_SEED = int(os.environ.get('MTS_FUZZ_SEED', datetime.date.today().strftime('%Y%m%d')))
SECONDS = float(os.environ.get('MTS_FUZZ_SECONDS', 30))

CASES = [
    # (id, script, seed offset, extra environment)
    ('codec_level6', 'fuzz_gpu.py', 0, {}),
    ('codec_levels4to9', 'fuzz_gpu.py', 1, {'FUZZ_LEVELS': '1'}),
    ('codec_levels1to9', 'fuzz_gpu.py', 2, {'FUZZ_LEVELS': '2'}),
    ('inflate_verdicts', 'fuzz_inflate_gpu.py', 3, {}),
    ('reader_slices', 'fuzz_reader_gpu.py', 4, {}),
]


@pytest.mark.parametrize('name,script,offset,extra', CASES, ids=[c[0] for c in CASES])
def test_fuzz(name, script, offset, extra, tmp_path):
    seed = BASE_SEED + offset
    env = dict(os.environ, PYTHONWARNINGS='ignore', HOME=str(tmp_path), **extra)
    env.pop('MTSCOMP_DEVICE_CACHE_GB', None)
    (ROOT / 'gpurun_out').mkdir(exist_ok=True)                       # (fuzz_gpu.py leaves a failing input there)
    cmd = [sys.executable, str(ROOT / 'tools' / script), str(seed), str(SECONDS)]
    print('\nfuzz %s: seed %d, %.0f s: %s %s' % (name, seed, SECONDS, ' '.join('%s=%s' % kv for kv in extra.items()), ' '.join(cmd[1:])), flush=True)
    r = subprocess.run(cmd, cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=SECONDS * 6 + 300)
    tail = '\n'.join((r.stdout + r.stderr).strip().splitlines()[-12:])
    print(tail, flush=True)
    assert r.returncode == 0, 'fuzz %s failed with seed %d (replay: MTS_FUZZ_SEED=%d): %s' % (name, seed, BASE_SEED, tail)
    summary = [ln for ln in r.stdout.splitlines() if 'mismatches' in ln]
    assert summary and summary[-1].rstrip().endswith(' 0 mismatches'), tail
    # the run did something: at least one chunk / stream / read went through the device
    counts = [int(tok) for tok in summary[-1].replace(',', ' ').replace(':', ' ').split() if tok.isdigit()]
    assert len(counts) >= 3 and counts[1] > 0, summary[-1]
