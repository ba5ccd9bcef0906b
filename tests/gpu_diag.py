"""Stand-alone stage-by-stage diagnostic for the GPU box: prints the first mismatch of each stage.
Usage: python tests/gpu_diag.py > gpurun_out/diag.log 2>&1"""
import sys
import time
import traceback
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mtscomp_amd import hip  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tests.inputs import cases_small  # noqa: E402


def first_diff(a, b):
    a = np.asarray(a).ravel(); b = np.asarray(b).ravel()
    n = min(a.size, b.size)
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) if d.size else (n if a.size != b.size else -1)), a.size, b.size


def main():
    print('devices', hip.device_count())
    cases = cases_small()
    names = sys.argv[1:] or ['three', 'aaa', 'zeros_1k', 'rand_300', 'first50', 'ar1_8ch', 'text_100k', 'repeats_200k']
    for name in names:
        data = cases[name]
        print('==== case', name, len(data))
        try:
            t0 = time.time()
            tf, tq = O.match_tables(data, 6)
            gf, gq = hip.debug_match_tables(data, 6)
            i, _, _ = first_diff(gf, tf)
            print(' tables t_full first diff', i, '' if i < 0 else (hex(int(gf[i])), hex(int(tf[i]))), 'ndiff', int((gf != tf).sum()))
            i, _, _ = first_diff(gq, tq)
            print(' tables t_quarter first diff', i, '' if i < 0 else (hex(int(gq[i])), hex(int(tq[i]))), 'ndiff', int((gq != tq).sum()))
            _, toks, tokpos, blocks = O.deflate(data, 6, report=True)
            gt = hip.debug_tokens(data, 6)
            i, na, nb = first_diff(gt, toks)
            print(' tokens', na // 2, nb // 2, 'first diff', i if i < 0 else (i // 2, gt.ravel()[i - i % 2:i - i % 2 + 2].tolist(), toks.ravel()[i - i % 2:i - i % 2 + 2].tolist()))
            want = zlib.compress(data, 6)
            got = hip.debug_deflate(data, 6)
            i, na, nb = first_diff(np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8))
            print(' deflate', na, nb, 'first diff byte', i, 'blocks', [(b['btype'], b['bit_start']) for b in blocks[:4]])
            st, out = hip.debug_inflate(want, len(data))
            i, na, nb = first_diff(np.frombuffer(out, np.uint8), np.frombuffer(data, np.uint8))
            print(' inflate status', st, 'first diff', i, na, nb)
            print(' stage times', hip.last_stage_times(), 'wall %.2fs' % (time.time() - t0))
        except Exception:
            traceback.print_exc()
        sys.stdout.flush()


if __name__ == '__main__':
    main()
