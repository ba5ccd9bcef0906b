"""The bench line's `extras.level_sweep` alone (BASELINE configs[4] shape: 1024 ch, chunk = 0.25 s, levels 1, 2, 3, 6, 9), for
profiling:  rocprofv3 --kernel-trace --stats -- python3 tools/level_sweep.py [seconds of recording, default 60] [levels]"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from mtscomp_amd import hip  # noqa: E402

if __name__ == '__main__':
    seconds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    levels = tuple(int(x) for x in sys.argv[2].split(',')) if len(sys.argv) > 2 else (1, 2, 3, 6, 7, 8, 9)
    hip.require_device()
    print(json.dumps(bench.extra_level_sweep(hip, 0, seconds, levels)))
