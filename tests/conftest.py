import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _has_gpu():
    return os.path.exists('/dev/kfd')


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The seeds of tests/test_gpu_fuzz.py in the log of every run that executed them (also with -q, also when they pass)."""
    ran = [r for r in terminalreporter.stats.get('passed', []) + terminalreporter.stats.get('failed', []) if 'test_gpu_fuzz' in r.nodeid]
    if ran:
        from tests import test_gpu_fuzz as F
        terminalreporter.write_line('fuzz: base seed %d (MTS_FUZZ_SEED; case k uses base + k), %.0f s per case (MTS_FUZZ_SECONDS): %s'
                                    % (F.BASE_SEED, F.SECONDS, ', '.join('%s=%d' % (c[0], F.BASE_SEED + c[2]) for c in F.CASES)))
