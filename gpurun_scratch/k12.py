import sys, numpy as np
sys.path.insert(0, '.')
from mtscomp_amd import hip
from oracle import oracle as O
r = np.random.RandomState(3)
for shape in [(30000, 64), (3000, 385), (64, 64), (128, 8), (200, 64), (64, 2)]:
    for dt in ['int16', 'uint8', 'int32']:
        info = np.iinfo(dt)
        x = r.randint(info.min, int(info.max) + 1, size=shape, dtype=np.int64).astype(dt)
        want = O.delta_transpose(x, 5)
        got = hip.delta_transpose(x, 5)
        ok1 = np.array_equal(got, want)
        d = np.flatnonzero(np.frombuffer(got, np.uint8) != np.frombuffer(want, np.uint8)) if not ok1 else []
        back = hip.cumsum_transpose(want, shape[0], shape[1], dt, 5)
        ok2 = np.array_equal(back, x)
        d2 = np.argwhere(back != x)[:3].tolist() if not ok2 else []
        print(shape, dt, ok1, list(d[:5]), len(d), ok2, d2)
