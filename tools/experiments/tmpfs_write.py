"""How fast can a fresh tmpfs file be filled?  pwrite threads x piece sizes, with and without fallocate running ahead."""
import os, sys, time, threading
import numpy as np
from multiprocessing.dummy import Pool
N = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 1386000000
path = '/dev/shm/_tw.bin'
src = np.random.randint(0, 255, N, dtype=np.uint8)
mv = memoryview(src)

def run(nthreads, piece, falloc=0, label=''):
    if os.path.exists(path): os.unlink(path)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    t0 = time.perf_counter()
    stop = []
    if falloc:
        def fa():
            step = 64 << 20
            for a in range(0, N, step):
                os.posix_fallocate(fd, a, min(step, N - a))
        fts = []
        if falloc == 1:
            th = threading.Thread(target=fa); th.start(); fts.append(th)
        else:
            def fa2(k):
                per = (N + falloc - 1) // falloc
                a0, a1 = k * per, min(N, (k + 1) * per)
                step = 16 << 20
                for a in range(a0, a1, step):
                    os.posix_fallocate(fd, a, min(step, a1 - a))
            for k in range(falloc):
                th = threading.Thread(target=fa2, args=(k,)); th.start(); fts.append(th)
    def one(a):
        b = min(a + piece, N)
        while a < b:
            a += os.pwrite(fd, mv[a:b], a)
    with Pool(nthreads) as p:
        p.map(one, range(0, N, piece), chunksize=1)
    if falloc:
        for th in fts: th.join()
    dt = time.perf_counter() - t0
    os.close(fd)
    print('%-28s threads %2d piece %4d MB falloc %d: %.2f GB/s' % (label, nthreads, piece >> 20, falloc, N / dt / 1e9), flush=True)

for nt in (1, 2, 4, 8, 16):
    run(nt, 16 << 20)
for piece in (1 << 20, 4 << 20, 64 << 20):
    run(4, piece)
run(4, 16 << 20, 1)
run(8, 16 << 20, 1)
run(4, 16 << 20, 4)
# rewrite into existing pages
fd = os.open(path, os.O_WRONLY)
t0 = time.perf_counter()
def one(a):
    b = min(a + (16 << 20), N)
    while a < b: a += os.pwrite(fd, mv[a:b], a)
with Pool(4) as p: p.map(one, range(0, N, 16 << 20))
print('existing pages, 4 threads: %.2f GB/s' % (N / (time.perf_counter() - t0) / 1e9))
os.close(fd); os.unlink(path)
