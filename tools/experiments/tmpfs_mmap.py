"""Fresh tmpfs file filled through a shared mapping by N threads (page faults in parallel, no inode lock) vs pwrite."""
import os, sys, time, mmap
import numpy as np
from multiprocessing.dummy import Pool
N = 1386000000
path = '/dev/shm/_tw.bin'
src = np.random.randint(0, 255, N, dtype=np.uint8)
mv = memoryview(src)
def run_pwrite(nt, piece=16 << 20):
    if os.path.exists(path): os.unlink(path)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    t0 = time.perf_counter()
    def one(a):
        b = min(a + piece, N)
        while a < b: a += os.pwrite(fd, mv[a:b], a)
    with Pool(nt) as p: p.map(one, range(0, N, piece), chunksize=1)
    dt = time.perf_counter() - t0
    os.close(fd)
    print('pwrite  threads %2d: %.2f GB/s' % (nt, N / dt / 1e9), flush=True)
def run_mmap(nt, piece=4 << 20, populate=False):
    if os.path.exists(path): os.unlink(path)
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    t0 = time.perf_counter()
    os.ftruncate(fd, N)
    mm = mmap.mmap(fd, N, flags=mmap.MAP_SHARED | (mmap.MAP_POPULATE if populate else 0))
    dst = np.frombuffer(mm, dtype=np.uint8)
    t1 = time.perf_counter()
    def one(a):
        b = min(a + piece, N)
        np.copyto(dst[a:b], src[a:b])
    with Pool(nt) as p: p.map(one, range(0, N, piece), chunksize=1)
    dt = time.perf_counter() - t0
    del dst; mm.close(); os.close(fd)
    print('mmap    threads %2d piece %2d MB populate %d: %.2f GB/s (map %.1f ms)' % (nt, piece >> 20, populate, N / dt / 1e9, (t1 - t0) * 1e3), flush=True)
for nt in (1, 2): run_pwrite(nt)
for nt in (1, 2, 4, 8, 16, 32): run_mmap(nt)
run_mmap(8, 1 << 20); run_mmap(8, 16 << 20); run_mmap(16, 1 << 20)
run_mmap(8, 4 << 20, True)
os.unlink(path)
