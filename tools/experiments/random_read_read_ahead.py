import os, sys, json, subprocess
sys.path.insert(0, '.')
if len(sys.argv) > 1:
    os.environ['MTSCOMP_READ_AHEAD'] = sys.argv[1]
    import bench
    from mtscomp_amd import hip
    r = bench.extra_random_read(hip, 0, 600)
    print('read-ahead max', sys.argv[1], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != 'workload'}, flush=True)
else:
    for ra in ('0', '4', '2', '0', '4'):
        print(subprocess.run([sys.executable, __file__, ra], capture_output=True, text=True).stdout.strip()[-700:], flush=True)
