"""Sums the PC samples of tools/pc_sample.sh per kernel, per source line and per instruction.

    python tools/pc_sample_summary.py gpurun_out/pcs [kernel substring, default k_match5]
"""
import collections
import csv
import glob
import sys

out = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else 'k_match5'
csv.field_size_limit(1 << 30)
kern = {}
for fn in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        kern[r.get('Dispatch_Id')] = r.get('Kernel_Name', '?').split('(')[0]
files = [f for f in glob.glob(out + '/**/*.csv', recursive=True) if 'pc_sampling' in f]
print('files:', files)
per_kernel = collections.Counter()
per_line = collections.Counter()
per_inst = collections.Counter()
per_reason = collections.Counter()
n = 0
for fn in files:
    rd = csv.DictReader(open(fn))
    print(fn, rd.fieldnames)
    for r in rd:
        n += 1
        k = kern.get(r.get('Dispatch_Id'), '?')
        per_kernel[k] += 1
        if want not in k:
            continue
        inst = r.get('Instruction', '?')
        line = r.get('Instruction_Comment', '?')
        per_line[line] += 1
        per_inst[(line, inst)] += 1
        for key in ('Stall_Reason', 'Wave_Issued', 'Instruction_Type', 'Snapshot_Stall_Reason'):
            if key in r:
                per_reason[(key, r[key])] += 1
print('samples', n)
for k, v in per_kernel.most_common(12):
    print('%8d  %s' % (v, k))
tot = sum(per_line.values()) or 1
print('--- %s: %d samples; per source line' % (want, tot))
for k, v in per_line.most_common(80):
    print('%6.2f%%  %s' % (100.0 * v / tot, k))
print('--- per instruction')
for (line, inst), v in per_inst.most_common(120):
    print('%6.2f%%  %-60s %s' % (100.0 * v / tot, inst[:60], line[-50:]))
print('--- reasons')
for k, v in per_reason.most_common(40):
    print('%6.2f%%  %s' % (100.0 * v / tot, k))
