"""Prints the kernels of the last MSM of a rocprofv3 kernel trace (development aid).
usage: python scripts/trace_tail.py DIR"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_digits' in r['Kernel_Name']][-1]
t0 = int(rows[idx]['Start_Timestamp'])
for r in rows[idx:]:
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('plonk::', '').replace('void ', '')
    if 'rocprim' in n:
        n = 'rocprim'
    if 'fillBuffer' in n:
        continue
    print('%-22s grid %9s wg %4s start %9.1f us dur %8.1f us' % (
        n[:22], r['Grid_Size_X'], r['Workgroup_Size_X'], (int(r['Start_Timestamp']) - t0) / 1e3,
        (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
