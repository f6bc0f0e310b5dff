"""Kernel sequence from the middle of a bench run under rocprofv3 --kernel-trace: name, duration, gap to the previous kernel.
usage: seq_dump.py <prof_dir> <rows>"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2])      # rows to print, from the middle of the trace (steady-state graph replays)
rows = rows[len(rows) // 2:len(rows) // 2 + n]
prev = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'\(.*', '', r['Kernel_Name'])
    name = re.sub(r'^void ', '', name)[:60]
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')):>4}  {name}")
    prev = e
