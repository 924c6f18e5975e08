"""Gaps around one kernel in a rocprofv3 --kernel-trace CSV: python tools/trace_gaps.py <dir> <kernel substring> [n]
Prints, for the first n launches of the kernel: the previous kernel, the gap between its end and this start, the duration."""
import csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
f = [p for p in glob.glob(d + '/**/*kernel_trace.csv', recursive=True)][0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
k = 0
for i, r in enumerate(rows):
    if pat in r['Kernel_Name'] and i > 0:
        p = rows[i - 1]
        q = rows[i + 1] if i + 1 < len(rows) else None
        print(f"{r['Kernel_Name'][:40]:40s} dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us | prev {p['Kernel_Name'][:36]:36s} "
              f"dur {(int(p['End_Timestamp']) - int(p['Start_Timestamp'])) / 1e3:8.1f} gap {(int(r['Start_Timestamp']) - int(p['End_Timestamp'])) / 1e3:7.1f} us"
              + (f" | next gap {(int(q['Start_Timestamp']) - int(r['End_Timestamp'])) / 1e3:7.1f}" if q else ''))
        k += 1
        if k >= n:
            break
