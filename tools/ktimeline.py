"""Timeline of the last extraction in a rocprofv3 kernel trace: python tools/ktimeline.py gpurun_out/<dir>/p_kernel_trace.csv [first-kernel-prefix]"""
import csv
import sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
first = sys.argv[2] if len(sys.argv) > 2 else 'k_pyramid'
names = [(r['Kernel_Name'][:18], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
last_desc = [i for i, n in enumerate(names) if n[0].startswith('k_describe')][-1]
idx = last_desc
while idx > 0 and not names[idx][0].startswith('k_describe') or idx == last_desc:
    idx -= 1
idx += 1
t0 = names[idx][1]
for n, s, e in names[idx:last_desc + 1]:
    print(f"{n:20s} start={(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:7.1f}")
print(f"span {(names[last_desc][2] - t0) / 1e3:.1f} us")
