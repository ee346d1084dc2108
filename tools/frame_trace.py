#!/usr/bin/env python3
"""Print one frame's kernel timeline from a rocprofv3 kernel trace CSV (tools/frame_trace.py <kernel_trace.csv> [frame])."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'splatClipDataKernel' in r['Kernel_Name']]
f = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[f], idx[f + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev = None
tot = 0
for r in rows[a - 4:b - 3]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = re.sub(r'<.*', '', r['Kernel_Name'].replace('rocprim::ROCPRIM_400200_NS::detail::', 'rp::').replace('gs::(anonymous namespace)::', ''))[:44]
    gap = (s - prev) / 1e3 if prev else 0
    tot += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:7.1f} gap={gap:6.1f} grid={r['Grid_Size_X']:>9s} wg={r['Workgroup_Size_X']:>5s} vgpr={r['VGPR_Count']:>4s} lds={r['LDS_Block_Size']:>6s} {name}")
    prev = e
print('sum of kernel durations (us):', round(tot, 1))
