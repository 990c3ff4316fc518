"""Development tool: timeline of one step from a rocprofv3 --kernel-trace csv (start / end of every kernel relative to the step's
first solve kernel, per queue).  Usage: python scratch/timeline.py <dir with *_kernel_trace.csv> [step index from the end, default 30]"""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0], r.get('Queue_Id', '?')) for r in rows))
names = set(e[2] for e in ev)
mark = next(m for m in ('k_solve_light_ow', 'k_solve_light', 'k_solve') if m in names)
starts = [i for i, e in enumerate(ev) if e[2] == mark]
for which in (back, back - 1):
    i0, i1 = starts[-which], starts[-which + 1]
    t0 = ev[i0][0]
    print("---- step starting at event %d (%.1f us to the next step)" % (i0, (ev[i1][0] - t0) / 1e3))
    for s, e, n, q in ev[i0:i1]:
        print("%-16s q%-3s %8.1f -> %8.1f  (%6.1f us)" % (n[:16], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
