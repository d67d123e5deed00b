"""Timeline of the LAST filter_lr of a `rocprofv3 --kernel-trace --output-format csv -- python3 tools/enc_time.py ...` run:
per queue, start / end / duration of every kernel (us since the first).   python tools/diag/enc_timeline.py <kernel_trace.csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
idx = [i for i, r in enumerate(rows) if 'gn_partial' in r['Kernel_Name']]
per = 3 if n == 0 else n          # gn_partial launches per filter_lr (fused: the first block's three)
seg = rows[idx[-per]:]
t0 = int(seg[0]['Start_Timestamp'])
for r in seg:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    nm = r['Kernel_Name'].replace('surs::enc::', '').replace('void ', '').split('(')[0][:44]
    print("q%-2s %8.1f %8.1f dur %6.1f  %-44s %s x %s x %s" % (r['Queue_Id'], s, e, e - s, nm, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']),
                                                             r['Grid_Size_Y'], r['Grid_Size_Z']))
