"""Dev helper: from a rocprofv3 kernel trace, the idle gap at the side-stream join (cc_emit end -> build_rois start) and
when the RPN NMS chain finished relative to it."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last_cc = last_nms = None
gaps = []
for r in rows:
    n = r['Kernel_Name']
    if n.startswith('cc_emit_kernel'): last_cc = int(r['End_Timestamp'])
    if n.startswith('nms_reduce_kernel'): last_nms_prev, last_nms = last_nms, int(r['End_Timestamp'])
    if n.startswith('build_rois_kernel') and last_cc:
        gaps.append(((int(r['Start_Timestamp']) - last_cc) / 1e3, (last_nms - last_cc) / 1e3))
print('per step: idle between cc_emit end and build_rois start (us), RPN-NMS end minus cc_emit end (us)')
for g in gaps[-8:]: print('  %8.1f %8.1f' % g)
