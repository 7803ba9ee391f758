# dev: rocprofv3 kernel durations of the same 60 steps with the submitting process on each NUMA node: are the KERNELS slower from the remote
# node, or the gaps between them?  (results -> gpurun_out/state_numa_prof.txt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/state_numa_prof.txt; : > $O
export NUHTC_HOST_AFFINITY=0
for n in 0 1; do
  C=$(cat /sys/devices/system/node/node$n/cpulist)
  rm -rf /tmp/np$n
  ( cd $R && timeout 240 taskset -c $C rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np$n -- python3 tools/dev/r04_state_numa2.py ) 2>&1 | grep "^gpu" >> $O
  F=$(find /tmp/np$n -name "*kernel_stats.csv" | head -1)
  if [ -z "$F" ]; then echo "node $n: no stats file" >> $O; continue; fi
  echo "node $n: $(grep 'gemm_split_kernel<1, 3, 0>' "$F" < /dev/null | cut -d, -f1-4)" >> $O
  T=$(find /tmp/np$n -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && timeout 120 python3 - "$T" >> $O <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 30 steps' worth of launches: kernel time and the gaps between consecutive kernels on the busiest queue
q = max(set(r['Queue_Id'] for r in rows), key=lambda x: sum(1 for r in rows if r['Queue_Id'] == x))
rows = [r for r in rows if r['Queue_Id'] == q][-6000:]
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rows, rows[1:])]
gaps = [g for g in gaps if 0 <= g < 200000]
g3 = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows if 'gemm_split_kernel<1, 3, 0>' in r['Kernel_Name']]
print(f'   main queue, last {len(rows)} launches: kernel time {busy / 1e6:.2f} ms, gaps {sum(gaps) / 1e6:.2f} ms (mean {sum(gaps) / len(gaps) / 1e3:.2f} us, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us); gemm_split<1,3,0> mean {sum(g3) / len(g3) / 1e3:.2f} us over {len(g3)}')
PY
done
cat $O
