#!/bin/bash
# Run on the GPU box: sample shader clocks / power of every visible GPU while bench.py runs (read-only queries).
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/clock_watch.txt
rm -f $out
python3 $root/bench.py --steps 990 --warmup 5 --no-cpu-baseline "$@" > $root/gpurun_out/clock_bench.json 2>&1 &
pid=$!
for i in $(seq 1 40); do
  echo "-- t=$i" >> $out
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)" >> $out
  kill -0 $pid 2>/dev/null || break
  sleep 1
done
wait $pid
tail -c 300 $root/gpurun_out/clock_bench.json
grep -E "sclk" $out | sort | uniq -c | sort -rn | head -20
