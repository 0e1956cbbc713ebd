cd "$GRAFT_REPO_ROOT"
for c in 1048576 262144 131072 65536 1048576; do
  EGNN_BWD_CHUNK=$c python bench.py --mode train --steps 10 --warmup 3 --reps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('chunk %-8s train ms/step %.3f' % ('$c', d['ms_per_step']))"
done
