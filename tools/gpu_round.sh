#!/bin/bash
# One GPU-box visit: GPU test suite, then one bench line per BASELINE config.  Usage (through gpurun, repo root):
#   tools/gpu_round.sh TAG [STEPS]
set -o pipefail
TAG=${1:-r02a}
STEPS=${2:-5}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests -m gpu -q > "$OUT/pytest.log" 2>&1
rc=$?
tail -15 "$OUT/pytest.log"
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest ended with $rc: stopping"; exit $rc; fi
for c in 3 2 4 5 parity; do
  timeout -k 10 400 python bench.py --config $c --steps $STEPS --warmup 2 > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err" || { echo "bench $c failed"; tail -5 "$OUT/bench_$c.err"; exit 1; }
  python - "$OUT/bench_$c.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print(f"config {d['config']['workload'][:28]:28s} {d['value']:10.1f} Msamples/s  {d['ms_per_step']:9.3f} ms  frac {r['frac']:.4f}  {r['kernel']}  "
      f"lane {r['lane_utilization']} walk {r['grid_walk_lane_utilization']}  verified {[v['equal'] for v in d.get('verified_rows', [])]}")
PY
done
echo "pytest rc=$rc"
