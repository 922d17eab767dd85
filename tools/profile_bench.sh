#!/bin/bash
# Profile one bench workload on the GPU box (run from the repo root through gpurun):
#   tools/profile_bench.sh TAG [CONFIG] [STEPS]
#   pass 0: the full bench line (cpu_baseline, verified_rows, HIP-event kernel time) -> gpurun_out/prof_TAG/bench.json
#   pass 1: rocprofv3 --kernel-trace --stats                                -> .../trace
#   pass 2..: PMC counters, each group in its own run (never mixed with other trace domains)
# Summaries land under gpurun_out/prof_<tag>/; tools/summarize_profile.py copies what is judged into profiles/.
set -o pipefail
TAG=${1:-r02}
CONFIG=${2:-3}
STEPS=${3:-3}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# extra bench.py arguments of this workload (e.g. BENCH_ARGS="--spp 2 --size 800x600" = parity mode at the reference's operating point)
BENCH="python3 $PWD/bench.py --config $CONFIG --steps $STEPS --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
# pass 0 is the FULL line -- cpu_baseline and verified_rows included, 10 timed steps: the one profiles/<tag>_bench.json keeps; the
# rocprofv3 passes below run the same workload without the CPU legs (they would only be profiled host time)
python3 $PWD/bench.py --config $CONFIG --steps ${FULL_STEPS:-10} --warmup 2 ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
# the trace pass runs the driver's default step counts (20 timed + 3 warm-up): its per-kernel AVERAGE is then the steady state the HIP events of
# pass 0 see, not a mean over three launches and a cold first one
BENCH_TRACE="python3 $PWD/bench.py --config $CONFIG --steps ${TRACE_STEPS:-20} --warmup 3 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH_TRACE > "$OUT/trace.log" 2>&1 || { tail -20 "$OUT/trace.log"; exit 1; }
i=0
GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE GRBM_COUNT")
if [ "${QUICK:-0}" != "1" ]; then
  GROUPS_+=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum")
fi
for PMC in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d "$OUT/pmc$i" -- $BENCH > "$OUT/pmc$i.log" 2>&1 || { echo "pmc pass $i failed"; tail -5 "$OUT/pmc$i.log"; }
done
echo "profiled config $CONFIG -> $OUT"
