#!/bin/bash
# Run on the GPU box (gpurun): for every BASELINE configuration a driver-shaped bench line, a rocprofv3 kernel-stats run of
# the same command and the two HBM counter passes (FETCH_SIZE, WRITE_SIZE: each its own run, kernel trace only -- never
# together with other trace domains); for configuration 2 also two SQ counter sets; the FP64 rate probe.
# Output under gpurun_out/$1 (default r04f); scripts/collect_profiles.py copies the judged summaries into profiles/.
#   bash scripts/refresh_profiles.sh r04f "1 2 3 4 5"
set -u
TAG=${1:-r04f}
CFGS=${2:-"1 2 3 4 5"}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ -x build/probes/dp_rate_probe ] || make -f scripts/probes.mk build/probes/dp_rate_probe 2> "$OUT/dp_rate_build.log"; then
  BIN=build/probes/dp_rate_probe
  $BIN > "$OUT/dp_rate.txt" 2>&1
fi
for c in $CFGS; do
  python3 bench.py --config $c > "$OUT/bench_cfg$c.json" 2> "$OUT/bench_cfg$c.err"
done
for c in $CFGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg$c" -o run -- \
    python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-others > "$OUT/prof_cfg$c.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_cfg$c" -o run -- \
    python3 bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-others > "$OUT/pmc_fetch_cfg$c.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write_cfg$c" -o run -- \
    python3 bench.py --config $c --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-others > "$OUT/pmc_write_cfg$c.log" 2>&1
done
if echo " $CFGS " | grep -q " 2 "; then
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o run -- \
    python3 bench.py --config 2 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-others > "$OUT/pmc_sq.log" 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 \
    --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -o run -- \
    python3 bench.py --config 2 --steps 3 --warmup 2 --no-cpu-baseline --no-sweep --no-others > "$OUT/pmc_sq2.log" 2>&1
fi
# keep what travels back small: the per-dispatch traces of the stats runs are not needed (the stats are)
find "$OUT" -name "run_kernel_trace.csv" -path "*prof_cfg*" -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
