#!/bin/bash
# the PMC passes of scripts/refresh_profiles.sh alone (cfg-2), into gpurun_out/$1
set -u
TAG=${1:-r02p}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --config 2 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg2" -o run -- python3 bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof_cfg2.log" 2>&1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE" "sq2 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64"; do
  set -- $pass; name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/pmc_$name" -o run -- \
    python3 bench.py --config 2 --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/pmc_$name.log" 2>&1
done
ls "$OUT"/*
