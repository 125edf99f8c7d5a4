#!/bin/bash
# Run on the GPU box (gpurun): driver-shaped bench lines, rocprofv3 kernel stats for every BASELINE configuration and the
# PMC passes of the cfg-2 step (FETCH_SIZE, WRITE_SIZE, two SQ sets; each its own run, kernel trace only).  Output under gpurun_out/$1 (default r02f);
# scripts/collect_profiles.py copies the judged summaries into profiles/.
set -u
TAG=${1:-r02f}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for c in 1 2 3 4 5; do
  python3 bench.py --config $c > "$OUT/bench_cfg$c.json" 2> "$OUT/bench_cfg$c.err"
done
for c in 1 2 3 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_cfg$c" -o run -- \
    python3 bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/prof_cfg$c.log" 2>&1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -o run -- \
  python3 bench.py --config 2 --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -o run -- \
  python3 bench.py --config 2 --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o run -- \
  python3 bench.py --config 2 --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 \
  --kernel-trace --output-format csv -d "$OUT/pmc_sq2" -o run -- \
  python3 bench.py --config 2 --steps 3 --warmup 2 --no-cpu-baseline > "$OUT/pmc_sq2.log" 2>&1
ls -R "$OUT" | head -60
