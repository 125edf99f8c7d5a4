# Run on the GPU box: the small evidence files DESIGN.md cites besides the per-configuration profiles -- per-kernel
# durations of one Caratheodory step (scripts/car_time.py), the bidiagonalisation's in-kernel stamps (stamp build), the
# multi-CU step's kernels, KMeans per kernel.  Output: gpurun_out/$1/*.txt (copy into profiles/ by hand).
TAG=${1:-r03e}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
bash scripts/car_ab.sh 2>&1 | grep -E "lib:|car" > $OUT/car_kernels.txt
if [ -f sober_amd/csrc/build_stamps/libsober_hip_stamps.so ]; then
  SOBER_HIP_LIB=sober_amd/csrc/build_stamps/libsober_hip_stamps.so python3 scripts/car_stamps.py 2>&1 | grep -E "wave|block" > $OUT/car_stamps.txt
fi
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mck && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mck -o run -- python3 $R/scripts/car_mc_time.py > /tmp/mck.log 2>&1; grep -i "ms" /tmp/mck.log | head -5; python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/mck/run_kernel_stats.csv')):
    if 'k_mc' in r['Name']: print('  ', r['Name'][:60], r['Calls'], 'avg us %.1f' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3))
PY
) > $OUT/car_mc_kernels.txt 2>&1
KM_MINPCT=0.3 bash scripts/kmeans_prof.sh 2>&1 | cut -c1-160 > $OUT/kmeans_kernels.txt
python3 scripts/kmeans_time.py >> $OUT/kmeans_kernels.txt 2>&1
ls -la $OUT
