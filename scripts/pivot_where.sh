# Timing-only builds of k_car_pivot_stream (results are wrong; only the duration matters): what each part of the
# producing wave's loop costs.  Build first:  for v in NODIV NOPUB NOELIM; do
#   make -C sober_amd/csrc BUILD=build_x$v EXTRA=-DSP_X_$v OUT=build_x$v/lib.so; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" NODIV NOPUB NOELIM; do
  if [ -n "$v" ]; then export SOBER_HIP_LIB=$R/sober_amd/csrc/build_x$v/lib.so; fi
  rm -rf /tmp/pw
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pw -o run -- python3 $R/scripts/car_time.py > /tmp/pw.log 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open('/tmp/pw/run_kernel_stats.csv')):
    if 'pivot_stream' in r['Name']: print('${v:-base}', 'k_car_pivot_stream avg us %.1f' % (float(r['AverageNs'])/1e3))
PY
done
