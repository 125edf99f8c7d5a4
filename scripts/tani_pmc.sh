# SQ counters of the fingerprint level kernel alone (level 0 of configuration 5: scripts/tani_kernel_time.py), two passes
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/tp_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/tp_$name -o run -- python3 $R/scripts/tani_kernel_time.py > /tmp/tp_$name.log 2>&1 || tail -3 /tmp/tp_$name.log
  python3 - <<PY
import csv, collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/tp_$name/run_counter_collection.csv')):
    if 'tani' in r['Kernel_Name'] and r['Grid_Size'] in ('119808', '259584', '39936'):
        acc[(r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
for (g,c),v in sorted(acc.items()):
    print(g, c, '%.4g' % (sum(v[-10:])/len(v[-10:])))
PY
done
