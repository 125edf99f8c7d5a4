# SQ counters of the kernels whose name contains $1, from a run of the python script $2 (two counter passes, kernel trace
# only): bash scripts/kernel_pmc.sh k_kmeans_screen scripts/kmeans_time.py
R=$GRAFT_REPO_ROOT
PAT=$1; SCRIPT=$2
cd /tmp && export TMPDIR=/tmp
for pass in "a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE" "b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/kp_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/kp_$name -o run -- python3 $R/$SCRIPT > /tmp/kp_$name.log 2>&1 || tail -3 /tmp/kp_$name.log
  PAT=$PAT python3 - <<PY
import csv, collections, os
acc=collections.defaultdict(list)
for r in csv.DictReader(open('/tmp/kp_$name/run_counter_collection.csv')):
    if os.environ['PAT'] in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:40], r['Grid_Size'], r['LDS_Block_Size'] if 'LDS_Block_Size' in r else '', r['Counter_Name'])].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()):
    print(*k, '%.4g' % (sum(v[-10:])/len(v[-10:])), 'n', len(v))
PY
done
