# kernel timeline of one step of a bench configuration:  bash scripts/timeline_cfg.sh 2
R=$GRAFT_REPO_ROOT
C=${1:-2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o run -- python3 $R/bench.py --config $C --steps 6 --warmup 2 --no-cpu-baseline --no-sweep --no-others > /tmp/tl.log 2>&1 || tail -5 /tmp/tl.log
python3 $R/scripts/trace_timeline.py /tmp/tl/run_kernel_trace.csv 3
