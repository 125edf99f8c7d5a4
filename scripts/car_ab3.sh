# three builds of the Caratheodory kernels on one box: in-tree, and the libraries named on the command line (relative paths)
R=$GRAFT_REPO_ROOT
cd $R
python scripts/car_dump.py /tmp/d0.npz 2>&1 | tail -1
i=1
for lib in "$@"; do
  SOBER_HIP_LIB=$R/$lib python scripts/car_dump.py /tmp/d$i.npz 2>&1 | tail -1
  echo "in-tree vs $lib:"; python scripts/car_dump.py --compare /tmp/d0.npz /tmp/d$i.npz
  i=$((i+1))
done
if [ $# -ge 2 ]; then echo "$1 vs $2:"; python scripts/car_dump.py --compare /tmp/d1.npz /tmp/d2.npz; fi
bash scripts/car_ab.sh "$@" 2>&1 | grep -v amdgpu.ids
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export SOBER_HIP_LIB=$R/$lib; else unset SOBER_HIP_LIB; fi
  echo "mc, lib: ${lib:-in-tree}"; python scripts/car_mc_time.py 2>&1 | tail -3
done
