# usage: bash scripts/level_kernel_sweep.sh "<lib> <args...>" ...   (lib: head | block | <variant built into sober_amd/csrc/build_<variant>/)
for spec in "$@"; do
  set -- $spec; lib=$1; shift
  case $lib in head) unset SOBER_HIP_LIB;; *) export SOBER_HIP_LIB=$PWD/sober_amd/csrc/build_$lib/libsober_hip_$lib.so;; esac
  echo "== $lib $*"
  python scripts/level_kernel_sweep.py "$@" 2>&1 | grep -v "amdgpu.ids"
done
