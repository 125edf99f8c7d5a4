for v in head xNODSREAD xNOMFMA xNOQUOT xNOEXPAND "$@"; do
  if [ $v = head ]; then unset SOBER_HIP_LIB; else export SOBER_HIP_LIB=$PWD/sober_amd/csrc/build_$v/libsober_hip_$v.so; fi
  python scripts/tani_kernel_time.py 2>&1 | grep -v "amdgpu.ids" | tail -1
done
