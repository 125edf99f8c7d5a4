# the screened KMeans E step with smaller margins than the 2^-12 that ships (diagnostic builds, on the GPU box):
#   for m in 14 16 18 20 22; do make -C sober_amd/csrc BUILD=build_km$m EXTRA="-DSOBER_DIAG_BUILD -DKM_MARGIN_LOG2=$m" OUT=build_km$m/libsober_hip_km$m.so; done
R=${GRAFT_REPO_ROOT:-$PWD}
echo "margin 2^-12 (the library)"; python3 $R/scripts/kmeans_margin.py 2>&1 | grep -v amdgpu.ids
for m in 14 16 18 20 22; do
  L=$R/sober_amd/csrc/build_km$m/libsober_hip_km$m.so
  [ -f $L ] || continue
  echo "margin 2^-$m"; SOBER_HIP_LIB=$L SOBER_ALLOW_DIAG_LIB=1 python3 $R/scripts/kmeans_margin.py 2>&1 | grep -v amdgpu.ids
done
