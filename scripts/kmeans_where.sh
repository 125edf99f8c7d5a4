# timing-only / tuning builds of the matrix-core KMeans E step.  build here: bash scripts/kmeans_where.sh build; run on the GPU box without argument
X="${XS:-KM_X_NOMFMA KM_X_NOSEL KM_PB_=8 KM_PB_=7 KM_PB_=2}"
C=sober_amd/csrc
if [ "$1" = build ]; then
  make -C $C > /dev/null
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function"
  OBJS=""; for o in level_reduce level_reduce_mfma level_reduce_tani level_gather misc dgemm car car_mc chol compact car_host host_rng level_exec nystrom_exec rccl_link; do OBJS="$OBJS $C/build/$o.o"; done
  mkdir -p $C/build_x
  for x in $X; do
    ( /opt/rocm/bin/hipcc $FL -D$x -c $C/kmeans.hip -o $C/build_x/kmeans_$x.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build_x/libsober_hip_$x.so $OBJS $C/build_x/kmeans_$x.o -ldl ) > $C/build_x/$x.log 2>&1 &
  done
  wait; ls $C/build_x/*.so; exit 0
fi
for x in "" $X; do
  if [ -n "$x" ]; then export SOBER_HIP_LIB=$PWD/$C/build_x/libsober_hip_$x.so; fi
  echo "== ${x:-default}"; bash scripts/kmeans_prof.sh 2>&1 | grep -E "assign"
done
