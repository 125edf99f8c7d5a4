# timing-only builds of the merged bidiagonalisation (wrong results): what each part of a step costs.
# build here:  bash scripts/bidiag_where.sh build     run on the GPU box:  bash scripts/bidiag_where.sh
X="${XS:-NOLARFG NORSUM NOHUPD NOGUPD NOQ NOY NOBAR NOD NOVSTORE}"
C=sober_amd/csrc
if [ "$1" = build ]; then
  make -C $C > /dev/null
  FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function"
  OBJS=""; for o in level_reduce level_reduce_mfma level_reduce_tani level_gather misc dgemm kmeans car_mc chol compact car_host host_rng level_exec rccl_link; do OBJS="$OBJS $C/build/$o.o"; done
  for x in $X; do
    mkdir -p $C/build_x
    ( /opt/rocm/bin/hipcc $FL -DCB2_X_$x -c $C/car.hip -o $C/build_x/car_$x.o && \
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $C/build_x/libsober_hip_$x.so $OBJS $C/build_x/car_$x.o -ldl ) > $C/build_x/$x.log 2>&1 &
  done
  wait
  exit 0
fi
libs=""
for x in $X; do libs="$libs $C/build_x/libsober_hip_$x.so"; done
bash scripts/car_ab.sh $libs | grep -E "lib:|bidiag"
