#!/bin/bash
# The GPU test suite against a build of libcusmc_hip.so whose HOST code (cusmc_abi.hip: handles, plans, the multi-device
# threads, the copy-out) is compiled with UBSan (undefined, bounds, float-divide-by-zero; first report aborts) and
# libstdc++'s assertions (_GLIBCXX_ASSERTIONS: every std::vector / std::array index checked).  Device code is not
# instrumented.  (AddressSanitizer was tried first: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and
# aborts on the first device allocation of a process that is not an xnack+ ASan build -- which this pool does not run.)
#     bash scripts/host_sanitizers.sh build             # -> cusmc_amd/libcusmc_hip_ubsan.so (git-ignored; travels with gpurun)
#     bash scripts/host_sanitizers.sh run [pytest args]  # e.g. run tests -m gpu -x -q
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
RTDIR=/opt/rocm/lib/llvm/lib/clang/22/lib/linux
case "$1" in
build)
  make -C "$R/cusmc_amd/csrc" -j8
  cd "$R/cusmc_amd/csrc" && mkdir -p build_san
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed \
    -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-atomic-optimizer-strategy=None \
    -Xarch_host -fsanitize=undefined,bounds,float-divide-by-zero -Xarch_host -fno-sanitize-recover=undefined \
    -Xarch_host -D_GLIBCXX_ASSERTIONS -Xarch_host -fno-omit-frame-pointer \
    -Ibuild -c cusmc_abi.hip -o build_san/cusmc_abi.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -fsanitize=undefined -shared-libsan \
    -o ../libcusmc_hip_ubsan.so build_san/cusmc_abi.o build/kernels/*.o
  ;;
run)
  shift
  cd "$R"
  CUSMC_LIBRARY="$R/cusmc_amd/libcusmc_hip_ubsan.so" LD_LIBRARY_PATH=$RTDIR:$LD_LIBRARY_PATH \
    UBSAN_OPTIONS=print_stacktrace=1 python3 -m pytest "$@"
  ;;
*) echo "usage: $0 build | run [pytest args]"; exit 2;;
esac
