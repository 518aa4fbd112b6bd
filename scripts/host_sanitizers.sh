#!/bin/bash
# The GPU test suite against a build of libcusmc_hip.so whose HOST code (cusmc_abi.hip: handles, plans, the multi-device
# threads, the copy-out) is compiled with UBSan (undefined, bounds, float-divide-by-zero; first report aborts) and
# libstdc++'s assertions (_GLIBCXX_ASSERTIONS: every std::vector / std::array index checked).  Device code is not
# instrumented.  (AddressSanitizer was tried first: ROCm's ASan runtime intercepts hsa_amd_memory_pool_allocate and
# aborts on the first device allocation of a process that is not an xnack+ ASan build -- which this pool does not run.)
#     bash scripts/host_sanitizers.sh build             # -> cusmc_amd/libcusmc_hip_ubsan.so (git-ignored; travels with gpurun)
#     bash scripts/host_sanitizers.sh run [pytest args]  # e.g. run tests -m gpu -x -q
# ThreadSanitizer over the multi-device entry points (host threads, events, the published[] handshake), with a small
# C++ driver instead of Python (scripts/calib/tsan_multi.cpp: three shards on device 0).  The HIP / HSA runtimes are not
# instrumented, so TSan reports their internal allocations as races: the run prints how many reports have a racing
# access inside libcusmc_hip itself (round 3: 0 of 86).
#     bash scripts/host_sanitizers.sh tsan-build        # -> cusmc_amd/libcusmc_hip_tsan.so, scripts/calib/tsan_multi
#     bash scripts/host_sanitizers.sh tsan-run          # on the GPU box; reports under gpurun_out/tsan/
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
tsan-build)
  make -C "$R/cusmc_amd/csrc" -j8
  cd "$R/cusmc_amd/csrc" && mkdir -p build_san
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed \
    -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-atomic-optimizer-strategy=None -Xarch_host -fsanitize=thread \
    -Ibuild -c cusmc_abi.hip -o build_san/cusmc_abi_tsan.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fno-gpu-sanitize -fsanitize=thread -shared-libsan \
    -o ../libcusmc_hip_tsan.so build_san/cusmc_abi_tsan.o build/kernels/*.o
  /opt/rocm/lib/llvm/bin/clang++ -O1 -g -fsanitize=thread -shared-libsan -I"$R/include" "$R/scripts/calib/tsan_multi.cpp" \
    -L"$R/cusmc_amd" -lcusmc_hip_tsan -o "$R/scripts/calib/tsan_multi"
  ;;
tsan-run)
  cd "$R" && rm -rf gpurun_out/tsan && mkdir -p gpurun_out/tsan
  LD_LIBRARY_PATH=$RTDIR:$R/cusmc_amd:$LD_LIBRARY_PATH \
    TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 history_size=4 log_path=$R/gpurun_out/tsan/log" \
    ./scripts/calib/tsan_multi || true
  python3 - "$R"/gpurun_out/tsan/log.* <<'PY'
import re, sys
reports = [r for f in sys.argv[1:] for r in open(f).read().split("==================\n") if "WARNING: ThreadSanitizer" in r]
mine = 0
for r in reports:
    tops = []
    for b in re.findall(r"((?:Previous )?(?:[Aa]tomic )?(?:[Ww]rite|[Rr]ead) of size \d+ at .*?\n(?:    #\d+ .*\n)+)", r):
        frames = [x for x in re.findall(r"    #\d+ (.*)\n", b) if "libclang_rt.tsan" not in x]
        tops.append(frames[0] if frames else "")
    if any("libcusmc_hip" in t for t in tops):
        mine += 1
        print(r)
print("%d ThreadSanitizer reports, %d with a racing access inside libcusmc_hip" % (len(reports), mine))
PY
  ;;
*) echo "usage: $0 build | run [pytest args] | tsan-build | tsan-run"; exit 2;;
esac
