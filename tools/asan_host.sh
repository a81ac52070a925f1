#!/bin/bash
# Host-sanitizer build of the library's HOST side: every translation unit compiled with AddressSanitizer +
# UBSan on the HOST code only (-fno-gpu-sanitize: the gfx950 code objects are the product's, unsanitized, and nothing here
# launches them), linked with the workspace fuzz driver tools/fuzz_workspaces.cpp.  CPU only -- never run on the GPU box.
#   bash tools/asan_host.sh [cases]      -> log in profiles/<round>_asan_host_fuzz.log (pass the name as $2)
set -e
cd "$(dirname "$0")/.."
B=/tmp/mobocmf_asan_host
rm -rf $B && mkdir -p $B
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -O1 -g"
pids=""
for f in gemm_f64 chol gram elementwise rff tiny_step coop_step api; do
  $HIPCC --offload-arch=gfx950 -fno-gpu-sanitize $SAN -DMOBOCMF_HOST_FUZZ -fPIC -std=c++17 -Wno-unused-result -c mobocmf_amd/csrc/$f.hip -o $B/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/lib/llvm/bin/clang++ $SAN -std=c++17 -c tools/fuzz_workspaces.cpp -o $B/fuzz.o
$HIPCC --offload-arch=gfx950 -fno-gpu-sanitize $SAN $B/*.o -o $B/fuzz_workspaces
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
if [ -n "$2" ]; then $B/fuzz_workspaces ${1:-4000} 2>&1 | tee "$2"; else $B/fuzz_workspaces ${1:-4000} 2>&1; fi
