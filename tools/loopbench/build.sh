#!/bin/bash
# [LB_SRC=tail_bench.hip] build.sh NAME [extra hipcc flags...]  ->  build/lb/NAME (travels to the GPU box: build/ is git-ignored, not gpurun-ignored)
# prints registers / scratch of the den_loop_kernel instantiations it holds
set -e
cd "$(dirname "$0")/../.."
NAME=$1; shift
mkdir -p build/lb
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-unused-function -I motion-latent-diffusion_amd/csrc/kernels \
  -DLB_NAME="\"$NAME\"" "$@" -Rpass-analysis=kernel-resource-usage -o build/lb/$NAME tools/loopbench/${LB_SRC:-loop_bench.hip} 2> build/lb/$NAME.ru || { tail -20 build/lb/$NAME.ru; exit 1; }
grep -A12 -E "Function Name: _ZN3mld(15den_loop|18den_cluster|19ffn_strip_x3|20strip_gemm_x3|20attn_flash_x3|11gemm_kernel|11gemm_big)" build/lb/$NAME.ru | grep -E "Function Name|VGPRs:|ScratchSize|SGPRs:" | sed 's/.*remark: *//' | paste - - - - 
