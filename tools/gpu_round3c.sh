#!/bin/bash
# Second late round-3 call: A/B of "nt_hints" (now a template parameter of the row-strip GEMMs) and "attn_tr" (transpose-read V /
# streaming hints in the key-blocked attention) at the headline call shape, under rocprofv3 --kernel-trace --stats so that every
# variant -- a kernel of its own name -- gets its own dispatch average; plus the GPU test that pins the options' results.
set -u
TAG=${1:-r03c}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$SECONDS
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "first_decoder_layer or key_blocked or split_f16_decode_mode" 2>&1 | tail -5
echo "tests done at $((SECONDS - T0)) s"
AB='[{"nt_hints": 0, "attn_tr": 0}, {"nt_hints": 1}, {"nt_hints": 0}, {"nt_hints": 1}, {"nt_hints": 0, "attn_tr": 1}, {"attn_tr": 0}, {"attn_tr": 1}, {"attn_tr": 2}, {"attn_tr": 3}, {"attn_tr": 0}, {"attn_tr": 3, "nt_hints": 1}, {"attn_tr": 0, "nt_hints": 0}, {"attn_tr": 1, "nt_hints": 1}, {"attn_tr": 0, "nt_hints": 0}]'
(cd /tmp && AB_OPTS="$AB" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG} -o ab -- python $R/tools/ab_decode.py > $R/gpurun_out/${TAG}_decoder_ab.log 2>&1)
tail -1 gpurun_out/${TAG}_decoder_ab.log | cut -c1-2500
F=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1)
[ -n "$F" ] && { cp $F gpurun_out/${TAG}_kernel_stats_ab.csv; grep -E "Name|attn_flash|strip_gemm_x3|ffn_strip|den_loop" $F | cut -c1-200; }
find gpurun_out/prof_${TAG} -name "*.csv" -size +4M -delete
echo "total seconds: $((SECONDS - T0))"
