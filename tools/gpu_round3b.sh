#!/bin/bash
# Late round-3 evidence in ONE gpurun call (the round's GPU budget was nearly spent): the loop kernel's PMC traffic passes on the
# final build (bench.py reads profiles/r03_pmc_traffic.json), then smoke + GPU parity tests + bench (tools/gpu_check.sh), then the A/B
# of the two decoder options added last ("dec_l0_once", "nt_hints") at the headline call shape.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
T0=$SECONDS
PMC_TRAFFIC_ONLY=1 PMC_TIMEOUT=120 bash tools/gpu_pmc.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
[ -s gpurun_out/${TAG}_pmc_traffic.json ] && cp gpurun_out/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
echo "pmc seconds: $((SECONDS - T0))"; tail -2 gpurun_out/${TAG}_pmc.log | cut -c1-400
bash tools/gpu_check.sh $TAG
echo "check done at: $((SECONDS - T0)) s"
AB_OPTS='[{"dec_l0_once": 0, "nt_hints": 0}, {"dec_l0_once": 1}, {"dec_l0_once": 0}, {"dec_l0_once": 1}, {"nt_hints": 1}, {"nt_hints": 2}, {"nt_hints": 4}, {"nt_hints": 8}, {"nt_hints": 0}, {"nt_hints": 3}, {"nt_hints": 12}, {"nt_hints": 15}, {"nt_hints": 0}, {"nt_hints": 15}]' \
  timeout 240 python tools/ab_decode.py > gpurun_out/${TAG}b_decoder_ab.log 2>&1
tail -1 gpurun_out/${TAG}b_decoder_ab.log | cut -c1-2500
echo "total seconds: $((SECONDS - T0))"
