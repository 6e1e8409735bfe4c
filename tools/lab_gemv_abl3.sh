# Lab: ablations of the light GEMV at cfg1 (see MI355Q_GEMV_LIGHT_ABL in woq_gemv_small.hip), graph-timed.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so MI355Q_GEMV_LIGHT=1
for m in ${MS:-1 16}; do
for abl in ${ABLS:-0 1 2 3 4 8 12 13 15 31}; do
  MI355Q_GEMV_LIGHT_ABL=$abl timeout -k 10 120 python3 tools/microbench.py gemv --m $m --n ${N:-4096} --k ${K:-4096} > gpurun_out/lab/o.txt 2>gpurun_out/lab/err.txt || { tail -5 gpurun_out/lab/err.txt; exit 1; }
  echo "m=$m abl=$abl $(python3 -c "import json;o=json.load(open('gpurun_out/lab/o.txt'));print(round(o['us_per_call'],2),'us')")"
done; done
