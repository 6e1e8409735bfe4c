cd $GRAFT_REPO_ROOT
for nt in ${NTS:-0 128}; do
echo "== NT $nt"
MI355Q_NORM_NT=$nt timeout -k 10 300 python -u -m pytest tests/test_gpu_norm_quant.py -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | tail -1 || exit 1
MI355Q_NORM_NT=$nt timeout -k 10 200 python3 tools/lab_norm.py 2>&1 | grep tokens || exit 1
done
