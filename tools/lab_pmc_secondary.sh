# HBM traffic (PMC) of the secondary kernels: GEMV 28672x8192, int8 GEMM cfg3, context attention, rmsnorm-quant
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc2
cat > /tmp/ctx_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
dev = torch.device("cuda", 0)
print(bench.bench_context_attention(dev, 3, B=2, H=64, Hkv=8, S=4096, kv="fp8"))
print(bench.bench_rmsnorm_quant(dev, 8, sets=4))
PY
run() { # name, kernel substring, counter, program...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc $1 --output-format csv -d gpurun_out/pmc2/$name -- "${@:2}" > gpurun_out/pmc2/$name.out 2>&1 || { tail -5 gpurun_out/pmc2/$name.out; exit 1; }
}
run gemv_f FETCH_SIZE python3 tools/microbench.py gemv --m 1 --n 28672 --k 8192
run gemv_w WRITE_SIZE python3 tools/microbench.py gemv --m 1 --n 28672 --k 8192
python3 tools/pmc_traffic.py gpurun_out/pmc2/gemv_f gpurun_out/pmc2/gemv_w woq_gemv_kernel gpurun_out/pmc2/gemv.json
run g8_f FETCH_SIZE python3 tools/microbench.py gemm8 --m 4096 --n 13824 --k 5120 --steps 4
run g8_w WRITE_SIZE python3 tools/microbench.py gemm8 --m 4096 --n 13824 --k 5120 --steps 4
python3 tools/pmc_traffic.py gpurun_out/pmc2/g8_f gpurun_out/pmc2/g8_w gemm8_phased_kernel gpurun_out/pmc2/gemm8.json
run cx_f FETCH_SIZE python3 /tmp/ctx_one.py
run cx_w WRITE_SIZE python3 /tmp/ctx_one.py
python3 tools/pmc_traffic.py gpurun_out/pmc2/cx_f gpurun_out/pmc2/cx_w ctx_attn_kernel gpurun_out/pmc2/ctx_attn.json
python3 tools/pmc_traffic.py gpurun_out/pmc2/cx_f gpurun_out/pmc2/cx_w ctx_prepare_fast_kernel gpurun_out/pmc2/ctx_prepare.json
python3 tools/pmc_traffic.py gpurun_out/pmc2/cx_f gpurun_out/pmc2/cx_w norm_quant_kernel gpurun_out/pmc2/norm_quant.json
